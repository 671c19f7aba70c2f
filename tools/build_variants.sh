#!/bin/bash
# tools/build_variants.sh NAME "-DFLAG=.. -DFLAG2=.." -- A/B build of ONE kinds translation unit with extra macros / options:
# writes variants/libfundsp_hip_NAME.so (select it with FUNDSP_HIP_LIB=...).  Design tool, not part of the product.
#   FILE=fd_kinds_fm (default: the oscillator -> filter chains incl. the headline kernel, ~15 s) | fd_kinds_fm_ts (their
#   three-way time-split kernels) | fd_kinds_graph (~90 s)
#   ILP=0 drops the fm unit's -mllvm -amdgpu-sched-strategy=iterative-ilp
set -e
cd "$(dirname "$0")/../fundsp_amd/csrc"
NAME=$1; shift
FILE=${FILE:-fd_kinds_fm}
mkdir -p ../../variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -Wno-unused-value"
if { [ "$FILE" = "fd_kinds_fm" ] || [ "$FILE" = "fd_kinds_fm_mix" ]; } && [ "${ILP:-1}" = "1" ]; then FLAGS="$FLAGS -mllvm -amdgpu-sched-strategy=iterative-ilp"; fi
if [ "$FILE" = "fd_kinds_fm_ts" ] && [ "${ILP:-1}" = "1" ]; then FLAGS="$FLAGS -mllvm -amdgpu-sched-strategy=max-ilp"; fi
/opt/rocm/bin/hipcc $FLAGS $@ -c $FILE.hip -o /tmp/${FILE}_$NAME.o
OBJS=""
for o in fd_capi fd_kinds_leaf fd_kinds_graph fd_kinds_graph_mix fd_kinds_fm fd_kinds_fm_mix fd_kinds_fm_ts fd_fdn fd_jit fd_comm fd_rust; do
  if [ "$o" = "$FILE" ]; then OBJS="$OBJS /tmp/${FILE}_$NAME.o"; else OBJS="$OBJS $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libfundsp_hip_$NAME.so $OBJS -lhiprtc -lrccl -ldl
echo built variants/libfundsp_hip_$NAME.so
