#!/bin/bash
# tools/build_variants.sh NAME "-DFLAG=.. -DFLAG2=.." -- A/B build of ONE kinds translation unit with extra macros / options:
# writes variants/libfundsp_hip_NAME.so (select it with FUNDSP_HIP_LIB=...).  Design tool, not part of the product.
#   FILE=fd_kinds_fm (default: the oscillator -> filter chains incl. the headline kernel, ~15 s) | fd_kinds_fm_ts (their
#   three-way time-split kernels) | fd_kinds_graph (~90 s)
#   ILP=0 drops the fm unit's -mllvm -amdgpu-sched-strategy=iterative-ilp
#   FILE2=fd_kinds_graph_mix: a second unit built with the same macros (plain flags) into the same library
#   EXPERIMENTS=1: build from a scratch copy of the source of commit 9d4be23 with tools/variants/r04_experiments.patch applied -- the
#   tree that still has the retired round-4 switches (FD_STAGE_SPLIT, FD_WT_PAIRS, FD_ROLE_CROSS, FD_KNOCK, ...; tools/variants/README.md)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
if [ "${EXPERIMENTS:-0}" = "1" ]; then
  rm -rf "$ROOT/variants/_exp" && mkdir -p "$ROOT/variants/_exp"
  git -C "$ROOT" archive 9d4be23 fundsp_amd/csrc include | tar -x -C "$ROOT/variants/_exp"
  (cd "$ROOT/variants/_exp" && patch -s -p1 < "$ROOT/tools/variants/r04_experiments.patch")
  make -s -C "$ROOT/variants/_exp/fundsp_amd/csrc" -j8
  cd "$ROOT/variants/_exp/fundsp_amd/csrc"
else
  cd "$ROOT/fundsp_amd/csrc"
fi
NAME=$1; shift
FILE=${FILE:-fd_kinds_fm}
mkdir -p "$ROOT/variants"
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -Wno-unused-value"
flags_of() {  # the Makefile's scheduling strategy per unit
  local f="$BASE"
  if { [ "$1" = "fd_kinds_fm" ] || [ "$1" = "fd_kinds_fm_mix" ]; } && [ "${ILP:-1}" = "1" ]; then f="$f -mllvm -amdgpu-sched-strategy=iterative-ilp"; fi
  if [ "$1" = "fd_kinds_fm_ts" ] && [ "${ILP:-1}" = "1" ]; then f="$f -mllvm -amdgpu-sched-strategy=max-ilp"; fi
  echo "$f"
}
/opt/rocm/bin/hipcc $(flags_of $FILE) $@ -c $FILE.hip -o /tmp/${FILE}_$NAME.o
if [ -n "$FILE2" ]; then /opt/rocm/bin/hipcc $(flags_of $FILE2) $@ -c $FILE2.hip -o /tmp/${FILE2}_$NAME.o; fi
OBJS=""
for o in fd_capi fd_kinds_leaf fd_kinds_graph fd_kinds_graph_mix fd_kinds_fm fd_kinds_fm_mix fd_kinds_fm_ts fd_fdn fd_jit fd_comm fd_rust; do
  if [ "$o" = "$FILE" ]; then OBJS="$OBJS /tmp/${FILE}_$NAME.o"; elif [ "$o" = "$FILE2" ]; then OBJS="$OBJS /tmp/${FILE2}_$NAME.o"; else OBJS="$OBJS $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/variants/libfundsp_hip_$NAME.so" $OBJS -lhiprtc -lrccl -ldl
echo built variants/libfundsp_hip_$NAME.so
