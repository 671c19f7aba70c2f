#!/bin/bash
# tools/build_variant_asm.sh NAME "PASS ARGS" [-DFLAGS...] -- like build_variants.sh, but fd_kinds_fm goes through its device assembly
# with tools/prio_pass.py in between (static priority mixes in the real kernel; design tool): variants/libfundsp_hip_NAME.so
set -e
cd "$(dirname "$0")/../fundsp_amd/csrc"
NAME=$1; PASSARGS=$2; shift; shift
LLVM=/opt/rocm/lib/llvm/bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -Wno-unused-value -mllvm -amdgpu-sched-strategy=iterative-ilp"
T=/tmp/fmv_$NAME
/opt/rocm/bin/hipcc $FLAGS $@ --cuda-device-only -S fd_kinds_fm.hip -o $T.dev.s 2>/dev/null
python3 ../../tools/prio_pass.py $T.dev.s $T.pp.s $PASSARGS
if [ "${ALIGN:-0}" = "1" ]; then python3 align_pass.py $T.pp.s $T.al.s; else cp $T.pp.s $T.al.s; fi
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T.al.s -o $T.dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T.hsaco $T.dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T.hsaco -output=$T.hipfb
/opt/rocm/bin/hipcc $FLAGS $@ --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T.hipfb -c fd_kinds_fm.hip -o $T.o
OBJS=""
for o in fd_capi fd_kinds_leaf fd_kinds_graph fd_kinds_fm fd_kinds_fm_ts fd_fdn fd_jit fd_comm fd_rust; do
  if [ "$o" = "fd_kinds_fm" ]; then OBJS="$OBJS $T.o"; else OBJS="$OBJS $o.o"; fi
done
mkdir -p ../../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libfundsp_hip_$NAME.so $OBJS -lhiprtc -lrccl -ldl
rm -f $T.dev.s $T.pp.s $T.al.s $T.dev.o $T.hsaco $T.hipfb
echo built variants/libfundsp_hip_$NAME.so
