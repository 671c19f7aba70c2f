#!/bin/bash
# tools/kernel_regs.sh OBJECT [PATTERN] -- LDS bytes, VGPRs and spills of the kernels in a hipcc object (its gfx950 code object's notes)
L=/opt/rocm/lib/llvm/bin; T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
$L/clang-offload-bundler -type=o -unbundle -targets=hipv4-amdgcn-amd-amdhsa--gfx950 -input=$T/fat.bin -output=$T/k.hsaco
$L/llvm-readelf --notes $T/k.hsaco | grep -E "^\s+\.name:|group_segment_fixed_size|\.vgpr_count|vgpr_spill|sgpr_spill" | paste - - - - - | sed 's/ \+/ /g' | grep -E "${2:-.}" | while read -r line; do
  n=$(echo "$line" | sed 's/.*\.name: \([^ \t]*\).*/\1/'); echo "$line" | sed "s/\.name: [^ \t]*//" | tr -d '\n'; echo " $(echo $n | c++filt | cut -c1-60) ... $(echo $n | grep -o 'Li[0-9]*ELi[0-9]*ELi[0-9]*ELi[0-9]*ELi[0-9]*E[A-Za-z0-9]*$')"; done
rm -rf $T
