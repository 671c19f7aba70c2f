#!/usr/bin/env python3
"""tools/pipe_cost.py ASM.s KERNEL_REGEX [min_instructions] -- static VALU-pipe cost of a kernel's loops (design tool).

Prices every loop of the kernels whose mangled name matches KERNEL_REGEX with the per-class SIMD costs measured on gfx950
(profiles/r03_ubench_issue_v3.txt read as co-run rates, profiles/r03_looplab.txt section 4):
    2 cycles   VOP1/VOP2/VOP3 with VGPR / inline-constant / literal sources (mul add fma lshl and cvt ...)
    4 cycles   packed f32 (v_pk_fma/mul/add_f32), three-operand integer/select (v_max3 v_min3 v_med3 v_bfi v_bfe v_bitop3
               v_perm v_alignbit v_lshl_add v_add3 v_and_or v_or3 v_xad), and any plain VALU with an SGPR / vcc source
    8 cycles   transcendental (v_exp v_log v_rcp v_rsq v_sqrt v_sin v_cos): quarter rate
and prints, per loop, the instruction count by class, the pipe cycles per trip, and the wave-issue cycles per trip (one
instruction per ~4.1 cycles per wave: the other bound when only one or two waves share a SIMD).
ASM.s = `hipcc <Makefile flags> --cuda-device-only -S` of a translation unit."""
import re
import sys
from collections import Counter

THREE_OP = ("v_max3", "v_min3", "v_med3", "v_bfi", "v_bfe", "v_bitop3", "v_perm", "v_alignbit", "v_lshl_add", "v_add3", "v_and_or",
            "v_or3", "v_xad", "v_lshl_or", "v_add_lshl", "v_mad_u32", "v_mad_i32", "v_mad_u64", "v_cndmask")
TRANS = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")


def classify(line):
    op = line.split()[0]
    args = line[len(op):].split(";")[0]
    if not op.startswith("v_"):
        if op.startswith("ds_"): return "lds", 0
        if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem", 0
        if op == "s_nop": return "s_nop", 0
        if op == "s_waitcnt": return "s_waitcnt", 0
        return "salu", 0
    if op.startswith(TRANS): return "trans", 8
    if op.startswith("v_pk_"): return "packed", 4
    if op.startswith(THREE_OP): return "3-operand", 4
    if op.startswith("v_cmp") or op.startswith("v_readfirstlane") or op.startswith("v_readlane"): return "to-sgpr", 4
    if re.search(r"\bs\d+\b|\bs\[\d+:\d+\]|\bvcc\b|\bexec\b", args): return "plain+sgpr", 4
    return "plain", 2


def main():
    asm = open(sys.argv[1]).read().split("\n")
    rx = re.compile(sys.argv[2])
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    n = 0
    while n < len(asm):
        m = re.match(r"^(_Z\S+):", asm[n])
        if not (m and rx.search(m.group(1))):
            n += 1
            continue
        name = m.group(1)
        end = next(k for k in range(n, len(asm)) if "s_endpgm" in asm[k])
        print(f"== {name[:150]}")
        labels = {mm.group(1): k for k in range(n, end) if (mm := re.match(r"^(\.LBB\d+_\d+):", asm[k]))}
        for k in range(n, end):
            mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", asm[k])
            if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
                a = labels[mm.group(1)]
                ins = [x.strip() for x in asm[a + 1:k + 1] if x.strip() and not x.strip().startswith((";", ".", "//"))]
                inner = any(re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", asm[j]) and labels.get(re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", asm[j]).group(1), 1 << 60) <= j
                            and labels.get(re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", asm[j]).group(1), -1) >= a for j in range(a + 1, k))
                if len(ins) < min_n or inner:  # innermost loops only
                    continue
                cls, cyc, ops = Counter(), 0, Counter()
                for x in ins:
                    c, w = classify(x)
                    cls[c] += 1
                    cyc += w
                    if c in ("plain+sgpr", "3-operand", "to-sgpr", "trans"):
                        ops[x.split()[0] + (" [s]" if c == "plain+sgpr" else "")] += 1
                nv = sum(v for c, v in cls.items() if c in ("plain", "plain+sgpr", "packed", "3-operand", "trans", "to-sgpr"))
                print(f"  loop {mm.group(1)}: {len(ins)} instructions, {dict(cls)}")
                print(f"      VALU pipe {cyc} cycles per trip; wave issue ~{4.1 * (len(ins) - cls['s_nop'] - cls['s_waitcnt']):.0f} cycles per trip ({nv} VALU)")
                print(f"      4-cycle singles: {dict(ops.most_common(12))}")
        n = end + 1


if __name__ == "__main__":
    main()
