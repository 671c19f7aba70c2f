#!/usr/bin/env python3
"""tools/ubench_issue_check.py -- build-host check of tools/ubench_issue.hip's device code (VERDICT r02 Next 3i: "disassembly
checked for s_nop / s_waitcnt"): every asm kind's timed block must be >= 512 consecutive instructions of the intended
mnemonics with NO s_nop / s_waitcnt / other instruction in between, and operand kinds (SGPR, SGPR pair, literal, vcc) must be
what the kind's name says.  Also prints the VALU instruction counts of the C++ streams' loops (arguments of the binary)."""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                      "-Wno-unused-value", "-I", os.path.join(ROOT, "fundsp_amd", "csrc"), "--cuda-device-only", "-S", "-o", "-",
                      os.path.join(ROOT, "tools", "ubench_issue.hip")], capture_output=True, text=True).stdout
src = open(os.path.join(ROOT, "tools", "ubench_issue.hip")).read()
m = re.search(r"enum Kind \{(.*?)NKINDS", src, re.S)
names = [x.strip().split()[0] for x in re.sub(r"//.*", "", m.group(1)).replace("\n", " ").split(",") if x.strip()]
expect = {  # kind -> (regex every instruction of the block must match, instructions per block)
    "FMA_S_K8": r"v_fma_f32 v\d+, v\d+, s\d+, v\d+", "MUL_S_K8": r"v_mul_f32_e32 v\d+, s\d+, v\d+", "MUL_LIT_K8": r"v_mul_f32_e32 v\d+, 0x3f000001, v\d+",
    "ADD_INL_K8": r"v_add_f32_e32 v\d+, 0\.5, v\d+", "ADD_E64_K8": r"v_add_f32_e64 v\d+, v\d+, v\d+", "CND_E64_K8": r"v_cndmask_b32_e64 v\d+, v\d+, v\d+, s\[",
    "CMP_K8": r"v_cmp_lt_f32_e32 vcc", "CMP_E64_K8": r"v_cmp_lt_f32_e64 s\[", "CMPCND_K8": r"v_cmp_lt_f32_e32 vcc|v_cndmask_b32_e32 v\d+, v\d+, v\d+, vcc",
    "PKMUL_S_K8": r"v_pk_mul_f32 v\[[\d:]+\], v\[[\d:]+\], s\[", "PKFMA_S_K8": r"v_pk_fma_f32 v\[[\d:]+\], v\[[\d:]+\], s\[[\d:]+\], v\[",
    "BITOP3_S_K8": r"v_bitop3_b32 v\d+, v\d+, v\d+, s\d+", "CND_K8": r"v_cndmask_b32_e32 v\d+, v\d+, v\d+, vcc", "FMA_D1": r"v_fma_f32 v\d+, v\d+, v\d+, v\d+",
    "PKFMA_K8": r"v_pk_fma_f32 v\[", "MUL_K8": r"v_mul_f32_e32 v\d+, v\d+, v\d+", "MAX3_K8": r"v_max3_f32", "BFE_K8": r"v_bfe_i32", "MAXABS_K8": r"v_max_f32_e64 v\d+, \|v\d+\|, \|v\d+\|",
}
bad = 0
for name, rx in expect.items():
    k = names.index(name)
    sym = f"_Z1kILi{k}ELi{k}EEvP3Reciffii:"
    i = asm.find("\n" + sym)
    if i < 0:
        print(f"{name}: kernel not found")
        bad += 1
        continue
    body = asm[i:asm.find("s_endpgm", i)]
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", ".", "_Z"))]
    best = run = 0
    for l in lines:
        if re.match(rx, l):
            run += 1
            best = max(best, run)
        else:
            run = 0
    ok = best >= 512
    bad += not ok
    print(f"{name:12s} longest uninterrupted run of the intended instruction(s): {best} {'ok' if ok else 'TOO SHORT (hazard nops / other instructions inside the block?)'}")
# the C++ streams: VALU instructions of their inner loops
for name in ("SINE4", "SVF8", "SINE4_SVF8"):
    k, idle = names.index(name), names.index("IDLE")
    sym = f"_Z1kILi{k}ELi{idle}EEvP3Reciffii:"
    i = asm.find("\n" + sym)
    body = asm[i:asm.find("s_endpgm", i)].split("\n")
    labels = {}
    for n, l in enumerate(body):
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            labels[mm.group(1)] = n
    best = (0, 0, 0)
    for n, l in enumerate(body):
        mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
            ins = [x.split()[0] for x in body[labels[mm.group(1)] + 1:n] if x.strip() and not x.strip().startswith((";", "."))]
            valu = [x for x in ins if x.startswith("v_")]
            if len(valu) > best[0]:
                best = (len(valu), sum(x.startswith("v_pk") for x in valu), ins.count("s_nop"))
    print(f"{name:12s} loop: {best[0]} VALU instructions ({best[1]} packed), {best[2]} s_nop")
sys.exit(1 if bad else 0)
