#!/usr/bin/env python3
"""tools/looplab/make_lab.py OUTDIR -- the loop lab (design tool; results in profiles/r03_looplab.txt).

Takes the two hot loops of the headline kernel exactly as the compiler emits them (fd_kinds_fm.hip's process-mode, two-stage
pipeline kernel of fm_svf, lowpass-specialised twin: stage 0 = role A, stage 1 = role B), splices them into the compiled
assembly of lab_template.hip and assembles one code object per VARIANT of the bodies:
    base          the loops as they are
    nostore       role B without its 8 buffer stores
    nosalu        role B without its scalar ALU instructions
    nolds         both roles without their LDS traffic (no hand-over)
    align         s_nop padding in front of runs of misaligned 8-byte instructions (both roles)
    align_b       the same, role B only
    vconst        packed instructions read their constants from VGPR pairs instead of SGPR pairs
tools/looplab/_run_lab times role A alone, role B alone, both, both with role B at s_setprio 1 (s_memtime per wave)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "looplab", "_out")
os.makedirs(OUT, exist_ok=True)
LLVM = "/opt/rocm/lib/llvm/bin"
CSRC = os.path.join(ROOT, "fundsp_amd", "csrc")

def sh(cmd, **kw):
    return subprocess.run(cmd, check=True, capture_output=True, text=True, **kw).stdout

mk = open(os.path.join(CSRC, "Makefile")).read()
flags = re.search(r"^FLAGS\s*=\s*(.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split()
flags = [f for f in flags if f != "-fPIC"] + ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
asm = sh(["/opt/rocm/bin/hipcc"] + flags + ["--cuda-device-only", "-S", "-o", "-", os.path.join(CSRC, "fd_kinds_fm.hip")], cwd=CSRC)
start = None
for m in re.finditer(r"^(_ZN2fd13k_render_pipeI\S+):", asm, re.M):
    if "SineFast" not in m.group(1) and "Unop" in m.group(1) and "ELi0ELi2ELi1ELi3ELi4EEE" in m.group(1):
        start = m.start()
        break
body = asm[start:asm.find("s_endpgm", start)].split("\n")
labels = {re.match(r"^(\.LBB\d+_\d+):", l).group(1): n for n, l in enumerate(body) if re.match(r"^(\.LBB\d+_\d+):", l)}
loops = []
for n, l in enumerate(body):
    mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
        a = labels[mm.group(1)]
        ins = [x.split(";")[0].rstrip() for x in body[a + 1:n] if x.strip() and not x.strip().startswith((";", "."))]
        ins = [x for x in ins if x.strip()]
        if sum(x.split()[0].startswith("v_pk") for x in ins) >= 100 and len(ins) < 400:
            loops.append(ins)
B = max(loops, key=lambda L: sum(x.split()[0].startswith("v_pk") for x in L))           # stage 1 (packed SVF: most packed)
A = min((L for L in loops if not any("buffer_store" in x for x in L)), key=len)          # stage 0

def sizes(ins):
    txt = "\n".join(ins) + "\n"
    out = subprocess.run([LLVM + "/llvm-mc", "-arch=amdgcn", "-mcpu=gfx950", "-show-encoding"], input=txt, capture_output=True, text=True).stdout
    sz = [len(l.split("encoding: [", 1)[1].split("]")[0].split(",")) for l in out.split("\n") if "encoding: [" in l]
    assert len(sz) == len(ins), (len(sz), len(ins))
    return sz

def used(ins):
    v, s = set(), set()
    for l in ins:
        for m in re.finditer(r"\bv(\d+)\b", l): v.add(int(m.group(1)))
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]", l): v.update(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r"\bs(\d+)\b", l): s.add(int(m.group(1)))
        for m in re.finditer(r"\bs\[(\d+):(\d+)\]", l): s.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return v, s

def pad_align(ins, min_run=3):
    """s_nop 0 (4 bytes) in front of a run of >= min_run consecutive 8-byte instructions that would start at 4 (mod 8)"""
    sz = sizes(ins)
    out, off, i = [], 0, 0
    while i < len(ins):
        if sz[i] == 8 and off % 8 == 4:
            j = i
            while j < len(ins) and sz[j] == 8: j += 1
            if j - i >= min_run:
                out.append("\ts_nop 0")
                off += 4
        out.append(ins[i]); off += sz[i]; i += 1
    return out

def vconst(ins):
    """SGPR-pair constants of packed instructions -> VGPR pairs v[100+..] (op_sel_hi kept: the pair's low half is broadcast)"""
    mp = {}
    out = []
    for l in ins:
        if l.split()[0].startswith("v_pk"):
            def rep(m):
                key = m.group(0)
                if key not in mp: mp[key] = 100 + 2 * len(mp)
                return f"v[{mp[key]}:{mp[key] + 1}]"
            l = re.sub(r"s\[\d+:\d+\]", rep, l)
        out.append(l)
    return out

VARIANTS = {
    "base": (A, B),
    "nostore": (A, [x for x in B if "buffer_store" not in x]),
    "nosalu": (A, [x for x in B if not x.split()[0].startswith(("s_add", "s_mul", "s_cmp", "s_cselect", "s_mov", "s_lshl", "s_and"))]),
    "nolds": ([x for x in A if not x.split()[0].startswith("ds_")], [x for x in B if not x.split()[0].startswith(("ds_", "s_waitcnt"))]),
    "align": (pad_align(A), pad_align(B)),
    "align_a": (pad_align(A), B),
    "align_a2": (pad_align(A, 2), B),
    "align_a1": (pad_align(A, 1), B),
    "align_a2_nostore": (pad_align(A, 2), [x for x in B if "buffer_store" not in x]),
    "align_a6": (pad_align(A, 6), B),
    "align_a12": (pad_align(A, 12), B),
    "align_ab6": (pad_align(A, 6), pad_align(B, 6)),
    "align_ab12": (pad_align(A, 12), pad_align(B, 12)),
    "align_b": (A, pad_align(B)),
    "align_b2": (A, pad_align(B, 2)),
    "vconst": (vconst(A), vconst(B)),
}
def prio_tail(ins, frac, hi=1, lo=0):
    """priority mix: the body runs at s_setprio hi for its first frac instructions, at lo for the rest (each s_setprio with an
    s_nop 0 so that the parity of every 8-byte run behind it is unchanged)"""
    k = int(len(ins) * frac)
    return [f"\ts_setprio {hi}", "\ts_nop 0"] + ins[:k] + [f"\ts_setprio {lo}", "\ts_nop 0"] + ins[k:]
for f in (95, 90, 85, 80, 70, 60, 50):
    VARIANTS[f"bmix{f}"] = (A, prio_tail(B, f / 100))           # role B: priority 1 for the first f %, 0 (A is older: A wins) after
    VARIANTS[f"bmix{f}_al"] = (pad_align(A), prio_tail(B, f / 100))
for f in (35, 50, 65, 80):
    VARIANTS[f"amix{f}"] = (prio_tail(A, f / 100, 2, 0), B)      # role A: priority 2 (above B's 1) for its first f %
    VARIANTS[f"bhead{f}"] = (A, prio_tail(B, 1 - f / 100, 0, 1))  # role B: priority 0 first, 1 for the last f %
def template(*defs):
    return sh(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", "-", *defs, os.path.join(ROOT, "tools", "looplab", "lab_template.hip")])
tmpl = template()
# carriers of other shapes: four waves per SIMD (16-wave workgroups, 32-frame tiles: the same LDS), and the unsplit voice
# (every wave runs A then B) at two and four waves per SIMD
SHAPES = {"": tmpl, "_w16": template("-DLAB_WAVES=16", "-DLAB_TILE=4"), "_merged": template("-DLAB_MERGED=1"),
          "_merged_w16": template("-DLAB_MERGED=1", "-DLAB_WAVES=16"), "_aab": template("-DLAB_WAVES=12", "-DLAB_A_SPLIT=2")}
va, sa = used(A); vb, sb = used(B)
init = ["\tv_mov_b32 v%d, 0.5" % r for r in sorted(va | vb | set(range(100, 120)))]
# LDS addresses (the registers the ds_ instructions use) inside the allocation; null buffer resource
for l in A + B:
    if l.split()[0].startswith("ds_"):
        m = re.search(r"ds_(?:read|write)\S*\s+(?:v\[\d+:\d+\],\s*)?(v\d+)", l)
        # address register: first operand of a write, second of a read
        ops = [o.strip() for o in l.split(None, 1)[1].split(",")]
        addr = ops[0] if "write" in l.split()[0] else ops[1]
        addr = addr.split()[0]
        init.append(f"\tv_and_b32 {addr}, 0x3ff8, {addr}" if False else f"\tv_mov_b32 {addr}, 0x400")
init += ["\ts_mov_b32 s%d, 0" % r for r in sorted(sa | sb) if r < 100]
# (a null resource -- all words zero -- drops every store; scalar offsets are then irrelevant)
JOBS = [(name, "", ra, rb) for name, (ra, rb) in VARIANTS.items()]
for shape in ("_w16", "_merged", "_merged_w16", "_aab"):
    JOBS += [("base", shape, A, B), ("align_a", shape, pad_align(A), B)]
for name, shape, ra, rb in JOBS:
    name += shape
    s = SHAPES[shape].replace("\t; LAB_INIT", "\n".join(init))
    hdr = "\t.p2align 3\n"
    s = s.replace("\t; LAB_BODY_A", hdr + "\n".join(ra)).replace("\t; LAB_BODY_B", hdr + "\n".join(rb))
    p = os.path.join(OUT, f"lab_{name}")
    open(p + ".s", "w").write(s)
    subprocess.run([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", p + ".s", "-o", p + ".o"], check=True)
    subprocess.run([LLVM + "/ld.lld", "-shared", p + ".o", "-o", p + ".hsaco"], check=True)
    na, nb = len(ra), len(rb)
    print(f"{name}: role A {na} instructions ({sum(sizes(ra))} bytes), role B {nb} instructions ({sum(sizes(rb))} bytes) -> {p}.hsaco")
