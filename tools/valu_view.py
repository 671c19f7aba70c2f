#!/usr/bin/env python3
"""tools/valu_view.py TAG -- the VALU view of the headline kernel ("report VALU utilisation too"; VERDICT r02 Next 4).

Inputs, both committed under profiles/:
  * profiles/<TAG>_pmc_sq_c3.txt          SQ counters of `python bench.py` (rocprofv3 --pmc pass, tools/pmc_summary.py)
  * the kernel's ISA: fd_kinds_fm.hip compiled here with the Makefile's flags (--cuda-device-only -S); the two hot loops
    (stage 0 = modulator, stage 1 = carrier + lowpass-specialised SVF; one 8-frame SIMD item per trip) are written to
    profiles/<TAG>_isa_fm_svf.txt with their instruction histograms.
Output: profiles/valu_latest.json, which bench.py quotes as roofline.valu (with its source, like roofline.traffic)."""
import json
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
VOICES, FRAMES = 65536, 48000
KERNEL = "k_render_pipe"

# ---- the ISA --------------------------------------------------------------------------------------------------------
mk = open(os.path.join(ROOT, "fundsp_amd", "csrc", "Makefile")).read()
flags = re.search(r"^FLAGS\s*=\s*(.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split()
flags = [f for f in flags if f != "-fPIC"] + ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
asm = subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["--cuda-device-only", "-S", "-o", "-", os.path.join(ROOT, "fundsp_amd", "csrc", "fd_kinds_fm.hip")],
                     capture_output=True, text=True, cwd=os.path.join(ROOT, "fundsp_amd", "csrc")).stdout
# the process-mode, 2-stage, 4-groups-per-workgroup pipeline kernel of FmSvf (exact arithmetic: the type without SineFast)
start = None
for m in re.finditer(r"^(_ZN2fd13k_render_pipeI\S+):", asm, re.M):
    name = m.group(1)
    if "SineFast" not in name and "Unop" in name and name.endswith("Li0ELi2ELi1ELi3ELi4EEEvPfmmPKfS9_mPKvS9_j") or ("SineFast" not in name and "Unop" in name and "ELi0ELi2ELi1ELi3ELi4EEE" in name):
        start = m.start()
        break
assert start is not None, "headline kernel not found in the ISA"
body = asm[start:asm.find(".Lfunc_end", start)].split("\n")  # (the whole function: a kernel may hold more than one s_endpgm)
labels = {}
for n, l in enumerate(body):
    mm = re.match(r"^(\.LBB\d+_\d+):", l)
    if mm:
        labels[mm.group(1)] = n
loops = []
for n, l in enumerate(body):
    mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
        a = labels[mm.group(1)]
        ins = [x.strip() for x in body[a + 1:n] if x.strip() and not x.strip().startswith((";", "."))]
        ops = [x.split()[0] for x in ins]
        pk = sum(o.startswith("v_pk") for o in ops)
        if pk >= 100 and len(ops) < 400:
            loops.append((a, n, ins, ops))
# the lowpass-specialised twin's loops: the one with packed SVF ops (most packed) is stage 1, the other stage 0
def flops(ops):
    f = 0
    for o in ops:
        if o.startswith(("v_pk_mul_f32", "v_pk_add_f32")): f += 2
        elif o.startswith("v_pk_fma_f32"): f += 4
        elif o.startswith(("v_mul_f32", "v_add_f32", "v_sub_f32")): f += 1
        elif o.startswith(("v_fma_f32", "v_fmac_f32")): f += 2
    return f
stage1 = max(loops, key=lambda L: sum(o.startswith("v_pk") for o in L[3]))
stage0 = min((L for L in loops if not any(o.startswith("buffer_store") or o.startswith("global_store") for o in L[3])), key=lambda L: len(L[3]))
out = [f"# {TAG}: hot loops of fd::k_render_pipe<fm_svf, process, 2 stages, 4 voice groups per workgroup> (gfx950), one 8-frame SIMD item per trip;",
       f"# compiled from fundsp_amd/csrc/fd_kinds_fm.hip with the Makefile's flags (tools/valu_view.py).  Lowpass-specialised twin (LpOf<G>)."]
summary = {}
for nm, L in (("stage 0: sine_hz(f) * f * m + f  (modulator sine, packed)", stage0), ("stage 1: >> sine() >> lowpass_hz  (carrier sine packed, SVF state equations packed, buffer stores)", stage1)):
    ops = L[3]
    valu = [o for o in ops if o.startswith("v_")]
    pk = [o for o in valu if o.startswith("v_pk")]
    summary[nm[:7]] = dict(valu=len(valu), packed=len(pk), flops=flops(ops), s_nop=ops.count("s_nop"))
    out.append(f"\n## {nm}\n# per 8 frames: {len(ops)} instructions, {len(valu)} VALU ({len(pk)} packed, {len(valu) - len(pk)} plain) = {len(valu) / 8:.1f} issue slots per frame, "
               f"{(len(valu) + len(pk)) / 8:.1f} plain-op equivalents per frame, {flops(ops) / 8:.1f} flops per voice-frame, {ops.count('s_nop')} s_nop")
    out.append("# " + ", ".join(f"{k} x{v}" for k, v in Counter(ops).most_common()))
    out += ["\t" + x for x in L[2]]
open(os.path.join(ROOT, "profiles", f"{TAG}_isa_fm_svf.txt"), "w").write("\n".join(out) + "\n")

# ---- the counters ---------------------------------------------------------------------------------------------------
pmc_path = os.path.join(ROOT, "profiles", f"{TAG}_pmc_sq_c3.txt")
txt = open(pmc_path).read()
blk = txt[txt.find("k_render_pipe"):]
blk = blk[:blk.find("\nkernel:", 10)]
ctr = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\w+)\s+dispatches=\s*\d+ mean=([\d.e+]+)", blk, re.M)}
groups = VOICES // 64
v0, v1 = summary["stage 0"], summary["stage 1"]
insts_static = (v0["valu"] + v1["valu"]) / 8
packed_frac = (v0["packed"] + v1["packed"]) / (v0["valu"] + v1["valu"])
insts = ctr["SQ_INSTS_VALU"] / (groups * FRAMES)
cycles_per_simd = ctr["GRBM_GUI_ACTIVE"] / 8          # the counter sums the 8 XCDs
kernel_ms = None
cyc_per_frame = cycles_per_simd / FRAMES
equiv = insts * (1 + packed_frac)
flops_vf = (v0["flops"] + v1["flops"]) / 8
rec = {
    "kernel": "fd::k_render_pipe<fm_svf, process, 2 stages, 4 groups per workgroup>, 65536 voices x 48000 frames",
    "insts_per_voice_group_frame": round(insts, 2),
    "insts_per_voice_group_frame_isa": round(insts_static, 2),
    "packed_fraction_isa": round(packed_frac, 3),
    "plain_op_equivalents_per_voice_group_frame": round(equiv, 1),
    "cycles_per_frame_per_simd": round(cyc_per_frame, 1),
    "issue_cycles_per_inst": round(cyc_per_frame / insts, 2),
    "frac_of_issue_peak": round(equiv * 2 / cyc_per_frame, 3),
    "issue_peak": "one plain wave64 VALU instruction per 2 cycles per SIMD, one packed-f32 per 4 (profiles/r03_ubench_issue_v2.txt: two waves of v_mul_f32 2.04 cycles per instruction per SIMD, four of v_pk_fma_f32 4.27)",
    "flops_per_voice_frame_isa": round(flops_vf, 1),
    "wait_inst_any_frac_of_wave_cycles": round(ctr["SQ_WAIT_INST_ANY"] / ctr["SQ_WAVE_CYCLES"], 3) if "SQ_WAIT_INST_ANY" in ctr else None,
    "wait_any_frac_of_wave_cycles": round(ctr["SQ_WAIT_ANY"] / ctr["SQ_WAVE_CYCLES"], 3) if "SQ_WAIT_ANY" in ctr else None,
    "voices": VOICES, "frames": FRAMES, "config": 3, "math": "exact",
    "source": f"profiles/{TAG}_pmc_sq_c3.txt (separate rocprofv3 --pmc pass of `python bench.py`: SQ_INSTS_VALU, GRBM_GUI_ACTIVE / 8 XCDs) + profiles/{TAG}_isa_fm_svf.txt (static instruction mix)",
}
json.dump(rec, open(os.path.join(ROOT, "profiles", "valu_latest.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
