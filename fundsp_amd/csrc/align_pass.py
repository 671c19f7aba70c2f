#!/usr/bin/env python3
"""align_pass.py IN.s OUT.s -- encoding-alignment pass over the device assembly of one translation unit (gfx950).

Why (profiles/r03_ubench_issue_v3.txt, profiles/r03_looplab.txt): a wave of this chip issues an 8-byte-encoded instruction
that sits at an address = 4 (mod 8) one cycle later than an aligned one (4.08 -> 5.08 cycles per instruction for a stream of
v_pk_fma_f32).  The voice kernels are packed-f32 / VOP3 streams that are VALU-issue bound, and whether a loop's long runs of
8-byte instructions start aligned is an accident of the 4-byte instructions in front of them: stage 0 of the headline kernel ran
13 % faster in the loop lab with ONE s_nop in front of its packed block.  The compiler has no pass for this, so the build runs
this one between `hipcc -S` and the assembler (Makefile: the *.aligned.s rule).

What it does: the instruction stream is a sequence of 4-, 8- and 12-byte encodings (sizes from `llvm-mc -show-encoding`); a
dynamic programme over (position, address parity) picks, per instruction, one of
    keep                                   cost 1 if it is an 8-byte encoding at 4 (mod 8)
    widen a VOP1/VOP2 `_e32` to `_e64`     the same operation in its 8-byte VOP3 encoding (no literal, no implicit vcc): moves
                                           the parity of everything behind it for free where it lands aligned itself
    put `s_nop 0` in front                 cost NOP_COST issue cycles
minimising the total.  The instructions, their order and their operands are untouched -- only encodings and padding change, so
results are bit-identical by construction (and the GPU parity suite runs on the padded library).  `.p2align >= 3` resets the
parity (kernels start 256-byte aligned).  Prints one summary line.
"""
import re
import subprocess
import sys

LLVM = "/opt/rocm/lib/llvm/bin"
NOP_COST = 4.0      # issue cycles of one s_nop 0 (loop lab: six of them cost stage 1 ~22 cycles per trip)
WIDEN_COST = 0.01   # prefer doing nothing where it is a tie
LOOP_W = 8          # an instruction inside d nested loops weighs LOOP_W ^ d (d capped at 4)
INSN = re.compile(r"^\s+([a-z][a-z0-9_]+)(\s|$)")
# VOP1 / VOP2 operations whose _e64 form is the same operation (no carry-in / carry-out, no implicit vcc, no DPP/SDWA)
WIDENABLE = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_max_f32", "v_min_f32", "v_and_b32", "v_or_b32", "v_xor_b32",
             "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_max_u32",
             "v_min_u32", "v_max_i32", "v_min_i32", "v_mul_u32_u24", "v_mul_i32_i24", "v_cvt_f32_i32", "v_cvt_f32_u32",
             "v_cvt_i32_f32", "v_cvt_u32_f32", "v_floor_f32", "v_fract_f32", "v_trunc_f32", "v_rndne_f32", "v_rcp_f32", "v_sqrt_f32"}


def enc_size(line):
    """bytes of one `-show-encoding` line (literal bytes that wait for a fixup print as A)"""
    return len(line.split("; encoding: [", 1)[1].split("]")[0].split(","))


def mc_sizes(text):
    """encoded size in bytes of every instruction of an assembly text, in order"""
    r = subprocess.run([LLVM + "/llvm-mc", "-triple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-show-encoding"], input=text, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("align_pass: llvm-mc failed:\n" + r.stderr[-2000:])
    return [enc_size(l) for l in r.stdout.split("\n") if "; encoding: [" in l]


def main(src, dst):
    lines = open(src).read().split("\n")
    idx = [n for n, l in enumerate(lines) if INSN.match(l) and not l.lstrip().startswith((".", ";", "//"))]
    sizes = mc_sizes("\n".join(lines))
    if len(sizes) != len(idx):
        raise SystemExit(f"align_pass: {len(idx)} instruction lines but {len(sizes)} encodings")
    # which 4-byte instructions can be widened: try them all in one assembler run
    cand = []
    for k, n in enumerate(idx):
        m = INSN.match(lines[n])
        op = m.group(1)
        if sizes[k] == 4 and op.endswith("_e32") and op[:-4] in WIDENABLE and "vcc" not in lines[n]:
            cand.append(k)
    wide = {}
    if cand:
        txt = "\n".join(re.sub(r"_e32\b", "_e64", lines[idx[k]].split(";")[0], count=1) for k in cand)
        r = subprocess.run([LLVM + "/llvm-mc", "-triple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-show-encoding"], input=txt, capture_output=True, text=True)
        out = [l for l in r.stdout.split("\n") if "; encoding: [" in l]
        if r.returncode == 0 and len(out) == len(cand):
            for k, l in zip(cand, out):
                if enc_size(l) == 8:
                    wide[k] = True
        # (an assembler error on any candidate: no widening at all -- padding alone still works)
    # segments between parity resets: a `.p2align >= 3` between two instructions starts a new one
    is_reset = [bool((m := re.match(r"^\s+\.p2align\s+(\d+)", l)) and int(m.group(1)) >= 3) for l in lines]
    seg_start = [0]
    for k in range(1, len(idx)):
        if any(is_reset[idx[k - 1] + 1:idx[k]]):
            seg_start.append(k)
    seg_start.append(len(idx))
    action = [0] * len(idx)  # 0 keep, 1 widen, 2 nop in front, 3 nop in front + widen
    # Dynamic weight of an instruction: LOOP_W ^ (number of loops around it), a loop = the span from a label to a branch back to it.
    # (Without it the pass spent its padding where the static count was, and a hot loop could come out worse than it went in:
    # config 4 ran 9.7 ms aligned against 9.45 ms plain once its tiles had doubled, profiles/r03_ab17_stage_loops.txt.)
    import bisect
    label_ix = {}
    for n, l in enumerate(lines):
        m = re.match(r"^(\.L[A-Za-z0-9_$.]+):", l)
        if m:
            label_ix[m.group(1)] = bisect.bisect_left(idx, n)
    diff = [0] * (len(idx) + 1)
    for k, n in enumerate(idx):
        op = lines[n].split()
        if op and (op[0].startswith("s_cbranch") or op[0] == "s_branch") and len(op) > 1 and op[1] in label_ix and label_ix[op[1]] <= k:
            diff[label_ix[op[1]]] += 1
            diff[k + 1] -= 1
    weight, d = [1.0] * len(idx), 0
    for k in range(len(idx)):
        d += diff[k]
        weight[k] = float(LOOP_W ** min(d, 4))
    before = after = 0.0
    INF = float("inf")
    for a, b in zip(seg_start[:-1], seg_start[1:]):
        n = b - a
        if n <= 0:
            continue
        # cost[i][p]: best cost of instructions i.. given parity p in front of instruction i; filled backwards
        cost = [[0.0, 0.0] for _ in range(n + 1)]
        choice = [[0, 0] for _ in range(n)]
        for i in range(n - 1, -1, -1):
            k = a + i
            w = sizes[k] // 4
            for p in (0, 1):
                best, bc = INF, 0
                for act in (0, 1, 2, 3):
                    if (act & 1) and k not in wide:
                        continue
                    c = 0.0
                    q = p
                    if act & 2:
                        c += NOP_COST * weight[k]
                        q ^= 1
                    ww = 2 if (act & 1) else w
                    if act & 1:
                        c += WIDEN_COST * weight[k]
                    if ww == 2 and q == 1:
                        c += 1.0 * weight[k]
                    c += cost[i + 1][q ^ (ww & 1)]
                    if c < best:
                        best, bc = c, act
                cost[i][p] = best
                choice[i][p] = bc
        p = 0
        for i in range(n):
            k = a + i
            act = choice[i][p]
            action[k] = act
            w = sizes[k] // 4
            q = p ^ (1 if act & 2 else 0)
            ww = 2 if (act & 1) else w
            if ww == 2 and q == 1:
                after += 1
            p = q ^ (ww & 1)
        # the untouched layout, for the summary
        p = 0
        for i in range(n):
            w = sizes[a + i] // 4
            if w == 2 and p == 1:
                before += 1
            p ^= (w & 1)
    # Branches are SIMM16 dword offsets and the compiler has already relaxed the ones that did not fit: a function whose
    # longest branch would leave the range once padded keeps its original layout
    label_at = {}
    for n, l in enumerate(lines):
        m = re.match(r"^(\.L[A-Za-z0-9_$.]+):", l)
        if m:
            label_at[m.group(1)] = bisect.bisect_left(idx, n)  # the instruction that follows the label
    reverted = 0
    for a, b in zip(seg_start[:-1], seg_start[1:]):
        off = [0] * (b - a + 1)
        for i in range(a, b):
            act = action[i]
            off[i - a + 1] = off[i - a] + (4 if act & 2 else 0) + (8 if act & 1 else sizes[i])
        far = False
        for i in range(a, b):
            op = lines[idx[i]].split()
            if op and (op[0].startswith("s_cbranch") or op[0] == "s_branch") and len(op) > 1 and op[1] in label_at:
                tgt = label_at[op[1]]
                if a <= tgt <= b:
                    # offset of the target (in front of its own s_nop, if any: conservative) against the end of the branch
                    if abs(off[tgt - a] - off[i - a + 1]) > 4 * 32000:
                        far = True
                        break
        if far:
            reverted += 1
            for i in range(a, b):
                action[i] = 0
    nops = widened = 0
    out = list(lines)
    for k, n in enumerate(idx):
        act = action[k]
        l = out[n]
        if act & 1:
            l = re.sub(r"_e32\b", "_e64", l, count=1)
            widened += 1
        if act & 2:
            l = "\ts_nop 0\n" + l
            nops += 1
        out[n] = l
    open(dst, "w").write("\n".join(out))
    n8 = sum(1 for s in sizes if s == 8)
    print(f"align_pass: {len(idx)} instructions, {n8} of them 8-byte; at 4 (mod 8): {int(before)} -> {int(after)}; {widened} widened to _e64, {nops} s_nop" + (f"; {reverted} function(s) with branches near the SIMM16 range left as they were" if reverted else ""))


if __name__ == "__main__":
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    main(sys.argv[1], sys.argv[2])
