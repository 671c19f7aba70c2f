#!/usr/bin/env python3
"""tools/prio_pass.py IN.s OUT.s [a=FRAC] [b=FRAC] -- design tool (A/B of static priority mixes in the real kernel, the loop lab's
amix / bmix variants): in every two-stage pipeline kernel of the assembly, the producer's item loop (packed, no buffer_store) gets
`s_setprio 2` at its top and `s_setprio 0` after FRAC of its instructions (a=), the consumer's item loop (packed, buffer_store)
`s_setprio 1` at its top and `s_setprio 0` after FRAC (b=).  Each s_setprio travels with an s_nop 0 (encoding parity unchanged)."""
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
opt = dict(x.split("=") for x in sys.argv[3:])
fa = float(opt["a"]) if "a" in opt else None
fb = float(opt["b"]) if "b" in opt else None
lines = open(src).read().split("\n")
INSN = re.compile(r"^\s+([a-z][a-z0-9_]+)(\s|$)")
label_at = {}
for n, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: label_at[m.group(1)] = n
func_at = [(n, l.split(":")[0]) for n, l in enumerate(lines) if re.match(r"^_Z\S+:", l)]
def func_of(n):
    f = ""
    for k, name in func_at:
        if k <= n: f = name
        else: break
    return f
ins_after = {}  # line index -> list of lines to insert AFTER it
done = []
for n, l in enumerate(lines):
    mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
    if not (mm and mm.group(1) in label_at and label_at[mm.group(1)] < n): continue
    a = label_at[mm.group(1)]
    if "k_render_pipe" not in func_of(n): continue
    idx = [k for k in range(a + 1, n) if INSN.match(lines[k])]
    if any(re.match(r"^\.LBB", lines[k]) for k in range(a + 1, n)): continue  # innermost, single block only
    pk = sum(lines[k].lstrip().startswith("v_pk") for k in idx)
    if pk < 100 or len(idx) > 400: continue
    cons = any(re.search(r"(buffer|global|flat)_store", lines[k]) for k in idx)
    f = fb if cons else fa
    if f is None: continue
    hi = 1 if cons else 2
    ins_after.setdefault(a, []).extend([f"\ts_setprio {hi}", "\ts_nop 0"])
    ins_after.setdefault(idx[int(len(idx) * f) - 1], []).extend(["\ts_setprio 0", "\ts_nop 0"])
    done.append(("consumer" if cons else "producer", len(idx), pk))
out = []
for n, l in enumerate(lines):
    out.append(l)
    out.extend(ins_after.get(n, []))
open(dst, "w").write("\n".join(out))
print(f"prio_pass: {len(done)} loops: {done}")
