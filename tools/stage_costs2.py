"""tools/stage_costs2.py -- knock-out experiments on the config-4 voice (run-time compiled variants of the graph, exact and
tolerance mode), 32 768 voices x 48 000 frames, seeded per voice like the bench: which stage bounds the fused pipeline
kernel.  Run-time compiled graphs always launch 4 voice groups per workgroup (the bench's ahead-of-time kind uses 2 at
this bank size).  Design tool."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import fundsp_amd as F
from fundsp_amd import graph as G
from fundsp_amd import workloads as W
SR = 48000.0
V, T = 32768, 48000
F.wavetable_build("saw")
p = W.saw_moog_params(V, SR)
gate = torch.from_numpy(W.gate_signal(T, SR)).cuda()[None, :, None].expand(1, T, V).contiguous()
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
A = (0.01, 0.1, 0.6, 0.2)
src = lambda: (G.dc(p["f"]) >> G.saw()) | G.dc(p["fc"]) | G.dc(p["q"])
cases = {
    "full: (saw|fc|q)>>moog * adsr >> pan": (lambda: (src() >> G.moog()) * G.adsr_live(*A) >> G.pan(p["pan"]), True),
    "no saw: (noise|fc|q)>>moog * adsr >> pan": (lambda: ((G.noise() | G.dc(p["fc"]) | G.dc(p["q"])) >> G.moog()) * G.adsr_live(*A) >> G.pan(p["pan"]), True),
    "no adsr: (saw|fc|q)>>moog * pass >> pan": (lambda: (src() >> G.moog()) * G.pass_() >> G.pan(p["pan"]), True),
    "no moog: (saw) * adsr >> pan": (lambda: (G.dc(p["f"]) >> G.saw()) * G.adsr_live(*A) >> G.pan(p["pan"]), True),
    "saw >> moog only": (lambda: src() >> G.moog(), False),
    "moog alone (noise|fc|q)>>moog": (lambda: (G.noise() | G.dc(p["fc"]) | G.dc(p["q"])) >> G.moog(), False),
    "saw alone": (lambda: G.dc(p["f"]) >> G.saw(), False),
}
for name, (build, has_in) in cases.items():
    row = []
    for math in (F.MATH_EXACT, F.MATH_FAST):
        b = F.Bank.from_graph(build(), V, sample_rate=SR)
        b.set_option("math", math)
        b.set_seed(p["seed"])   # per-voice hashes as in the bench: the envelopes' jittered segments end at different samples
        out = torch.empty((b.outputs(), T, V), dtype=torch.float32, device="cuda")
        row.append(timeit(lambda: b.process(T, gate if has_in else None, out)) * 1e3)
        del b, out
    print(f"{name:45s} exact {row[0]:8.2f} ms   fast {row[1]:8.2f} ms", flush=True)
