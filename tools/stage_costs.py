"""tools/stage_costs.py -- time the three stages of the config-4 voice separately (run-time compiled sub-graphs),
32 768 voices x 48 000 frames, to see which stage bounds the pipeline.  Design tool."""
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
cases = {
    "saw stack (dc(f)>>saw | dc(fc) | dc(q))": (lambda: (G.dc(p["f"]) >> G.saw()) | G.dc(p["fc"]) | G.dc(p["q"]), False),
    "moog (noise | dc(fc) | dc(q)) >> moog()": (lambda: (G.noise() | G.dc(p["fc"]) | G.dc(p["q"])) >> G.moog(), False),
    "tail noise * adsr_live >> pan": (lambda: G.noise() * G.adsr_live(0.01, 0.1, 0.6, 0.2) >> G.pan(p["pan"]), True),
    "adsr_live alone": (lambda: G.adsr_live(0.01, 0.1, 0.6, 0.2), True),
}
for name, (build, has_in) in cases.items():
    b = F.Bank.from_graph(build(), V, sample_rate=SR)
    out = torch.empty((b.outputs(), T, V), dtype=torch.float32, device="cuda")
    dt = timeit(lambda: b.process(T, gate if has_in else None, out))
    print(f"{name:45s} {dt*1e3:8.2f} ms")
