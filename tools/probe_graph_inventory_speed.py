"""Probe (GPU box): every graph of the run-time compiler's inventory (tests/test_gpu_jit.py GRAPHS: leaves, combinators, composed opcodes) and a few
README idioms as banks of 16 384 instances x 1 s at 48 kHz -- a scan for kinds that render orders of magnitude below their neighbours (how the wide
sum, the reverb behind generators and the limiter's tree were found).  python tools/probe_graph_inventory_speed.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fundsp_amd as F
from fundsp_amd import graph as GR
from test_gpu_jit import GRAPHS

EXTRA = {
    "readme_busi20_noise_resonators": (lambda m: m.busi(20, lambda i: m.noise() >> m.resonator_hz(1000.0 + 50.0 * i, 20.0)), 0, 0),   # README.md:1357
    "readme_busi20_harmonics": (lambda m: m.busi(20, lambda i: m.mul(float(i + 1)) >> m.sine()), 1, 0),                                # README.md:1152
    "echo_1s": (lambda m: m.pass_() & m.feedback(m.delay(1.0) * 0.5), 1, 65536),
    "organ_lowpass": (lambda m: m.organ_hz(110.0) >> m.lowpass_hz(1000.0, 1.0), 0, 0),
    "dsf_saw": (lambda m: m.dc(110.0) >> m.dsf_saw_r(0.9), 0, 0),
}
V, T, SR = 16384, 48000, 48000.0
rows = []
for name, (build, ni, ring) in list(GRAPHS.items()) + list(EXTRA.items()):
    try:
        g = build(GR)
        for kind in GR.uses_wavetables(g):
            F.wavetable_build(kind)
        t0 = time.perf_counter()
        b = F.Bank.from_graph(g, V, ring_frames=ring, sample_rate=SR)
        tc = time.perf_counter() - t0
        b.set_seed(np.arange(V, dtype=np.uint64) + 1)
        x = None
        if ni:
            x = torch.rand((ni, T, V), dtype=torch.float32, device="cuda") * 2 - 1
            if "harmonics" in name:
                x = x.abs() * 300 + 40
        out = torch.empty((g.nout, T, V), dtype=torch.float32, device="cuda")
        b.process(T, x, out); torch.cuda.synchronize()
        t0 = time.perf_counter(); b.process(T, x, out); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        rows.append((V * T / ms / 1e3, name, ms, b.get_option("last_kernel"), tc))
        del b, out, x
    except Exception as e:
        print(f"{name:34s} FAILED {repr(e)[:200]}", flush=True)
for v, name, ms, k, tc in sorted(rows):
    print(f"{name:34s} {ms:9.2f} ms  {v:10.1f} Msamples/s  kernel {k}  (compile + create {tc:5.1f} s)", flush=True)
