"""Probe (GPU box): the reference's `oversample` bench graph (benches/benchmark.rs:62: noise() >> oversample(pass())) on 65 536 instances, one rendered
second at 44.1 kHz -- the number the Oversampler's process walk is judged by.  python tools/probe_oversample_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fundsp_amd as F
from fundsp_amd import graph as G
V, T = 65536, 44100
b = F.Bank.from_graph(G.noise() >> G.oversample(G.pass_()), V, sample_rate=44100.0)
b.set_seed(np.arange(V, dtype=np.uint64) + 1)
out = torch.empty((1, T, V), dtype=torch.float32, device="cuda")
b.process(T, None, out); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); b.process(T, None, out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"noise() >> oversample(pass()), {V} instances x {T} frames: {min(ts):.2f} ms (kernel {b.get_option('last_kernel')})")
