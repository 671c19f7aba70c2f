"""Probe (GPU box): an additive voice with its gain, filter and panner -- sumi(N, |i| sine_hz(f (i + 1))) * g >> lowpass_hz(fc, q) >> pan(p) -- rendered
with the sum branch-major and the rest of the graph as its frame-major tail (fd_device.hpp WideSplit), next to the bare sum (what the tail costs) and to
the frame-major rendering such a graph got before (the same voice with the sum written as two half sums added together, a shape WideSplit does not
take: every branch's state in registers across the frame loop).  python tools/probe_wide_tail.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fundsp_amd as F
from fundsp_amd import graph as G

SR, T = 44100.0, 44100


def timed(g, V):
    b = F.Bank.from_graph(g, V, sample_rate=SR)
    b.set_seed(np.arange(V, dtype=np.uint64) + 1)
    out = torch.empty((b.outputs(), T, V), dtype=torch.float32, device="cuda")
    b.process(T, None, out); torch.cuda.synchronize()
    ts = []
    for _ in range(2):
        t0 = time.perf_counter(); b.process(T, None, out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), b.get_option("last_kernel")


for N in (32, 64):
    voice = lambda: G.sumi(N, lambda i: G.sine_hz(55.0 * (i + 1))) * (1.0 / N) >> G.lowpass_hz(2000.0, 1.0) >> G.pan(0.2)
    bare = lambda: G.sumi(N, lambda i: G.sine_hz(55.0 * (i + 1)))
    halves = lambda: (G.sumi(N // 2, lambda i: G.sine_hz(55.0 * (i + 1))) + G.sumi(N // 2, lambda i: G.sine_hz(55.0 * (i + 1 + N // 2)))) * (1.0 / N) >> G.lowpass_hz(2000.0, 1.0) >> G.pan(0.2)
    for V in (1024, 16384):
        a, ka = timed(voice(), V)
        b, kb = timed(bare(), V)
        c, kc = (timed(halves(), V) if N == 32 else (float("nan"), 0))
        print(f"{N} partials, {V:6d} instances x {T} frames: voice (sum >> gain >> lowpass >> pan) {a:8.2f} ms (kernel {ka})   bare sum {b:8.2f} ms (kernel {kb})   "
              f"frame-major proxy {c:8.2f} ms (kernel {kc})", flush=True)
