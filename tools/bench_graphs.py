"""Throughput of run-time compiled graphs beyond the BASELINE configs (informational).
Run on the GPU box: python tools/bench_graphs.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fundsp_amd as F
from fundsp_amd import graph as G

SR = 48000.0
CASES = [
    # name, graph, voices, frames, ring_frames, bytes of ring traffic per voice-sample (read + write)
    ("busi 8 sines", lambda: G.busi(8, lambda i: G.sine_hz(110.0 * (i + 1))) * 0.125, 65536, 12000, 0, 0),
    ("pulse wave", lambda: (G.sine_hz(3.0) * 50.0 + 220.0 | G.dc(0.3)) >> G.pulse() * 0.2, 32768, 12000, 0, 0),
    ("flanger", lambda: G.noise() >> G.flanger(0.6, 0.002, 0.006, "EnvSineHz", hz=0.7, lo=0.002, hi=0.006), 32768, 12000, 512, 24),
    ("phaser", lambda: G.noise() >> G.phaser(0.5, "EnvSineHz", hz=0.7, lo=0.0, hi=1.0), 32768, 12000, 0, 0),
    ("limiter_stereo 2 ms", lambda: (G.noise() | G.noise()) >> G.limiter_stereo(0.002, 0.02), 32768, 12000, 256, 0),
    ("allpass chain, 6 rings", lambda: G.noise() >> G.pipei(6, lambda i: G.allnest_c(0.6, G.delay(0.003 + 0.0007 * i))), 32768, 12000, 512, 48),
    ("echo feedback(delay)", lambda: G.noise() >> G.feedback(G.delay(0.005) * 0.5), 32768, 12000, 512, 8),
    ("reverb4_stereo", lambda: (G.noise() | G.noise()) >> G.reverb4_stereo(20.0, 2.0), 2048, 12000, 8192, 256),
    ("reverb3_stereo", lambda: (G.noise() | G.noise()) >> G.reverb3_stereo(2.0, 0.6, lambda: G.lowpole_hz(1600.0)), 2048, 12000, 2048, 608),
]
for name, make, V, T, ring, rb in CASES:
    g = make()
    b = F.Bank.from_graph(g, V, ring_frames=ring, sample_rate=SR)
    b.set_seed(np.arange(V, dtype=np.uint64))
    out = b.process(T)
    torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        b.process(T, out=out)
        torch.cuda.synchronize()
        ms.append(b.last_kernel_ms())
    m = min(ms)
    alg = (4 * g.nout + rb) * V * T / (m * 1e-3) / 1e9
    print(f"graphs {name:22s} V={V:6d} T={T:6d}: {m:8.2f} ms  {V * T / m / 1e3:10.1f} Msamples/s  {alg:8.1f} GB/s algorithmic")
    b.close()
