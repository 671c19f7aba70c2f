"""Planar layout on a heavy graph (config 4 voice: saw >> moog * adsr >> pan): planar pipeline kernel vs the single-wave
planar kernel.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fundsp_amd as F
from fundsp_amd import workloads as W
V, T, SR = 32768, 12000, 48000.0
F.wavetable_build("saw")
for ps in (1, 0):
    F.lib().fdsp_set_option(b"pipe_split", ps)
    b = W.make_saw_moog_bank(V, SR)
    gate = torch.ones((V, 1, T), dtype=torch.float32, device="cuda")
    out = torch.empty((V, 2, T), dtype=torch.float32, device="cuda")
    ms = []
    for _ in range(4):
        b.process(T, gate, out=out, layout=F.LAYOUT_PLANAR, frame_stride=T)
        torch.cuda.synchronize()
        ms.append(b.last_kernel_ms())
    print(f"planar_heavy pipe_split={ps}: {min(ms):.2f} ms  {V * T / min(ms) / 1e3:.0f} Msamples/s")
