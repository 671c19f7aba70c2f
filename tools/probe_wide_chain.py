"""Probe (GPU box): the reference's 100-sine bench graph at several bank sizes through the chain of waves ("pipe_split" 2 forces it) and through
one wave per voice group ("pipe_split" 0): where the chain stops paying.  python tools/probe_wide_chain.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fundsp_amd as F
from fundsp_amd import graph as GR
import oracle as O, criterion_graphs as CG

g = CG.table(GR, O)["sine"][0]
T = CG.FRAMES
for V in (64, 256, 1024, 4096, 16384, 32768, 49152, 65536):
    b = F.Bank.from_graph(g, V, sample_rate=CG.SAMPLE_RATE)
    b.set_seed(np.arange(V, dtype=np.uint64) + 1)
    out = torch.empty((1, T, V), dtype=torch.float32, device="cuda")
    row = []
    for split in (2, 0):
        b.set_option("pipe_split", split)
        b.process(T, None, out); torch.cuda.synchronize()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter(); b.process(T, None, out); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        row.append((min(ts), b.get_option("last_kernel")))
    print(f"V={V:6d}  chain of waves {row[0][0]:8.2f} ms (kernel {row[0][1]})   one wave per voice group {row[1][0]:8.2f} ms (kernel {row[1][1]})", flush=True)
    del b, out
