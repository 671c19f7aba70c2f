"""Config 2 (1024 voices noise >> biquad): kernel time by launch length and kernel family, and the 64-frame block replayed from a HIP
graph.  Prints one JSON line per row.  Design tool."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import fundsp_amd as F
from fundsp_amd import workloads as W

V, sr = int(os.environ.get("V", 1024)), 48000.0
for name, opts in (("choice", {}), ("pipeline", {"time_split": 0, "pipe_split": 2}), ("single", {"pipe_split": 0})):
    for T in (64, 256, 4096, 48000):
        b = W.make_noise_biquad_bank(V, sr)
        for k, v in opts.items():
            b.set_option(k, v)
        out = torch.empty((1, T, V), dtype=torch.float32, device="cuda")
        for _ in range(5):
            b.process(T, None, out)
        torch.cuda.synchronize()
        n = 100 if T <= 256 else 10
        ks = []
        t0 = time.perf_counter()
        for _ in range(n):
            b.process(T, None, out)
            ks.append(b.last_kernel_ms())
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e6
        ks.sort()
        row = {"family": name, "last_kernel": b.get_option("last_kernel"), "T": T, "kernel_us_median": round(ks[len(ks) // 2] * 1e3, 2), "wall_us": round(wall, 2)}
        if T == 64:
            NB = 32
            outs = [torch.empty_like(out) for _ in range(NB)]
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    for kk in range(NB):
                        b.process(64, None, outs[kk])
                g.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    g.replay()
                torch.cuda.synchronize()
                us = (time.perf_counter() - t0) / 50 / NB * 1e6
            row["hip_graph_replay_us_per_block"] = round(us, 2)
            row["hip_graph_Msamples_s"] = round(V * 64 / us, 1)
            del g, outs
        print(json.dumps(row), flush=True)
