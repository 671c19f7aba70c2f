"""Config 3 (65 536 voices): kernel time by launch length, call by call (event pair) and replayed from a HIP graph -- to separate the fixed cost
of a launch from the per-block work.  With FUNDSP_HIP_LIB=variants/libfundsp_hip_c3_k3.so (both stages idle: FD_KNOCK=3, NOT a renderer)
the same launches do no arithmetic at all.  Design tool."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import fundsp_amd as F
from fundsp_amd import workloads as W

V = int(os.environ.get("V", 65536))
for T in (64, 128, 256, 512):
    b = W.make_fm_svf_bank(V, 48000.0)
    out = torch.empty((1, T, V), dtype=torch.float32, device="cuda")
    for _ in range(5):
        b.process(T, None, out)
    torch.cuda.synchronize()
    ks = []
    for _ in range(60):
        b.process(T, None, out)
        ks.append(b.last_kernel_ms())
    ks.sort()
    NB = 16
    outs = [torch.empty_like(out) for _ in range(NB)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for kk in range(NB):
                b.process(T, None, outs[kk])
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            g.replay()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 30 / NB * 1e6
    print(json.dumps({"V": V, "T": T, "last_kernel": b.get_option("last_kernel"), "event_pair_us_median": round(ks[len(ks) // 2] * 1e3, 2),
                      "hip_graph_replay_us_per_launch": round(us, 2)}), flush=True)
    del g, outs
