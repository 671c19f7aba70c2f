"""A run-time compiled FM voice (the README's example graph) on the strong-scaling shard sizes: kernel time with and without the time-split kernels."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fundsp_amd as F
from fundsp_amd import graph as GR, workloads as W
for V in (8192, 16384, 32768):
    p = W.fm_svf_params(V, 48000.0)
    g = GR.sine_hz(p["f"]) * p["f"] * p["m"] + p["f"] >> GR.sine() >> GR.lowpass_hz(p["fc"], p["q"])
    row = {"V": V}
    for split in (1, 0):
        b = F.Bank.from_graph(g, V, sample_rate=48000.0)
        b.set_seed(p["seed"])
        b.set_option("time_split", split)
        out = torch.empty((1, 48000, V), dtype=torch.float32, device="cuda")
        for _ in range(2):
            b.process(48000, None, out)
        ks = []
        for _ in range(4):
            b.process(48000, None, out)
            ks.append(b.last_kernel_ms())
        row["time_split" if split else "pipeline"] = {"kernel_ms": round(sum(ks) / len(ks), 4), "last_kernel": b.get_option("last_kernel")}
    print(json.dumps(row), flush=True)
