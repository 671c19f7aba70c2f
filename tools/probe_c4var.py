"""Config 4: the stream-gate kind (gate = HBM input [frames][voices]) against the Var-gate kind (`var(gate) >> adsr_live`, no input),
per-GPU shard 32 768 voices x 48 000 frames; voice-out and mode B.  Prints one JSON line per measurement."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
import fundsp_amd as F
from fundsp_amd import workloads as W

V, T, sr = int(os.environ.get("V", 32768)), 48000, 48000.0
mode = F.MODE_PROCESS
only = os.environ.get("ONLY")
for math in ("exact", "fast"):
    if only and math != "exact":
        continue
    for cfg in ("4v", 4):
        if only and str(cfg) != only:
            continue
        wl = B.make_workload(F, W, torch, cfg, V, T, sr, 0, F.LAYOUT_VOICE_MINOR, math)
        ms, kms = B.quick(F, torch, wl, T, mode, steps=4, warmup=1)
        row = {"cfg": cfg, "math": math, "V": V, "voice_out_ms": round(ms, 4), "kernel_ms": round(kms, 4)}
        bank = wl["bank"]
        wl["out"] = None
        wl["outs"] = None
        torch.cuda.empty_cache()
        if cfg == "4v":
            bank.mix_reserve(max(n for _, n in wl["plan"]))
            mixbufs = [torch.empty((2, n), dtype=torch.float32, device="cuda") for _, n in wl["plan"]]
            B.run_plan(wl, mode, F.MIX_SUM, mixbufs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                B.run_plan(wl, mode, F.MIX_SUM, mixbufs)
            torch.cuda.synchronize()
            row["mode_b_ms"] = round((time.perf_counter() - t0) / 4 * 1e3, 4)
        else:
            bank.mix_reserve(T)
            mixbuf = torch.empty((2, T), dtype=torch.float32, device="cuda")
            bank.process_mix(T, wl["inp"], mix=F.MIX_SUM, out=mixbuf, mode=mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                bank.process_mix(T, wl["inp"], mix=F.MIX_SUM, out=mixbuf, mode=mode)
            torch.cuda.synchronize()
            row["mode_b_ms"] = round((time.perf_counter() - t0) / 4 * 1e3, 4)
        print(json.dumps(row), flush=True)
        del wl, bank
        torch.cuda.empty_cache()
