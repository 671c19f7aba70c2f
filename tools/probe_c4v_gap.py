"""Where the 0.2-0.3 ms between the Var-gate step's wall time and its two kernels go: the same two launches with / without the setter in between,
with / without the per-launch event pair.  Design tool."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B, fundsp_amd as F
from fundsp_amd import workloads as W
V, T, sr = 32768, 48000, 48000.0
wl = B.make_workload(F, W, torch, "4v", V, T, sr, 0, F.LAYOUT_VOICE_MINOR, "exact")
bank = wl["bank"]
def step(setter, kind):
    for k, (value, n) in enumerate(wl["plan"]):
        if setter == "all":
            bank.set_param(wl["gate_slot"], float(value))
        bank.process(n, None, wl["outs"][k])
for timing in (1, 0):
    bank.set_option("timing", timing)
    for setter in ("all", "none"):
        for _ in range(2):
            step(setter, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            step(setter, 0)
        torch.cuda.synchronize()
        print(json.dumps({"timing": timing, "setter": setter, "ms_per_step": round((time.perf_counter() - t0) / 6 * 1e3, 4)}), flush=True)
