"""Probe (GPU box): README.md:436's `multipass() & 0.2 * reverb_stereo(20.0, 2.0, 1.0)` on 2 048 instances x 48 000 frames (BASELINE config 5's size) --
the reverb's lane-per-frame bank alone, the same bank with the gain and dry bus folded into its epilogue (fdsp_bank_set_bus: what Bank.from_graph
builds for the graph), and the whole graph compiled at run time (one lane per instance, its Bus / Unop / MultiPass nodes as device code) on a
tenth of the frames.  python tools/probe_reverb_bus.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import fundsp_amd as F
from fundsp_amd import graph as G

V, T, SR = 2048, 48000, 48000.0


def whole(node):
    return G.multipass(2) & 0.2 * node


def timed(bank, frames, x, out, n=3):
    bank.process(frames, x, out, layout=F.LAYOUT_PLANAR, frame_stride=frames); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); bank.process(frames, x, out, layout=F.LAYOUT_PLANAR, frame_stride=frames); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts)


x = (torch.rand((V, 2, T), device="cuda") * 2 - 1).contiguous()
out = torch.empty((V, 2, T), device="cuda")
for name, node, bytes_per in (("reverb_stereo(20, 2, 1)", lambda: G.reverb_stereo(20.0, 2.0, 1.0), 272),
                              ("reverb3_stereo(2, 0.5, lowpole_hz(8000))", lambda: G.reverb3_stereo(2.0, 0.5, lambda: G.lowpole_hz(8000.0)), 624)):
    plain = F.Bank.from_graph(node(), V, sample_rate=SR)
    bus = F.Bank.from_graph(whole(node()), V, sample_rate=SR)
    assert bus.kind == plain.kind and bus.get_bus()[0] == F.BUS_DRY_WET
    a, b = timed(plain, T, x, out), timed(bus, T, x, out)
    gb = V * T * bytes_per / 1e9
    print(f"{name}: the node alone {a:7.2f} ms ({gb / a:5.2f} TB/s)   multipass() & 0.2 * node, folded {b:7.2f} ms ({gb / b:5.2f} TB/s, same algorithmic bytes)", flush=True)
    t10 = T // 10
    jit = F.Bank.from_graph(whole(node()), V, sample_rate=SR, fdn_kernel=False, ring_frames=8192 if "reverb3" in name else 0)
    c = timed(jit, t10, x[:, :, :t10].contiguous(), out[:, :, :t10].contiguous(), n=1)
    print(f"    the same graph compiled at run time (lane per instance), {t10} frames: {c:8.2f} ms = {c / t10 / (b / T):6.1f} x per frame", flush=True)
    del plain, bus, jit
