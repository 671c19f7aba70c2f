#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes: 3 launches of the bench kernel (config 3, 65536 x 48000) plus two
calibration kernels with a KNOWN byte count on the same 12.58 GB buffer (a fill = pure write, a copy = read+write),
so FETCH_SIZE / WRITE_SIZE can be calibrated as MI355X_MICROARCH.md prescribes.
Run under:  rocprofv3 --pmc FETCH_SIZE --output-format csv ...   and again with  --pmc WRITE_SIZE
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import fundsp_amd as F
from fundsp_amd import workloads as W

WHICH = os.environ.get("TRAFFIC_WORKLOAD", "c3")   # c3 (default: the headline kernel) | c2 | c4v | c5 | c5r4 | fdn16 | rv3: the other HBM-side kernels
T = 48000
if WHICH == "c3":
    V = 65536
    bank = W.make_fm_svf_bank(V, 48000.0)
    out = torch.empty((1, T, V), dtype=torch.float32, device="cuda")
    for _ in range(3):
        bank.process(T, None, out)
elif WHICH == "c2":        # noise >> biquad on a bank that fills the chip: 4 B per voice-sample out
    V = 65536
    bank = W.make_noise_biquad_bank(V, 48000.0)
    out = torch.empty((1, T, V), dtype=torch.float32, device="cuda")
    for _ in range(3):
        bank.process(T, None, out)
elif WHICH == "c4v":       # config 4, the gate a Var slot: 8 B per voice-sample out, nothing in; one launch per gate value
    V = 32768
    F.wavetable_build("saw")
    bank = W.make_saw_moog_var_bank(V, 48000.0)
    out = torch.empty((2, T // 2, V), dtype=torch.float32, device="cuda")
    for _ in range(3):
        for gate in (1.0, 0.0):
            bank.set_param(W.C4V_SLOTS["gate"], gate)
            bank.process(T // 2, None, out)
elif WHICH == "fdn16":     # the prelude's fdn example through the generic lane-per-frame kernel: 136 B per instance-frame (16 rings r/w + 1 in + 1 out)
    import numpy as np
    from fundsp_amd import graph as G

    V = 4096
    r = W.rnd1(np.arange(16, dtype=np.uint64))
    d = [float(np.float32(np.float32(0.01) * (np.float32(1) - np.float32(x)) + np.float32(0.03) * np.float32(x))) for x in r]
    bank = F.Bank.from_graph(G.split(16) >> G.fdn(G.stacki(16, lambda i: G.delay(d[i]) >> G.fir(0.2, 0.4, 0.2))) >> G.join(16), V, sample_rate=48000.0)
    assert bank.kind == "fdn"
    inp = torch.rand((V, 1, T), dtype=torch.float32, device="cuda") * 2 - 1
    outp = torch.empty((V, 1, T), dtype=torch.float32, device="cuda")
    for _ in range(3):
        bank.process(T, inp, outp, layout=F.LAYOUT_PLANAR, frame_stride=T)
elif WHICH == "rv3":       # reverb3_stereo(2, 0.5, lowpole_hz(8000)): 624 B per instance-frame (76 rings r/w + 2 in + 2 out)
    from fundsp_amd import graph as G

    V = 2048
    bank = F.Bank.from_graph(G.reverb3_stereo(2.0, 0.5, lambda: G.lowpole_hz(8000.0)), V, sample_rate=48000.0)
    assert bank.kind == "reverb3_stereo"
    inp = torch.rand((V, 2, T), dtype=torch.float32, device="cuda") * 2 - 1
    outp = torch.empty((V, 2, T), dtype=torch.float32, device="cuda")
    for _ in range(3):
        bank.process(T, inp, outp, layout=F.LAYOUT_PLANAR, frame_stride=T)
else:                      # c5: reverb_stereo(10, 2, 0.5); c5r4: reverb4_stereo(20, 2): 272 B per instance-frame (rings + I/O)
    V = 2048
    bank = F.Bank.reverb_stereo(V, 10.0, 2.0, 0.5) if WHICH == "c5" else F.Bank.reverb4_stereo(V, 20.0, 2.0)
    bank.set_sample_rate(48000.0)
    inp = torch.rand((V, 2, T), dtype=torch.float32, device="cuda") * 2 - 1
    outp = torch.empty((V, 2, T), dtype=torch.float32, device="cuda")
    for _ in range(3):
        bank.process(T, inp, outp, layout=F.LAYOUT_PLANAR, frame_stride=T)
torch.cuda.synchronize()
if WHICH != "c3":          # the calibration kernels below keep their 12.58 GB buffer whatever was rendered
    V = 65536
    out = torch.empty((1, T, V), dtype=torch.float32, device="cuda")
out.fill_(1.0)          # known: V*T*4 bytes written
torch.cuda.synchronize()
dst = torch.empty_like(out)
dst.copy_(out)          # known: V*T*4 read + V*T*4 written
torch.cuda.synchronize()
print("done", V * T * 4)
