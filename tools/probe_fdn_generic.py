"""The documented `fdn` network (prelude.rs:1334: split >> fdn::<U16>(stacki(delay >> fir)) >> join) on 2 048 / 16 384 instances: the lane-per-frame
FDN kernel (fdsp_fdn_create, what Bank.from_graph builds for this shape) against the run-time compiled lane-per-voice rendering of the
same graph.  Run on the GPU box: python tools/probe_fdn_generic.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import fundsp_amd as F
from fundsp_amd import graph as G
from fundsp_amd import workloads as W

SR = 48000.0


def net(n, w, nin, nout):
    r = W.rnd1(np.arange(n, dtype=np.uint64))
    d = [float(np.float32(np.float32(0.01) * (np.float32(1) - np.float32(x)) + np.float32(0.03) * np.float32(x))) for x in r]
    line = G.stacki(n, lambda i: G.delay(d[i]) >> G.fir(*w))
    return (G.split(n) if nin == 1 else G.multisplit(2, n // 2)) >> G.fdn(line) >> (G.join(n) if nout == 1 else G.multijoin(2, n // 2))


def timed(b, x, T, reps=3, layout=F.LAYOUT_PLANAR):
    kw = dict(layout=layout, frame_stride=T) if layout == F.LAYOUT_PLANAR else dict(layout=layout)
    out = b.process(T, x, **kw)
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        b.process(T, x, out=out, **kw)
        torch.cuda.synchronize()
        ms.append(b.last_kernel_ms())
    return min(ms), out


for n, w, nin, nout, V in [(16, (0.2, 0.4, 0.2), 1, 1, 2048), (16, (0.2, 0.4, 0.2), 1, 1, 16384), (32, (0.2, 0.4, 0.2), 2, 2, 2048), (8, (0.5, 0.4), 2, 2, 8192), (4, (0.9,), 1, 1, 16384)]:
    T = 48000
    x = torch.rand((V, nin, T), device="cuda") * 2 - 1          # planar [instance][channel][frame]: what a lane = frame kernel reads coalesced
    fast = F.Bank.from_graph(net(n, w, nin, nout), V, sample_rate=SR)
    assert fast.kind == "fdn"
    ms, o1 = timed(fast, x, T)
    by = (8 * n + 4 * (nin + nout)) * V * T
    print(f"fdn<{n}> fir{len(w)} {nin}->{nout} V={V:6d} T={T}: lane-per-frame {ms:8.3f} ms = {by / ms / 1e6:8.1f} GB/s algorithmic ({by / ms / 1e6 / 8000:.3f} of 8 TB/s), "
          f"{V * T / ms / 1e3:10.1f} M instance-frames/s", flush=True)
    xv = x.permute(1, 2, 0).contiguous()                          # voice-minor [channel][frame][instance]
    ms_v, _ = timed(fast, xv, T, layout=F.LAYOUT_VOICE_MINOR)
    print(f"    the same bank with voice-minor I/O (through the planar staging copy: transpose in, render, transpose out): {ms_v:8.3f} ms", flush=True)
    if V <= 2048:
        Ts = 4800
        slow = F.Bank.from_graph(net(n, w, nin, nout), V, sample_rate=SR, fdn_kernel=False)
        xs = xv[:, :Ts].contiguous()
        ms2, _ = timed(slow, xs, Ts, reps=2, layout=F.LAYOUT_VOICE_MINOR)
        fast.reset()
        slow.reset()
        o1 = fast.process(Ts, xs)
        o2 = slow.process(Ts, xs)
        same = torch.equal(o1.view(torch.int32), o2.view(torch.int32))
        print(f"    run-time compiled lane-per-voice, T={Ts}: {ms2:8.3f} ms = {(8 * n + 4 * (nin + nout)) * V * Ts / ms2 / 1e6:8.1f} GB/s; per frame {ms2 / Ts / (ms / T):.0f} x the lane-per-frame kernel; identical samples: {same}", flush=True)
        slow.close()
    fast.close()
