"""reverb3_stereo(2.0, 0.5, lowpole_hz(8000)) (the reference's own example, prelude.rs:1850-1856) on 2 048 instances: the lane-per-frame kernel
(fdsp_reverb3_stereo_create, what Bank.from_graph builds for the stock node) against the run-time compiled lane-per-voice rendering.
624 B per instance-frame (76 ring reads + 76 ring writes + 2 in + 2 out).  Run on the GPU box: python tools/probe_reverb3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fundsp_amd as F
from fundsp_amd import graph as G

SR = 48000.0


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


mk = lambda: G.reverb3_stereo(2.0, 0.5, lambda: G.lowpole_hz(8000.0))
for V in (256, 2048, 8192):
    T = 48000
    x = torch.rand((V, 2, T), device="cuda") * 2 - 1
    fast = F.Bank.from_graph(mk(), V, sample_rate=SR)
    assert fast.kind == "reverb3_stereo"
    ms, _ = timed(fast, x, T)
    by = 624 * V * T
    print(f"reverb3_stereo V={V:5d} T={T}: lane-per-frame {ms:8.3f} ms = {by / ms / 1e6:8.1f} GB/s algorithmic ({by / ms / 1e6 / 8000:.3f} of 8 TB/s), {V * T / ms / 1e3:9.1f} M instance-frames/s", flush=True)
    if V == 2048:
        xv = x.permute(1, 2, 0).contiguous()
        ms_v, _ = timed(fast, xv, T, layout=F.LAYOUT_VOICE_MINOR)
        print(f"    voice-minor I/O (through the planar staging copy): {ms_v:8.3f} ms", flush=True)
        Ts = 2400
        slow = F.Bank.from_graph(mk(), V, ring_frames=2048, sample_rate=SR, fdn_kernel=False)
        xs = xv[:, :Ts].contiguous()
        ms2, _ = timed(slow, xs, Ts, reps=2, layout=F.LAYOUT_VOICE_MINOR)
        # (reset() does not clear the input diffusers, reverb.rs:211-224: the comparison needs banks without a past)
        slow.close()
        f2 = F.Bank.from_graph(mk(), V, sample_rate=SR)
        slow = F.Bank.from_graph(mk(), V, ring_frames=2048, sample_rate=SR, fdn_kernel=False)
        same = torch.equal(f2.process(Ts, xs).view(torch.int32), slow.process(Ts, xs).view(torch.int32))
        f2.close()
        print(f"    run-time compiled lane-per-voice, T={Ts}: {ms2:8.3f} ms = {624 * V * Ts / ms2 / 1e6:8.1f} GB/s; per frame {ms2 / Ts / (ms / T):.0f} x the lane-per-frame kernel; identical samples: {same}", flush=True)
        slow.close()
    fast.close()

# the loop filter of the reference's examples (examples/keys.rs:134): highshelf_hz(5000, 1, db_amp(-1)) -- sixteen SVF recurrences on the serial lanes
V, T = 2048, 48000
x = torch.rand((V, 2, T), device="cuda") * 2 - 1
b = F.Bank.from_graph(G.reverb3_stereo(2.0, 0.5, lambda: G.highshelf_hz(5000.0, 1.0, 10.0 ** (-1.0 / 20.0))), V, sample_rate=SR)
assert b.kind == "reverb3_stereo"
ms, _ = timed(b, x, T)
print(f"reverb3_stereo(2, 0.5, highshelf_hz(5000, 1, db_amp(-1))) V={V} T={T}: lane-per-frame {ms:8.3f} ms = {624 * V * T / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
b.close()
b = F.Bank.from_graph(G.reverb3_stereo(2.0, 0.5, lambda: G.highshelf_hz(5000.0, 1.0, 10.0 ** (-1.0 / 20.0))), 256, sample_rate=SR)
ms, _ = timed(b, x[:256].contiguous(), T)
print(f"    256 instances (one wave per instance, one per SIMD): {ms:8.3f} ms", flush=True)
