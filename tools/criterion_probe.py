"""Probe (GPU box): the reference's criterion bench graphs (tests/criterion_graphs.py) as banks -- compile time, parity of a few
instances against the oracle, ms per rendered second.  python tools/criterion_probe.py [names..]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fundsp_amd as F
from fundsp_amd import graph as GR
import oracle as O, criterion_graphs as CG

names = sys.argv[1:] or list(CG.table(O, O))
V = int(os.environ.get("V", "4096"))
for name in names:
    g, ring, line = CG.table(GR, O)[name]
    for kind in GR.uses_wavetables(g):
        F.wavetable_build(kind)
    t0 = time.perf_counter()
    try:
        b = F.Bank.from_graph(g, V, ring_frames=ring, sample_rate=CG.SAMPLE_RATE)
    except Exception as e:
        print(name, "COMPILE/CREATE FAILED", repr(e)[:400], flush=True); continue
    tc = time.perf_counter() - t0
    seeds = np.arange(V, dtype=np.uint64) * 7919 + 13
    b.set_seed(seeds)
    T = CG.FRAMES
    out = b.process(T, None, layout=F.LAYOUT_VOICE_MINOR, mode=F.MODE_PROCESS)
    torch.cuda.synchronize()
    got = out.cpu().numpy()          # [nout][T][V]
    bad = 0
    for v in (0, 1, V // 2, V - 1):
        n = CG.table(O, O)[name][0]
        n.set_sample_rate(CG.SAMPLE_RATE); n.set_seed(int(seeds[v]))
        want = n.render_blocks(None, length=T, block=64)
        d = (got[:, :, v].view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got[:, :, v]) & np.isnan(want))
        bad += int(d.sum())
    ms = []
    for _ in range(3):
        b.reset(); b.set_seed(seeds)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b.process(T, None, out, layout=F.LAYOUT_VOICE_MINOR, mode=F.MODE_PROCESS)
        torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    print(f"{name:11s} V={V} kind={b.kind[:40]} compile {tc:6.1f} s  last_kernel {b.get_option('last_kernel')}  mismatching samples {bad}  ms per rendered second {min(ms):9.3f}"
          f"  = {V * T / min(ms) / 1e3:10.1f} Msamples/s = {V / (min(ms) * 1e-3):10.0f} x real time", flush=True)
    del b, out
