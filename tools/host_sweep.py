"""fdsp_bank_process_host: per-call latency of a 64-frame block, staged through HBM vs zero-copy (the option
"host_zero_copy_max" picks the cross-over).  Run on the GPU box: python tools/host_sweep.py"""
import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import fundsp_amd as F
from fundsp_amd import workloads as W
from fundsp_amd._lib import lib
L = lib()
def run(kind, V, T, layout, n=400):
    if kind == "fm":
        b = W.make_fm_svf_bank(V, 48000.0); x = None
    else:
        b = F.Bank("fixed_svf", V); b.set_sample_rate(48000.0)
        x = np.random.rand(*((1, T, V) if layout == 0 else (V, 1, T))).astype(np.float32)
    out = np.zeros((1, T, V) if layout == 0 else (V, 1, T), dtype=np.float32)
    b.process_host(T, x, layout=layout, out=out)
    t0 = time.perf_counter()
    for _ in range(n): b.process_host(T, x, layout=layout, out=out)
    return (time.perf_counter() - t0) / n * 1e6
for kind in ("fm", "svf"):
  for layout in (1, 0):
    for V in (1, 16, 64, 256, 1024, 4096, 16384):
        r = []
        for zc in (0, 1 << 30):   # staged through HBM (pageable copies) vs kernel reading / writing pinned host memory
            assert L.fdsp_set_option(b"host_zero_copy_max", zc) == 0
            r.append(run(kind, V, 64, layout, 300 if V < 10000 else 100))
        print(f"hb2 {kind} layout={layout} V={V:6d}: staged {r[0]:8.1f}  zerocopy {r[1]:8.1f} us")
