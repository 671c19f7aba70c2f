import sys, os, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import fundsp_amd as F
from fundsp_amd import workloads as W
V, T = 65536, 12000
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for kind in ("sine", "fixed_svf"):
    b = F.Bank(kind, V); b.set_sample_rate(48000.0)
    for layout, shape_in, shape_out in ((F.LAYOUT_VOICE_MINOR, (1, T, V), (1, T, V)), (F.LAYOUT_PLANAR, (V, 1, T), (V, 1, T))):
        inp = torch.full(shape_in, 440.0, dtype=torch.float32, device="cuda"); out = torch.empty(shape_out, dtype=torch.float32, device="cuda")
        dt = timeit(lambda: b.process(T, inp, out, layout=layout, frame_stride=T if layout else 0))
        print(kind, "layout", layout, f"{dt*1e3:.3f} ms", f"{V*T*8/dt/1e9:.0f} GB/s")
b = W.make_fm_svf_bank(V, 48000.0)
for layout, shape in ((F.LAYOUT_VOICE_MINOR, (1, T, V)), (F.LAYOUT_PLANAR, (V, 1, T))):
    out = torch.empty(shape, dtype=torch.float32, device="cuda")
    dt = timeit(lambda: b.process(T, None, out, layout=layout, frame_stride=T if layout else 0))
    print("fm_svf layout", layout, f"{dt*1e3:.3f} ms", f"{V*T/dt/1e6:.0f} Msamples/s")
