"""tools/store_bound.py -- how fast can the [frame][voice] output stream be WRITTEN at all?  (design input)
Times (a) the render kernel of a near-free graph (noise: ~10 VALU/sample) in the headline geometry, (b) torch fill_
and zero_ (hipMemset) of the same 12.6 GB, (c) a copy.  Prints GB/s of each."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import fundsp_amd as F

V, T = 65536, 48000
out = torch.empty((1, T, V), dtype=torch.float32, device="cuda")
nbytes = out.numel() * 4


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for kind in ("noise", "sine"):
    b = F.Bank(kind, V)
    b.set_sample_rate(48000.0)
    inp = None
    if b.inputs():
        inp = torch.full((1, T, V), 440.0, dtype=torch.float32, device="cuda")
    for mode, mname in ((F.MODE_PROCESS, "process"),):
        dt = timeit(lambda: b.process(T, inp, out, layout=F.LAYOUT_VOICE_MINOR, mode=mode))
        print(f"render {kind:6s} {mname}: {dt*1e3:.3f} ms  write {nbytes/dt/1e9:.0f} GB/s  kernel {b.last_kernel_ms():.3f} ms")
    del b, inp
dt = timeit(lambda: out.fill_(1.0)); print(f"fill_   : {dt*1e3:.3f} ms  {nbytes/dt/1e9:.0f} GB/s")
dt = timeit(lambda: out.zero_()); print(f"zero_   : {dt*1e3:.3f} ms  {nbytes/dt/1e9:.0f} GB/s")
src = torch.empty_like(out)
dt = timeit(lambda: out.copy_(src)); print(f"copy_   : {dt*1e3:.3f} ms  write {nbytes/dt/1e9:.0f} GB/s (+ same read)")
