"""tools/small_t.py -- config 3 at the other launch lengths of SURVEY 8(d): T = 64, 4096 (48 000 is the headline)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import fundsp_amd as F
from fundsp_amd import workloads as W
V = 65536
bank = W.make_fm_svf_bank(V, 48000.0)
for T, n in ((64, 2000), (4096, 100), (48000, 10)):
    out = torch.empty((1, T, V), dtype=torch.float32, device="cuda")
    for _ in range(3): bank.process(T, None, out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): bank.process(T, None, out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"T={T:6d}: {dt*1e6:9.1f} us per launch  {V*T/dt/1e6:10.0f} Msamples/s  kernel {bank.last_kernel_ms()*1e3:8.1f} us  real-time factor {T/48000/dt:8.1f}x")
