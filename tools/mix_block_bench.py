"""tools/mix_block_bench.py -- one 64-frame block per launch (the real-time pattern): voice-out render vs render + mix-down in the launch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fundsp_amd as F
from fundsp_amd import workloads as W
SR = 48000.0
def t_us(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for name, make, V in (("config3", W.make_fm_svf_bank, 65536), ("config3", W.make_fm_svf_bank, 8192), ("config2", W.make_noise_biquad_bank, 1024)):
    b = make(V, SR)
    b.set_option("timing", 0)
    out = torch.empty((1, 64, V), dtype=torch.float32, device="cuda")
    mix = torch.empty((2, 64), dtype=torch.float32, device="cuda")
    b.mix_reserve(64)
    r = t_us(lambda: b.process(64, None, out))
    lk1 = b.get_option("last_kernel")
    m = t_us(lambda: b.process_mix(64, mix=F.MIX_PAN, out=mix))
    lk2 = b.get_option("last_kernel")
    u = t_us(lambda: (b.process(64, None, out), F.mix_stereo(out[0])))
    print(f"{name} V={V}: voice-out {r:.1f} us/block (kernel family {lk1}); fused mix {m:.1f} us/block (family {lk2}); voice-out + mix_stereo {u:.1f} us/block", flush=True)
