"""tools/bench_extra.py -- throughput of the paths next to the headline (informational; results in profiles/).
Same geometry as the headline unless noted: 65 536 voices x 48 000 frames, voice-minor layout, process semantics."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import fundsp_amd as F
from fundsp_amd import workloads as W

SR = 48000.0


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def report(name, V, T, dt, bytes_per_sample):
    print(f"{name:58s} {V:6d} x {T:5d}  {dt*1e3:8.3f} ms  {V*T/dt/1e6:10.0f} Msamples/s  {V*T*bytes_per_sample/dt/1e9:7.0f} GB/s algorithmic")


def main():
    V, T = 65536, 48000
    out = torch.empty((1, T, V), dtype=torch.float32, device="cuda")
    inp = torch.full((1, T, V), 440.0, dtype=torch.float32, device="cuda")

    b = F.Bank("sine", V); b.set_sample_rate(SR)
    report("sine leaf, frequency input from HBM (loader wave)", V, T, timeit(lambda: b.process(T, inp, out)), 8)
    b = F.Bank("fixed_svf", V); b.set_sample_rate(SR)
    report("fixed_svf leaf, audio input from HBM (loader wave)", V, T, timeit(lambda: b.process(T, inp, out)), 8)
    b = F.Bank("rez_hz", V); b.set_sample_rate(SR)
    report("rez_hz leaf (tanh per sample), audio input from HBM", V, T, timeit(lambda: b.process(T, inp, out)), 8)
    del b

    p = W.fm_svf_params(V, SR)
    b = W.make_fm_svf_bank(V, SR, params=p)
    rng = np.random.default_rng(1)
    start = rng.random(V) * 0.5
    b.set_events(start, start + 0.1 + rng.random(V) * 0.4, 0.01, 0.05)
    def ev():
        b.events_rewind(0.0)
        b.process_events(T, None, out)
    report("fm_svf with per-voice events + fades (voice scheduler)", V, T, timeit(ev), 4)
    del b

    b = F.Bank("oversample_fm", V)
    b.set_param("0.0.0.0.0.0:value[0]", p["f"]); b.set_param("0.0.0.0:scalar", p["f"])
    b.set_param("0.0.0:scalar", p["m"]); b.set_param("0.0:scalar", p["f"])
    b.set_sample_rate(SR)
    T2 = 12000
    o2 = out[:, :T2].contiguous()
    report("oversample(sine_hz(f)*f*m+f >> sine())  (2x oversampled FM)", V, T2, timeit(lambda: b.process(T2, None, o2), 3), 4)
    del b, inp

    V4 = 16384
    i4 = torch.rand((4, T, V4), dtype=torch.float32, device="cuda")
    i4[1] = 500.0 + 3000.0 * i4[1]; i4[2] = 0.5 + i4[2]; i4[3] = 1.0 + i4[3]
    o4 = torch.empty((1, T, V4), dtype=torch.float32, device="cuda")
    b = F.Bank("svf4", V4); b.set_param(":mode", 6.0); b.set_sample_rate(SR)
    report("svf4 (bell, audio + cutoff + q + gain inputs: coefs per sample)", V4, T, timeit(lambda: b.process(T, i4, o4), 3), 20)


if __name__ == "__main__":
    main()
