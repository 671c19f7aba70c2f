"""Latency / throughput of the host-buffer boundary (fdsp_bank_process_host), the call a Rust AudioNode::process
shim makes once per block.  Run on the GPU box: python tools/host_bench.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fundsp_amd as F
from fundsp_amd import workloads as W
for V, T, n in ((1, 64, 1000), (64, 64, 1000), (1024, 64, 1000), (4096, 64, 500), (16384, 64, 200), (65536, 64, 50), (4096, 48000, 3)):
    b = W.make_fm_svf_bank(V, 48000.0)
    out = np.zeros((V, 1, T), dtype=np.float32)
    b.process_host(T, out=out)
    t0 = time.perf_counter()
    for _ in range(n): b.process_host(T, out=out)
    dt = (time.perf_counter() - t0) / n
    print(f"process_host V={V:6d} T={T:6d}: {dt*1e6:10.1f} us/call  {V*T/dt/1e6:9.1f} Msamples/s  {V*T*4/dt/1e9:6.2f} GB/s out")
