#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes: 3 launches of the bench kernel (config 3, 65536 x 48000) plus two
calibration kernels with a KNOWN byte count on the same 12.58 GB buffer (a fill = pure write, a copy = read+write),
so FETCH_SIZE / WRITE_SIZE can be calibrated as MI355X_MICROARCH.md prescribes.
Run under:  rocprofv3 --pmc FETCH_SIZE --output-format csv ...   and again with  --pmc WRITE_SIZE
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import fundsp_amd as F
from fundsp_amd import workloads as W

V, T = 65536, 48000
bank = W.make_fm_svf_bank(V, 48000.0)
out = torch.empty((1, T, V), dtype=torch.float32, device="cuda")
for _ in range(3):
    bank.process(T, None, out)
torch.cuda.synchronize()
out.fill_(1.0)          # known: V*T*4 bytes written
torch.cuda.synchronize()
dst = torch.empty_like(out)
dst.copy_(out)          # known: V*T*4 read + V*T*4 written
torch.cuda.synchronize()
print("done", V * T * 4)
