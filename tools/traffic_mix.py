#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes of the fused mix-down ("WRITE_SIZE of the launch ~ the partials"):
BASELINE config 4's per-GPU shard (32768 voices x 48000 frames): 2 voice-out renders, 2 renders with the fused mix-down
(fdsp_bank_process_mix: k_render_pipe_mix + k_mix_tree), then two calibration kernels with a KNOWN byte count on the 12.58 GB
voice-out buffer (a fill = pure write, a copy = read + write), so FETCH_SIZE / WRITE_SIZE can be calibrated as MI355X_MICROARCH.md
prescribes.  Run under:  rocprofv3 --pmc FETCH_SIZE --output-format csv ...   and again with  --pmc WRITE_SIZE
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import fundsp_amd as F
from fundsp_amd import workloads as W

V, T, SR = 32768, 48000, 48000.0
F.wavetable_build("saw")
bank = W.make_saw_moog_bank(V, SR)
gate = torch.from_numpy(W.gate_signal(T, SR)).cuda()[None, :, None].expand(1, T, V).contiguous()
out = torch.empty((2, T, V), dtype=torch.float32, device="cuda")
mix = torch.empty((2, T), dtype=torch.float32, device="cuda")
bank.mix_reserve(T)
for _ in range(2):
    bank.process(T, gate, out)
torch.cuda.synchronize()
for _ in range(2):
    bank.process_mix(T, gate, mix=F.MIX_SUM, out=mix)
torch.cuda.synchronize()
out.fill_(1.0)          # known: 2*V*T*4 bytes written
torch.cuda.synchronize()
dst = torch.empty_like(out)
dst.copy_(out)          # known: 2*V*T*4 read + 2*V*T*4 written
torch.cuda.synchronize()
print("done", 2 * V * T * 4, "partials", (V // 64) * 2 * T * 4, "gate", V * T * 4)
