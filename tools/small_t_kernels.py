"""tools/small_t_kernels.py -- which kernel family a short launch should take: config 3 (65 536 voices and its shards), config 2
(1 024 voices) and config 4 (32 768 voices) at T = 16 ... 512 with the single-wave kernel ("pipe_split" 0), the stage pipeline forced
("pipe_split" 2) and the library's own choice; kernel time from the per-launch event pair, median of 150 launches.
FUNDSP_HIP_LIB selects a variant library (-DFD_PIPE_MIN_T / -DFD_TS_MIN_T: the choice's thresholds)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import fundsp_amd as F
from fundsp_amd import workloads as W
SR = 48000.0
F.wavetable_build("saw")
CASES = [("config 3", W.make_fm_svf_bank, 65536, 0, 1), ("config 3", W.make_fm_svf_bank, 16384, 0, 1), ("config 3", W.make_fm_svf_bank, 8192, 0, 1),
         ("config 2", W.make_noise_biquad_bank, 1024, 0, 1), ("config 4", W.make_saw_moog_bank, 32768, 1, 2)]
PLANAR = "planar" in sys.argv[1:]   # small_t_kernels.py planar: the reference's per-voice layout [voice][channel][T] instead of voice-minor
only = [a for a in sys.argv[1:] if a != "planar"] or None
for name, make, V, ni, no in CASES:
    for T in (16, 64, 128, 192, 256, 512):
        row = []
        out = torch.empty((V, no, T) if PLANAR else (no, T, V), dtype=torch.float32, device="cuda")
        inp = torch.ones((V, ni, T) if PLANAR else (ni, T, V), dtype=torch.float32, device="cuda") if ni else None
        for label, ps in (("choice", 1), ("single-wave", 0), ("pipeline", 2)):
            if only and label not in only: continue
            bank = make(V, SR)
            bank.set_option("pipe_split", ps)
            ms = []
            for i in range(170):
                bank.process(T, inp, out, layout=F.LAYOUT_PLANAR if PLANAR else F.LAYOUT_VOICE_MINOR)
                if i >= 20: ms.append(bank.last_kernel_ms())
            row.append(f"{label}: {np.median(ms)*1e3:7.1f} us (family {bank.get_option('last_kernel')})")
            del bank
        print(f"{name} V={V:6d} T={T:4d}: " + " | ".join(row), flush=True)
