"""tools/shards_knock.py -- which role bounds the three-way time-split kernel: kernel ms of the 8 192- / 16 384- / 32 768-voice shards
with roles of k_render_ts3 idled (variant libraries built with -DFD_KNOCK_TS=n; FUNDSP_HIP_LIB selects one)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fundsp_amd as F
from fundsp_amd import workloads as W

T, sr = 48000, 48000.0
row = []
for V in (32768, 16384, 8192):
    b = W.make_fm_svf_bank(V, sr)
    out = torch.empty((1, T, V), dtype=torch.float32, device="cuda")
    ms = []
    for i in range(4):
        b.process(T, None, out, layout=F.LAYOUT_VOICE_MINOR, mode=F.MODE_PROCESS)
        ms.append(b.last_kernel_ms())
    row.append(f"{V}: {min(ms[1:]):.3f} ms")
    del b, out
print(os.environ.get("FUNDSP_HIP_LIB", "default").split("_")[-1], " | ".join(row), flush=True)
