"""tools/shards_bench.py -- the per-GPU shards of the 65 536-voice headline (32 768 / 16 384 / 8 192 voices x 48 000 frames) on
one GPU, per time_split option: 1 = 3 + 3 + 1 waves per voice group (round 3), 2 = round 2's 2 + 2 + 1 / 2 + 1 + 1 layouts,
0 = the plain pipeline kernel.  Prints kernel ms (HIP events) per shard size; checks that the kernels agree bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fundsp_amd as F
from fundsp_amd import workloads as W

T, sr = 48000, 48000.0
TS = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 0]   # shards_bench.py 1 = the default kernels only
for V in (32768, 16384, 8192):
    ref = None
    row = []
    for ts in TS:
        b = W.make_fm_svf_bank(V, sr)
        b.set_option("time_split", ts)
        out = torch.empty((1, T, V), dtype=torch.float32, device="cuda")
        ms = []
        for i in range(5):
            b.process(T, None, out, layout=F.LAYOUT_VOICE_MINOR, mode=F.MODE_PROCESS)
            ms.append(b.last_kernel_ms())
            if i == 0:
                first = out[0, :256].cpu().numpy().copy()
        if ref is None:
            ref = first
        same = np.array_equal(ref.view(np.uint32), first.view(np.uint32))
        row.append(f"time_split={ts}: {min(ms[1:]):.3f} ms (kernel family {b.get_option('last_kernel')}, first launch {'==' if same else '!='} option 1)")
        del b, out
    print(f"{V} voices: " + " | ".join(row), flush=True)
