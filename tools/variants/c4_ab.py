"""tools/c4_ab.py -- config 4 per-GPU shard (32 768 voices x 48 000 frames), same box: stage_split 0 / 1, voice-out and fused mix.
FUNDSP_HIP_LIB selects a variant library (tools/build_variants.sh).  --check compares the voice-out bits of the two settings."""
import sys, os, time, json, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fundsp_amd as F
from fundsp_amd import workloads as W

ap = argparse.ArgumentParser()
ap.add_argument("--splits", default="0,1")
ap.add_argument("--maths", default="0")
ap.add_argument("--reps", type=int, default=1)
ap.add_argument("--check", action="store_true")
ap.add_argument("--mix", action="store_true")
ap.add_argument("--uniform", action="store_true", help="every voice the same frequency and seed: all lanes of a wave gather from the same cache lines")
ap.add_argument("--fmin", type=float, default=0.0, help="raise every voice's frequency to at least this (short tables only)")
ap.add_argument("--label", default=os.path.basename(os.environ.get("FUNDSP_HIP_LIB", "default")))
a = ap.parse_args()
SR, T, V = 48000.0, 48000, 32768
F.wavetable_build("saw")
gate = torch.from_numpy(W.gate_signal(T, SR)).cuda()[None, :, None].expand(1, T, V).contiguous()
out = torch.empty((2, T, V), dtype=torch.float32, device="cuda")
mix = torch.empty((2, T), dtype=torch.float32, device="cuda")
ref = None
for rep in range(a.reps):
    for math in [int(x) for x in a.maths.split(",")]:
        for split in [int(x) for x in a.splits.split(",")]:
            p = W.saw_moog_params(V, SR)
            if a.uniform:
                p["f"][:] = 220.0
                p["seed"][:] = 7
            if a.fmin > 0:
                p["f"] = np.maximum(p["f"], np.float32(a.fmin))
            b = W.make_saw_moog_bank(V, SR, params=p)
            b.set_option("math", math)
            b.set_option("stage_split", split)
            b.mix_reserve(T)
            for _ in range(2):
                b.process(T, gate, out)
            torch.cuda.synchronize()
            k = []
            for _ in range(5):
                b.process(T, gate, out)
                k.append(b.last_kernel_ms())
            line = f"{a.label}: math={math} stage_split={split}: voice-out kernel {sum(k)/len(k):.4f} ms (min {min(k):.4f})"
            if a.mix:
                km = []
                for _ in range(4):
                    b.process_mix(T, gate, mix=F.MIX_SUM, out=mix)
                    km.append(b.last_kernel_ms())
                line += f"; fused mix {sum(km[1:])/len(km[1:]):.4f} ms"
            print(line, flush=True)
            if a.check and rep == 0 and math == F.MATH_EXACT:
                b2 = W.make_saw_moog_bank(V, SR)
                b2.set_option("stage_split", split)
                o = b2.process(T, gate, out)
                h = torch.sum(o.view(torch.int32).to(torch.int64)).item()
                if ref is None:
                    ref = h
                print(f"  checksum of the voice-out bits (fresh bank): {h} {'== first' if h == ref else '!= first  MISMATCH'}", flush=True)
                del b2
            del b
