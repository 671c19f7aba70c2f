"""tools/mix_bench.py -- the fused mix-down next to the voice-out render and the unfused mix (one GPU).
Config 4 per-GPU shard (32 768 voices, MIX_SUM) and config 3 (65 536 voices, MIX_PAN; and its 2/4/8-GPU shards)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fundsp_amd as F
from fundsp_amd import workloads as W

SR, T = 48000.0, 48000


def timeit(fn, steps=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    res = {}
    F.wavetable_build("saw")
    V = 32768
    b = W.make_saw_moog_bank(V, SR)
    gate = torch.from_numpy(W.gate_signal(T, SR)).cuda()[None, :, None].expand(1, T, V).contiguous()
    out = torch.empty((2, T, V), dtype=torch.float32, device="cuda")
    mix = torch.empty((2, T), dtype=torch.float32, device="cuda")
    b.mix_reserve(T)
    r = {}
    r["voice_out_ms"] = timeit(lambda: b.process(T, gate, out))
    r["voice_out_then_sum_voices_ms"] = timeit(lambda: (b.process(T, gate, out), F.sum_voices(out)))
    r["fused_mix_ms"] = timeit(lambda: b.process_mix(T, gate, mix=F.MIX_SUM, out=mix))
    r["fused_kernel_ms_incl_tree"] = b.last_kernel_ms()
    res["config4_32768"] = {k: round(v, 4) for k, v in r.items()}
    del b, out, gate
    for V in (65536, 32768, 16384, 8192):
        b = W.make_fm_svf_bank(V, SR)
        out = torch.empty((1, T, V), dtype=torch.float32, device="cuda")
        pan = torch.zeros(V, dtype=torch.float32, device="cuda")
        b.mix_reserve(T)
        r = {}
        r["voice_out_ms"] = timeit(lambda: b.process(T, None, out))
        r["voice_out_then_mix_stereo_ms"] = timeit(lambda: (b.process(T, None, out), F.mix_stereo(out[0], pan)))
        r["fused_pan_mix_ms"] = timeit(lambda: b.process_mix(T, mix=F.MIX_PAN, out=mix))
        r["fused_sum_mix_ms"] = timeit(lambda: b.process_mix(T, mix=F.MIX_SUM, out=mix))
        r["last_kernel"] = b.get_option("last_kernel")
        res[f"config3_{V}"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}
        del b, out
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
