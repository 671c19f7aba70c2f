#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X voice-bank engine.

Metric (BASELINE.json): Msamples/s (whole node) for the 65 536-voice SVF+FM graph
    sine_hz(f) * f * m + f >> sine() >> lowpass_hz(fc, q)           (BASELINE config 3)
One step = one pass of the hot path: render FRAMES samples of every voice of the rank's bank into an
HBM-resident [frame][voice] f32 buffer (voice-out mode A, AudioNode::process semantics, 64-sample blocks).
Multi-GPU: voices shard as contiguous ranges, one process per GPU, no data-path collective (weak scaling:
65 536 voices per GPU); `--mix` adds the on-device stereo mix-down + one RCCL all-reduce per step.

Prints ONE JSON line on rank 0 (contract in the round brief) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def cpu_baseline(voices_per_gpu, frames, sample_rate, target_seconds):
    """Time the CPU oracle ("port" of the reference's process() path) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle as O
    from fundsp_amd import workloads as W

    cores = os.cpu_count() or 1
    # calibrate on a small sample, then size the timed sample for ~target_seconds of wall time
    p = W.fm_svf_params(2 * cores, sample_rate)
    _, s = O.bank_render(3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], frames, sample_rate, True, 0, cores, store=False)
    rate = 2 * cores * frames / max(s, 1e-6)
    n = int(min(voices_per_gpu, max(cores, rate * target_seconds / frames)))
    n = max(cores, n // cores * cores)
    p = W.fm_svf_params(n, sample_rate)
    out = np.zeros((n, frames), dtype=np.float32)
    job_params = [p["f"], p["m"], p["fc"], p["q"]]
    _, s = O.bank_render(3, job_params, p["seed"], frames, sample_rate, True, 0, cores, store=False)
    del out
    return {
        "value": round(n * frames / s / 1e6, 3),
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n} of the {voices_per_gpu} config-3 voices x {frames} frames, oracle process() path "
                  f"(C restatement, gcc -O2 -ffp-contract=off), {cores} threads, {s:.2f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, choices=[3, 4, 5],
                    help="3 = headline FM+SVF voices (default); 4 = saw>>moog*adsr>>pan voices (informational)")
    ap.add_argument("--voices", type=int, default=None, help="voices per GPU (weak scaling); default 65536 (config 3) / 32768 (config 4)")
    ap.add_argument("--frames", type=int, default=48000, help="frames per step (1 s @ 48 kHz)")
    ap.add_argument("--sample-rate", type=float, default=48000.0)
    ap.add_argument("--layout", choices=["voice_minor", "planar"], default="voice_minor")
    ap.add_argument("--mode", choices=["process", "tick"], default="process")
    ap.add_argument("--mix", action="store_true", help="add on-device stereo mix-down + all-reduce per step")
    ap.add_argument("--pipe-split", type=int, default=1, choices=[0, 1, 2, 3],
                    help="pipeline split of Pipe-chain kinds: 0 off, 1 best plan (default), 2 / 3 = that many stages")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="wall-time budget of the cpu_baseline leg (0 = skip)")
    args = ap.parse_args()

    import torch

    import fundsp_amd as F
    from fundsp_amd import dist as fdist
    from fundsp_amd import workloads as W

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback for the product path)"
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    if args.pipe_split != 1:
        from fundsp_amd import _lib
        assert _lib.lib().fdsp_set_option(b"pipe_split", args.pipe_split) == 0
    if args.voices is None:
        args.voices = {3: 65536, 4: 32768, 5: 2048}[args.config]
    V, T, sr = args.voices, args.frames, args.sample_rate
    layout = F.LAYOUT_VOICE_MINOR if args.layout == "voice_minor" else F.LAYOUT_PLANAR
    mode = F.MODE_PROCESS if args.mode == "process" else F.MODE_TICK
    first = rank * V  # contiguous voice ranges of the N*V-voice whole-node bank
    inp = None
    if args.config == 3:
        bank = W.make_fm_svf_bank(V, sr, voice0=first)
        n_out, bytes_per_sample = 1, 4
    elif args.config == 5:
        # 16 384 x reverb_stereo(10, 2, 0.5) over 8 GPUs = 2048 instances per GPU; stereo white noise resident in HBM;
        # planar [instance][channel][frame] I/O (lane = frame in the kernel's staging phases)
        layout = F.LAYOUT_PLANAR
        bank = F.Bank.reverb_stereo(V, 10.0, 2.0, 0.5)
        bank.set_sample_rate(sr)
        g = torch.Generator(device="cuda").manual_seed(1234 + first)
        inp = torch.rand((V, 2, T), dtype=torch.float32, device="cuda", generator=g) * 2 - 1
        n_out, bytes_per_sample = 2, 272  # 32 ring reads + 32 ring writes + 2 in + 2 out, x 4 B
    else:
        assert layout == F.LAYOUT_VOICE_MINOR, "config 4 bench uses the device-native layout"
        F.wavetable_build("saw")
        bank = W.make_saw_moog_bank(V, sr, voice0=first)
        gate = torch.from_numpy(W.gate_signal(T, sr)).cuda()
        inp = gate[None, :, None].expand(1, T, V).contiguous()  # [1][frame][voice] gate resident in HBM
        n_out, bytes_per_sample = 2, 12
    fs = T if layout == F.LAYOUT_PLANAR else 0
    out = torch.empty((n_out, T, V) if layout == F.LAYOUT_VOICE_MINOR else (V, n_out, fs), dtype=torch.float32,
                      device="cuda")

    def step():
        bank.process(T, inp, out, layout=layout, frame_stride=fs, mode=mode)
        if args.mix:
            if args.config == 3:
                mix = F.mix_stereo(out[0] if layout == F.LAYOUT_VOICE_MINOR else out[:, 0, :].t().contiguous())
            elif args.config == 5:
                mix = out.sum(dim=0)     # [2][T] sum over instances (planar layout)
            else:
                mix = F.sum_voices(out)  # voices are already panned to stereo
            fdist.allreduce_mix(mix)

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # HIP events recorded by the C ABI on the launch stream around the render kernel (read after the loop
        # would only see the last launch; reading here synchronises on that launch, which the next step's
        # launch on the same stream is ordered behind anyway)
        kernel_ms.append(bank.last_kernel_ms())
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # also relate the kernel to what this box's HBM delivers to plain streaming kernels (a 2 GiB
    # device-to-device copy = read + write, and a fill = write only, the kernel's own traffic shape), outside the timed region
    measured = None
    if rank == 0 and world == 1:
        n = 1 << 29
        a = torch.empty(n, dtype=torch.float32, device="cuda")
        b2 = torch.empty(n, dtype=torch.float32, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        def timed(fn, reps=5):
            fn()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e-3
        t_copy = timed(lambda: b2.copy_(a))
        t_fill = timed(lambda: a.fill_(1.0))
        measured = {"copy_gbs": round(2 * n * 4 / t_copy / 1e9, 1), "fill_gbs": round(n * 4 / t_fill / 1e9, 1)}
        del a, b2

    if rank == 0:
        total_samples = float(world) * V * T * args.steps
        value = total_samples / elapsed / 1e6
        avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        # config 3 = 4 B per voice-sample out + 64 B state/params per voice per launch;
        # config 4 = 4 B gate in + 8 B stereo out per voice-sample (+ 188 B of slots per voice per launch)
        algo_bytes = V * T * bytes_per_sample + V * {3: 64, 4: 188, 5: 512}[args.config]
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("voices") == V and rec.get("frames") == T and args.config == 3:
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": {3: "Msamples/s (whole node) for 65536-voice SVF+FM graph",
                       4: "Msamples/s (whole node) for saw>>moog*adsr>>pan voices (BASELINE config 4, informational)",
                       5: "M instance-frames/s (whole node) for reverb_stereo FDN instances (BASELINE config 5, informational)"}[args.config],
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": {3: "BASELINE config 3: sine_hz(f)*f*m+f >> sine() >> lowpass_hz(fc,q), ",
                             4: "BASELINE config 4 voice: ((dc(f)>>saw()|dc(fc)|dc(q))>>moog())*adsr_live(.01,.1,.6,.2)>>pan(p), gate in, ",
                             5: "BASELINE config 5: reverb_stereo(10.0, 2.0, 0.5) 32-line FDN, stereo noise in, planar I/O, "}[args.config] +
                            f"{V} voices/GPU x {T} frames/step @ {sr:g} Hz, voice-out ([frame][voice] f32), "
                            f"{args.mode} semantics, per-voice params from rnd1(4v+k), phases via set_seed(v)",
                "voices_per_gpu": V,
                "frames_per_step": T,
                "layout": args.layout,
                "mix_allreduce": bool(args.mix),
                "parallelism": f"voice-shard x{world}",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "kernel": {3: (f"fd::k_render_pipe<fm_svf, {args.mode}, 2 compute stages cut after the modulator> (voice_minor)"
                               if args.layout == "voice_minor" and args.mode == "process" and args.pipe_split in (1, 2) and T % 8 == 0
                               else f"fd::k_render_pipe_planar<fm_svf, {args.mode}, 2 compute stages + storer wave> (planar)"
                               if args.layout == "planar" and args.pipe_split == 1 and T >= 256 and T % 4 == 0
                               else f"fd::k_render<fm_svf, {args.mode}, {args.layout}> / pipe_split={args.pipe_split}"),
                           4: f"fd::k_render_pipe<saw_moog_adsr_pan, {args.mode}, loader wave + 3 compute stages (saw | moog | *adsr >> pan)> ({args.layout})",
                           5: "fd::k_fdn_render"}[args.config],
                "kernel_ms_avg": round(avg_ms, 4),
                "algorithmic_bytes_per_launch": algo_bytes,
                "measured_streaming": measured,
                "frac_of_measured_fill": round(achieved / measured["fill_gbs"], 4) if measured else None,
            },
        }
        if world == 1 and args.cpu_seconds > 0 and args.config == 3:
            res["cpu_baseline"] = cpu_baseline(V, T, sr, args.cpu_seconds)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)

    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
