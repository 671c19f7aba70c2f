#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X voice-bank engine.

Metric (BASELINE.json): Msamples/s (whole node) for the 65 536-voice SVF+FM graph at 1/2/4/8 MI355X
    sine_hz(f) * f * m + f >> sine() >> lowpass_hz(fc, q)           (BASELINE config 3)
One step = one pass of the hot path: render FRAMES samples of every voice of the rank's shard into an HBM-resident
[frame][voice] f32 buffer (voice-out mode A, AudioNode::process semantics, 64-sample blocks), exact arithmetic
(bit-identical to the CPU oracle).

Multi-GPU (`--gpus N`, one process per GPU): the headline is STRONG scaling -- the metric's 65 536 voices in total,
sharded as contiguous ranges of 65 536 / N voices per GPU, no data-path collective.  `--scaling weak` keeps 65 536 voices
per GPU instead; at N > 1 the weak figure is also measured (outside the timed region) and reported as
`config.scaling_alt`.  `--mix` adds the on-device stereo mix-down + one RCCL all-reduce of [2][frames] per step.

Prints ONE JSON line on rank 0 (contract in the round brief) with `roofline`, `cpu_baseline` and -- at N = 1, measured
after the timed region -- a `secondary` list: the tolerance mode of the same workload, and BASELINE configs 2, 4, 5.
"""
import argparse
import ctypes
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
TOTAL_VOICES = 65536   # BASELINE.json: "65536-voice SVF+FM graph"


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (C port of the reference's process()/tick() path) built for THIS host
# ---------------------------------------------------------------------------------------------------------------------
NATIVE_FLAGS = "-O3 -march=native -ffp-contract=off -fno-fast-math"


def _native_oracle():
    """(oracle module, ctypes handle of the -O3 -march=native build of oracle/ made on THIS host)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native"])
    L = C.CDLL(os.path.join(ROOT, "oracle", "_native", "libfundsp_oracle_native.so"))
    L.o_bank_render.restype = C.c_double
    L.o_bank_render.argtypes = [C.POINTER(O.BankJob), C.POINTER(C.c_float)]
    L.o_fast_simd_flavour.restype = C.c_char_p
    return O, L


def host_cpu_budget(cgroup_root="/sys/fs/cgroup"):
    """What this process may actually use of the host: logical CPUs, the affinity mask, and the cgroup CPU quota (v2 cpu.max /
    v1 cpu.cfs_quota_us).  `effective` = min(affinity, ceil(quota)): the thread count the cpu_baseline legs run with and report
    as `cores` (os.cpu_count() said 256 on a box whose run behaved like ~20)."""
    logical = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = logical
    quota = None
    try:
        txt = open(os.path.join(cgroup_root, "cpu.max")).read().split()
        if txt and txt[0] != "max":
            quota = float(txt[0]) / float(txt[1])
    except (OSError, ValueError, IndexError):
        try:
            q = float(open(os.path.join(cgroup_root, "cpu", "cpu.cfs_quota_us")).read())
            p = float(open(os.path.join(cgroup_root, "cpu", "cpu.cfs_period_us")).read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            quota = None
    effective = affinity if quota is None else max(1, min(affinity, int(quota + 0.999)))
    load = None
    try:
        load = os.getloadavg()[0]
    except OSError:
        pass
    return {"logical_cpus": logical, "affinity_cpus": affinity, "cgroup_quota_cpus": None if quota is None else round(quota, 2),
            "effective_cpus": effective, "loadavg_1min_before": None if load is None else round(load, 1)}


def cpu_baseline(voices, frames, sample_rate, target_seconds):
    """Time the CPU restatement of the reference path on a bounded sample of the same workload, on this host's cores.
    `value` = the MONOMORPHISED process() path (oracle/o_fast.c: what rustc makes of the statically typed graph -- no node
    tree, both sines one 8-lane `wide` operation per 8 frames on the host's SIMD unit, the SVF per sample; bit-identical
    to the generic oracle, tests/test_oracle_fast.py).  Also reported: the generic tree-walking oracle in the same shape
    (`tree_walk_value`, round 1-2's figure) and tick-shaped (`tick_shaped_value`: AudioNode::tick per sample, libm sinf).
    Built on this host with NATIVE_FLAGS (oracle/Makefile `native`).
    `cores` = the threads used = what the process may use (host_cpu_budget: affinity mask and cgroup quota, not os.cpu_count()),
    each pinned to its own CPU of the mask; `one_thread_value` = the same binary on ONE pinned thread for >= 1 s, and
    `scaling_efficiency` = value / (cores x one_thread_value) -- well below 1 means the threads did not get a core each
    (SMT siblings, an oversubscribed or throttled host), whatever the count says.  Every leg runs >= 3 s."""
    from fundsp_amd import workloads as W

    O, L = _native_oracle()
    budget = host_cpu_budget()
    cores = budget["effective_cpus"]
    L.o_bank_pin_threads.argtypes = [C.c_int]
    L.o_bank_pin_threads(1)

    def timed(n, process, fast, threads):
        p = W.fm_svf_params(n, sample_rate)
        return O.bank_render(3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], frames, sample_rate, process, 0, threads,
                             store=False, lib=L, fast=fast)[1]

    def leg(process, fast, threads, seconds):
        n = 2 * threads
        s = timed(n, process, fast, threads)  # calibrate on a small sample, then size the timed sample for its share of the budget
        for _ in range(3):                    # (a short calibration run under-estimates a loaded host: re-size until the leg is long enough)
            rate = n * frames / max(s, 1e-6)
            n = int(max(threads, rate * seconds * 1.15 / frames))   # (beyond the bank's 65 536 voices the family simply continues: voice index -> parameters)
            n = max(threads, n // threads * threads)
            s = timed(n, process, fast, threads)
            if s >= seconds:
                break
        return (n * frames / s / 1e6, n, s)

    per_leg = max(3.0, target_seconds / 4.0)
    out = {"fast": leg(True, True, cores, per_leg), "tree": leg(True, False, cores, per_leg), "tick": leg(False, False, cores, per_leg)}
    one = leg(True, True, 1, max(1.0, target_seconds / 12.0))
    L.o_bank_pin_threads(0)
    v, n, s = out["fast"]
    eff = v / (cores * one[0]) if one[0] > 0 else None
    return {
        "value": round(v, 3),
        "unit": "Msamples/s",
        "cores": cores,
        "host": budget,
        "threads_pinned": True,
        "one_thread_value": round(one[0], 3),
        "scaling_efficiency": None if eff is None else round(eff, 3),
        "kind": "port (monomorphised)",
        "flags": f"gcc {NATIVE_FLAGS}",
        "simd": L.o_fast_simd_flavour().decode(),
        "tree_walk_value": round(out["tree"][0], 3),
        "tick_shaped_value": round(out["tick"][0], 3),
        "sample": f"{n} voices of the config-3 family (the bank has {voices}) x {frames} frames, monomorphised process() path of the reference restated in C "
                  f"(oracle/o_fast.c: f32x8 sines as 8-lane vector code, SVF per sample; gcc {NATIVE_FLAGS}), {cores} pinned threads, {s:.2f} s; "
                  f"one thread: {one[1]} voices, {one[2]:.2f} s; "
                  f"generic tree-walking oracle, same shape: {out['tree'][1]} voices, {out['tree'][2]:.2f} s; "
                  f"tick-shaped: {out['tick'][1]} voices, {out['tick'][2]:.2f} s",
    }


def cpu_baseline_config(config, sample_rate, frames, target_seconds=3.0):
    """cpu_baseline of a secondary entry (configs 2 / 4 / "4v" / 5): the C restatement of the reference's process() path on a bounded sample
    of the same workload, built on this host with NATIVE_FLAGS, `cores` pinned threads with MANY voices per thread (the C bank drivers of
    oracle/o_bank.c / o_fast.c split the voices over the threads), every leg >= 3 s.  Configs 4 / 5: the MONOMORPHISED process() -- the
    statically dispatched, inlined form rustc makes of the typed graph (fundsp_oracle.c o_c4_block / o_reverb_stereo_block: the tree walk's
    own node functions without the tree; bit-equal to it, tests/test_oracle_fast.py) -- with the generic tree walk beside it
    (`tree_walk_value`).  Config 2: the reference's own SIMD form, BiquadBank<f32x8> -- EIGHT voices per vector instruction (o_fast.c
    o_biquad_bank8_render; bit-equal to the scalar voices) -- with one scalar Biquad per voice beside it (`scalar_voice_value`).
    "rv3" / "fdn16" (round 6: reverb3_stereo and the prelude's fdn example): their monomorphised block forms and the tree walk (o_fast.c o_graph_bank_render)."""
    import numpy as np

    from fundsp_amd import workloads as W

    O, L = _native_oracle()
    cores = host_cpu_budget()["effective_cpus"]
    L.o_bank_pin_threads.argtypes = [C.c_int]
    L.o_bank_pin_threads(1)
    try:
        if config == 2:
            def timed(n, fast):   # fast: BiquadBank<f32x8>, eight voices per SIMD instruction -- the reference's own form of this config
                p = W.noise_biquad_params(n, sample_rate)
                return O.bank_render(2, [p["fc"], p["q"]], p["seed"], frames, sample_rate, True, 0, cores, store=False, lib=L, fast=fast)[1]
            unit, what, legs = "Msamples/s", "config-2 voices (8 x noise >> BiquadBank<f32x8> lanes)", (("value", True), ("scalar_voice_value", False))
        elif config in (4, "4v"):
            adsr = (0.01, 0.1, 0.6, 0.2)
            gate = W.gate_signal(frames, sample_rate) if config == 4 else None
            plan = W.gate_plan(frames, sample_rate) if config == "4v" else None

            def timed(n, fast):
                p = W.saw_moog_params(n, sample_rate)
                return O.c4_bank_render(p, adsr, frames, sample_rate, gate=gate, plan=plan, threads=cores, fast=fast, store=False, lib=L)[1]
            unit = "Msamples/s"
            what = ("config-4 voices, gate = " + ("an audio-rate input stream" if config == 4 else "var(gate) >> adsr_live, the variable set between the two halves of the note"))
            legs = (("value", True), ("tree_walk_value", False))
        elif config in ("rv3", "fdn16"):
            # round 6's lane-per-frame kinds (oracle/o_fast.c o_graph_bank_render): the monomorphised block forms (fundsp_oracle.c o_reverb3_block /
            # o_fdn16_block: the tree walk's ticks written out on plain state, bit-equal to it -- tests/test_oracle_fast.py) and the generic tree walk
            rng = np.random.default_rng(5)
            nch = 2 if config == "rv3" else 1
            x = (rng.random((nch, frames), dtype=np.float32) * 2 - 1).astype(np.float32)
            if config == "rv3":
                params, which, what = (2.0, 0.5, 8000.0), "reverb3", "reverb3_stereo(2, 0.5, lowpole_hz(8000)) instances on one stereo noise input"
            else:
                r = W.rnd1(np.arange(16, dtype=np.uint64))
                params = [float(np.float32(np.float32(0.01) * (np.float32(1) - np.float32(v)) + np.float32(0.03) * np.float32(v))) for v in r] + [float(np.float32(w)) for w in (0.2, 0.4, 0.2)]
                which, what = "fdn16", "instances of the prelude's fdn example (16 lines) on one mono noise input"

            def timed(n, fast):
                return O.graph_bank_render(which, params, n, x, sample_rate, threads=cores, store=False, lib=L, fast=fast)[1]
            unit, legs = "M instance-frames/s", (("value", True), ("tree_walk_value", False))
        else:
            rng = np.random.default_rng(5)
            x = (rng.random((2, frames), dtype=np.float32) * 2 - 1).astype(np.float32)

            def timed(n, fast):
                return O.reverb_bank_render(n, x, sample_rate, 10.0, 2.0, 0.5, threads=cores, fast=fast, store=False, lib=L)[1]
            unit, what, legs = "M instance-frames/s", "reverb_stereo(10, 2, 0.5) instances on one stereo noise input", (("value", True), ("tree_walk_value", False))
        out = {"unit": unit, "cores": cores, "threads_pinned": True,
               "kind": "port (monomorphised)", "flags": f"gcc {NATIVE_FLAGS}"}
        notes = []
        for key, fast in legs:
            n = cores
            s = timed(n, fast)
            for _ in range(3):   # size the sample for >= target_seconds (many voices per thread), re-size while the estimate was short
                n = max(cores, int(n * target_seconds * 1.15 / max(s, 1e-4)) // cores * cores)
                s = timed(n, fast)
                if s >= target_seconds:
                    break
            out[key] = round(n * frames / s / 1e6, 3)
            notes.append(f"{('BiquadBank<f32x8> (biquad_bank.rs:73-84), 8 voices per vector' if config == 2 else 'monomorphised process()') if fast else ('one scalar Biquad per voice, tree walk' if config == 2 else 'generic tree walk')}: {n} {what} x {frames} frames "
                         f"= {n // cores} per thread, {s:.2f} s")
        out["sample"] = "; ".join(notes) + f"; {cores} pinned threads"
        return out
    finally:
        L.o_bank_pin_threads(0)


# ---------------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------------
def make_workload(F, W, torch, config, V, T, sr, first, layout, math, voice_out=True):
    """-> dict(bank, inp, out, layout, fs, n_out, bytes_per_sample, slot_bytes, kernel); voice_out=False: no per-voice output
    buffer (a run that only ever takes the fused mix-down, fdsp_bank_process_mix)"""
    inp = None
    if config == 3:
        bank = W.make_fm_svf_bank(V, sr, voice0=first)
        n_out, bps, slot_bytes = 1, 4, 64          # 4 B/voice-sample out + 64 B state/params per launch
        kernel = ("fd::k_render_pipe<fm_svf, process, 2 compute stages (modulator | carrier + lowpass-specialised SVF)>"
                  if layout == F.LAYOUT_VOICE_MINOR else "fd::k_render_pipe_planar<fm_svf, process, 2 compute stages + storer wave>")
    elif config == 2:
        bank = W.make_noise_biquad_bank(V, sr, voice0=first)
        n_out, bps, slot_bytes = 1, 4, 48          # noise generated in-kernel: 4 B/voice-sample out
        kernel = "fd::k_render_ts3<noise_biquad> (whole blocks, small banks: noise | biquad feed-forward half | recurrence) / fd::k_render_pipe / fd::k_render (ragged launches)"
    elif config in (5, "5r4"):
        # 16 384 x reverb_stereo(10, 2, 0.5) over 8 GPUs = 2048 instances per GPU; stereo white noise resident in HBM;
        # planar [instance][channel][frame] I/O (the kernel is lane = frame).  "5r4": reverb4_stereo(20, 2) -- two 16-line networks in
        # series (prelude.rs:1873-1941) -- through the same kernel family
        layout = F.LAYOUT_PLANAR
        bank = F.Bank.reverb_stereo(V, 10.0, 2.0, 0.5) if config == 5 else F.Bank.reverb4_stereo(V, 20.0, 2.0)
        bank.set_sample_rate(sr)
        g = torch.Generator(device="cuda").manual_seed(1234 + first)
        inp = torch.rand((V, 2, T), dtype=torch.float32, device="cuda", generator=g) * 2 - 1
        n_out, bps, slot_bytes = 2, 272, 512       # 32 ring reads + 32 ring writes + 2 in + 2 out, x 4 B
        kernel = "fd::k_fdn_render_frames (lane = frame, one wave per instance)" + ("" if config == 5 else ", two 16-line networks in series")
    elif config == "5bus":
        # the reverb as the reference's documentation uses it: `multipass() & 0.2 * reverb_stereo(20.0, 2.0, 1.0)` ("to add 20% reverb to a stereo signal",
        # README.md:436), built from the GRAPH: Bank.from_graph recognises the bus (graph.bus_plan), builds the reverb's lane-per-frame bank and folds the
        # Unop, MultiPass and Bus nodes into its kernel's epilogue (fdsp_bank_set_bus); planar I/O, stereo noise in
        from fundsp_amd import graph as G

        layout = F.LAYOUT_PLANAR
        bank = F.Bank.from_graph(G.multipass(2) & 0.2 * G.reverb_stereo(20.0, 2.0, 1.0), V, sample_rate=sr)
        assert bank.kind == "reverb_stereo" and bank.get_bus()[0] == F.BUS_DRY_WET, "Bank.from_graph did not fold the bus into the reverb's lane-per-frame bank"
        g = torch.Generator(device="cuda").manual_seed(555 + first)
        inp = torch.rand((V, 2, T), dtype=torch.float32, device="cuda", generator=g) * 2 - 1
        n_out, bps, slot_bytes = 2, 272, 512       # the bare reverb's bytes: the bus reads and writes nothing of its own
        kernel = "fd::k_fdn_render_frames (lane = frame, one wave per instance), dry / wet bus in the epilogue"
    elif config == "rv3":
        # reverb3_stereo(2.0, 0.5, lowpole_hz(8000.0)) -- the reference's own example (prelude.rs:1850-1856) -- built from the GRAPH: Bank.from_graph
        # sees the stock node and takes its lane-per-frame kernel (fdsp_reverb3_stereo_create); planar I/O, stereo noise in
        from fundsp_amd import graph as G

        layout = F.LAYOUT_PLANAR
        bank = F.Bank.from_graph(G.reverb3_stereo(2.0, 0.5, lambda: G.lowpole_hz(8000.0)), V, sample_rate=sr)
        assert bank.kind == "reverb3_stereo", "Bank.from_graph did not take the lane-per-frame kernel for reverb3_stereo"
        g = torch.Generator(device="cuda").manual_seed(777 + first)
        inp = torch.rand((V, 2, T), dtype=torch.float32, device="cuda", generator=g) * 2 - 1
        n_out, bps, slot_bytes = 2, 624, 1024      # 76 ring reads + 76 ring writes + 2 in + 2 out, x 4 B
        kernel = "fd::k_rv3_render (lane = frame, one wave per instance; the sixteen loop filters on eight lanes between the allpass layers)"
    elif config == "fdn16":
        # the Hadamard network the prelude documents (prelude.rs:1334 "Mono Reverb"): split >> fdn::<U16>(stacki(delay(lerp(0.01, 0.03, rnd1(i)))
        # >> fir((0.2, 0.4, 0.2)))) >> join, built from the GRAPH: Bank.from_graph recognises the shape (graph.fdn_plan) and takes the
        # lane-per-frame FDN kernel (fdsp_fdn_create) instead of compiling a lane-per-voice Feedback graph; planar I/O, mono noise in
        import numpy as np
        from fundsp_amd import graph as G

        layout = F.LAYOUT_PLANAR
        r = W.rnd1(np.arange(16, dtype=np.uint64))
        d = [float(np.float32(np.float32(0.01) * (np.float32(1) - np.float32(x)) + np.float32(0.03) * np.float32(x))) for x in r]
        bank = F.Bank.from_graph(G.split(16) >> G.fdn(G.stacki(16, lambda i: G.delay(d[i]) >> G.fir(0.2, 0.4, 0.2))) >> G.join(16), V, sample_rate=sr)
        assert bank.kind == "fdn", "Bank.from_graph did not take the lane-per-frame FDN kernel for the documented fdn graph"
        g = torch.Generator(device="cuda").manual_seed(4321 + first)
        inp = torch.rand((V, 1, T), dtype=torch.float32, device="cuda", generator=g) * 2 - 1
        n_out, bps, slot_bytes = 1, 136, 256       # 16 ring reads + 16 ring writes + 1 in + 1 out, x 4 B
        kernel = "fd::k_fdn_frames_generic<16, 3> (lane = frame, one wave per instance, ring capacity at run time)"
    elif config == "4v":
        # config 4 in the reference's own gate shape: `var(gate) >> adsr_live` (examples/live_adsr.rs:72; SURVEY 8(d) `dc(gate)`): the gate is a
        # per-voice Var slot, read once per block like Var::process (shared.rs:122-125) -- the graph has NO input.  One step = one note per
        # second = the launches of W.gate_plan (gate high for 0.5 s, then low), the slot set on the device between them.
        layout = F.LAYOUT_VOICE_MINOR
        F.wavetable_build("saw")
        bank = W.make_saw_moog_var_bank(V, sr, voice0=first)
        plan = W.gate_plan(T, sr)
        n_out, bps, slot_bytes = 2, 8, 192         # 8 B stereo out per voice-sample, nothing in
        kernel = "fd::k_render_pipe<saw_moog_var_adsr_pan, process, 3 compute stages (saw | moog | *(var >> adsr) >> pan), no loader wave>"
        if math == "fast":
            bank.set_option("math", F.MATH_FAST)
        outs = [torch.empty((n_out, n, V), dtype=torch.float32, device="cuda") for _, n in plan] if voice_out else None
        return dict(bank=bank, inp=None, out=None, outs=outs, plan=plan, gate_slot=W.C4V_SLOTS["gate"], layout=layout, fs=0, n_out=n_out, bps=bps,
                    slot_bytes=slot_bytes, kernel=kernel)
    else:
        layout = F.LAYOUT_VOICE_MINOR
        F.wavetable_build("saw")
        bank = W.make_saw_moog_bank(V, sr, voice0=first)
        gate = torch.from_numpy(W.gate_signal(T, sr)).cuda()
        inp = gate[None, :, None].expand(1, T, V).contiguous()  # [1][frame][voice] gate resident in HBM
        n_out, bps, slot_bytes = 2, 12, 188        # 4 B gate in + 8 B stereo out per voice-sample
        kernel = "fd::k_render_pipe<saw_moog_adsr_pan, process, loader wave + 3 compute stages (saw | moog | *adsr >> pan)>"
    if math == "fast":
        bank.set_option("math", F.MATH_FAST)
    fs = T if layout == F.LAYOUT_PLANAR else 0
    out = None
    if voice_out:
        out = torch.empty((n_out, T, V) if layout == F.LAYOUT_VOICE_MINOR else (V, n_out, fs), dtype=torch.float32, device="cuda")
    return dict(bank=bank, inp=inp, out=out, layout=layout, fs=fs, n_out=n_out, bps=bps, slot_bytes=slot_bytes, kernel=kernel)


def run_plan(wl, mode, mix=None, mixbufs=None, kernel_ms=False):
    """One step of a workload whose step is a PLAN of launches with a shared variable set in between (config "4v"); kernel_ms: wait for
    every launch and return the sum of their HIP-event times (otherwise nothing waits: the step is enqueued back to back)"""
    bank, kms = wl["bank"], 0.0
    for k, (value, n) in enumerate(wl["plan"]):
        bank.set_param(wl["gate_slot"], float(value))     # fdsp_bank_set_param_all: filled on the device, in stream order, no host wait
        if mix is None:
            bank.process(n, None, wl["outs"][k], layout=wl["layout"], mode=mode)
        else:
            bank.process_mix(n, None, mix=mix, out=mixbufs[k], mode=mode)
        if kernel_ms:
            kms += bank.last_kernel_ms()
    return kms


def quick(F, torch, wl, T, mode, steps=6, warmup=2):
    """Secondary measurements (outside the headline's timed region): wall ms per step and the kernel's own HIP-event ms."""
    bank = wl["bank"]
    if "plan" in wl:
        for _ in range(warmup):
            run_plan(wl, mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run_plan(wl, mode)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        k = [run_plan(wl, mode, kernel_ms=True) for _ in range(max(2, steps // 2))]
        return ms, sum(k) / len(k)
    # as the headline's timed region: the launches back to back on one (non-default) stream, each between a pair of HIP events on that stream,
    # nothing waits until all of them are queued
    qs = torch.cuda.Stream()
    with torch.cuda.stream(qs):
        for _ in range(warmup):
            bank.process(T, wl["inp"], wl["out"], layout=wl["layout"], frame_stride=wl["fs"], mode=mode)
        torch.cuda.synchronize()
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for a, b in pairs:
            a.record()
            bank.process(T, wl["inp"], wl["out"], layout=wl["layout"], frame_stride=wl["fs"], mode=mode)
            b.record()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    k = [a.elapsed_time(b) for a, b in pairs]
    return ms, sum(k) / len(k)


def secondary(F, W, torch, sr, mode):
    """Configs the headline does not time (N = 1 only; each entry says what it measured)."""
    out = []
    V, T = TOTAL_VOICES, 48000
    wl = make_workload(F, W, torch, 3, V, T, sr, 0, F.LAYOUT_VOICE_MINOR, "fast")
    ms, kms = quick(F, torch, wl, T, mode)
    algo = V * T * 4 + V * 64
    out.append({"name": "config3_math_fast", "what": "the headline workload in tolerance mode (FDSP_MATH_FAST: FMA sine polynomial, "
                "recurrences exact; within 1e-4 of the exact mode over 441 samples; over the full second asserted <= 1e-3 max / 5e-5 rms, measured 2.6e-4 / 1.2e-5: tests/test_gpu_math_fast.py)", "ms_per_step": round(ms, 4),
                "kernel_ms_avg": round(kms, 4), "value": round(V * T / ms / 1e3, 1), "unit": "Msamples/s",
                "roofline_frac": round(algo / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
    del wl
    # the strong-scaling shards of the headline on ONE GPU: what each of N GPUs renders when the 65 536 voices are split N
    # ways (no collective on the data path, so the N-GPU step time is the shard's time): small banks take the time-split kernel
    shards = {"name": "config3_strong_scaling_shards", "what": "the per-GPU shard of the 65 536-voice headline at N = 2 / 4 / 8 GPUs, "
              "rendered on this one GPU (exact arithmetic; banks of <= 32 768 voices take the three-way time-split kernel k_render_ts3); "
              "implied_value = 65536 voices x frames / shard time", "unit": "Msamples/s"}
    for n in (2, 4, 8):
        Vs = TOTAL_VOICES // n
        wl = make_workload(F, W, torch, 3, Vs, T, sr, 0, F.LAYOUT_VOICE_MINOR, "exact")
        ms, kms = quick(F, torch, wl, T, mode)
        shards[f"N{n}"] = {"voices_per_gpu": Vs, "ms_per_step": round(ms, 4), "kernel_ms_avg": round(kms, 4),
                           "implied_value": round(TOTAL_VOICES * T / ms / 1e3, 1)}
        del wl
    out.append(shards)
    # config 2: 1024-voice biquad bank on white noise, 64-sample blocks (the launch-latency config)
    V = 1024
    c2 = {"name": "config2_biquad_bank_1024", "what": "BASELINE config 2: 1024 voices noise >> lowpass biquad (one BiquadBank<f32x8> lane "
          "per voice), noise generated in-kernel, voice-out; T = frames per launch.  The DF1 biquad is a chain of two stages cut at the seam of its expression "
          "(feed-forward half | recurrence, biquad.rs:186-188), so launches of whole blocks take the three-way time-split kernel (last_kernel 4): noise and the "
          "feed-forward half in three waves each, the serial wave carries the recurrence alone", "unit": "Msamples/s"}
    for T in (64, 128, 256, 4096, 48000):
        wl = make_workload(F, W, torch, 2, V, T, sr, 0, F.LAYOUT_VOICE_MINOR, "exact")
        ms, kms = quick(F, torch, wl, T, mode, steps=50 if T <= 256 else 10, warmup=5)
        c2[f"T{T}"] = {"us_per_launch": round(ms * 1e3, 2), "kernel_us": round(kms * 1e3, 2), "value": round(V * T / ms / 1e3, 2),
                       "last_kernel": wl["bank"].get_option("last_kernel")}
        if T == 48000:
            try:
                c2["cpu_baseline"] = cpu_baseline_config(2, sr, 48000)
            except Exception as e:
                c2["cpu_baseline"] = {"error": repr(e)}
        if T <= 256:  # the real-time pattern: one launch per callback, call by call vs replayed from a HIP graph
            NB = 32
            outs = [torch.empty_like(wl["out"]) for _ in range(NB)]
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    for kk in range(NB):
                        wl["bank"].process(T, None, outs[kk], layout=wl["layout"], mode=mode)
                g.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    g.replay()
                torch.cuda.synchronize()
                us = (time.perf_counter() - t0) / 20 / NB * 1e6
            c2[f"T{T}"]["hip_graph_replay_us_per_launch"] = round(us, 2)
            c2[f"T{T}"]["hip_graph_value"] = round(V * T / us, 2)
            del g, outs
        del wl
    # who wins at the block size the config names (64 frames per launch), and from which launch length the GPU does: the CPU's process()
    # path works in 64-sample blocks whatever the caller's buffer, so its figure does not depend on T
    cpu = c2.get("cpu_baseline", {}).get("value")
    if cpu:
        c2["cpu_vs_gpu"] = {"cpu_value": cpu, "cpu_cores": c2["cpu_baseline"].get("cores")}
        for T in (64, 128, 256):
            gv = c2[f"T{T}"]["hip_graph_value"]
            c2["cpu_vs_gpu"][f"T{T}"] = {"gpu_hip_graph_value": gv, "winner": "gpu" if gv > cpu else "cpu", "gpu_over_cpu": round(gv / cpu, 3)}
        c2["cpu_vs_gpu"]["T48000"] = {"gpu_value": c2["T48000"]["value"], "winner": "gpu" if c2["T48000"]["value"] > cpu else "cpu",
                                      "gpu_over_cpu": round(c2["T48000"]["value"] / cpu, 3)}
        c2["cpu_vs_gpu"]["note"] = (f"the host renders one 64-frame block of the 1024 voices in {1024 * 64 / cpu:.2f} us (BiquadBank<f32x8>: eight voices per vector instruction, "
                                    f"{c2['cpu_baseline'].get('cores')} cores); a dependent chain of EMPTY kernels replays at ~1.6 us per node on this platform, a kernel of "
                                    "d us of work at d + 0.9, and an event pair around an empty kernel reads ~6 us (tools/ubench_launch.hip, profiles/r05_ubench_launch_*.txt): "
                                    "a 1024-voice bank is 16 wavefronts on a chip of 1024 SIMDs -- at one block per launch the launch is the cost, the GPU wins from launches "
                                    "of a few blocks on and by the bank size (voices_65536 below: the same kernel family on a bank that fills the chip)")
    # ... the same graph on a bank that fills the machine (64 x the voices): what the engine does with config 2's arithmetic when it has the lanes
    try:
        Vb, Tb = 65536, 4096
        wl = make_workload(F, W, torch, 2, Vb, Tb, sr, 0, F.LAYOUT_VOICE_MINOR, "exact")
        ms, kms = quick(F, torch, wl, Tb, mode, steps=10, warmup=3)
        c2["voices_65536"] = {"voices": Vb, "T": Tb, "us_per_launch": round(ms * 1e3, 2), "kernel_us": round(kms * 1e3, 2), "value": round(Vb * Tb / ms / 1e3, 1),
                              "last_kernel": wl["bank"].get_option("last_kernel"),
                              "roofline_frac": round((Vb * Tb * 4 + Vb * 48) / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # ... and that bank at the config's own block size, one 64-frame block per launch replayed from a HIP graph: the launch that costs the
        # 1024-voice bank the comparison carries 64 x the voices here
        NB = 16
        outs = [torch.empty((1, 64, Vb), dtype=torch.float32, device="cuda") for _ in range(NB)]
        s_ = torch.cuda.Stream()
        with torch.cuda.stream(s_):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s_):
                for kk in range(NB):
                    wl["bank"].process(64, None, outs[kk], layout=wl["layout"], mode=mode)
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / 20 / NB * 1e6
        c2["voices_65536"]["T64_hip_graph_replay_us_per_launch"] = round(us, 2)
        c2["voices_65536"]["T64_hip_graph_value"] = round(Vb * 64 / us, 1)
        if cpu:
            c2["voices_65536"]["T64_gpu_over_cpu"] = round(Vb * 64 / us / cpu, 2)
        del g, outs, wl
    except Exception as e:
        c2["voices_65536"] = {"error": repr(e)}
    out.append(c2)
    # ... the same launch pattern from a COMPILED host, straight through the C ABI (tools/launch_overhead.cpp, built here with g++):
    # what the Python binding adds per 64-frame block is the difference to T64.us_per_launch above
    try:
        exe = os.path.join(ROOT, "tools", "_launch_overhead")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "launch_overhead.cpp"), "-o", exe,
                               "-L", os.path.join(ROOT, "fundsp_amd"), "-lfundsp_hip", "-Wl,-rpath," + os.path.join(ROOT, "fundsp_amd"),
                               "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        txt = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
        rows = [json.loads(l[5:]) for l in txt.splitlines() if l.startswith("JSON ")]
        c2["T64"]["compiled_host_cpp"] = {("timing_on" if r["timing"] else "timing_off"): {k: v for k, v in r.items() if k != "timing"} for r in rows}
        c2["T64"]["compiled_host_cpp"]["what"] = ("tools/launch_overhead.cpp: fdsp_bank_process call by call from C++ (us): host time inside the call, 5000 back-to-back "
                                                  "launches per launch, launch + fdsp_bank_synchronize per block; with / without the per-launch HIP event pair")
    except Exception as e:
        c2["T64"]["compiled_host_cpp"] = {"error": repr(e)}
    # config 3 is quoted at T in {64, 4096, 48000}; the headline is T = 48000, here the other two (exact, voice-out).
    # Algorithmic bytes V*T*4 + V*64: 5 B per voice-sample at T = 64.  T = 64 also replayed from a HIP graph (the real-time pattern).
    c3 = {"name": "config3_frames_per_launch", "what": "BASELINE config 3 (65536 voices, exact, voice-out) at the other launch lengths of SURVEY 8(d): "
          "T = 64 (one AudioNode::process block per launch; `last_kernel` says which family ran: 2 = the stage pipeline) and T = 4096; algorithmic bytes V*T*4 + V*64 per launch", "unit": "Msamples/s"}
    V = TOTAL_VOICES
    for T in (64, 4096):
        wl = make_workload(F, W, torch, 3, V, T, sr, 0, F.LAYOUT_VOICE_MINOR, "exact")
        ms, kms = quick(F, torch, wl, T, mode, steps=50 if T == 64 else 20, warmup=5)
        algo = V * T * 4 + V * 64
        c3[f"T{T}"] = {"us_per_launch": round(ms * 1e3, 2), "kernel_us": round(kms * 1e3, 2), "value": round(V * T / ms / 1e3, 1),
                       "algorithmic_bytes_per_voice_sample": round(algo / (V * T), 3),
                       "roofline_frac": round(algo / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "last_kernel": wl["bank"].get_option("last_kernel")}
        if T == 64:
            NB = 16
            outs = [torch.empty_like(wl["out"]) for _ in range(NB)]
            s_ = torch.cuda.Stream()
            with torch.cuda.stream(s_):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s_):
                    for kk in range(NB):
                        wl["bank"].process(64, None, outs[kk], layout=wl["layout"], mode=mode)
                g.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    g.replay()
                torch.cuda.synchronize()
                us = (time.perf_counter() - t0) / 20 / NB * 1e6
            # ... and the block a real-time callback actually wants: 64 frames of the stereo MIX (fdsp_bank_process_mix: the pipeline
            # kernel with the fused mix-down + the tree pass; [2][64] f32 = 512 bytes leave the launch instead of 16.8 MB of voices)
            wl["bank"].mix_reserve(64)
            mixo = torch.empty((2, 64), dtype=torch.float32, device="cuda")
            for _ in range(5):
                wl["bank"].process_mix(64, mix=F.MIX_PAN, out=mixo, mode=mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            km = []
            for _ in range(50):
                wl["bank"].process_mix(64, mix=F.MIX_PAN, out=mixo, mode=mode)
                km.append(wl["bank"].last_kernel_ms())
            torch.cuda.synchronize()
            c3["T64"]["fused_mix_us_per_launch"] = round((time.perf_counter() - t0) / 50 * 1e6, 2)
            c3["T64"]["fused_mix_kernels_us"] = round(sum(km) / len(km) * 1e3, 2)
            mixes = [torch.empty_like(mixo) for _ in range(16)]
            sm = torch.cuda.Stream()
            with torch.cuda.stream(sm):
                gm = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gm, stream=sm):
                    for kk in range(16):
                        wl["bank"].process_mix(64, mix=F.MIX_PAN, out=mixes[kk], mode=mode)
                gm.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    gm.replay()
                torch.cuda.synchronize()
                c3["T64"]["fused_mix_hip_graph_replay_us_per_block"] = round((time.perf_counter() - t0) / 20 / 16 * 1e6, 2)
            del gm, mixes
            c3["T64"]["hip_graph_replay_us_per_block"] = round(us, 2)
            c3["T64"]["hip_graph_value"] = round(V * 64 / us, 1)
            c3["T64"]["hip_graph_roofline_frac"] = round(algo / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            del g, outs
        del wl
    out.append(c3)
    for cfg, V, name, unit, math in (("4v", 32768, "config4_var_gate_32768", "Msamples/s", "exact"),
                                     ("4v", 32768, "config4_var_gate_math_fast", "Msamples/s", "fast"),
                                     (4, 32768, "config4_saw_moog_adsr_pan_32768", "Msamples/s", "exact"),
                                     (4, 32768, "config4_math_fast", "Msamples/s", "fast"),
                                     (5, 2048, "config5_reverb_stereo_2048", "M instance-frames/s", "exact"),
                                     ("5r4", 2048, "reverb4_stereo_2048", "M instance-frames/s", "exact"),
                                     ("5bus", 2048, "reverb_stereo_with_its_documented_bus_2048", "M instance-frames/s", "exact"),
                                     ("fdn16", 4096, "fdn16_mono_reverb_from_graph_4096", "M instance-frames/s", "exact"),
                                     ("rv3", 2048, "reverb3_stereo_from_graph_2048", "M instance-frames/s", "exact")):
        T = 48000
        wl = make_workload(F, W, torch, cfg, V, T, sr, 0, F.LAYOUT_VOICE_MINOR, math)
        ms, kms = quick(F, torch, wl, T, mode, steps=6 if cfg == "4v" else 4, warmup=3 if cfg == "4v" else 1)
        algo = V * T * wl["bps"] + V * wl["slot_bytes"]
        arith = "exact arithmetic" if math == "exact" else ("tolerance mode (FDSP_MATH_FAST: the ladder's tanh on the hardware exp2 / reciprocal, "
                                                            "recurrence exact; within 1e-4 of the exact mode: tests/test_gpu_math_fast.py)")
        shape = {"4v": " -- the reference's gate shape `var(gate) >> adsr_live` (examples/live_adsr.rs:72): no graph input, the step = two launches "
                       "(gate high 24000 frames, low 24000) with the Var slot set on the device in between; 8 B per voice-sample (stereo out)",
                 4: " -- the gate as an audio-rate HBM input stream [frames][voices] (hosts that modulate the gate per sample): 4 B in + 8 B out per voice-sample"}.get(cfg, "")
        what = (f"the reference's allpass-loop reverb, reverb3_stereo(2.0, 0.5, lowpole_hz(8000)) (prelude.rs:1850-1871, reverb.rs:152-279), {V} instances x {T} frames, built with "
                "Bank.from_graph: the stock node is rendered by its lane-per-frame kernel (fdsp_reverb3_stereo_create) -- the run-time compiled lane-per-voice Reverb3 node renders the "
                "same samples ~160 x slower (profiles/r06_reverb3_probe.txt); 624 B per instance-frame (76 ring reads + 76 ring writes + 2 in + 2 out)"
                if cfg == "rv3" else
                f"the reverb as the reference's documentation uses it, `multipass() & 0.2 * reverb_stereo(20.0, 2.0, 1.0)` (README.md:436 'to add 20% reverb to a stereo signal'; "
                f"Bus audionode.rs:1842-1877, Unop<X, FrameMulScalar> combinator.rs:477-488), {V} instances x {T} frames, built with Bank.from_graph: the bus is recognised and folded "
                "into the epilogue of the reverb's lane-per-frame kernel (fdsp_bank_set_bus: out = in + 0.2 * y on the input frames the block still holds in registers) -- the same 272 B "
                "per instance-frame as the bare reverb; the whole graph compiled as one lane-per-voice kernel renders the same samples ~220 x slower (profiles/r06_reverb_bus_probe.txt)"
                if cfg == "5bus" else
                f"the generic Hadamard network of the prelude's own example (prelude.rs:1334: split >> fdn::<U16>(stacki(delay >> fir)) >> join), {V} instances x {T} frames, built "
                "with Bank.from_graph: the graph's shape is recognised and rendered by the lane-per-frame FDN kernel (fdsp_fdn_create) -- the run-time compiled lane-per-voice "
                "form of the same graph renders the same samples ~130 x slower (profiles/r06_fdn_generic_probe.txt); 136 B per instance-frame (16 ring reads + 16 ring writes + 1 in + 1 out)"
                if cfg == "fdn16" else
                f"BASELINE config {str(cfg)[0]} per-GPU shard ({V} {'voices' if cfg != 5 else 'instances'} x {T} frames), {arith}{shape}" if cfg != "5r4" else
                f"the other Hadamard FDN reverb of the reference, reverb4_stereo(20, 2) (prelude.rs:1873-1941: two fdn::<U16> in series), {V} instances x {T} frames "
                "through the lane-per-frame FDN kernel generalised to networks in series; same 272 B per instance-frame (32 ring reads + 32 ring writes + 2 in + 2 out)")
        out.append({"name": name, "what": what,
                    "ms_per_step": round(ms, 4), "kernel_ms_avg": round(kms, 4), "value": round(V * T / ms / 1e3, 1), "unit": unit,
                    "algorithmic_bytes_per_unit": wl["bps"], "roofline_frac": round(algo / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "kernel": wl["kernel"]})
        if cfg == 4:   # both accountings of the stream-gate kind: with its own 4 B/voice-sample input stream, and at the stereo output alone
            out[-1]["roofline_frac_at_8B_stereo_out_only"] = round((V * T * 8 + V * wl["slot_bytes"]) / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if cfg == "4v" and math == "exact":   # mode B of the same step: the stereo mix of the shard leaves the launches, [2][T] f32
            try:
                wl["outs"] = None
                torch.cuda.empty_cache()
                wl["bank"].mix_reserve(max(n for _, n in wl["plan"]))
                mixbufs = [torch.empty((2, n), dtype=torch.float32, device="cuda") for _, n in wl["plan"]]
                run_plan(wl, mode, F.MIX_SUM, mixbufs)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(4):
                    run_plan(wl, mode, F.MIX_SUM, mixbufs)
                torch.cuda.synchronize()
                ms_b = (time.perf_counter() - t0) / 4 * 1e3
                kb = [run_plan(wl, mode, F.MIX_SUM, mixbufs, kernel_ms=True) for _ in range(2)]
                out[-1]["mode_b_fused_mix"] = {"what": "the same step through fdsp_bank_process_mix (FDSP_MIX_SUM): the shard's stereo mix [2][frames] leaves "
                                               "the launches, no voice-out buffer; not a bandwidth test", "ms_per_step": round(ms_b, 4),
                                               "kernel_ms_incl_tree": round(sum(kb) / len(kb), 4), "value": round(V * T / ms_b / 1e3, 1)}
                del mixbufs
            except Exception as e:
                out[-1]["mode_b_fused_mix"] = {"error": repr(e)}
        if math == "exact" and cfg not in ("5r4", "5bus"):
            try:
                out[-1]["cpu_baseline"] = cpu_baseline_config(cfg, sr, T)
            except Exception as e:
                out[-1]["cpu_baseline"] = {"error": repr(e)}
        del wl
    # The path's one exchange step, on one GPU: config 4's stereo mix-down + ONE all-reduce(sum) of [2][frames] through the product's
    # collective (fdsp_mix_allreduce: RCCL inside libfundsp_hip.so, on the communicator's side stream, 1 rank here), overlapped with
    # the next render.  Round 4: the mix-down is FUSED into the render kernel (fdsp_bank_process_mix: the last stage of a voice group
    # reduces its 64 voices through LDS, one float per channel and frame leaves for HBM) -- the voice-out buffer
    # (12.6 GB) is neither written nor re-read.  The unfused pair (voice-out render + fdsp_sum_voices, same summation order) beside it.
    for cfg4, name4 in (("4v", "config4_var_gate_mix_single_rank"), (4, "config4_mix_single_rank")):
        try:
            V, T = 32768, 48000
            wl = make_workload(F, W, torch, cfg4, V, T, sr, 0, F.LAYOUT_VOICE_MINOR, "exact")
            plan = wl.get("plan")
            comm = F.Comm.local([0])
            bank = wl["bank"]
            bank.mix_reserve(T)
            ms_render, kms = quick(F, torch, wl, T, mode, steps=3, warmup=1)

            def wall(fn, n):
                fn()
                comm.wait(0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                comm.wait(0)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e3
            keep = []

            fused_k = [0.0]

            def fused_step():
                if plan is not None:   # the Var-gate shape: the note's launches, their [2][n] pieces joined, ONE all-reduce per step
                    pieces = [torch.empty((2, n), dtype=torch.float32, device="cuda") for _, n in plan]
                    run_plan(wl, mode, F.MIX_SUM, pieces)          # (enqueued back to back: nothing waits inside the step)
                    keep.append(torch.cat(pieces, dim=1))
                else:
                    keep.append(bank.process_mix(T, wl["inp"], mix=F.MIX_SUM, mode=mode))
                comm.allreduce(keep[-1], slot=0)
                del keep[:-2]

            def unfused_step():
                if plan is not None:
                    run_plan(wl, mode)
                    keep.append(torch.cat([F.sum_voices(o) for o in wl["outs"]], dim=1))
                else:
                    bank.process(T, wl["inp"], wl["out"], layout=wl["layout"], frame_stride=wl["fs"], mode=mode)
                    keep.append(F.sum_voices(wl["out"]))
                comm.allreduce(keep[-1], slot=0)
                del keep[:-2]
            ms_fused = wall(fused_step, 4)
            if plan is not None:
                fused_k[0] = run_plan(wl, mode, F.MIX_SUM, [torch.empty((2, n), dtype=torch.float32, device="cuda") for _, n in plan], kernel_ms=True)
            fused_kernel_ms = fused_k[0] if plan is not None else bank.last_kernel_ms()
            ms_unfused = wall(unfused_step, 3)
            mix = keep[-1]
            t0 = time.perf_counter()
            for _ in range(20):
                comm.allreduce(mix, slot=0)
                comm.wait(0)
            ar_us = (time.perf_counter() - t0) / 20 * 1e6
            groups = (V + 63) // 64
            out.append({"name": name4, "what": ("BASELINE config 4 in the reference's gate shape (var(gate) >> adsr_live: two launches per step, the Var slot set in between), " if plan is not None else "BASELINE config 4, gate as an HBM input stream, ") + "per-GPU shard (32768 voices x 48000 frames) with the path's exchange "
                        "step: the stereo mix-down FUSED into the render kernel (fdsp_bank_process_mix, FDSP_MIX_SUM: group partials [512][2][48000] f32 "
                        "+ fd k_mix_tree) + one fdsp_mix_allreduce of [2][48000] f32 (RCCL inside the library, side stream, 1-rank communicator), "
                        "overlapped with the next render; the unfused pair (voice-out render + fdsp_sum_voices, same order) beside it",
                        "ms_per_step_render_only": round(ms_render, 4), "ms_per_step_with_mix_and_allreduce": round(ms_fused, 4),
                        "fused_render_plus_tree_kernel_ms": round(fused_kernel_ms, 4),
                        "ms_per_step_unfused_mix_and_allreduce": round(ms_unfused, 4),
                        "partial_mix_bytes_per_step": groups * 2 * T * 4, "voice_out_bytes_not_written": V * T * 8,
                        "allreduce_blocking_us_1rank": round(ar_us, 1),
                        "value": round(V * T / ms_fused / 1e3, 1), "unit": "Msamples/s"})
            comm.close()
            del wl, keep, mix
        except Exception as e:
            out.append({"name": name4, "error": repr(e)})
    # ... and the headline's voices in mode B: every voice panned (FDSP_MIX_PAN) and summed in the render launch
    try:
        V, T = TOTAL_VOICES, 48000
        wl = make_workload(F, W, torch, 3, V, T, sr, 0, F.LAYOUT_VOICE_MINOR, "exact", voice_out=False)
        bank = wl["bank"]
        bank.mix_reserve(T)
        mixbuf = torch.empty((2, T), dtype=torch.float32, device="cuda")
        for _ in range(2):
            bank.process_mix(T, mix=F.MIX_PAN, out=mixbuf, mode=mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = []
        for _ in range(6):
            bank.process_mix(T, mix=F.MIX_PAN, out=mixbuf, mode=mode)
            k.append(bank.last_kernel_ms())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 6 * 1e3
        out.append({"name": "config3_mix_pan_fused", "what": "the headline's 65536 voices x 48000 frames in mode B (SURVEY 8(d)): per-voice equal-power pan + "
                    "sum over the voices inside the render launch (fdsp_bank_process_mix, FDSP_MIX_PAN), [2][48000] f32 out; not a bandwidth test",
                    "ms_per_step": round(ms, 4), "kernel_ms_avg_incl_tree": round(sum(k) / len(k), 4), "value": round(V * T / ms / 1e3, 1), "unit": "Msamples/s"})
        del wl, mixbuf
    except Exception as e:
        out.append({"name": "config3_mix_pan_fused", "error": repr(e)})
    # the reference's own harness (benches/benchmark.rs) as banks -- `python bench.py --criterion` is the same block alone, with longer CPU legs
    try:
        torch.cuda.empty_cache()
        out.append({"name": "reference_criterion_benches", **criterion_benches(F, torch, cpu_seconds=0.5, steps=2)})
    except Exception as e:
        out.append({"name": "reference_criterion_benches", "error": repr(e)})
    return out


# ---------------------------------------------------------------------------------------------------------------------
# the reference's own benches (benches/benchmark.rs, criterion) as banks
# ---------------------------------------------------------------------------------------------------------------------
CRITERION_VOICES = {"sine": 65536, "pass": 65536, "wavetable": 65536, "envelope": 65536, "oversample": 65536, "equalizer": 65536,
                    "reverb": 4096, "limiter": 65536, "phaser": 65536}


def _oracle_on_native_build():
    """tests/oracle.py's graph notation on the -O3 -march=native build of the oracle made on this host (oracle/Makefile `native`)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native"])
    native = os.path.join(ROOT, "oracle", "_native", "libfundsp_oracle_native.so")
    if getattr(O, "_native_path", None) != native:
        O._lib = None
        O.build = lambda: native
        O._native_path = native
    return O


def criterion_benches(F, torch, names=None, cpu_seconds=1.0, steps=3):
    """The reference's OWN benchmark harness (benches/benchmark.rs: criterion, each bench = Wave::render of 1 s of ONE graph at 44.1 kHz on one
    thread; no figures are published) with the graphs as BANKS of instances: ms per rendered second of V instances, the
    instance-seconds of audio per wall second that is (`x_real_time`), next to the oracle's C restatement of the same graph on this host --
    ONE instance on one thread (= what criterion times: `cpu_ms_per_instance_second`) and one instance per core on all cores.  Nine of the
    thirteen benches are graphs of nodes on the path (tests/criterion_graphs.py, each instance bit-equal to the oracle:
    tests/test_gpu_criterion.py); `reverb` also as the chain of two banks (fundsp_amd.Chain: generator kernel + lane-per-frame network kernel)."""
    import threading

    import numpy as np

    from fundsp_amd import graph as GR

    O = _oracle_on_native_build()
    import criterion_graphs as CG

    cores = host_cpu_budget()["effective_cpus"]
    T, sr = CG.FRAMES, CG.SAMPLE_RATE
    res = {"what": "benches/benchmark.rs (criterion: 1 s of one graph at 44.1 kHz per iteration) as banks of V instances with per-instance seeds; "
                   "x_real_time = instance-seconds of audio per wall second; cpu = the oracle's tree walk (gcc " + NATIVE_FLAGS + "), one instance per thread",
           "sample_rate": sr, "frames": T, "cpu_cores": cores, "not_on_the_path": CG.NOT_ON_THE_PATH, "benches": {}}

    def gpu_time(run):
        for _ in range(1):
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            run()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return min(ts)

    def cpu_leg(name):
        def make():
            n = CG.table(O, O)[name][0]
            n.set_sample_rate(sr)
            return n
        n = make()
        n.render_blocks(None, length=T, block=64)     # (tables, page faults)
        t0, k = time.perf_counter(), 0
        while time.perf_counter() - t0 < cpu_seconds or k < 2:
            n.render_blocks(None, length=T, block=64)
            k += 1
        one = (time.perf_counter() - t0) / k * 1e3
        nodes, counts = [make() for _ in range(cores)], [0] * cores
        stop = time.perf_counter() + cpu_seconds

        def work(j):
            while time.perf_counter() < stop or counts[j] < 1:
                nodes[j].render_blocks(None, length=T, block=64)   # (ctypes releases the GIL inside the C call)
                counts[j] += 1
        th = [threading.Thread(target=work, args=(j,)) for j in range(cores)]
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        return one, sum(counts) / (time.perf_counter() - t0)

    for name in (names or list(CG.table(O, O))):
        V = CRITERION_VOICES[name]
        e = {"benchmark_rs_line": CG.table(O, O)[name][2], "instances": V}
        try:
            g, ring, _ = CG.table(GR, O)[name]
            for kind in GR.uses_wavetables(g):
                F.wavetable_build(kind)
            t0 = time.perf_counter()
            seeds = np.arange(V, dtype=np.uint64) * 7919 + 13
            if name == "reverb":   # as one lane-per-voice graph (fewer instances: it is two orders slower) and as the chain
                Vg = 256
                b = F.Bank.from_graph(g, Vg, ring_frames=ring, sample_rate=sr, fdn_kernel=False)
                b.set_seed(seeds[:Vg])
                out = torch.empty((g.nout, T, Vg), dtype=torch.float32, device="cuda")
                ms1 = gpu_time(lambda: b.process(T, None, out, layout=F.LAYOUT_VOICE_MINOR))
                e["as_one_graph"] = {"instances": Vg, "ms_per_rendered_second": round(ms1, 3), "x_real_time": round(Vg / (ms1 * 1e-3), 1), "last_kernel": b.get_option("last_kernel")}
                del b, out
                t0 = time.perf_counter()
                ch = F.Bank.from_graph(g, V, sample_rate=sr)   # `generator >> stock reverb`: from_graph builds the chain by itself
                assert isinstance(ch, F.Chain) and ch.effect.kind == "reverb_stereo"
                ch.set_seed(seeds)
                e["compile_and_create_s"] = round(time.perf_counter() - t0, 2)
                out = torch.empty((V, 2, ch.frame_stride(T)), dtype=torch.float32, device="cuda")
                ms = gpu_time(lambda: ch.process(T, None, out, layout=F.LAYOUT_PLANAR))
                e["form"] = "Bank.from_graph -> fundsp_amd.Chain(noise | noise bank, reverb_stereo bank: fd::k_fdn_render_frames, lane = frame)"
                del ch, out
            else:
                b = F.Bank.from_graph(g, V, ring_frames=ring, sample_rate=sr)
                b.set_seed(seeds)
                e["compile_and_create_s"] = round(time.perf_counter() - t0, 2)
                out = torch.empty((g.nout, T, V), dtype=torch.float32, device="cuda")
                ms = gpu_time(lambda: b.process(T, None, out, layout=F.LAYOUT_VOICE_MINOR))
                e["last_kernel"] = b.get_option("last_kernel")
                del b, out
                if name == "sine":   # ... and on a small bank; both through the chain of waves per voice group (last_kernel 8) and through one wave per group (1)
                    Vs = 1024
                    bs = F.Bank.from_graph(g, Vs, sample_rate=sr)
                    bs.set_seed(seeds[:Vs])
                    outs = torch.empty((g.nout, T, Vs), dtype=torch.float32, device="cuda")
                    small = {"instances": Vs}
                    for key, split in (("chain_of_waves", 1), ("one_wave_per_voice_group", 0)):
                        bs.set_option("pipe_split", split)
                        m1 = gpu_time(lambda: bs.process(T, None, outs, layout=F.LAYOUT_VOICE_MINOR))
                        small[key] = {"ms_per_rendered_second": round(m1, 3), "x_real_time": round(Vs / (m1 * 1e-3), 1), "last_kernel": bs.get_option("last_kernel")}
                    e["small_bank"] = small
                    del bs, outs
            e["ms_per_rendered_second"] = round(ms, 3)
            e["x_real_time"] = round(V / (ms * 1e-3), 1)
            e["value"] = round(V * T / ms / 1e3, 1)
            e["unit"] = "Msamples/s"
        except Exception as ex:
            e["error"] = repr(ex)[:300]
        try:
            one, rate = cpu_leg(name)
            e["cpu_ms_per_instance_second"] = round(one, 3)
            e["cpu_x_real_time_all_cores"] = round(rate, 1)
            if "x_real_time" in e:
                e["gpu_over_cpu_all_cores"] = round(e["x_real_time"] / rate, 1)
        except Exception as ex:
            e["cpu_error"] = repr(ex)[:300]
        res["benches"][name] = e
    return res


class quiet_stdout:
    """RCCL prints a version banner to the C-level stdout when a communicator is created; the contract of this script is ONE
    JSON line on stdout.  Inside this context fd 1 points at stderr; C stdio is flushed before fd 1 is restored."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


class Peers:
    """How the ranks of one bench run meet: barrier, max-over-ranks, and the mix-down communicator.

    * world == 1: nothing to meet.
    * one process per GPU (the driver's `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`): torch.distributed
      carries the barrier, the max and -- once -- the 128-byte RCCL id; the mix-down itself goes through the product's own
      collective, fdsp_mix_allreduce (RCCL inside libfundsp_hip.so), NOT through torch.distributed.
    * one process driving N GPUs (`python bench.py --gpus N` as typed, no RANK in the environment): N host threads, a
      threading.Barrier, one local communicator (fdsp_comm_create_local = ncclCommInitAll), slot k = device k."""

    def __init__(self, world, rank=0, dist=None, thread_barrier=None, shared=None, comm=None, slot=0):
        self.world, self.rank, self.dist, self.tb, self.shared, self.comm, self.slot = world, rank, dist, thread_barrier, shared, comm, slot

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        elif self.tb is not None:
            self.tb.wait()

    def max(self, x, torch):
        if self.dist is not None:
            t = torch.tensor([x], dtype=torch.float64, device="cuda")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return float(t.item())
        if self.tb is not None:
            self.shared[self.rank] = x
            self.tb.wait()
            m = max(self.shared)
            self.tb.wait()
            return m
        return x


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spin-up", type=float, default=0.5, help="seconds of untimed rendering before the warm-up steps (clock ramp); 0 = none")
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5],
                    help="3 = headline FM+SVF voices (default); 2 / 4 / 5 = the other BASELINE configs as the timed workload (informational)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong (default, BASELINE metric): the config's voice count in total, split over the GPUs; weak: that many per GPU")
    ap.add_argument("--voices", type=int, default=None, help="total voices (strong) / voices per GPU (weak); default = the config's own")
    ap.add_argument("--frames", type=int, default=48000, help="frames per step (1 s @ 48 kHz)")
    ap.add_argument("--sample-rate", type=float, default=48000.0)
    ap.add_argument("--layout", choices=["voice_minor", "planar"], default="voice_minor")
    ap.add_argument("--mode", choices=["process", "tick"], default="process")
    ap.add_argument("--mix-mode", choices=["fused", "unfused"], default="fused", help="--mix: fused = the render kernel reduces over the voices itself (fdsp_bank_process_mix, no voice-out buffer); unfused = voice-out render + a second kernel")
    ap.add_argument("--mix", action="store_true", help="add the on-device stereo mix-down + fdsp_mix_allreduce (RCCL inside the library, side stream) per step")
    ap.add_argument("--pipe-split", type=int, default=1, choices=[0, 1, 2, 3],
                    help="pipeline split of Pipe-chain kinds: 0 off, 1 best plan (default), 2 / 3 = that many stages")
    ap.add_argument("--math", choices=["exact", "fast"], default="exact",
                    help="exact = the reference's arithmetic, bit-identical to the oracle (headline); fast = tolerance mode (FDSP_MATH_FAST)")
    ap.add_argument("--gate", choices=["var", "stream"], default="var",
                    help="--config 4: var = the reference's own gate shape `var(gate) >> adsr_live` (examples/live_adsr.rs:72): no graph input, one step = the "
                         "launches of one note with the Var slot set on the device in between (default); stream = the gate as an audio-rate HBM input [frames][voices]")
    ap.add_argument("--reverb", choices=["stereo", "4", "3"], default="stereo",
                    help="--config 5: stereo = reverb_stereo(10, 2, 0.5), BASELINE config 5 (default); 4 = reverb4_stereo(20, 2), two 16-line networks in series; 3 = reverb3_stereo(2, 0.5, lowpole_hz(8000)), the allpass-loop reverb (624 B per instance-frame)")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="wall-time budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--criterion", action="store_true", help="only the reference's own criterion benches (benches/benchmark.rs) as banks: one JSON line {\"criterion\": ..}")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements (configs 2 / 4 / 5, tolerance mode)")
    return ap.parse_args(argv)


def launch_plan(args, env, device_count):
    """How `bench.py --gpus N` runs.  -> ("single", None) | ("torchrun-rank", (rank, world, local_rank)) | ("threads", N).
    Raises SystemExit with a message about DEVICES (never about launchers) when the box has too few."""
    world = int(env.get("WORLD_SIZE", "1"))
    if "RANK" in env and world > 1:
        if world != args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus} was started under a launcher with WORLD_SIZE={world}: the two must agree")
        return "torchrun-rank", (int(env["RANK"]), world, int(env.get("LOCAL_RANK", "0")))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    n = device_count()
    if n < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: {args.gpus} HIP device{'s' if args.gpus > 1 else ''} needed, {n} present "
                         f"(the product path has no CPU fallback)")
    return ("single", None) if args.gpus == 1 else ("threads", args.gpus)


def main(argv=None):
    args = parse_args(argv)

    import torch

    plan, info = launch_plan(args, os.environ, lambda: torch.cuda.device_count() if torch.cuda.is_available() else 0)

    import fundsp_amd as F
    from fundsp_amd import _lib

    if args.pipe_split != 1:
        assert _lib.lib().fdsp_set_option(b"pipe_split", args.pipe_split) == 0
    if args.criterion:   # the reference's own harness as banks, nothing else (one GPU)
        torch.cuda.set_device(0)
        with quiet_stdout():
            res = criterion_benches(F, torch, cpu_seconds=max(0.5, min(2.0, args.cpu_seconds / 8.0)))
        print(json.dumps({"criterion": res}), flush=True)
        return
    if plan == "threads":   # one process, N GPUs, N host threads: `python bench.py --gpus N` as typed
        import threading

        n = info
        with quiet_stdout():
            comm = F.Comm.local(list(range(n))) if args.mix else None
        tb, shared, results, errors = threading.Barrier(n), [0.0] * n, [None] * n, []

        def body(k):
            try:
                torch.cuda.set_device(k)
                results[k] = run_rank(args, torch, F, Peers(n, k, thread_barrier=tb, shared=shared, comm=comm, slot=k), k)
            except BaseException as e:   # a dead thread must not leave the others waiting at the barrier
                errors.append(e)
                tb.abort()
        threads = [threading.Thread(target=body, args=(k,)) for k in range(n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        with quiet_stdout():
            if comm is not None:
                comm.close()
        print(json.dumps(results[0]), flush=True)
        return
    if plan == "torchrun-rank":
        import torch.distributed as dist

        rank, world, local_rank = info
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        with quiet_stdout():
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        comm = None
        if args.mix:   # the RCCL id travels over torch.distributed ONCE; every all-reduce of the run is the library's own
            idt = torch.zeros(128, dtype=torch.uint8, device="cuda")
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(F.Comm.unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, src=0)
            with quiet_stdout():
                comm = F.Comm.rank(bytes(idt.cpu().numpy().tobytes()), world, rank, local_rank)
        with quiet_stdout():
            res = run_rank(args, torch, F, Peers(world, rank, dist=dist, comm=comm), local_rank)
        if rank == 0:
            print(json.dumps(res), flush=True)
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()
        return
    torch.cuda.set_device(0)
    with quiet_stdout():   # everything C-level that talks (RCCL's banner) goes to stderr; the JSON line is printed last
        comm = F.Comm.local([0]) if args.mix else None
        res = run_rank(args, torch, F, Peers(1, 0, comm=comm), 0)
        if comm is not None:
            comm.close()
    print(json.dumps(res), flush=True)


HEADLINE_KERNEL_SOURCES = ("fd_math.hpp", "fd_nodes.hpp", "fd_device.hpp", "fd_engine.hpp", "fd_opts.hpp", "fd_kinds_fm.hpp", "fd_kinds_fm.hip")


def headline_kernel_source_hash():
    """sha256 over the files the headline kernel (fd::k_render_pipe<fm_svf ..>, fd_kinds_fm.hip) is compiled from, in a fixed order:
    what profiles/pmc_latest.json must carry for its HBM traffic to be quoted as this kernel's (tools/pmc_latest.py writes it)."""
    import hashlib

    h = hashlib.sha256()
    for name in HEADLINE_KERNEL_SOURCES:
        h.update(name.encode() + b"\0")
        h.update(open(os.path.join(ROOT, "fundsp_amd", "csrc", name), "rb").read())
    # ... and from the COMMAND that compiles it (compiler, flags, per-TU scheduling strategy), as `make -n` spells it -- not the Makefile's bytes:
    # adding an unrelated translation unit to the library does not make the headline kernel another kernel (round 6)
    csrc = os.path.join(ROOT, "fundsp_amd", "csrc")
    try:
        cmd = subprocess.run(["make", "-C", csrc, "-n", "-B", "fd_kinds_fm.o"], capture_output=True, text=True, timeout=60).stdout
        cmd = "\n".join(line.strip() for line in cmd.splitlines() if "fd_kinds_fm" in line and not line.startswith("make"))
    except Exception:
        cmd = ""
    if not cmd:   # no make on this box: the Makefile's bytes (a stricter stamp)
        cmd = open(os.path.join(csrc, "Makefile")).read()
    h.update(b"build\0" + cmd.encode())
    return h.hexdigest()


def cpu_baseline_wanted(args, rank, world):
    """The headline's cpu_baseline leg runs on rank 0 of EVERY run of the headline config, whatever the number of GPUs."""
    return rank == 0 and world >= 1 and args.cpu_seconds > 0 and args.config == 3


def fused_mix(args, F, layout):
    """--mix takes the fused mix-down (the render kernel reduces over the voices itself) wherever the bank has it: the voice-minor
    configs 2 / 3 / 4.  --mix-mode unfused keeps round 3's shape (voice-out render, then a second kernel over it)."""
    return bool(args.mix) and args.mix_mode == "fused" and args.config in (2, 3, 4) and layout == F.LAYOUT_VOICE_MINOR


def mix_step(F, torch, args, wl, peers, mode):
    """The path's one exchange step: the per-GPU stereo mix-down, then ONE all-reduce(sum) of [2][frames] f32 through
    fdsp_mix_allreduce on the communicator's side stream -- the next render does not wait for it.  Fused (default): the mix-down
    IS the render launch (fdsp_bank_process_mix; config 4's voices end in a Panner -> FDSP_MIX_SUM, the mono configs are panned
    per voice -> FDSP_MIX_PAN).  Unfused: a second kernel over the voice-out buffer, same summation order."""
    if wl["out"] is None:
        mix = wl["bank"].process_mix(args.frames, wl["inp"], mix=F.MIX_SUM if args.config == 4 else F.MIX_PAN, mode=mode)
    elif args.config in (2, 3):
        mix = F.mix_stereo(wl["out"][0] if wl["layout"] == F.LAYOUT_VOICE_MINOR else wl["out"][:, 0, :].t().contiguous())
    elif args.config == 5:
        mix = F.sum_instances(wl["out"])   # [2][T] sum over the instances (planar layout), the mix-down's tree order
    else:
        mix = F.sum_voices(wl["out"])  # voices are already panned to stereo
    peers.comm.allreduce(mix, slot=peers.slot)
    return mix


class PowerMeter:
    """Socket energy / power / shader clock over the timed region, straight from librocm_smi64 (ctypes; measurement only).
    Why it is in the line: the headline kernel runs the socket AT ITS POWER CAP (profiles/r03_power_bound.txt) -- the shader
    clock during the run is what the power manager leaves, so joules per step bound the step time, not cycles.  Everything
    here degrades to None when the library or a counter is missing."""

    class _Freq(ctypes.Structure):
        _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32),
                    ("frequency", ctypes.c_uint64 * 33)]

    def __init__(self, device=0):
        self.dev, self.lib, self.samples = int(device), None, []
        try:
            lib = ctypes.CDLL("librocm_smi64.so")
            if lib.rsmi_init(ctypes.c_uint64(0)) == 0:
                self.lib = lib
        except OSError:
            self.lib = None

    def energy_j(self):
        if self.lib is None:
            return None
        c, res, ts = ctypes.c_uint64(0), ctypes.c_float(0), ctypes.c_uint64(0)
        if self.lib.rsmi_dev_energy_count_get(ctypes.c_uint32(self.dev), ctypes.byref(c), ctypes.byref(res), ctypes.byref(ts)) != 0:
            return None
        return c.value * float(res.value) * 1e-6

    def cap_w(self):
        if self.lib is None:
            return None
        cap = ctypes.c_uint64(0)
        if self.lib.rsmi_dev_power_cap_get(ctypes.c_uint32(self.dev), ctypes.c_uint32(0), ctypes.byref(cap)) != 0:
            return None
        return cap.value * 1e-6

    def sclk_mhz(self):
        if self.lib is None:
            return None
        f = PowerMeter._Freq()
        if self.lib.rsmi_dev_gpu_clk_freq_get(ctypes.c_uint32(self.dev), ctypes.c_uint32(0), ctypes.byref(f)) != 0 or f.current >= 33:
            return None
        return f.frequency[f.current] * 1e-6

    def start(self):
        """begin of the window (the spin-up: the same launches as the timed steps, back to back)"""
        self.samples = []
        self.e0, self.t0 = self.energy_j(), time.perf_counter()

    def sample_clock(self):
        """one shader-clock reading while the GPU is still working through the queued steps"""
        v = self.sclk_mhz()
        if v:
            self.samples.append(v)

    def stop(self, steps_in_window):
        e1, t1 = self.energy_j(), time.perf_counter()
        if self.e0 is None or e1 is None or e1 <= self.e0:
            return None
        watts, cap = (e1 - self.e0) / (t1 - self.t0), self.cap_w()
        return {"socket_watts": round(watts, 1), "cap_watts": cap, "frac_of_cap": round(watts / cap, 4) if cap else None,
                "joules_per_step": round((e1 - self.e0) / steps_in_window, 4), "window_s": round(t1 - self.t0, 3),
                "sclk_mhz": round(sum(self.samples) / len(self.samples), 0) if self.samples else None, "sclk_mhz_peak": 2400,
                "source": "librocm_smi64: the socket's energy accumulator from the start of the spin-up to the end of the timed region "
                          "(three library calls in all: a 100 Hz sampler thread cost the run 1-2 %), sclk read once while the last steps were queued",
                "reading": "power-bound when socket_watts is within a few per cent of cap_watts and sclk is below its peak: "
                           "profiles/r03_power_bound.txt (energy per instruction class and per HBM byte, the step's energy budget)"}


def run_rank(args, torch, F, peers, device):
    """One rank's share of the run (a process under torch.distributed.run, or a host thread of a one-process run).
    Returns the result record on rank 0, None elsewhere."""
    from fundsp_amd import dist as fdist
    from fundsp_amd import workloads as W

    rank, world = peers.rank, peers.world
    distributed = world > 1
    base_voices = args.voices or {2: 1024, 3: TOTAL_VOICES, 4: 32768, 5: 2048}[args.config]
    T, sr = args.frames, args.sample_rate
    layout = F.LAYOUT_VOICE_MINOR if args.layout == "voice_minor" else F.LAYOUT_PLANAR
    mode = F.MODE_PROCESS if args.mode == "process" else F.MODE_TICK

    def fence():
        torch.cuda.synchronize()
        peers.barrier()
        torch.cuda.synchronize()

    def timed_run(scaling, steps, warmup):
        """K steps of the rank's shard between two fences; returns (max-over-ranks seconds, kernel ms list, workload, V)."""
        if scaling == "strong":
            first, V = fdist.shard_range(base_voices, rank, world)   # contiguous ranges of the whole-node bank
        else:
            first, V = rank * base_voices, base_voices
        fused = fused_mix(args, F, layout)
        cfg = "4v" if args.config == 4 and args.gate == "var" else "5r4" if args.config == 5 and args.reverb == "4" else "rv3" if args.config == 5 and args.reverb == "3" else args.config
        wl = make_workload(F, W, torch, cfg, V, T, sr, first, layout, args.math, voice_out=not fused)
        bank = wl["bank"]
        mixes = []
        plan = wl.get("plan")
        if fused:
            bank.mix_reserve(T)
            # mode B of SURVEY 8(d): the algorithmic bytes are the inputs and the [2][T] mix -- "not a bandwidth test"
            wl["bps"] = 4 if cfg == 4 else 0
            wl["kernel"] = wl["kernel"].replace("fd::k_render_pipe<", "fd::k_render_pipe_mix<") + " + fd::k_mix_tree (fused mix-down: no voice-out buffer)"
        step_kernel_ms = [0.0]
        plan_kernel_ms = [False]   # inside the timed region the plan's launches are enqueued back to back; their HIP-event times are read in extra steps after it

        def step():
            if plan is not None:
                # the Var-gate shape: one note = the plan's launches with the shared variable set before each; the step's mix (mode B) is
                # assembled from the launches' [2][n] pieces before its ONE all-reduce
                if fused:
                    pieces = [torch.empty((2, n), dtype=torch.float32, device="cuda") for _, n in plan]
                    step_kernel_ms[0] = run_plan(wl, mode, F.MIX_SUM, pieces, kernel_ms=plan_kernel_ms[0])
                    mix = torch.cat(pieces, dim=1)
                else:
                    step_kernel_ms[0] = run_plan(wl, mode, kernel_ms=plan_kernel_ms[0])
                    mix = None
                if args.mix:
                    if mix is None:
                        mix = torch.cat([F.sum_voices(o) for o in wl["outs"]], dim=1)
                    peers.comm.allreduce(mix, slot=peers.slot)
                    mixes.append(mix)
                    del mixes[:-2]
                return
            if not fused:
                bank.process(T, wl["inp"], wl["out"], layout=wl["layout"], frame_stride=wl["fs"], mode=mode)
            if args.mix:
                mixes.append(mix_step(F, torch, args, wl, peers, mode))
                del mixes[:-2]   # the side stream may still be summing the previous one

        # untimed spin-up before the W warm-up steps: an idle MI355X sits at a 600 MHz shader clock and needs a few
        # hundred ms of work to reach its operating point (W = 2 steps are 10 ms; measured 5.27 vs 5.18 ms/step)
        meter = PowerMeter(device) if rank == 0 else None
        if meter is not None:
            meter.start()
        window_steps = 0
        t_spin = time.perf_counter() + args.spin_up
        while time.perf_counter() < t_spin:
            step()
            torch.cuda.synchronize()
            window_steps += 1
        for _ in range(warmup):
            step()
        fence()
        kernel_ms = []
        # Plain voice-out steps (the headline): the K launches are enqueued back to back on ONE stream, each between a pair of HIP events
        # recorded on that stream, and nothing waits inside the timed region -- the durations are read after the closing fence.  (Rounds 1-5
        # read the library's own event pair after every step, which made the host wait for each launch: 20-30 us of idle GPU per step on a
        # quiet host, 0.9 ms per step on a box whose host was busy -- profiles/r06_bench_default_f.json.)  Steps with a mix-down keep the
        # per-step read: their kernel figure is the render launch alone, which only the library's pair brackets.
        pairs = None
        if plan is None and not args.mix:
            pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for i in range(steps):
            if pairs is not None:
                pairs[i][0].record()
            step()
            if pairs is not None:
                pairs[i][1].record()
            if meter is not None and i == steps // 2:
                meter.sample_clock()   # (a library call, no device wait)
            # mix-down steps: HIP events recorded by the C ABI on the launch stream around the render kernel (reading here synchronises
            # on that launch, which the next step's launch on the same stream is ordered behind anyway)
            if plan is None and pairs is None:
                kernel_ms.append(bank.last_kernel_ms())
        if args.mix:
            peers.comm.wait(peers.slot)   # the last all-reduce belongs to the timed region
        fence()
        elapsed = peers.max(time.perf_counter() - t0, torch)
        if pairs is not None:
            kernel_ms = [a.elapsed_time(b) for a, b in pairs]
        if plan is not None:   # the kernels' own HIP-event times: three more steps of the same work, waiting for every launch
            plan_kernel_ms[0] = True
            for _ in range(3):
                step()
                kernel_ms.append(step_kernel_ms[0])
            if args.mix:
                peers.comm.wait(peers.slot)
            fence()
        wl["power"] = meter.stop(window_steps + warmup + steps) if meter is not None else None
        return elapsed, kernel_ms, wl, V

    # everything of the timed run goes to ONE non-default stream: Bank.process launches on torch's current stream when that is not the
    # default one (the default stream's handle is 0 = "the bank's own stream"), and the event pairs around the launches must sit on it too
    run_stream = torch.cuda.Stream()
    with torch.cuda.stream(run_stream):
        elapsed, kernel_ms, wl, V = timed_run(args.scaling, args.steps, args.warmup)
    total_voices = base_voices if args.scaling == "strong" else base_voices * world
    scaling_alt = None
    if distributed and args.config == 3:   # the other scaling law, outside the timed region, a few steps
        other = "weak" if args.scaling == "strong" else "strong"
        kernel_label, shard_bps, shard_slots, power = wl["kernel"], wl["bps"], wl["slot_bytes"], wl.get("power")
        del wl
        e2, _, wl2, V2 = timed_run(other, max(3, args.steps // 2), 1)
        tv2 = base_voices if other == "strong" else base_voices * world
        n2 = max(3, args.steps // 2)
        scaling_alt = {"scaling": other, "total_voices": tv2, "voices_per_gpu": V2, "value": round(tv2 * T * n2 / e2 / 1e6, 3),
                       "ms_per_step": round(e2 / n2 * 1e3, 4)}
        del wl2
    else:
        kernel_label, shard_bps, shard_slots, power = wl["kernel"], wl["bps"], wl["slot_bytes"], wl.get("power")
        del wl
    if rank != 0:
        return None

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
        total_samples = float(total_voices) * T * args.steps
        value = total_samples / elapsed / 1e6
        avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        # ALGORITHMIC bytes of one launch of this rank's shard: per-unit figure x units + slots
        algo_bytes = V * T * shard_bps + V * shard_slots
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
        # HBM traffic comes from separate rocprofv3 --pmc passes of this same command (it cannot be collected inside this
        # process): the committed summary is quoted, with its source, when it was taken on this very shard shape
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("voices") == V and rec.get("frames") == T and rec.get("config", 3) == args.config and rec.get("math", "exact") == args.math:
                    # ... and on THIS kernel: the record names the kernel it counted and carries the hash of the sources that kernel is
                    # compiled from; a record taken on another build of the headline kernel is not quoted
                    if rec.get("kernel_source_sha256") == headline_kernel_source_hash() and "k_render_pipe" in rec.get("kernel", ""):
                        traffic = rec.get("hbm_bytes_per_launch")
                        traffic_source = rec.get("source", "profiles/pmc_latest.json") + " (separate rocprofv3 --pmc passes, not this run; same kernel sources: sha256 " + rec["kernel_source_sha256"][:12] + ")"
                    else:
                        traffic_source = ("profiles/pmc_latest.json was counted on another build of the kernel (its kernel_source_sha256 differs from this tree's): "
                                          "not quoted -- rerun tools/pmc_hbm_pass.sh + tools/pmc_latest.py")
            except Exception:
                traffic = None
        # the VALU view (the kernel is issue-bound, so the HBM fraction alone says little): instruction counts
        # and cycles from the committed SQ pass + ISA listing (tools/valu_view.py), the flop rate from THIS run's kernel time
        valu = None
        vpath = os.path.join(ROOT, "profiles", "valu_latest.json")
        if os.path.exists(vpath):
            try:
                rec = json.load(open(vpath))
                if rec.get("voices") == V and rec.get("frames") == T and rec.get("config", 3) == args.config and rec.get("math", "exact") == args.math:
                    flops = rec["flops_per_voice_frame_isa"] * V * T / (avg_ms * 1e-3)
                    valu = {k: rec[k] for k in ("insts_per_voice_group_frame", "packed_fraction_isa", "plain_op_equivalents_per_voice_group_frame",
                                                "issue_cycles_per_inst", "frac_of_issue_peak", "issue_peak", "wait_inst_any_frac_of_wave_cycles",
                                                "wait_any_frac_of_wave_cycles", "source")}
                    valu["tflops_this_run"] = round(flops / 1e12, 2)
                    valu["flops_frac_of_157TF"] = round(flops / 157.3e12, 4)
            except Exception:
                valu = None
        unit_name = {2: "voices", 3: "voices", 4: "voices", 5: "instances"}[args.config]
        res = {
            "metric": {3: "Msamples/s (whole node) for 65536-voice SVF+FM graph",
                       2: "Msamples/s (whole node) for the 1024-voice biquad_bank on white noise (BASELINE config 2, informational)",
                       4: "Msamples/s (whole node) for saw>>moog*adsr>>pan voices (BASELINE config 4, informational)",
                       5: f"M instance-frames/s (whole node) for {'reverb4_stereo' if args.reverb == '4' else 'reverb3_stereo' if args.reverb == '3' else 'reverb_stereo'} instances (BASELINE config 5{' shape, the four-channel-network reverb' if args.reverb == '4' else ' shape, the allpass-loop reverb' if args.reverb == '3' else ''}, informational)"}[args.config],
            "value": round(value, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "spin_up_s": args.spin_up,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": {3: "BASELINE config 3: sine_hz(f)*f*m+f >> sine() >> lowpass_hz(fc,q), ",
                             2: "BASELINE config 2: noise() >> lowpass biquad (BiquadBank<f32x8> lane per voice), ",
                             4: ("BASELINE config 4 voice: ((dc(f)>>saw()|dc(fc)|dc(q))>>moog())*(var(gate)>>adsr_live(.01,.1,.6,.2))>>pan(p), the gate a shared variable "
                                 "set between the launches of one note (high 0.5 s, low 0.5 s), " if args.gate == "var" else
                                 "BASELINE config 4 voice: ((dc(f)>>saw()|dc(fc)|dc(q))>>moog())*adsr_live(.01,.1,.6,.2)>>pan(p), gate in as an audio-rate stream, "),
                             5: ("reverb4_stereo(20.0, 2.0): two 16-line FDNs in series (prelude.rs:1873-1941), stereo noise in, planar I/O, " if args.reverb == "4" else
                                 "reverb3_stereo(2.0, 0.5, lowpole_hz(8000.0)): the allpass-loop reverb (prelude.rs:1850-1871, reverb.rs:152-279), stereo noise in, planar I/O, " if args.reverb == "3" else
                                 "BASELINE config 5: reverb_stereo(10.0, 2.0, 0.5) 32-line FDN, stereo noise in, planar I/O, ")}[args.config] +
                            f"{total_voices} {unit_name} in total = {V} per GPU x {T} frames/step @ {sr:g} Hz, "
                            f"{'planar ([' + unit_name[:-1] + '][channel][frame] f32)' if args.config == 5 or args.layout == 'planar' else ('mix-out ([2][frame] f32, fused stereo mix-down: mode B)' if fused_mix(args, F, layout) else 'voice-out ([frame][voice] f32)')}, "
                            f"{args.mode} semantics, {args.math} arithmetic, per-voice params from rnd1(4v+k), phases via set_seed(v)",
                "total_voices": total_voices,
                "voices_per_gpu": V,
                "frames_per_step": T,
                "layout": "planar" if args.config == 5 else args.layout,
                "mix_allreduce": bool(args.mix),
                "mix_mode": (("fused (fdsp_bank_process_mix)" if fused_mix(args, F, layout) else "unfused (voice-out render + a second kernel)") if args.mix else None),
                "mix_collective": ("fdsp_mix_allreduce (RCCL inside libfundsp_hip.so, communicator side stream)" if args.mix else None),
                "launch": ("one process per GPU (torch.distributed.run)" if peers.dist is not None else
                           "one process, one host thread per GPU" if world > 1 else "one process, one GPU"),
                "math": args.math,
                "parallelism": f"voice-shard x{world}",
                "scaling_alt": scaling_alt,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": kernel_label,
                "kernel_ms_avg": round(avg_ms, 4),
                "algorithmic_bytes_per_launch": algo_bytes,
                "measured_streaming": measured,
                "frac_of_measured_fill": round(achieved / measured["fill_gbs"], 4) if measured else None,
                "valu": valu,
                "power": power,
            },
        }
        # the CPU path "in the same run" (north_star): on rank 0, after the timed region -- at N > 1 too (the other ranks wait at the
        # run's last barrier)
        if cpu_baseline_wanted(args, peers.rank, world):
            res["cpu_baseline"] = cpu_baseline(total_voices, T, sr, args.cpu_seconds)
        else:
            res["cpu_baseline"] = None
        if world == 1 and not args.no_secondary and args.config == 3 and args.math == "exact":
            try:
                res["secondary"] = secondary(F, W, torch, sr, mode)
                if measured:  # the HBM-side entries next to what this box's memory system streams (the fill / copy kernels timed above)
                    for e in res["secondary"]:
                        if isinstance(e, dict) and "roofline_frac" in e:
                            e["frac_of_measured_fill"] = round(e["roofline_frac"] * HBM_PEAK_GBS / measured["fill_gbs"], 4)   # write-only kernels
                            e["frac_of_measured_copy"] = round(e["roofline_frac"] * HBM_PEAK_GBS / measured["copy_gbs"], 4)   # 1:1 read/write (the reverbs' rings)
            except Exception as e:  # never lose the headline line to a secondary measurement
                res["secondary"] = [{"error": repr(e)}]
        return res


if __name__ == "__main__":
    main()
