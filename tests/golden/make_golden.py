#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (NOT from the reference: /root/reference is Rust and there is
no rustc/cargo in the build image, so it cannot be executed to produce vectors).

The fixtures freeze the oracle's current behaviour so that (a) any later edit of the oracle that changes a bit is
caught by the CPU suite and (b) the GPU suite has committed vectors to compare against besides the live oracle.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O  # noqa: E402
from fundsp_amd import workloads as W  # noqa: E402

SR = 48000.0


def main():
    # config 1: Wave::render of sine_hz(440) >> lowpass_hz(1000, 1), first 2048 samples at 48 kHz
    g = O.sine_hz(440.0) >> O.lowpass_hz(1000.0, 1.0)
    c1 = O.wave_render(SR, 2048 / SR, g)[0]
    # config 2: 64 noise>>biquad voices, 200 frames, process and tick path
    p2 = W.noise_biquad_params(64, SR)
    c2p, _ = O.bank_render(2, [p2["fc"], p2["q"]], p2["seed"], 200, SR, True, 0)
    c2t, _ = O.bank_render(2, [p2["fc"], p2["q"]], p2["seed"], 200, SR, False, 0)
    # config 3: 64 FM voices, 333 frames (ragged tail), process and tick path
    p3 = W.fm_svf_params(64, SR)
    c3p, _ = O.bank_render(3, [p3["f"], p3["m"], p3["fc"], p3["q"]], p3["seed"], 333, SR, True, 0)
    c3t, _ = O.bank_render(3, [p3["f"], p3["m"], p3["fc"], p3["q"]], p3["seed"], 333, SR, False, 0)
    np.savez_compressed(os.path.join(HERE, "configs_v1.npz"), config1=c1, config2_process=c2p, config2_tick=c2t,
                        config3_process=c3p, config3_tick=c3t)
    print("wrote", os.path.join(HERE, "configs_v1.npz"))
    gv = graph_vectors()
    np.savez_compressed(os.path.join(HERE, "graphs_v1.npz"), **gv)
    print("wrote", os.path.join(HERE, "graphs_v1.npz"))
    harness_inputs(p2, p3, gv)
    cv = criterion_vectors()
    np.savez_compressed(os.path.join(HERE, "criterion_v1.npz"), **cv)
    print("wrote", os.path.join(HERE, "criterion_v1.npz"))


def kat_args():
    """Argument list of the transcendental known-answer tables (rust_harness: libm::*f and wide::f32x8::*): every
    quadrant case of musl's sinf/cosf/tanf, the medium and the Payne-Hanek range, tanhf/expf breakpoints, specials."""
    rng = np.random.default_rng(777)
    parts = [
        np.linspace(-7.2, 7.2, 1441),                       # explicit quadrant cases up to 9pi/4
        np.linspace(-200.0, 200.0, 801),                    # Sine::process unwrapped phases
        rng.uniform(-3000.0, 3000.0, 512),                  # __rem_pio2f medium path
        10.0 ** rng.uniform(3.5, 8.5, 256),                 # up to the medium / large boundary
        10.0 ** rng.uniform(8.5, 38.4, 512),                # __rem_pio2_large
        -(10.0 ** rng.uniform(8.5, 38.4, 128)),
        2.0 ** np.arange(-30, 128, dtype=np.float64),       # powers of two
        np.array([0.0, -0.0, 1e-45, -1e-45, 1e-41, 1e-38, 1.17549435e-38, 3.4028235e38, -3.4028235e38, np.inf, -np.inf,
                  np.nan, 0.5493, -0.5493, 9.0, 10.0, 88.72, 88.73, -103.9, -104.0, 1e-5, 2.0 ** -12, 0.785398, 2.356194,
                  3.926990, 5.497787, 7.068583]),
    ]
    return np.concatenate(parts).astype(np.float32)


C4_V, C4_ADSR = 16, (0.005, 0.01, 0.6, 0.01)
C4_PLAN = [(0.0, 64), (1.0, 64 * 9), (0.0, 64 * 3 + 7), (0.5, 64 * 5), (0.0, 64 * 6 + 3)]
C4_FRAMES = sum(n for _, n in C4_PLAN)
C5_FRAMES = 64 * 200 + 13


def config45_inputs():
    """What rust_harness reads for configs 4 / 5 (and what tests/test_golden.py renders through the oracle): config-4 voice parameters,
    the gate stream of the stream-gate shape, the (value, frames) plan of the Var shape, one stereo noise input for the reverbs."""
    p4 = W.saw_moog_params(C4_V, SR)
    gate = W.gate_signal(C4_FRAMES, SR, on_frame=1, off_seconds=700 / SR)
    rng = np.random.default_rng(4242)
    x5 = (rng.random((2, C5_FRAMES), dtype=np.float32) * 2 - 1).astype(np.float32)
    x5[:, 2 * C5_FRAMES // 3:] = 0.0
    return p4, gate, x5


def harness_inputs(p2, p3, gv):
    """The exact arrays rust_harness/ reads (raw little-endian): per-voice parameters of configs 2 / 3, graph inputs,
    the KAT argument list -- so that the real reference renders from identical bits."""
    d = os.path.join(os.path.dirname(os.path.dirname(HERE)), "rust_harness", "inputs")
    os.makedirs(d, exist_ok=True)
    for k in ("f", "m", "fc", "q"):
        p3[k].astype("<f4").tofile(os.path.join(d, f"config3_{k}.f32"))
    p3["seed"].astype("<u8").tofile(os.path.join(d, "config3_seed.u64"))
    for k in ("fc", "q"):
        p2[k].astype("<f4").tofile(os.path.join(d, f"config2_{k}.f32"))
    p2["seed"].astype("<u8").tofile(os.path.join(d, "config2_seed.u64"))
    kat_args().astype("<f4").tofile(os.path.join(d, "kat_args.f32"))
    p4, gate, x5 = config45_inputs()
    for k in ("f", "fc", "q", "pan"):
        p4[k].astype("<f4").tofile(os.path.join(d, f"config4_{k}.f32"))
    p4["seed"].astype("<u8").tofile(os.path.join(d, "config4_seed.u64"))
    gate.astype("<f4").tofile(os.path.join(d, "config4_gate.f32"))
    np.array([[v, n] for v, n in C4_PLAN], dtype="<f4").tofile(os.path.join(d, "config4_plan.f32"))
    x5.astype("<f4").tofile(os.path.join(d, "config5_in.f32"))
    from test_gpu_jit import GRAPHS
    for name, (_b, ni, _r) in GRAPHS.items():
        if ni:
            gv[name + "__in"][:ni].astype("<f4").tofile(os.path.join(d, f"graph_{name}_in.f32"))
    print("wrote", d)


GRAPH_FRAMES = 64 * 3 + 9


def graph_vectors():
    """One short oracle render (process and tick semantics) of every graph of tests/test_gpu_jit.py::GRAPHS -- the leaf,
    combinator and composed-opcode inventory in one place -- with seed 12345 and a fixed noise input."""
    from test_gpu_jit import GRAPHS

    out = {}
    rng = np.random.default_rng(2024)
    for name, (build, ni, _ring) in GRAPHS.items():
        x = (rng.random((max(ni, 1), GRAPH_FRAMES), dtype=np.float32) * 2 - 1).astype(np.float32)
        if name == "saw_filter_env":
            x[0] = 0.0
            x[0, 3:150] = 1.0
        out[name + "__in"] = x
        for mode in ("process", "tick"):
            n = build(O)
            n.set_sample_rate(SR)
            n.set_seed(12345)
            xin = x[:ni] if ni else None
            out[f"{name}__{mode}"] = n.render_blocks(xin, length=GRAPH_FRAMES) if mode == "process" else n.render_ticks(xin, length=GRAPH_FRAMES)
    return out


CRITERION_WINDOWS = {"limiter": (4352, 5440)}   # (the limiter looks 4 410 frames ahead: silence before); every other bench: the first 1 088 frames


def criterion_vectors():
    """The reference's own bench graphs (benches/benchmark.rs; tests/criterion_graphs.py) at 44.1 kHz: a window of the second each bench renders,
    AS CONSTRUCTED (no set_seed: the hashes the constructors' pings hand down -- the reverb bench pins the walk of the tree the oracle's
    reverb_stereo node stands for) and after set_seed(7), process executor; the first 300 frames in the tick executor."""
    import criterion_graphs as CG

    out = {}
    for name in CG.table(O, O):
        a, b = CRITERION_WINDOWS.get(name, (0, 1088))
        for tag, seed in (("ctor", None), ("seed7", 7)):
            n = CG.table(O, O)[name][0]
            n.set_sample_rate(CG.SAMPLE_RATE)
            if seed is not None:
                n.set_seed(seed)
            out[f"{name}__{tag}__process"] = n.render_blocks(None, length=b, block=64)[:, a:b]
        n = CG.table(O, O)[name][0]
        n.set_sample_rate(CG.SAMPLE_RATE)
        out[f"{name}__ctor__tick"] = n.render_ticks(None, length=300)
    return out


if __name__ == "__main__":
    main()
