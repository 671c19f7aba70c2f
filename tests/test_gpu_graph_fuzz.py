"""Randomly composed voice graphs: every combinator of the notation over a pool of leaves, built twice from one
expression tree -- with the engine's notation (fundsp_amd.graph, compiled at run time) and with the oracle's -- and
compared bit for bit in process and tick semantics.  What this pins beyond the hand-written cases is the interplay:
arity bookkeeping, parameter paths and, above all, the ping order that gives every oscillator / noise source its hash
(audionode.rs:156-161 and each combinator's ping), for shapes nobody wrote down by hand."""
import os

import numpy as np
import pytest

import oracle as O
from fundsp_amd import LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MODE_PROCESS, MODE_TICK
from fundsp_amd import graph as GR
from test_gpu_parity import assert_bit_equal, noise_input, oracle_render, run_bank

pytestmark = pytest.mark.gpu
SR = 48000.0


def leaf(rng, nin, nout):
    r = lambda lo, hi: float(np.float32(rng.uniform(lo, hi)))
    pool = {
        (0, 1): [lambda: ("sine_hz", r(50, 3000)), lambda: ("noise",), lambda: ("dc", r(-1, 1)), lambda: ("impulse",),
                 lambda: ("poly_saw_hz", r(50, 2000)), lambda: ("mls",)],
        (1, 1): [lambda: ("lowpass_hz", r(200, 8000), r(0.5, 3)), lambda: ("highpole_hz", r(20, 2000)),
                 lambda: ("lowpole_hz", r(100, 9000)), lambda: ("shape_tanh", r(0.5, 3)), lambda: ("pass_",),
                 lambda: ("mul", r(-1.5, 1.5)), lambda: ("follow", r(0.001, 0.05)), lambda: ("tick",),
                 lambda: ("declick_s", r(0.001, 0.01)), lambda: ("bell_hz", r(200, 5000), r(0.5, 2), r(0.5, 2)),
                 lambda: ("moog_hz", r(300, 4000), r(0.0, 0.6)), lambda: ("meter_rms", r(0.005, 0.05)),
                 # delay lines (ring memory per voice) and feedback loops (the whole graph then renders with flushed denormals)
                 lambda: ("delay", r(0.0001, 0.004)), lambda: ("allnest_delay", r(-0.7, 0.7), r(0.0001, 0.003)),
                 lambda: ("echo", r(0.0002, 0.003), r(-0.8, 0.8)), lambda: ("tap_dc", r(0.0005, 0.003))],
        (1, 2): [lambda: ("pan", r(-1, 1)), lambda: ("split", 2)],
        (2, 1): [lambda: ("join", 2)],
        (2, 2): [lambda: ("reverse", 2), lambda: ("multipass", 2), lambda: ("rotate", r(0, 3), r(0.3, 1.0))],
        (0, 2): [lambda: ("dc2", r(-1, 1), r(-1, 1))],
        (1, 0): [lambda: ("sink",)],
        (2, 0): [lambda: ("multisink", 2)],
    }
    opts = pool.get((nin, nout))
    return None if not opts else opts[rng.integers(len(opts))]()


def gen(rng, nin, nout, depth):
    """Expression tree for a node with `nin` inputs and `nout` outputs (0 <= nin, nout <= 2)."""
    choices = []
    if depth > 0:
        choices += ["pipe"] * 3
        if nin + nout >= 2 and (nin > 0 or nout > 1):
            choices.append("stack")
        if nout >= 1:
            choices += ["bus", "binop", "unop"]
        if nout == 2:
            choices.append("branch")
        if nin == nout and nin > 0:
            choices.append("thru")
    lf = leaf(rng, nin, nout)
    if lf is not None:
        choices += ["leaf"] * (2 if depth > 0 else 50)
    for _ in range(50):
        c = choices[rng.integers(len(choices))]
        if c == "leaf":
            return lf
        if c == "pipe":
            k = int(rng.integers(1, 3))
            return ("pipe", gen(rng, nin, k, depth - 1), gen(rng, k, nout, depth - 1))
        if c == "stack":
            a, b = int(rng.integers(0, nin + 1)), int(rng.integers(0, nout + 1))
            if (a, b) in ((0, 0), (nin, nout)):
                continue
            return ("stack", gen(rng, a, b, depth - 1), gen(rng, nin - a, nout - b, depth - 1))
        if c == "bus":
            return ("bus", gen(rng, nin, nout, depth - 1), gen(rng, nin, nout, depth - 1))
        if c == "branch":
            return ("branch", gen(rng, nin, 1, depth - 1), gen(rng, nin, 1, depth - 1))
        if c == "thru":
            return ("thru", gen(rng, nin, int(rng.integers(0, nin + 1)), depth - 1))
        if c == "binop":
            a = int(rng.integers(0, nin + 1))
            return ("binop", "+-*"[rng.integers(3)], gen(rng, a, nout, depth - 1), gen(rng, nin - a, nout, depth - 1))
        if c == "unop":
            return ("unop", ["mul", "add", "neg", "rsub"][rng.integers(4)], float(np.float32(rng.uniform(-1.5, 1.5))),
                    gen(rng, nin, nout, depth - 1))
    raise AssertionError("no production fits")


def build(t, m):
    k = t[0]
    if k == "pipe": return build(t[1], m) >> build(t[2], m)
    if k == "stack": return build(t[1], m) | build(t[2], m)
    if k == "bus": return build(t[1], m) & build(t[2], m)
    if k == "branch": return build(t[1], m) ^ build(t[2], m)
    if k == "thru": return ~build(t[1], m)
    if k == "binop":
        a, b = build(t[2], m), build(t[3], m)
        return a + b if t[1] == "+" else a - b if t[1] == "-" else a * b
    if k == "unop":
        x = build(t[3], m)
        return x * t[2] if t[1] == "mul" else x + t[2] if t[1] == "add" else -x if t[1] == "neg" else t[2] - x
    if k == "shape_tanh": return m.shape("tanh", t[1])
    if k == "dc2": return m.dc(t[1], t[2])
    if k == "meter_rms": return m.meter("rms", t[1])
    if k == "allnest_delay": return m.allnest_c(t[1], m.delay(t[2]))
    if k == "echo": return m.feedback(m.delay(t[1]) * t[2])
    if k == "tap_dc": return (m.pass_() | m.dc(t[1])) >> m.tap(0.0004, 0.004)
    return getattr(m, k)(*t[1:])


@pytest.mark.parametrize("seed", range(int(os.environ.get("FUNDSP_FUZZ_GRAPHS", "16"))))   # more for a bug hunt
def test_random_graph_matches_oracle(gpu, seed):
    rng = np.random.default_rng(int(os.environ.get("FUNDSP_FUZZ_SEED0", "1000")) + seed)
    nin, nout = int(rng.integers(0, 3)), int(rng.integers(1, 3))
    tree = gen(rng, nin, nout, depth=int(rng.integers(3, 6)))
    g = build(tree, GR)
    assert (g.nin, g.nout) == (nin, nout), tree
    V, T = 5, 64 * 4 + 19
    seeds = np.arange(V, dtype=np.uint64) * 977 + seed
    x = noise_input(V, nin, T, seed=seed) if nin else None
    for mode, layout in ((MODE_PROCESS, LAYOUT_VOICE_MINOR), (MODE_TICK, LAYOUT_PLANAR), (MODE_PROCESS, LAYOUT_PLANAR), (MODE_PROCESS, "planar, tight rows"), (MODE_TICK, "planar, tight rows")):
        b = gpu.Bank.from_graph(g, V, ring_frames=256 if g.rings else 0, sample_rate=SR)
        b.set_seed(seeds)
        if isinstance(layout, str):   # rows exactly T = 275 floats apart: no 16-byte runs, so the single-wave planar kernel renders them (aligned rows: the planar pipeline)
            import torch

            xi = None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
            out = b.process(T, xi, layout=LAYOUT_PLANAR, frame_stride=T, mode=mode)
            torch.cuda.synchronize()
            assert b.get_option("last_kernel") == 1
            got = out.cpu().numpy()
        else:
            got = run_bank(b, x, T, layout, mode)
        for v in (0, V - 1):
            n = build(tree, O)
            n.set_sample_rate(SR)
            n.set_seed(int(seeds[v]))
            assert_bit_equal(got[v], oracle_render(n, None if x is None else x[v], T, mode), f"seed {seed} voice {v} mode {mode} layout {layout}: {tree}")


def wide_tree(rng):
    """A wide sum (sumi / busi over 8..14 branches of one random shape, parameters varying with the branch index) at the head of a random spine of
    Pipe / Unop nodes -- the shapes fd_device.hpp WideSplit takes branch-major, with the rest of the graph as the tail."""
    n = int(rng.integers(8, 15))
    bus = bool(rng.integers(2))
    nin = int(rng.integers(0, 2)) if bus else int(rng.integers(0, 2))
    nout = int(rng.integers(1, 3))
    branch = gen(rng, nin, nout, depth=int(rng.integers(0, 3)))
    spine = ("wide", "busi" if bus else "sumi", n, branch)
    width = nout
    for _ in range(int(rng.integers(0, 4))):
        if rng.integers(3) == 0:
            spine = ("unop", ["mul", "add", "neg", "rsub"][rng.integers(4)], float(np.float32(rng.uniform(-1.5, 1.5))), spine)
        else:
            k = int(rng.integers(1, 3))
            spine = ("pipe", spine, gen(rng, width, k, depth=int(rng.integers(0, 3))))
            width = k
    gin = nin if bus else n * nin
    return spine, gin, width


def scale_tree(t, i):
    """branch i of a wide sum: the same shape, every frequency-like leaf parameter moved with the branch index (sumi(|i| sine_hz(f * (i + 1))))"""
    if not isinstance(t, tuple):
        return t
    if t[0] in ("sine_hz", "poly_saw_hz", "lowpass_hz", "lowpole_hz", "highpole_hz", "bell_hz", "moog_hz"):
        return (t[0], float(np.float32(t[1] * (1.0 + 0.13 * i))),) + t[2:]
    return tuple(scale_tree(x, i) for x in t)


def build_wide(t, m):
    if t[0] == "wide":
        return getattr(m, t[1])(t[2], lambda i: build_wide(scale_tree(t[3], i), m))
    if t[0] == "pipe":
        return build_wide(t[1], m) >> build(t[2], m)
    if t[0] == "unop":
        x = build_wide(t[3], m)
        return x * t[2] if t[1] == "mul" else x + t[2] if t[1] == "add" else -x if t[1] == "neg" else t[2] - x
    return build(t, m)


@pytest.mark.parametrize("seed", range(int(os.environ.get("FUNDSP_FUZZ_WIDE", "10"))))   # more for a bug hunt
def test_random_wide_sum_with_a_tail_matches_oracle(gpu, seed):
    """both wide kernels (the chain of waves, and with "pipe_split" 0 one wave per voice group), both executors, both layouts, ragged launch"""
    rng = np.random.default_rng(int(os.environ.get("FUNDSP_FUZZ_SEED0", "1000")) + 50000 + seed)
    tree, nin, nout = wide_tree(rng)
    g = build_wide(tree, GR)
    assert (g.nin, g.nout) == (nin, nout), tree
    V, T = 67, 64 * 3 + 19
    seeds = np.arange(V, dtype=np.uint64) * 977 + seed
    x = noise_input(V, nin, T, seed=seed) if nin else None
    for mode, layout, split in ((MODE_PROCESS, LAYOUT_VOICE_MINOR, 1), (MODE_TICK, LAYOUT_PLANAR, 1), (MODE_PROCESS, LAYOUT_PLANAR, 0), (MODE_TICK, LAYOUT_VOICE_MINOR, 0)):
        b = gpu.Bank.from_graph(g, V, ring_frames=256 if g.rings else 0, sample_rate=SR)
        b.set_option("pipe_split", split)
        b.set_seed(seeds)
        got = run_bank(b, x, T, layout, mode)
        assert b.get_option("last_kernel") == (8 if split else 1), (b.get_option("last_kernel"), tree)
        for v in (0, 63, V - 1):
            n = build_wide(tree, O)
            n.set_sample_rate(SR)
            n.set_seed(int(seeds[v]))
            assert_bit_equal(got[v], oracle_render(n, None if x is None else x[v], T, mode), f"seed {seed} voice {v} mode {mode} split {split}: {tree}")


@pytest.mark.parametrize("seed", range(int(os.environ.get("FUNDSP_FUZZ_CHAINS", "6"))))   # more for a bug hunt
def test_random_front_into_a_lane_per_frame_node_with_a_bus(gpu, seed):
    """A random front graph (generators, filters on inputs, delay lines, feedback loops, hashed nodes) piped into a reverb / network with one of
    the documented buses around it: Bank.from_graph builds a chain of two banks -- the front's fused kernel, seeded from the construction hash of
    the WHOLE graph and flushing denormals when the network half has a Feedback node, and the network's lane-per-frame kernel with the bus in
    its epilogue.  Against the oracle's ONE graph, as constructed and after set_seed, inputs in the normal and in the denormal range."""
    import test_gpu_reverb_bus as RB

    rng = np.random.default_rng(int(os.environ.get("FUNDSP_FUZZ_SEED0", "1000")) + 90000 + seed)
    node = list(RB.NODES)[rng.integers(len(RB.NODES))]
    bus = (list(RB.BUSES) + ["bare"])[rng.integers(len(RB.BUSES) + 1)]
    mid = 1 if node == "fdn8_mono" else 2
    nin = int(rng.integers(0, 3))
    tree = gen(rng, nin, mid, depth=int(rng.integers(1, 4)))

    def whole(m):
        back = RB.NODES[node][0](m) if bus == "bare" else RB.build(m, node, bus)
        return build(tree, m) >> back

    front_rings = build(tree, GR).rings
    V, T = 6, 64 * 40 + 7
    x = None
    if nin:
        x = noise_input(V, nin, T, seed=seed)
        x[1] = (x[1] * np.float32(1e-38)).astype(np.float32)     # an instance in the denormal range
        x[2, :, T // 2:] = 0.0                                    # tails that decay
    seeds = np.arange(V, dtype=np.uint64) * 131 + seed
    for mode, layout, seeded in ((MODE_PROCESS, LAYOUT_PLANAR, False), (MODE_TICK, LAYOUT_VOICE_MINOR, True)):
        ch = gpu.Bank.from_graph(whole(GR), V, ring_frames=256 if front_rings else 0, sample_rate=SR)
        assert isinstance(ch, gpu.Chain) and ch.effect.kind == RB.NODES[node][1], (node, bus, tree)
        if seeded:
            ch.set_seed(seeds)
        got = RB.run(ch, x if x is not None else np.zeros((V, 0, T), dtype=np.float32), layout, mode, [0, 64 * 9 + 5, T]) if nin else None
        if not nin:
            import torch

            parts = [ch.process(n, layout=layout, frame_stride=n if layout == LAYOUT_PLANAR else None, mode=mode) for n in (64 * 9 + 5, T - 64 * 9 - 5)]
            torch.cuda.synchronize()
            got = np.concatenate([p.cpu().numpy() if layout == LAYOUT_PLANAR else p.cpu().numpy().transpose(2, 0, 1) for p in parts], axis=2)
        for v in (0, 1, 2, V - 1):
            n = whole(O)
            n.set_sample_rate(SR)
            if seeded:
                n.set_seed(int(seeds[v]))
            cuts = ((0, 64 * 9 + 5), (64 * 9 + 5, T))
            if mode == MODE_PROCESS:
                want = [n.render_blocks(None if x is None else x[v][:, a:e], length=e - a) for a, e in cuts]
            else:
                want = [n.render_ticks(None if x is None else x[v][:, a:e], length=e - a) for a, e in cuts]
            assert_bit_equal(got[v], np.concatenate(want, axis=1), f"seed {seed} {node} {bus} instance {v} mode {mode} seeded {seeded}: {tree}")


@pytest.mark.parametrize("seed", range(int(os.environ.get("FUNDSP_FUZZ_TINY", "8"))))   # more for a bug hunt
def test_random_graph_on_inputs_in_the_denormal_range(gpu, seed):
    """The general fuzzer's graphs with inputs, fed tiny signals: 1e-38 (denormal on arrival), 1e-30 and 1e-20 (products and filter tails go
    denormal inside).  Graphs with a Feedback node render flushed (the run-time compiler's module flag on the device, MXCSR FTZ + DAZ in the
    oracle: denormal.rs:18), all others keep IEEE denormals -- the two must agree on which is which, node by node."""
    rng = np.random.default_rng(int(os.environ.get("FUNDSP_FUZZ_SEED0", "1000")) + 70000 + seed)
    nin, nout = int(rng.integers(1, 3)), int(rng.integers(1, 3))
    tree = gen(rng, nin, nout, depth=int(rng.integers(2, 5)))
    g = build(tree, GR)
    V, T = 5, 64 * 4 + 19
    seeds = np.arange(V, dtype=np.uint64) * 977 + seed
    x = noise_input(V, nin, T, seed=seed)
    for v, s in ((1, 1e-38), (2, 1e-30), (3, 1e-20)):
        x[v] = (x[v] * np.float32(s)).astype(np.float32)
    x[4, :, T // 3:] = 0.0
    for mode, layout in ((MODE_PROCESS, LAYOUT_VOICE_MINOR), (MODE_TICK, LAYOUT_PLANAR)):
        b = gpu.Bank.from_graph(g, V, ring_frames=256 if g.rings else 0, sample_rate=SR)
        b.set_seed(seeds)
        got = run_bank(b, x, T, layout, mode)
        for v in range(V):
            n = build(tree, O)
            n.set_sample_rate(SR)
            n.set_seed(int(seeds[v]))
            assert_bit_equal(got[v], oracle_render(n, x[v], T, mode), f"seed {seed} voice {v} mode {mode} ({'flushed' if 'Feedback' in g.type else 'IEEE denormals'}): {tree}")


@pytest.mark.parametrize("seed", range(int(os.environ.get("FUNDSP_FUZZ_ROUTES", "6"))))   # more for a bug hunt
def test_random_graph_renders_the_same_through_every_kernel_route(gpu, seed):
    """One graph, one executor, every route the dispatcher has: "pipe_split" 0 (the single-wave kernels), 1 (the default choice by launch length) and
    2 (the stage / planar pipelines forced) x voice-minor, planar with 16-byte rows, planar with tight odd rows -- launched whole and in ragged
    chunks.  All must give the same bits; the first is held against the oracle.  (A route only ever taken at sizes no parity test uses is where the
    miscompiled single-wave planar kernel had hidden.)"""
    import torch

    rng = np.random.default_rng(int(os.environ.get("FUNDSP_FUZZ_SEED0", "1000")) + 30000 + seed)
    nin, nout = int(rng.integers(0, 3)), int(rng.integers(1, 3))
    tree = gen(rng, nin, nout, depth=int(rng.integers(2, 5)))
    g = build(tree, GR)
    V, T = 67, 64 * 5 + 11
    seeds = np.arange(V, dtype=np.uint64) * 977 + seed
    x = noise_input(V, nin, T, seed=seed) if nin else None
    for mode in (MODE_PROCESS, MODE_TICK):
        ref = {}
        for split in (0, 1, 2):
            for layout in ("voice-minor", "planar", "planar, tight rows"):
                for cuts in ((0, T), (0, 64 * 2 + 5, 64 * 2 + 5 + 9, T)):
                    b = gpu.Bank.from_graph(g, V, ring_frames=256 if g.rings else 0, sample_rate=SR)
                    b.set_option("pipe_split", split)
                    b.set_seed(seeds)
                    parts = []
                    for a, e in zip(cuts[:-1], cuts[1:]):
                        n = e - a
                        if layout == "voice-minor":
                            xi = None if x is None else torch.from_numpy(np.ascontiguousarray(x[:, :, a:e].transpose(1, 2, 0))).cuda()
                            parts.append(b.process(n, xi, mode=mode).cpu().numpy().transpose(2, 0, 1))
                        else:
                            fs = n if layout.endswith("tight rows") else (n + 63) // 64 * 64
                            xi = None
                            if x is not None:
                                buf = np.zeros((V, nin, fs), dtype=np.float32)
                                buf[:, :, :n] = x[:, :, a:e]
                                xi = torch.from_numpy(buf).cuda()
                            parts.append(b.process(n, xi, layout=LAYOUT_PLANAR, frame_stride=fs, mode=mode).cpu().numpy()[:, :, :n])
                    got = np.concatenate(parts, axis=2)
                    if cuts not in ref:   # (a launch walks its own blocks and remainder, like a process() call of that size: every chunking has its own samples)
                        ref[cuts] = got
                        for v in (0, 63, V - 1):
                            n_ = build(tree, O)
                            n_.set_sample_rate(SR)
                            n_.set_seed(int(seeds[v]))
                            want = np.concatenate([oracle_render(n_, None if x is None else x[v][:, a:e], e - a, mode) for a, e in zip(cuts[:-1], cuts[1:])], axis=1)
                            assert_bit_equal(got[v], want, f"seed {seed} voice {v} mode {mode} cuts {cuts}: {tree}")
                    else:
                        assert_bit_equal(got, ref[cuts], f"seed {seed} mode {mode} pipe_split {split} {layout} cuts {cuts} != the first route: {tree}")


@pytest.mark.parametrize("seed", range(int(os.environ.get("FUNDSP_FUZZ_MIX", "6"))))   # more for a bug hunt
def test_random_graph_fused_mix_equals_the_mix_of_its_voice_out_render(gpu, seed):
    """fdsp_bank_process_mix of a random run-time compiled graph -- the mix kernels are a module of their own, compiled on first use -- against
    fdsp_sum_voices / fdsp_mix_stereo of the voice-out render of a clone: the same fixed summation order, so bit for bit; whole and chunked, both
    executors.  (Kinds without a fused mix-down say so: "has_fused_mix" 0.)"""
    import torch
    from fundsp_amd import MIX_PAN, MIX_SUM

    rng = np.random.default_rng(int(os.environ.get("FUNDSP_FUZZ_SEED0", "1000")) + 110000 + seed)
    nin, nout = int(rng.integers(0, 3)), int(rng.integers(1, 3))
    tree = gen(rng, nin, nout, depth=int(rng.integers(2, 5)))
    g = build(tree, GR)
    V, T = (67, 300, 1030)[seed % 3], 64 * 3 + 7
    seeds = np.arange(V, dtype=np.uint64) * 977 + seed
    x = noise_input(V, nin, T, seed=seed) if nin else None
    xi = None if x is None else torch.from_numpy(np.ascontiguousarray(x.transpose(1, 2, 0))).cuda()
    pan = np.linspace(-1, 1, V).astype(np.float32)
    for mode in (MODE_PROCESS, MODE_TICK):
        b = gpu.Bank.from_graph(g, V, ring_frames=256 if g.rings else 0, sample_rate=SR)
        if b.get_option("has_fused_mix") != 1:
            return
        b.set_seed(seeds)
        ref = b.clone()
        out = ref.process(T, xi, mode=mode)
        how = MIX_PAN if nout == 1 and seed % 2 else MIX_SUM
        if how == MIX_PAN:
            b.set_pan(pan)
            want = gpu.mix_stereo(out[0], torch.from_numpy(pan).cuda()).cpu().numpy()
        else:
            want = gpu.sum_voices(out).cpu().numpy()
        chunked = b.clone()
        if how == MIX_PAN:
            chunked.set_pan(pan)
        assert_bit_equal(b.process_mix(T, xi, mix=how, mode=mode).cpu().numpy(), want, f"seed {seed} mode {mode} mix {how} V {V}: {tree}")
        if mode == MODE_TICK:   # (a tick-executor launch has no block structure: chunks continue the whole)
            parts = [chunked.process_mix(e - a, None if xi is None else xi[:, a:e].contiguous(), mix=how, mode=mode).cpu().numpy() for a, e in ((0, 64 + 3), (64 + 3, T))]
            assert_bit_equal(np.concatenate(parts, axis=1), want, f"seed {seed} chunked mix, tick executor: {tree}")


@pytest.mark.parametrize("seed", range(int(os.environ.get("FUNDSP_FUZZ_EVENTS", "6"))))   # more for a bug hunt
def test_random_generator_under_the_voice_scheduler(gpu, seed):
    """A random run-time compiled generator graph as the unit of every Sequencer event (sequencer.rs: per-voice start / end / fades, off-grid times):
    fdsp_bank_process_events (the kind's jit_events kernels) against the oracle's Sequencer restatement, every voice's faded contribution bit for bit,
    both executors; the fused Sequencer output (process_events_mix) against the sum of those contributions."""
    import torch
    from test_gpu_sequencer import random_events

    rng = np.random.default_rng(int(os.environ.get("FUNDSP_FUZZ_SEED0", "1000")) + 130000 + seed)
    nout = int(rng.integers(1, 3))
    tree = gen(rng, 0, nout, depth=int(rng.integers(1, 4)))
    g = build(tree, GR)
    V, T = 40, 64 * 7 + 21
    seeds = np.arange(V, dtype=np.uint64) * 977 + seed
    start, end, fin, fout, fade = random_events(V, T, np.random.default_rng(seed))
    for mode in (MODE_PROCESS, MODE_TICK):
        b = gpu.Bank.from_graph(g, V, ring_frames=256 if g.rings else 0, sample_rate=SR)
        b.set_seed(seeds)
        b.set_events(start, end, fin, fout, fade)
        got = b.process_events(T, mode=mode)
        torch.cuda.synchronize()
        got = got.cpu().numpy().transpose(2, 0, 1)
        seq = O.Sequencer(0, nout, SR)
        for v in range(V):
            n = build(tree, O)
            n.set_sample_rate(SR)
            n.set_seed(int(seeds[v]))
            seq.push(start[v], end[v], int(fade[v]), fin[v], fout[v], n)
        _mix, per = seq.render(T, process=(mode == MODE_PROCESS))
        for v in range(V):
            assert_bit_equal(got[v], per[v], f"seed {seed} event voice {v} mode {mode}: {tree}")
        assert abs(b.events_time() - seq.time()) == 0.0
        if b.get_option("has_fused_mix") == 1:
            b2 = gpu.Bank.from_graph(g, V, ring_frames=256 if g.rings else 0, sample_rate=SR)
            b2.set_seed(seeds)
            b2.set_events(start, end, fin, fout, fade)
            fused = b2.process_events_mix(T, mode=mode).cpu().numpy()
            summed = gpu.sum_voices(torch.from_numpy(np.ascontiguousarray(got.transpose(1, 2, 0))).cuda()).cpu().numpy()
            assert_bit_equal(fused, summed, f"seed {seed} mode {mode}: Sequencer output fused vs sum of the events: {tree}")


@pytest.mark.parametrize("seed", range(int(os.environ.get("FUNDSP_FUZZ_LIFE", "6"))))   # more for a bug hunt
def test_random_graph_through_a_life_of_lifecycle_calls(gpu, seed):
    """render, set_sample_rate, render, reset, render, set_seed, render -- the AudioNode lifecycle in mid-stream (what each node keeps and what it
    recomputes or clears on set_sample_rate / reset / set_hash differs from node to node: delay lines resize and empty, filters recompute
    coefficients and keep state, oscillators re-derive their phase from the hash) -- with the oracle's graph taken through the same calls; a clone
    taken in mid-stream continues like the original."""
    rng = np.random.default_rng(int(os.environ.get("FUNDSP_FUZZ_SEED0", "1000")) + 150000 + seed)
    nin, nout = int(rng.integers(0, 3)), int(rng.integers(1, 3))
    tree = gen(rng, nin, nout, depth=int(rng.integers(2, 5)))
    g = build(tree, GR)
    V = 5
    seeds = np.arange(V, dtype=np.uint64) * 977 + seed
    seeds2 = seeds * np.uint64(31) + np.uint64(7)
    steps = [("render", 150), ("rate", 44100.0), ("render", 64 * 2 + 9), ("reset", None), ("render", 100), ("seed", None), ("render", 64 + 30), ("rate", 96000.0), ("render", 70)]
    total = sum(n for k, n in steps if k == "render")
    x = noise_input(V, nin, total, seed=seed) if nin else None
    for mode, layout in ((MODE_PROCESS, LAYOUT_VOICE_MINOR), (MODE_TICK, LAYOUT_PLANAR)):
        b = gpu.Bank.from_graph(g, V, ring_frames=1024 if g.rings else 0, sample_rate=SR)
        b.set_seed(seeds)
        nodes = []
        for v in (0, V - 1):
            n = build(tree, O)
            n.set_sample_rate(SR)
            n.set_seed(int(seeds[v]))
            nodes.append((v, n))
        t, twin = 0, None
        for k, (what, arg) in enumerate(steps):
            if what == "render":
                xs = None if x is None else x[:, :, t:t + arg]
                got = run_bank(b, xs, arg, layout, mode)
                if twin is not None:
                    assert_bit_equal(run_bank(twin, xs, arg, layout, mode), got, f"seed {seed} step {k}: the clone continues like the original: {tree}")
                    twin = None
                for v, n in nodes:
                    assert_bit_equal(got[v], oracle_render(n, None if xs is None else xs[v], arg, mode), f"seed {seed} step {k} ({what}) voice {v} mode {mode}: {tree}")
                t += arg
                if k == 0:
                    twin = b.clone()
            elif what == "rate":
                b.set_sample_rate(arg)
                if twin is not None:
                    twin.set_sample_rate(arg)
                for _v, n in nodes:
                    n.set_sample_rate(arg)
            elif what == "reset":
                b.reset()
                for _v, n in nodes:
                    n.reset()
            else:
                b.set_seed(seeds2)
                for v, n in nodes:
                    n.set_seed(int(seeds2[v]))


# ---- the wider leaf pool ---------------------------------------------------------------------------------------------------------------------------
# The fuzzers above draw from a pool of some thirty leaves.  This one adds the rest of the fixed-parameter opcodes both notations spell alike: the
# wavetable oscillators (shared tables on the device), the PolyBLEP family, coloured noises, every SVF mode, the biquad family, one-pole and
# resonant filters, followers, shapers, the look-ahead limiter (a reduce tree in ring memory), 2x oversampling around a sub-graph.
def wide_leaves(rng):
    r = lambda lo, hi: float(np.float32(rng.uniform(lo, hi)))
    return [
        ("saw_hz", r(40, 3000)), ("organ_hz", r(40, 2000)), ("hammond_hz", r(40, 2000)), ("soft_saw_hz", r(40, 3000)), ("ramp_hz", r(1, 2000)),
        ("poly_square_hz", r(50, 2000)), ("poly_pulse_hz", r(50, 2000), r(0.1, 0.9)), ("pink",), ("brown",), ("white",),
        ("highpass_hz", r(50, 8000), r(0.5, 3)), ("bandpass_hz", r(100, 6000), r(0.5, 4)), ("notch_hz", r(100, 6000), r(0.5, 4)),
        ("peak_hz", r(100, 6000), r(0.5, 4)), ("allpass_hz", r(100, 6000), r(0.5, 4)), ("lowshelf_hz", r(100, 3000), r(0.5, 2), r(0.3, 3)),
        ("highshelf_hz", r(1000, 9000), r(0.5, 2), r(0.3, 3)), ("butterpass_hz", r(100, 9000)), ("resonator_hz", r(100, 6000), r(10, 800)),
        ("dcblock_hz", r(5, 100)), ("allpole_delay", r(0.1, 1.9)), ("pinkpass",), ("lowrez_hz", r(100, 6000), r(0.0, 0.9)),
        ("bandrez_hz", r(100, 6000), r(0.0, 0.9)), ("morph_hz", r(100, 6000), r(0.5, 3), r(-1, 1)), ("afollow", r(0.001, 0.02), r(0.01, 0.1)),
        ("clip",), ("clip_to", r(-0.8, -0.1), r(0.1, 0.8)), ("shape_k", ["softsign", "atan", "clip", "crush", "soft_crush"][rng.integers(5)], r(0.5, 8)),
        ("limiter", r(0.0005, 0.002), r(0.005, 0.05)),
    ]


_WIDE_ARITY = {}


def build2(t, m):
    k = t[0]
    if k == "oversample": return m.oversample(build2(t[1], m))
    if k == "shape_k": return m.shape(t[1], t[2])
    if k in ("pipe", "stack", "bus", "branch", "thru", "binop", "unop"):
        sub = [build2(x, m) if isinstance(x, tuple) else x for x in t[1:]]
        if k == "pipe": return sub[0] >> sub[1]
        if k == "stack": return sub[0] | sub[1]
        if k == "bus": return sub[0] & sub[1]
        if k == "branch": return sub[0] ^ sub[1]
        if k == "thru": return ~sub[0]
        if k == "binop":
            a, b = sub[1], sub[2]
            return a + b if t[1] == "+" else a - b if t[1] == "-" else a * b
        x = sub[2]
        return x * t[2] if t[1] == "mul" else x + t[2] if t[1] == "add" else -x if t[1] == "neg" else t[2] - x
    return build(t, m)


def gen2(rng, nin, nout, depth):
    """gen() with the wider pool at the leaves and `oversample` among the combinators"""
    if depth > 0 and nin == nout and nin >= 1 and rng.integers(8) == 0:
        return ("oversample", gen2(rng, nin, nout, depth - 1))
    if depth > 0 and rng.integers(3) > 0:
        t = gen(rng, nin, nout, 1)   # one combinator (or a leaf) of the base grammar ...
        if t[0] in ("pipe", "stack", "bus", "branch", "thru", "binop", "unop"):   # ... whose children are drawn again, from this grammar
            def redraw(x):
                if not isinstance(x, tuple):
                    return x
                gx = build(x, GR)
                return gen2(rng, gx.nin, gx.nout, depth - 1)
            return tuple(redraw(x) if isinstance(x, tuple) else x for x in t)
        return t
    cands = []
    for lf in wide_leaves(rng):
        if lf[0] not in _WIDE_ARITY:
            gl = build2(lf, GR)
            _WIDE_ARITY[lf[0]] = (gl.nin, gl.nout)
        if _WIDE_ARITY[lf[0]] == (nin, nout):
            cands.append(lf)
    if cands and rng.integers(4) > 0:
        return cands[rng.integers(len(cands))]
    return gen(rng, nin, nout, 0)


@pytest.mark.parametrize("seed", range(int(os.environ.get("FUNDSP_FUZZ_WIDER", "8"))))   # more for a bug hunt
def test_random_graph_of_the_wider_leaf_pool_matches_oracle(gpu, seed):
    rng = np.random.default_rng(int(os.environ.get("FUNDSP_FUZZ_SEED0", "1000")) + 170000 + seed)
    nin, nout = int(rng.integers(0, 3)), int(rng.integers(1, 3))
    tree = gen2(rng, nin, nout, depth=int(rng.integers(3, 7)))
    g = build2(tree, GR)
    assert (g.nin, g.nout) == (nin, nout), tree
    V, T = 5, 64 * 4 + 19
    seeds = np.arange(V, dtype=np.uint64) * 977 + seed
    x = noise_input(V, nin, T, seed=seed) if nin else None
    for mode, layout in ((MODE_PROCESS, LAYOUT_VOICE_MINOR), (MODE_TICK, LAYOUT_PLANAR), (MODE_PROCESS, LAYOUT_PLANAR)):
        b = gpu.Bank.from_graph(g, V, ring_frames=512 if g.rings else 0, sample_rate=SR)
        b.set_seed(seeds)
        got = run_bank(b, x, T, layout, mode)
        for v in (0, V - 1):
            n = build2(tree, O)
            n.set_sample_rate(SR)
            n.set_seed(int(seeds[v]))
            assert_bit_equal(got[v], oracle_render(n, None if x is None else x[v], T, mode), f"seed {seed} voice {v} mode {mode} layout {layout}: {tree}")
