"""Wide sums of generators -- Reduce<N, X, OP> / MultiBus<N, X> over N >= 8 input-free branches of one type (sumi / busi of oscillators) --
are rendered branch-major, one 64-frame block of one branch at a time, like the reference's process() (audionode.rs:2406-2462, 2123-2134;
fd_device.hpp render_body_wide).  Every instance bit-equal to the oracle's frame-by-frame tree walk: both executors, both layouts, ragged
launches, partially filled waves, branches with delay rings, two-channel branches, a branch that trips the packed sine's guard, chunked
launches, and the frame-major form of the same sum (N below the threshold) as the neighbour."""
import numpy as np
import pytest

import oracle as O
from fundsp_amd import LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MODE_PROCESS, MODE_TICK
from fundsp_amd import graph as GR
from test_gpu_parity import assert_bit_equal, noise_input, oracle_render, run_bank

pytestmark = pytest.mark.gpu
SR = 48000.0

GRAPHS = {
    # name: (builder, ring_frames)
    "sumi12_sines": (lambda m: m.sumi(12, lambda i: m.sine_hz(55.0 * (i + 1))), 0),
    "busi9_sines": (lambda m: m.busi(9, lambda i: m.sine_hz(110.0 * (i + 1)) * 0.1), 0),                 # MultiBus: tick folds from a zero frame
    "sumi8_noise_delays": (lambda m: m.sumi(8, lambda i: m.noise() >> m.delay(0.0005 * (i + 1))), 256),   # branches with rings
    "sumi8_panned": (lambda m: m.sumi(8, lambda i: m.sine_hz(100.0 * (i + 1)) >> m.pan(-0.7 + 0.2 * i)), 0),   # two channels per branch
    "sumi10_one_trips": (lambda m: m.sumi(10, lambda i: m.sine_hz(3.0e6 if i == 4 else 200.0 * (i + 1))), 0),  # branch 4 leaves the packed sine's domain every block
    "sumi7_sines_frame_major": (lambda m: m.sumi(7, lambda i: m.sine_hz(55.0 * (i + 1))), 0),          # below the threshold: the frame-major walk
    # branches WITH inputs: a MultiBus hands every branch the graph's inputs, a Reduce gives every branch its own
    "busi20_harmonics_of_an_input": (lambda m: m.busi(20, lambda i: m.mul(float(i + 1)) >> m.sine()), 0),   # README.md:1152: input = frequency
    "sumi8_lowpasses_own_inputs": (lambda m: m.sumi(8, lambda i: m.lowpass_hz(150.0 * (i + 1), 1.0 + 0.25 * i)), 0),
    "busi10_resonators_panned": (lambda m: m.busi(10, lambda i: m.resonator_hz(300.0 * (i + 1), 25.0) >> m.pan(-0.9 + 0.2 * i)), 0),   # 1 in, 2 out
    # the sum at the HEAD of a chain of Pipe / Unop nodes: the rest of the graph is the tail the finishing wave walks frame-major
    "additive_voice_gain_filter_pan": (lambda m: m.sumi(12, lambda i: m.sine_hz(82.4 * (i + 1))) * 0.05 >> m.lowpass_hz(900.0, 1.5) >> m.pan(0.3), 0),   # mono sum, stereo out
    "harmonics_of_an_input_gain": (lambda m: m.busi(20, lambda i: m.mul(float(i + 1)) >> m.sine()) * 0.05, 0),
    "sum_shaped_with_a_seeded_tail": (lambda m: m.sumi(9, lambda i: m.sine_hz(110.0 * (i + 1))) >> m.shape("tanh", 0.5) >> (m.pass_() + m.noise() * 0.01), 0),
    "rings_in_the_sum_and_in_the_tail": (lambda m: m.sumi(8, lambda i: m.noise() >> m.delay(0.0005 * (i + 1))) >> m.delay(0.001) >> m.lowpole_hz(2000.0), 256),
    "tail_whose_packed_sine_trips": (lambda m: m.sumi(8, lambda i: m.sine_hz(50.0 * (i + 1))) * 2.0e6 + 3.0e6 >> m.sine(), 0),   # the tail's Sine leaves its packed domain every block
}


@pytest.mark.parametrize("name", list(GRAPHS))
def test_wide_sum_matches_the_oracle(gpu, name):
    build, ring = GRAPHS[name]
    g = build(GR)
    V, T = 130, 64 * 6 + 13
    seeds = np.arange(V, dtype=np.uint64) * 7919 + 13
    x = None
    if g.nin:
        x = noise_input(V, g.nin, T, seed=5)
        if "harmonics" in name:
            x = (np.abs(x) * np.float32(300.0) + np.float32(40.0)).astype(np.float32)   # a frequency input
    wide = "frame_major" not in name
    for mode in (MODE_PROCESS, MODE_TICK):
        for layout in (LAYOUT_VOICE_MINOR, LAYOUT_PLANAR):
            # a small bank takes the chain of waves (last_kernel 8); with "pipe_split" 0 the one-wave-per-voice-group kernel renders it (1)
            for split, kernel in ((1, 8 if wide else None), (0, 1)):   # (the frame-major neighbour takes whatever the stage pipelines offer)
                b = gpu.Bank.from_graph(g, V, ring_frames=ring, sample_rate=SR)
                b.set_option("pipe_split", split)
                b.set_seed(seeds)
                got = run_bank(b, x, T, layout, mode)
                assert kernel is None or b.get_option("last_kernel") == kernel, (name, mode, layout, split, b.get_option("last_kernel"))
                for v in (0, 15, 16, 64, 129):
                    n = build(O)
                    n.set_sample_rate(SR)
                    n.set_seed(int(seeds[v]))
                    assert_bit_equal(got[v], oracle_render(n, None if x is None else x[v], T, mode), f"{name} instance {v} mode {mode} layout {layout} pipe_split {split}")


def test_wide_sum_chunked_launches_and_reset(gpu):
    """ragged launches one after the other (each walks its own 64-frame blocks + remainder, like consecutive process() calls), then reset"""
    build, _ = GRAPHS["sumi12_sines"]
    V, chunks = 70, (64, 200, 1, 311)
    b = gpu.Bank.from_graph(build(GR), V, sample_rate=SR)
    seeds = np.arange(V, dtype=np.uint64) + 99

    def oracle(v):
        n = build(O)
        n.set_sample_rate(SR)
        n.set_seed(int(seeds[v]))
        return np.concatenate([n.render_blocks(None, length=k, block=64) for k in chunks], axis=1)

    for _round in range(2):
        b.reset()
        b.set_seed(seeds)
        got = np.concatenate([run_bank(b, None, k, LAYOUT_VOICE_MINOR, MODE_PROCESS) for k in chunks], axis=2)
        for v in (3, 69):
            assert_bit_equal(got[v], oracle(v), f"chunked launches, instance {v}")


def test_wide_sum_on_a_bank_larger_than_the_chip(gpu):
    """more voice groups than CUs (the chain's workgroups run in batches), ragged bank size: the chain of waves and the one-wave kernel"""
    build, _ = GRAPHS["sumi12_sines"]
    V, T = 64 * 520 + 5, 64 * 2 + 3
    seeds = np.arange(V, dtype=np.uint64) * 3 + 1
    for split, kernel in ((1, 8), (0, 1)):
        b = gpu.Bank.from_graph(build(GR), V, sample_rate=SR)
        b.set_option("pipe_split", split)
        b.set_seed(seeds)
        got = run_bank(b, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
        assert b.get_option("last_kernel") == kernel
        for v in (0, 64 * 519 + 63, V - 1):
            n = build(O)
            n.set_sample_rate(SR)
            n.set_seed(int(seeds[v]))
            assert_bit_equal(got[v], oracle_render(n, None, T, MODE_PROCESS), f"instance {v} pipe_split {split}")


def test_wide_sum_one_block_launches_and_fast_math(gpu):
    """a launch of ONE block has no chain to fill (the one-wave kernel takes it); the tolerance-mode twin of a wide sum renders through the same
    two kernels and stays within its stated tolerance of the exact bank"""
    build, _ = GRAPHS["sumi12_sines"]
    V = 70
    seeds = np.arange(V, dtype=np.uint64) + 7
    b = gpu.Bank.from_graph(build(GR), V, sample_rate=SR)
    b.set_seed(seeds)
    n = build(O)
    n.set_sample_rate(SR)
    n.set_seed(int(seeds[9]))
    for k in range(3):
        got = run_bank(b, None, 64, LAYOUT_VOICE_MINOR, MODE_PROCESS)
        assert b.get_option("last_kernel") == 1
        assert_bit_equal(got[9], n.render_blocks(None, length=64, block=64), f"block {k}")
    exact = gpu.Bank.from_graph(build(GR), V, sample_rate=SR)
    fast = gpu.Bank.from_graph(build(GR), V, sample_rate=SR)
    fast.set_option("math", gpu.MATH_FAST)
    exact.set_seed(seeds); fast.set_seed(seeds)
    a, f = run_bank(exact, None, 441, LAYOUT_VOICE_MINOR, MODE_PROCESS), run_bank(fast, None, 441, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    assert fast.get_option("last_kernel") == 8 and np.abs(a - f).max() <= 12 * 1e-4   # 12 unit sines, each within the mode's 1e-4 over 441 samples
