"""GPU parity of the gain and dry bus around the reverb / network banks (fdsp_bank_set_bus) -- the shapes the reference's documentation gives its
reverbs:

    multipass() & 0.2 * reverb_stereo(20.0, 2.0, 1.0)                README.md:436
    0.2 * reverb_stereo(10.0, 1.0, 0.5) & multipass()                wave.rs:514
    wet * reverb_stereo(10.0, time) & (1.0 - wet) * multipass()      CHANGES.md:203
    multipass() & reverb_stereo(10.0, 1.0, 0.5)                      net.rs:681

Bank.from_graph recognises them (graph.bus_plan) around every node with a lane-per-frame kernel and folds the Unop / MultiPass / Bus nodes into that
kernel's epilogue.  Bit-exact against the oracle's rendering of the WHOLE graph (its own Bus, Unop and MultiPass nodes), both executors, both
layouts, ragged and chunked launches, the denormal range (graphs with a Feedback node flush, reverb3_stereo does not), clone, reset; against the
run-time compiled lane-per-voice rendering of the same graph; and in front of a generator (the chain of two banks)."""
import numpy as np
import pytest

import oracle as O
from fundsp_amd import BUS_DRY_WET, BUS_NONE, BUS_WET, LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MODE_PROCESS, MODE_TICK
from fundsp_amd import graph as GR
from test_gpu_fdn import delays_of, run
from test_gpu_parity import assert_bit_equal, oracle_render, run_bank

pytestmark = pytest.mark.gpu
SR = 48000.0


def net16(m):   # the prelude's fdn example with two channels in and out (prelude.rs:1334)
    d = delays_of(16)
    return m.multisplit(2, 8) >> m.fdn(m.stacki(16, lambda i: m.delay(d[i]) >> m.fir(0.2, 0.4, 0.2))) >> m.multijoin(2, 8)


def mono8(m):
    d = delays_of(8)
    return m.split(8) >> m.fdn(m.stacki(8, lambda i: m.delay(d[i]) >> m.fir(0.5, 0.4))) >> m.join(8)


NODES = {   # name -> (builder in either notation, the kind of the bank, pass node of its arity)
    "reverb_stereo": (lambda m: m.reverb_stereo(20.0, 2.0, 1.0), "reverb_stereo", lambda m: m.multipass(2)),
    "reverb4_stereo": (lambda m: m.reverb4_stereo(10.0, 3.0), "reverb4_stereo", lambda m: m.multipass(2)),
    "reverb3_stereo": (lambda m: m.reverb3_stereo(2.0, 0.6, lambda: m.lowpole_hz(6000.0)), "reverb3_stereo", lambda m: m.multipass(2)),
    "fdn16_stereo": (net16, "fdn", lambda m: m.multipass(2)),
    "fdn8_mono": (mono8, "fdn", lambda m: m.pass_()),
}
BUSES = {   # name -> (graph around node r with the pass node p, the bus it must become)
    "pass_and_wet": (lambda m, r, p: p & 0.2 * r, (BUS_DRY_WET, np.float32(0.2), np.float32(1.0))),             # README.md:436
    "wet_and_pass": (lambda m, r, p: 0.2 * r & p, (BUS_DRY_WET, np.float32(0.2), np.float32(1.0))),             # wave.rs:514
    "wet_and_dry": (lambda m, r, p: 0.3 * r & (1.0 - 0.3) * p, (BUS_DRY_WET, np.float32(0.3), np.float32(0.7))),   # CHANGES.md:203
    "pass_and_node": (lambda m, r, p: p & r, (BUS_DRY_WET, np.float32(1.0), np.float32(1.0))),                  # net.rs:681
    "wet_only": (lambda m, r, p: r * 0.25, (BUS_WET, np.float32(0.25), np.float32(1.0))),
}


def build(m, node, bus):
    mk, _kind, mkpass = NODES[node]
    return BUSES[bus][0](m, mk(m), mkpass(m))


def inputs(V, nin, T, seed):
    rng = np.random.default_rng(seed)
    x = (rng.random((V, nin, T), dtype=np.float32) * 2 - 1).astype(np.float32)
    x[:, :, 2 * T // 3:] = 0.0
    x[1] *= np.float32(1e-30)       # an instance in the denormal range: dry * in and wet * y flush where the graph has a Feedback node
    x[2] = 0.0
    x[2, 0, 0] = 1.0                # an impulse
    return x


@pytest.mark.parametrize("bus", list(BUSES))
@pytest.mark.parametrize("node", list(NODES))
def test_bus_around_a_lane_per_frame_bank_matches_the_oracles_whole_graph(gpu, node, bus):
    V, T = 5, 64 * 70 + 13
    nin = 1 if node == "fdn8_mono" else 2
    x = inputs(V, nin, T, seed=len(node) * 10 + len(bus))
    cuts = [0, 64 * 9, 64 * 9 + 7, 64 * 40 + 7, T]
    for mode in (MODE_PROCESS, MODE_TICK):
        for layout in (LAYOUT_PLANAR, LAYOUT_VOICE_MINOR):
            b = gpu.Bank.from_graph(build(GR, node, bus), V, sample_rate=SR)   # (a fresh bank: Reverb::reset leaves reverb3_stereo's input diffusers alone)
            assert b.kind == NODES[node][1] and b.inputs() == nin, b.kind
            mode_, wet, dry = b.get_bus()
            assert (mode_, np.float32(wet), np.float32(dry)) == BUSES[bus][1]
            got = run(b, x, layout, mode, cuts)
            assert b.get_option("last_kernel") == 6
            for v in range(V):
                n = build(O, node, bus)
                n.set_sample_rate(SR)
                want = [n.render_blocks(x[v][:, a:e]) if mode == MODE_PROCESS else n.render_ticks(x[v][:, a:e]) for a, e in zip(cuts[:-1], cuts[1:])]
                assert_bit_equal(got[v], np.concatenate(want, axis=1), f"{node} {bus} mode {mode} layout {layout} instance {v}")
    assert node == "reverb4_stereo" or np.abs(got[2, :, 1:]).max() > 1e-7   # the impulse's tail is in the output (reverb4_stereo's two networks in series: later than this render)


@pytest.mark.parametrize("node", ["reverb_stereo", "reverb3_stereo", "fdn16_stereo"])
def test_bus_bank_equals_the_run_time_compiled_graph_and_larger_banks_take_the_staging_copy(gpu, node):
    """the same graph compiled at run time (its Bus / Unop / MultiPass nodes as device code, one lane per voice) renders the same samples;
    a bank of a tile of instances or more takes voice-minor buffers through the staging copy, the bus included"""
    V, T = 70, 64 * 12 + 5
    g = build(GR, node, "wet_and_dry")
    fast = gpu.Bank.from_graph(g, V, sample_rate=SR)
    slow = gpu.Bank.from_graph(build(GR, node, "wet_and_dry"), V, sample_rate=SR, fdn_kernel=False,
                               ring_frames=4096 if node == "reverb3_stereo" else 0)
    assert fast.kind == NODES[node][1] and slow.kind.startswith("jit_")
    x = inputs(V, 2, T, seed=3)
    a = run_bank(fast, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    assert_bit_equal(a, run_bank(slow, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS), f"{node}: lane-per-frame bank with the bus == the compiled graph")
    fast = gpu.Bank.from_graph(g, V, sample_rate=SR)
    assert_bit_equal(run_bank(fast, x, T, LAYOUT_PLANAR, MODE_PROCESS), a, "planar == voice-minor (staged)")


def test_set_bus_by_hand_clone_reset_and_errors(gpu):
    V, T = 4, 64 * 20 + 3
    x = inputs(V, 2, T, seed=11)
    b = gpu.Bank.reverb_stereo(V, 10.0, 1.0, 0.5)
    b.set_sample_rate(SR)
    assert b.get_bus() == (BUS_NONE, 1.0, 1.0)
    plain = run_bank(b, x, T, LAYOUT_PLANAR, MODE_PROCESS)
    b.reset()
    b.set_bus(BUS_DRY_WET, 0.2, 1.0)
    first = run_bank(b, x[:, :, :700], 700, LAYOUT_PLANAR, MODE_PROCESS)
    twin = b.clone()
    assert twin.get_bus() == b.get_bus()
    rest = run_bank(b, x[:, :, 700:], T - 700, LAYOUT_PLANAR, MODE_PROCESS)
    assert_bit_equal(run_bank(twin, x[:, :, 700:], T - 700, LAYOUT_PLANAR, MODE_PROCESS), rest, "the clone carries the bus and continues alike")
    got = np.concatenate([first, rest], axis=2)
    want = (x + np.float32(0.2) * plain).astype(np.float32)     # one rounding per operation: dry * in (in * 1.0 = in) + wet * y
    assert_bit_equal(got[0], want[0], "the bus is x + 0.2 * (the bank without it)")
    assert_bit_equal(got[3], want[3], "the bus is x + 0.2 * (the bank without it)")
    # the other formulation of reverb_stereo (one lane per delay line, "fdn_kernel" 1) reads the block's inputs back from its LDS tile
    for V2, T2 in ((V, T), (4100, 64 * 6 + 3)):                 # one instance per wave | two (banks of 4 096 instances and more)
        x2 = inputs(V2, 2, T2, seed=11)
        lines, frames_ = gpu.Bank.reverb_stereo(V2, 10.0, 1.0, 0.5), gpu.Bank.reverb_stereo(V2, 10.0, 1.0, 0.5)
        for bank, k in ((lines, 1), (frames_, 0)):
            bank.set_sample_rate(SR)
            bank.set_option("fdn_kernel", k)
            bank.set_bus(BUS_DRY_WET, 0.2, 0.9)
        got_lines = run_bank(lines, x2, T2, LAYOUT_PLANAR, MODE_PROCESS)
        assert lines.get_option("last_kernel") == 7
        assert_bit_equal(got_lines, run_bank(frames_, x2, T2, LAYOUT_PLANAR, MODE_PROCESS), f"lane = line kernel with the bus == lane = frame kernel ({V2} instances)")
    b.set_sample_rate(44100.0)                                  # the setting survives a re-configuration of the rings
    assert b.get_bus()[0] == BUS_DRY_WET
    b.set_bus(BUS_NONE)
    b.set_sample_rate(SR)
    b.reset()
    assert_bit_equal(run_bank(b, x, T, LAYOUT_PLANAR, MODE_PROCESS), plain, "BUS_NONE: the bank as created")
    with pytest.raises(gpu.FdspError):
        b.set_bus(7)
    one_to_two = gpu.Bank.fdn(V, 4, delays_of(4), 1, (0.9,), 1, 2)
    one_to_two.set_bus(BUS_WET, 0.5)
    with pytest.raises(gpu.FdspError):
        one_to_two.set_bus(BUS_DRY_WET, 0.5, 0.5)              # a Bus needs as many outputs as inputs
    sine = gpu.Bank("sine", V)
    with pytest.raises(gpu.FdspError):
        sine.set_bus(BUS_WET, 0.5)                              # a compiled kind carries its bus in its type


def test_generator_into_a_bussed_reverb_is_a_chain_of_two_banks(gpu):
    """(noise() | noise()) >> (multipass() & 0.2 * reverb_stereo(..)): the generator in its fused kernel, the reverb WITH its bus in the
    lane-per-frame kernel; seeded from the construction hash of the whole graph (Bus, Unop and MultiPass pings included: probe_hash)"""
    import torch

    V, T = 70, 64 * 30 + 9

    def whole(m):
        return (m.noise() | m.noise()) >> (m.multipass(2) & 0.2 * m.reverb_stereo(10.0, 1.0, 0.5))

    ch = gpu.Bank.from_graph(whole(GR), V, sample_rate=SR)
    assert isinstance(ch, gpu.Chain) and ch.effect.kind == "reverb_stereo" and ch.effect.get_bus()[0] == BUS_DRY_WET
    out = ch.process(T, layout=LAYOUT_PLANAR)
    torch.cuda.synchronize()
    got = out.cpu().numpy()[:, :, :T]
    n = whole(O)
    n.set_sample_rate(SR)
    want = oracle_render(n, None, T, MODE_PROCESS)
    for v in (0, 69):
        assert_bit_equal(got[v], want, f"as constructed, instance {v}")
    seeds = np.arange(V, dtype=np.uint64) * 5 + 1
    ch.reset()
    ch.set_seed(seeds)
    out = ch.process(T, layout=LAYOUT_PLANAR, mode=MODE_TICK)
    torch.cuda.synchronize()
    got = out.cpu().numpy()[:, :, :T]
    for v in (3, 64):
        n = whole(O)
        n.set_sample_rate(SR)
        n.set_seed(int(seeds[v]))
        assert_bit_equal(got[v], oracle_render(n, None, T, MODE_TICK), f"set_seed, tick executor, instance {v}")


def test_effect_chain_with_inputs_and_rings_in_front_of_a_bussed_reverb(gpu):
    """An effect chain WITH inputs and delay rings in front of the reverb -- (lowpass_hz | (pass() & delay)) >> (multipass() & 0.3 * reverb_stereo(..)) --
    is a chain of two banks too: the front in its fused kernel, compiled with flushed denormals because the one graph has a Feedback node (the
    reference renders ALL of it under FTZ + DAZ once Feedback::new has run, feedback.rs:96); an instance in the denormal range is where a front
    that kept its denormals would differ.  reverb3_stereo has no Feedback node: its front keeps IEEE denormals."""
    V, T = 70, 64 * 25 + 11

    def with_feedback(m):
        return (m.lowpass_hz(800.0, 1.0) | (m.pass_() & m.delay(0.002))) >> (m.multipass(2) & 0.3 * m.reverb_stereo(10.0, 1.0, 0.5))

    def without(m):
        return (m.lowpole_hz(900.0) | m.lowpole_hz(700.0)) >> (m.multipass(2) & 0.3 * m.reverb3_stereo(2.0, 0.6, lambda: m.lowpole_hz(6000.0)))

    for build_, effect in ((with_feedback, "reverb_stereo"), (without, "reverb3_stereo")):
        x = inputs(V, 2, T, seed=21)
        x[1] = (x[1] * np.float32(1e-8)).astype(np.float32)     # 1e-38 .. 1e-45: denormal products inside the front's filters
        x[3, :, 300:] = 0.0                                     # a filter tail that decays through the denormal range
        for mode, layout in ((MODE_PROCESS, LAYOUT_PLANAR), (MODE_TICK, LAYOUT_VOICE_MINOR)):
            ch = gpu.Bank.from_graph(build_(GR), V, sample_rate=SR)
            assert isinstance(ch, gpu.Chain) and ch.effect.kind == effect and (ch.inputs(), ch.outputs()) == (2, 2)
            assert ch.effect.get_bus()[0] == BUS_DRY_WET
            got = run(ch, x, layout, mode, [0, 64 * 7 + 5, T])
            for v in (0, 1, 2, 3, 69):
                n = build_(O)
                n.set_sample_rate(SR)
                want = [n.render_blocks(x[v][:, a:e]) if mode == MODE_PROCESS else n.render_ticks(x[v][:, a:e]) for a, e in ((0, 64 * 7 + 5), (64 * 7 + 5, T))]
                assert_bit_equal(got[v], np.concatenate(want, axis=1), f"{effect} behind an effect chain, mode {mode} layout {layout} instance {v}")


def test_flushed_front_differs_from_an_unflushed_one_only_in_the_denormal_range(gpu):
    """the alias that makes the run-time compiler flush a kind (Bank.from_graph(.., flush_denormals=True)) changes nothing but that"""
    V, T = 8, 64 * 6
    g = lambda: GR.lowpole_hz(900.0) >> GR.lowpole_hz(500.0)
    a, b = gpu.Bank.from_graph(g(), V, sample_rate=SR), gpu.Bank.from_graph(g(), V, sample_rate=SR, flush_denormals=True)
    assert a.kind != b.kind
    x = inputs(V, 1, T, seed=5)
    x[1] = (x[1] * np.float32(1e-8)).astype(np.float32)
    ya, yb = run_bank(a, x, T, LAYOUT_PLANAR, MODE_PROCESS), run_bank(b, x, T, LAYOUT_PLANAR, MODE_PROCESS)
    assert_bit_equal(ya[0], yb[0], "normal range: identical")
    assert np.abs(ya[1]).max() > 0 and np.abs(yb[1]).max() == 0   # the denormal instance: kept | flushed
