"""The reference's OWN benchmark graphs (benches/benchmark.rs -- a criterion harness that renders 1 s of one graph at 44.1 kHz) as banks of
instances: every instance of the bank bit-equal to the oracle's rendering of the same graph with the same seed, for the full second the
bench renders (Wave::render: 689 blocks of 64 + a remainder of 4) in the process executor, and a shorter stretch in the tick executor.
Graph table: tests/criterion_graphs.py (nine of the thirteen benches; the other four are not graphs of nodes on the path)."""
import numpy as np
import pytest

import criterion_graphs as CG
import oracle as O
from fundsp_amd import LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MODE_PROCESS, MODE_TICK
from fundsp_amd import graph as GR
from test_gpu_parity import assert_bit_equal, oracle_render, run_bank

pytestmark = pytest.mark.gpu
NAMES = list(CG.table(O, O))


def _bank(gpu, name, V):
    g, ring, _line = CG.table(GR, O)[name]
    for kind in GR.uses_wavetables(g):   # the oracle's numpy-built tables, so both sides read identical table bits
        t = O.Wavetable.get(kind)
        offs = np.concatenate([[0], np.cumsum(t.lengths)])
        gpu.wavetable_upload(kind, t.pitches, [t.data[offs[i]:offs[i + 1]] for i in range(len(t.lengths))])
    return gpu.Bank.from_graph(g, V, ring_frames=ring, sample_rate=CG.SAMPLE_RATE, fdn_kernel=False), g


def _oracle(name, seed):
    n = CG.table(O, O)[name][0]
    n.set_sample_rate(CG.SAMPLE_RATE)
    n.set_seed(int(seed))
    return n


@pytest.mark.parametrize("name", NAMES)
def test_criterion_graph_one_second_matches_the_oracle(gpu, name):
    V, T = 130, CG.FRAMES
    b, g = _bank(gpu, name, V)
    seeds = np.arange(V, dtype=np.uint64) * 104729 + 5
    b.set_seed(seeds)
    got = run_bank(b, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    assert got.shape == (V, g.nout, T)
    for v in (0, 63, 64, 129):
        assert_bit_equal(got[v], oracle_render(_oracle(name, seeds[v]), None, T, MODE_PROCESS), f"{name} instance {v}")
    if name not in ("pass",):   # (a constant: every instance the same)
        assert (got[0] != got[1]).any(), "instances with different seeds render different audio"


@pytest.mark.parametrize("name", NAMES)
def test_criterion_graph_tick_executor_and_planar_layout(gpu, name):
    V, T = 70, 64 * 7 + 5
    b, _g = _bank(gpu, name, V)
    seeds = np.arange(V, dtype=np.uint64) * 31 + 1
    for mode, layout in ((MODE_TICK, LAYOUT_VOICE_MINOR), (MODE_PROCESS, LAYOUT_PLANAR), (MODE_TICK, LAYOUT_PLANAR)):
        b.reset()
        b.set_seed(seeds)
        got = run_bank(b, None, T, layout, mode)
        for v in (0, 69):
            assert_bit_equal(got[v], oracle_render(_oracle(name, seeds[v]), None, T, mode), f"{name} instance {v} mode {mode} layout {layout}")


def test_reverb_bench_as_a_chain_of_two_banks(gpu):
    """(noise() | noise()) >> reverb_stereo(10, 1, 0.5): Bank.from_graph renders `generator >> stock reverb` as gpu.Chain(generator bank,
    lane-per-frame network bank).  Bit-equal to the oracle's rendering of the ONE graph and to the run-time compiled one-graph bank: as
    constructed (the hash Pipe::new's probe ping hands down, read from the device by a probe kind), after set_seed, in both executors and
    both layouts, across chunked launches, and for a chain the host puts together from two banks."""
    import torch

    V, T = 130, 64 * 40 + 17
    g = CG.table(GR, O)["reverb"][0]
    ch = gpu.Bank.from_graph(g, V, sample_rate=CG.SAMPLE_RATE)
    assert isinstance(ch, gpu.Chain) and ch.effect.kind == "reverb_stereo" and (ch.inputs(), ch.outputs()) == (0, 2)
    one, _g = _bank(gpu, "reverb", V)
    assert one.kind.startswith("jit_")

    def planar(bank, mode):
        out = bank.process(T, layout=LAYOUT_PLANAR, mode=mode)
        torch.cuda.synchronize()
        return out.cpu().numpy()[:, :, :T]

    # as constructed: every instance is the reference's freshly constructed graph
    got = planar(ch, MODE_PROCESS)
    n = CG.table(O, O)["reverb"][0]
    n.set_sample_rate(CG.SAMPLE_RATE)
    want = oracle_render(n, None, T, MODE_PROCESS)
    for v in (0, 64, 129):
        assert_bit_equal(got[v], want, f"as constructed, instance {v}")
    assert_bit_equal(got, run_bank(one, None, T, LAYOUT_PLANAR, MODE_PROCESS), "as constructed: chain == one graph")
    seeds = np.arange(V, dtype=np.uint64) * 7 + 3
    for mode in (MODE_PROCESS, MODE_TICK):
        ch.reset(); one.reset()
        ch.set_seed(seeds); one.set_seed(seeds)
        got = planar(ch, mode)
        assert_bit_equal(got, run_bank(one, None, T, LAYOUT_PLANAR, mode), f"chain == one graph, mode {mode}")
        for v in (0, 64, 129):
            assert_bit_equal(got[v], oracle_render(_oracle("reverb", seeds[v]), None, T, mode), f"chain instance {v} mode {mode}")
    # voice-minor buffers (the network bank's staging copy), and set_seed(None) = the construction hash again
    ch.reset(); ch.set_seed(seeds)
    vm = ch.process(T)
    torch.cuda.synchronize()
    ch.reset(); ch.set_seed(seeds)
    assert_bit_equal(vm.cpu().numpy().transpose(2, 0, 1), planar(ch, MODE_PROCESS), "voice-minor == planar")
    ch.reset(); ch.set_seed(None)
    assert_bit_equal(planar(ch, MODE_PROCESS)[77], want, "set_seed(None) re-applies the construction hash")
    # chunked launches continue the tail
    ch.reset(); ch.set_seed(seeds)
    a = ch.process(1000, layout=LAYOUT_PLANAR).cpu().numpy()[:, :, :1000]
    b2 = ch.process(T - 1000, layout=LAYOUT_PLANAR).cpu().numpy()[:, :, :T - 1000]
    n = _oracle("reverb", seeds[5])
    w2 = np.concatenate([n.render_blocks(None, length=1000, block=64), n.render_blocks(None, length=T - 1000, block=64)], axis=1)
    assert_bit_equal(np.concatenate([a, b2], axis=2)[5], w2, "chunked launches, instance 5")
    # Clone in mid-tail: the clone and the original continue alike
    ch.reset(); ch.set_seed(seeds)
    planar(ch, MODE_PROCESS)
    twin = ch.clone()
    assert_bit_equal(planar(twin, MODE_PROCESS), planar(ch, MODE_PROCESS), "clone continues like the original")
    # a handful of instances (below the network bank's staging threshold): voice-minor buffers gathered directly
    few = gpu.Bank.from_graph(g, 3, sample_rate=CG.SAMPLE_RATE)
    assert isinstance(few, gpu.Chain)
    out3 = few.process(T)
    torch.cuda.synchronize()
    assert_bit_equal(out3.cpu().numpy()[:, :, 2], want, "three instances, voice-minor, as constructed")
    # a chain put together by the host: two stand-alone nodes piped by hand
    src = gpu.Bank.from_graph(GR.noise() | GR.noise(), V, sample_rate=CG.SAMPLE_RATE)
    rev = gpu.Bank.from_graph(GR.reverb_stereo(10.0, 1.0, 0.5), V, sample_rate=CG.SAMPLE_RATE)
    by_hand = gpu.Chain(src, rev)
    ns, nr = O.noise() | O.noise(), O.reverb_stereo(10.0, 1.0, 0.5)
    ns.set_sample_rate(CG.SAMPLE_RATE); nr.set_sample_rate(CG.SAMPLE_RATE)
    assert_bit_equal(planar(by_hand, MODE_PROCESS)[3], nr.render_blocks(ns.render_blocks(None, length=T, block=64), block=64), "two stand-alone nodes")
