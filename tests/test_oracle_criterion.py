"""The reference's own criterion bench graphs (benches/benchmark.rs; tests/criterion_graphs.py), CPU side: the oracle renders all nine,
its C closures are the Python-callback closures bit for bit, tick == process within the reference's own check_wave bar, and the engine's
notation builds graphs of the same arity."""
import numpy as np
import pytest

import criterion_graphs as CG
import oracle as O
from fundsp_amd import graph as GR

f32 = np.float32
TAU = f32(6.2831855)


def test_oracle_renders_every_criterion_graph_and_both_notations_agree_on_arity():
    eng, ora = CG.table(GR, O), CG.table(O, O)
    assert list(eng) == list(ora) and len(eng) == 9
    assert set(CG.NOT_ON_THE_PATH) == {"resynth", "chorus", "wrap", "netpass"}       # 9 + 4 = the 13 benches of benchmark.rs:99-137
    for name, (n, _ring, _line) in ora.items():
        g = eng[name][0]
        assert (g.nin, g.nout) == (n.inputs(), n.outputs()) == (0, 2 if name == "reverb" else 1), name
        n.set_sample_rate(CG.SAMPLE_RATE)
        y = n.render_blocks(None, length=6000, block=64)    # (the limiter looks 0.1 s = 4 410 frames ahead: silence until then)
        assert y.shape == (g.nout, 6000) and np.isfinite(y).all() and np.abs(y).max() > 0, name


def test_pass_bench_is_six_everywhere():
    """dc((1.0, 2.0)) * 2.0 >> pass() + pass() >> pass()  (benchmark.rs:25-31): (1 * 2) + (2 * 2)"""
    n = CG.table(O, O)["pass"][0]
    assert (n.render_blocks(None, length=200, block=64) == f32(6.0)).all()


@pytest.mark.parametrize("name", ["envelope", "phaser"])
def test_c_closures_equal_the_python_closures(name):
    """o_envfn_criterion_* (fundsp_oracle.c) against the same closures played through the Python callback path the other tests use"""
    if name == "envelope":
        py = O.noise() * O.envelope(lambda t: O.m_expf(-t) * O.m_sinf(t * f32(1.0) * TAU))
    else:
        py = O.noise() >> O.phaser(0.5, lambda t: O.m_sinf(t * f32(0.1) * TAU) * f32(0.5) + f32(0.5))
    c = CG.table(O, O)[name][0]
    for n in (py, c):
        n.set_sample_rate(CG.SAMPLE_RATE)
        n.set_seed(77)
    a, b = py.render_blocks(None, length=6000, block=64), c.render_blocks(None, length=6000, block=64)
    assert (a.view(np.uint32) == b.view(np.uint32)).all()


@pytest.mark.parametrize("name", ["sine", "wavetable", "equalizer", "oversample"])
def test_tick_and_process_agree_like_check_wave(name):
    """the reference's own bar for two executors of one graph: 1e-4 per sample (tests/test_basic.rs:21-47)"""
    a, b = CG.table(O, O)[name][0], CG.table(O, O)[name][0]
    for n in (a, b):
        n.set_sample_rate(CG.SAMPLE_RATE)
        n.set_seed(5)
    # (Oversampler::process leaves the last sample of an ODD block unwritten, oversample.rs:163-200: whole blocks for that graph)
    T = 448 if name == "oversample" else 441
    ya, yb = a.render_blocks(None, length=T, block=64), b.render_ticks(None, length=T)
    scale = 100.0 if name == "sine" else 1.0    # the sum of 100 unit sines
    assert np.abs(ya - yb).max() <= 1e-4 * scale


def _generic_reverb_stereo(m, room_size, time, damping):
    """reverb_stereo spelled out of its parts (prelude.rs:1744-1763), as fundsp_amd.graph.reverb_stereo does, in notation `m`"""
    a = f32(GR._db_amp(-60.0) ** (0.03 * room_size / 10.0 / time))
    gain = f32(1.0) - f32(damping)
    alpha = (gain + f32(1.0)) / f32(2.0)
    beta = (f32(1.0) - alpha) / f32(2.0)
    w = (beta * a, alpha * a, beta * a)
    line = m.stacki(32, lambda i: m.delay(float(f32(GR.REVERB_DELAYS[i] * room_size / 10.0))) >> m.fir(*w))
    pans = m.sumf(32, lambda x: m.pan(f32(-1.0) * (f32(1.0) - GR._smooth9(x)) + f32(1.0) * GR._smooth9(x)))
    return m.multisplit(2, 16) >> m.fdn(line) >> pans * m.dc(1.0 / 16.0, 1.0 / 16.0)


def test_the_reverb_node_hands_on_the_hash_of_the_tree_it_stands_for():
    """The oracle's reverb_stereo is ONE native node standing for a tree of 100 nodes; AudioNode::ping walks that tree.  The hash it hands on
    reaches, through the probe ping of a constructor, every hashed node of the graph -- here the two noise generators in front of it, as
    constructed (the reverb bench's own situation) -- and the hashed node behind it after set_seed."""
    for tail in (False, True):
        def build(rev):
            g = (O.noise() | O.noise()) >> rev
            return (g >> (O.pass_() | O.pass_() * O.noise())) if tail else g
        a, b = build(O.reverb_stereo(10.0, 1.0, 0.5)), build(_generic_reverb_stereo(O, 10.0, 1.0, 0.5))
        for n in (a, b):
            n.set_sample_rate(CG.SAMPLE_RATE)
        ya, yb = a.render_blocks(None, length=4000, block=64), b.render_blocks(None, length=4000, block=64)
        assert np.abs(ya).max() > 0 and (ya.view(np.uint32) == yb.view(np.uint32)).all(), f"as constructed, tail={tail}"
        for n in (a, b):
            n.reset()
            n.set_seed(12345)
        ya, yb = a.render_blocks(None, length=4000, block=64), b.render_blocks(None, length=4000, block=64)
        assert (ya.view(np.uint32) == yb.view(np.uint32)).all(), f"after set_seed, tail={tail}"


def test_the_reverb_node_flushes_the_whole_graph_like_the_tree_it_stands_for():
    """Constructing reverb_stereo constructs a Feedback node, and Feedback::new switches the thread to FTZ + DAZ for good (feedback.rs:96,
    denormal.rs:18): the reference then renders the WHOLE graph flushed -- a filter in front of the reverb, a dry bus around it --, not just the
    reverb's own arithmetic.  The native node must mark its graph the way the generic tree's Feedback node does: an input in the denormal range
    through `(lowpass_hz | pass) >> (multipass() & 0.3 * reverb)` is where a node that only flushed inside its own tick gave the dry path's
    denormals away (found by the chain-of-two-banks test of exactly this graph, tests/test_gpu_reverb_bus.py)."""
    rng = np.random.default_rng(3)
    x = ((rng.random((2, 700), dtype=np.float32) * 2 - 1) * f32(1e-38)).astype(np.float32)
    x[:, 400:] *= f32(1e8)          # ... and back into the normal range
    for executor in ("render_blocks", "render_ticks"):
        outs = []
        for rev in (O.reverb_stereo(10.0, 1.0, 0.5), _generic_reverb_stereo(O, 10.0, 1.0, 0.5)):
            g = (O.lowpass_hz(800.0, 1.0) | O.pass_()) >> (O.multipass(2) & 0.3 * rev)
            g.set_sample_rate(CG.SAMPLE_RATE)
            outs.append(getattr(g, executor)(x))
        assert (outs[0].view(np.uint32) == outs[1].view(np.uint32)).all(), executor
        assert not outs[0][:, :400].any() and np.abs(outs[0][:, 400:]).max() > 1e-31, executor
