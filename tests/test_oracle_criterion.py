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
