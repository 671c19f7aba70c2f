"""Limiter / Meter / Monitor / Feedback of the oracle against the reference's own tests for them -- CPU only.

tests/test_dynamics.rs:30-49 (limiter: a +100 dB edge never exceeds 1.0 and settles in 0.90..1.00), :51-75 (monitor and
meter agree), tests/test_basic.rs:646 (feedback arity), :348-351 (multitap tick == process), test_flow.rs:264-267
(allnest_c allpass responses, checked as |H| = 1 through energy preservation).
"""
import numpy as np
import pytest

import oracle as O


@pytest.mark.parametrize("samples", [2, 17, 480, 5000])
def test_limiter_edge(samples):  # test_dynamics.rs:30-49 (the reference draws 20 random lengths in 2..200 000)
    sr = 48000.0
    x = O.limiter(np.float32(samples) / np.float32(sr), np.float32(samples) / np.float32(sr))
    x.set_sample_rate(sr)
    x.render_ticks(np.zeros((1, samples), dtype=np.float32))
    edge = np.float32(10.0 ** (100.0 / 20.0))           # db_amp(100.0)
    y = x.render_ticks(np.full((1, samples), edge, dtype=np.float32))
    assert (y <= 1.0).all()
    value = x.tick([edge])[0]
    assert 0.90 <= value <= 1.00


def test_limiter_process_is_tick_and_reset():
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((2, 1500)) * 2).astype(np.float32)
    n = O.limiter_stereo(0.002, 0.02)
    a = n.render_blocks(x)
    b = O.limiter_stereo(0.002, 0.02).render_ticks(x)   # a fresh node: Limiter::reset leaves its follower's state and
    assert np.array_equal(a, b)                         # "first sample" coefficients alone (dynamics.rs:184-199)
    n.reset()
    c = n.render_blocks(x)
    assert not c[:, :round(44100.0 * float(np.float32(0.002)))].any() and np.abs(c).max() <= 1.0
    length = round(44100.0 * float(np.float32(0.002)))
    assert not a[:, :length].any() and a[:, length:].any()      # look-ahead delay = buffer length (:228-235)
    assert np.abs(a).max() <= 1.0


def test_monitor_and_meter_agree():  # test_dynamics.rs:51-75
    rng = np.random.default_rng(1)
    m1, m2 = O.monitor("sample"), O.meter("sample")
    for _ in range(1000):
        x = np.float32(rng.random())
        assert m1.tick([x])[0] == x and m2.tick([x])[0] == x and O.meter_level(m1) == x
    m1, m2 = O.monitor("peak", 0.1), O.meter("peak", 0.1)
    for _ in range(1000):
        x = np.float32(rng.random())
        x1, x2 = m1.tick([x])[0], m2.tick([x])[0]
        assert x1 == x and x2 >= 0.0 and x2 == O.meter_level(m1)


def test_feedback_arity_and_echo():
    fb = O.feedback(O.delay(0.5) * 0.5)                # test_basic.rs:646
    assert (fb.inputs(), fb.outputs()) == (1, 1)
    n = O.feedback(O.delay(10.0 / 44100.0) * 0.5)
    x = np.zeros((1, 45), dtype=np.float32)
    x[0, 0] = 1.0
    y = n.render_blocks(x)
    want = np.zeros(45, dtype=np.float32)
    want[10], want[21], want[32], want[43] = 0.5, 0.25, 0.125, 0.0625   # the loop adds one sample (value of the previous tick)
    assert np.array_equal(y[0], want)
    with pytest.raises(ValueError):
        O.fdn(O.stacki(3, lambda i: O.pass_()))        # FrameHadamard needs a power of two (feedback.rs:27)


def test_fdn_hadamard_is_orthonormal():
    """fdn(x) with x = unit delays: the feedback matrix is Hadamard / sqrt(N) (feedback.rs:35-57), so the total energy
    circulating in the loop is preserved from pass to pass."""
    n = O.fdn(O.multitick(4))
    x = np.zeros((4, 200), dtype=np.float32)
    x[:, 0] = [1.0, -0.5, 0.25, 2.0]
    y = n.render_blocks(x)
    e = (y.astype(np.float64) ** 2).sum(axis=0)
    assert np.allclose(e[1::2], e[1], rtol=1e-5) and not e[0::2].any()   # unit delay + the loop's own sample: period 2


def test_multitap_tick_equals_process():  # test_basic.rs:346-351 (the lfo closures replaced by sine LFOs)
    rng = np.random.default_rng(2)
    x = (rng.random((1, 2000)) * 2 - 1).astype(np.float32)
    for make in (lambda: (O.pass_() | O.sine_hz(0.9) * 0.4 + 0.5 | O.sine_hz(1.3) * 0.3 + 0.4) >> O.multitap(2, 0.0, 1.0),
                 lambda: (O.pass_() | O.sine_hz(0.7) * 0.4 + 0.5 | O.sine_hz(1.1) * 0.3 + 0.4) >> O.multitap_linear(2, 0.0, 1.0)):
        n = make()
        a = n.render_blocks(x)
        n.reset()
        assert np.max(np.abs(a - n.render_ticks(x))) <= 1e-4


def test_allnest_is_allpass():  # test_flow.rs:251-283: |H| = 1
    rng = np.random.default_rng(4)
    x = np.zeros((1, 8192), dtype=np.float32)
    x[0, :64] = rng.standard_normal(64)
    for n in (O.allnest_c(0.5, O.pass_()), O.allnest_c(0.6, O.tick()), O.allnest_c(0.7, O.allpole_delay(0.5)),
              (O.pass_() | O.dc(-0.6)) >> O.allnest(O.allpass_hz(3000.0, 3.0))):
        y = n.render_blocks(x)
        assert abs(float((y.astype(np.float64) ** 2).sum()) / float((x.astype(np.float64) ** 2).sum()) - 1.0) < 1e-4
