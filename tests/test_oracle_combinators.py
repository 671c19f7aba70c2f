"""Routing leaves and the Bus / Branch / Thru / N-fold combinators of the oracle, pinned by the reference's own tests
for them -- CPU only.

tests/test_basic.rs: check_wave cases :170-218 (Wave::render == tick rendering within 1e-4, reset restores), bus vs.
branch equivalence :391-401 (exact), arities of the operator table :604-640.  audionode.rs doc semantics for
Split/Join (:527-660), MultiBus fold (:2117-2134).
"""
import numpy as np
import pytest

import oracle as O


def check_wave(make, frames=441, tol=1e-4):
    """check_wave (test_basic.rs:21-47), any number of output channels."""
    g = make()
    w = O.wave_render(44100.0, frames / 44100.0, g)
    g.reset()
    t = g.render_ticks(length=frames)
    assert w.shape == t.shape and np.max(np.abs(w - t)) <= tol
    g.reset()
    assert np.array_equal(O.wave_render(44100.0, frames / 44100.0, g), w)


def test_check_wave_combinator_cases():
    check_wave(lambda: O.noise() >> O.declick() | O.noise() + O.noise())                                   # :170
    check_wave(lambda: O.noise().seed(1) * O.noise() | O.busi(4, lambda i: O.mls_bits(10 + i)))            # :171
    check_wave(lambda: O.pink() & O.noise() | O.sine_hz(440.0) & -O.noise())                               # :172
    check_wave(lambda: O.dc(110.0, 220.0) >> O.multipass(2)
               >> -O.stackf(2, lambda f: (O.sine() * (float(f) - 0.5))))                                   # :182
    check_wave(lambda: O.dc(110.0, 220.0, 440.0, 880.0) >> O.multipass(4)
               >> (O.sink() | -O.sine().phase(0.0) | O.sink() | O.sine()))                                 # :183-187
    check_wave(lambda: O.dc(110.0, 220.0) >> O.declick_s(0.1) + O.pass_() >> (O.saw() ^ O.dsf_square_r(0.9)))  # :188
    check_wave(lambda: O.dc(20.0, 40.0) >> O.reverse(2) >> O.pass_() * O.pass_()
               >> (O.dsf_saw_r(0.999) ^ O.square() * 0.1))                                                 # :189-191
    check_wave(lambda: O.dc(880.0, 440.0) >> O.pass_() - O.pass_()
               >> O.branchf(2, lambda f: O.triangle() * (float(f) - 0.5)))                                 # :194-196
    check_wave(lambda: (O.noise() | O.dc(440.0)) >> O.pipei(3, lambda _: ~O.lowpole()) >> O.lowpole()
               | ((O.mls() | O.dc(880.0)) >> ~O.butterpass() >> O.butterpass()))                           # :197-200
    check_wave(lambda: (O.brown() | O.dc(440.0)) >> O.pipei(4, lambda _: ~O.peak_q(1.0)) >> O.bell_q(1.0, 2.0)
               | ((O.mls() | O.dc(880.0)) >> ~O.lowshelf_q(1.0, 0.5) >> O.highshelf_q(2.0, 2.0)))          # :201-204
    check_wave(lambda: (O.dc(110.0) >> O.square().wave_phase(0.25) | O.dc(440.0))
               >> O.pipei(4, lambda _: ~O.lowpass_q(1.0)) >> O.highpass_q(1.0)
               | ((O.mls() | O.dc(880.0)) >> ~O.bandpass_q(1.0) >> O.notch_q(2.0)))                        # :205-210
    check_wave(lambda: O.dc(440.0, 880.0) >> O.multisplit(2, 5) >> O.sumi(10, lambda _: O.saw() * 0.1)
               | O.saw_hz(220.0) * 0.1)                                                                    # :211-214
    check_wave(lambda: O.dc(440.0, 880.0) >> O.multisplit(2, 3) >> O.multijoin(2, 3) >> (O.sine() | O.sine()))  # :215-217
    check_wave(lambda: (O.noise() >> O.split(16) >> O.join(16)) | (O.noise() >> O.split(11) >> O.join(11)))     # :218


def test_check_wave_pulse_and_table_family():
    check_wave(lambda: O.dc(110.0, 0.5) >> O.pulse() * 0.2 >> O.delay(0.1), frames=4410)   # test_basic.rs:237
    check_wave(lambda: O.organ_hz(110.0) * 0.5)
    check_wave(lambda: O.soft_saw_hz(220.0) | O.hammond_hz(55.0) * 0.5)
    a, b = O.pulse() | O.pulse(), None                                                      # outputs_diverge :606
    y = a.render_ticks(np.tile(np.array([[110.0], [0.5], [110.0], [0.5]], dtype=np.float32), (1, 64)))
    assert not np.array_equal(y[0], y[1])


def is_equal(x, y, trials=1000, seed=0):
    """is_equal (test_basic.rs:95-110): random frames from {-1, 0, 1}, exact tick outputs."""
    rng = np.random.default_rng(seed)
    assert (x.inputs(), x.outputs()) == (y.inputs(), y.outputs())
    for _ in range(trials):
        frame = rng.integers(-1, 2, size=x.inputs()).astype(np.float32)
        if not np.array_equal(x.tick(frame), y.tick(frame)):
            return False
    return True


def test_bus_vs_branch_equivalence():  # test_basic.rs:385-401
    w, x, y, z = -2.0, 3.0, -4.0, 5.0
    assert is_equal((O.pass_() ^ O.mul(y)) >> O.add(z) + O.sub(x), O.add(z) & O.mul(y) >> O.sub(x))
    assert is_equal((O.pass_() ^ O.mul(y) ^ O.add(w)) >> O.add(z) + O.sub(x) + O.mul(y),
                    O.add(z) & O.mul(y) >> O.sub(x) & O.add(w) >> O.mul(y))


def inouts(n): return n.inputs(), n.outputs()


def test_operator_table_arities():  # test_basic.rs:604-640
    assert inouts(-(-O.sink()) - 42.0 ^ O.sink() & -(-(-O.sink())) * 3.15) == (1, 0)
    assert inouts(O.pass_() ^ O.pass_()) == (1, 2)
    assert inouts(O.mul(0.5) + O.mul(0.5)) == (2, 1)
    assert inouts(O.pass_() ^ O.pass_() ^ O.pass_()) == (1, 3)
    assert inouts(O.sink() | O.zero()) == (1, 1)
    assert inouts(O.sink() | O.pass_()) == (2, 1)
    assert inouts(O.sink() | O.zero() | O.pass_()) == (2, 2)
    assert inouts(O.mul(0.0, 1.0)) == (2, 2)
    assert inouts(~O.butterpass() >> O.lowpole()) == (2, 1)
    assert inouts(~O.butterpass() >> ~O.butterpass() >> O.butterpass()) == (2, 1)
    with pytest.raises(ValueError):
        O.pass_() & O.sink()                       # Bus needs equal arities
    with pytest.raises(ValueError):
        O.pass_() ^ O.dc(1.0)                      # Branch needs equal inputs


def test_join_tick_divides_process_scales():
    """Join: tick = (x0 + x1 + x2) / 3 (audionode.rs:638-644), process = x0*z + x1*z + x2*z, z = 1/3 (:649-659)."""
    f32 = np.float32
    x = np.array([[0.1], [0.7], [-0.33]], dtype=np.float32)
    j = O.join(3)
    t = j.tick(x[:, 0])
    assert t[0] == (x[0, 0] + x[1, 0] + x[2, 0]) / f32(3)
    blk = np.zeros((3, 64), dtype=np.float32)
    blk[:, 0] = x[:, 0]
    z = f32(1.0) / f32(3)
    assert j.process(1, blk)[0, 0] == x[0, 0] * z + x[1, 0] * z + x[2, 0] * z


def test_multibus_zero_sign():
    """MultiBus::tick folds from a +0.0 frame (audionode.rs:2117-2121): a lone -0.0 becomes +0.0; process starts from
    the first node's output (:2123-2134) and keeps the sign."""
    b = O.busi(2, lambda i: O.pass_())
    nz = np.float32(-0.0)
    assert not np.signbit(b.tick([nz])[0])
    blk = np.zeros((1, 64), dtype=np.float32)
    blk[0, 0] = nz
    assert np.signbit(b.process(1, blk)[0, 0])


def test_thru_and_impulse():
    t = ~O.lowpole_hz(100.0)                        # 1 in, 1 out: plain filter
    assert inouts(t) == (1, 1)
    cut = ~(O.pass_() ^ O.pass_())                  # X has more outputs than inputs: surplus output cut (:1990-1999)
    assert inouts(cut) == (1, 1)
    keep = ~O.sink()                                # X has no outputs: input passes through
    x = np.arange(70, dtype=np.float32).reshape(1, -1)
    assert np.array_equal(keep.render_blocks(x), x) and np.array_equal(cut.render_blocks(x), x)
    imp = O.impulse(2)
    y = imp.render_blocks(length=70)
    assert y.shape == (2, 70) and (y[:, 0] == 1).all() and not y[:, 1:].any()
    imp.reset()
    assert np.array_equal(imp.render_ticks(length=70), y)
