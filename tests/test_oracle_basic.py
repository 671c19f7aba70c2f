"""Structural / self-consistency invariants of the oracle, re-expressed from the reference's tests -- CPU only.

tests/test_basic.rs: check_wave :21-47 (Wave::render == tick path within 1e-4 after reset), exact delay identity
:520-529 (tick >> tick >> tick == delay(3/44100)), constants :365-378, outputs_diverge :532-612 (identical
generators in one graph get different pseudorandom phases), doc-test audionode.rs:44-49 (reset determinism).
"""
import numpy as np
import pytest

import oracle as O


def check_wave(make, frames=441, tol=1e-4):
    """check_wave (test_basic.rs:21-47) for mono generators."""
    g = make()
    w = O.wave_render(44100.0, frames / 44100.0, g)
    g.reset()
    t = g.render_ticks(length=frames)
    assert np.max(np.abs(w - t)) <= tol
    g.reset()
    assert np.array_equal(O.wave_render(44100.0, frames / 44100.0, g), w)  # reset restores the initial state


def test_check_wave_hot_path_generators():
    check_wave(lambda: O.sine_hz(440.0))
    check_wave(lambda: O.noise())
    check_wave(lambda: O.noise().seed(1) * O.noise())
    check_wave(lambda: O.sine_hz(220.0) >> O.lowpass_hz(1000.0, 1.0))
    check_wave(lambda: O.noise() >> O.moog_hz(1500.0, 0.5))                       # test_basic.rs:276-290 (moog_hz)
    check_wave(lambda: O.noise() >> O.resonator_hz(440.0, 110.0))                 # :343-346
    check_wave(lambda: (O.noise() | O.dc(1000.0, 2.0)) >> O.svf("lowpass"))       # :201-210 (lowpass_q etc.)
    check_wave(lambda: (O.noise() | O.dc(1000.0, 2.0, 3.0)) >> O.svf("bell"))
    check_wave(lambda: (O.sine_hz(110.0) * 110.0 * 2.0 + 110.0) >> O.sine() >> O.lowpass_hz(2000.0, 1.0))  # config 3


def test_check_wave_filter_biquad_bank():
    """check_wave_filter (test_basic.rs:72-92) + biquad_bank case :314-317: a bank equals 8 independent biquads
    bit-for-bit in both the tick and the process path (lane arithmetic is element-wise f32x8)."""
    rng = np.random.default_rng(0)
    x = (rng.random((8, 2000), dtype=np.float32) * 2 - 1).astype(np.float32)
    coefs = [O.biquad_coefs("lowpass", 44100.0, 200.0 * (i + 1), 0.7 + 0.3 * i) for i in range(8)]
    bank = O.biquad_bank()
    for i in range(8):
        O.set_biquad_bank(bank, i, coefs[i])
    yb = bank.render_blocks(x)
    bank.reset()
    assert np.array_equal(bank.render_ticks(x), yb)
    for i in range(8):
        assert np.array_equal(O.biquad(*coefs[i]).render_blocks(x[i]), yb[i:i + 1])


def test_tick_chain_equals_delay():  # test_basic.rs:520-529
    rng = np.random.default_rng(1)
    x = rng.integers(-100, 100, size=(1, 500)).astype(np.float32)
    a = (O.tick() >> O.tick() >> O.tick()).render_blocks(x)
    b = O.delay(3.0 / 44100.0).render_blocks(x)
    assert np.array_equal(a, b)
    assert np.array_equal(a[0, 3:], x[0, :-3]) and not a[0, :3].any()


def test_constants():  # test_basic.rs:365-378
    g = O.dc(2.0) * 3.0 + 0.5
    assert g.tick()[0] == 6.5
    assert np.all(O.wave_render(44100.0, 100 / 44100.0, g) == 6.5)
    s = O.dc(1.0, 2.0) | O.dc(3.0)
    assert list(s.tick()) == [1.0, 2.0, 3.0]
    assert (3.0 - O.dc(1.0)).tick()[0] == 2.0 and (O.dc(1.0) - 3.0).tick()[0] == -2.0 and (-O.dc(1.0)).tick()[0] == -1.0


def test_outputs_diverge():  # test_basic.rs:532-612: identical generators in one graph get different phases
    g = O.sine_hz(440.0) | O.sine_hz(440.0)
    w = O.wave_render(44100.0, 0.01, g)
    assert np.max(np.abs(w[0] - w[1])) > 0.1
    g2 = O.noise() | O.noise()
    w2 = O.wave_render(44100.0, 0.01, g2)
    assert not np.array_equal(w2[0], w2[1])
    # ... while two separately built identical graphs are identical (deterministic hashing)
    assert np.array_equal(O.wave_render(44100.0, 0.01, O.sine_hz(440.0) | O.sine_hz(440.0)), w)


def test_set_seed_gives_distinct_reproducible_voices():
    def voice(seed):
        g = (O.sine_hz(200.0) * 200.0 * 1.5 + 200.0) >> O.sine() >> O.lowpass_hz(3000.0, 1.0)
        g.set_sample_rate(48000.0)
        g.set_seed(seed)
        return g.render_blocks(length=300)
    assert np.array_equal(voice(5), voice(5))
    assert not np.array_equal(voice(5), voice(6))


def test_process_block_size_edge_cases():
    """size = 0 is a no-op, ragged sizes go through process_remainder (audionode.rs:81-126)."""
    g = O.sine_hz(440.0) >> O.lowpass_hz(1000.0, 1.0)
    ref = O.sine_hz(440.0) >> O.lowpass_hz(1000.0, 1.0)
    out = []
    for size in (0, 1, 7, 8, 9, 63, 64, 0, 13):
        blk = g.process(size)
        out.append(blk[0, :size])
    got = np.concatenate(out)
    # same chunking through the generic executor
    want = []
    for size in (0, 1, 7, 8, 9, 63, 64, 0, 13):
        want.append(ref.process(size)[0, :size])
    assert np.array_equal(got, np.concatenate(want))
    assert got.shape[0] == 165 and np.isfinite(got).all()


def test_sample_rate_semantics():
    """Biquad keeps raw coefficients across set_sample_rate (biquad.rs:179-181); SVF / Moog / Resonator /
    ButterLowpass recompute (svf.rs:989-992, moog.rs:76-79, biquad.rs:263-267,349-352) -- SURVEY.md App. B.14."""
    c44 = O.svf_coefs("lowpass", 44100.0, 1000.0, 1.0)
    c48 = O.svf_coefs("lowpass", 48000.0, 1000.0, 1.0)
    assert not np.array_equal(c44, c48)
    rng = np.random.default_rng(3)
    x = (rng.random((1, 256), dtype=np.float32) - 0.5).astype(np.float32)
    n = O.lowpass_hz(1000.0, 1.0)
    y44 = n.render_blocks(x)
    n.reset()
    n.set_sample_rate(48000.0)
    y48 = n.render_blocks(x)
    assert not np.array_equal(y44, y48)
    b = O.biquad(*O.biquad_coefs("lowpass", 44100.0, 1000.0, 1.0))
    y1 = b.render_blocks(x)
    b.reset()
    b.set_sample_rate(96000.0)
    assert np.array_equal(b.render_blocks(x), y1)


def test_moog_with_inputs_recomputes_every_sample():
    """moog() (3 inputs) with constant control inputs equals moog_hz (moog.rs:83-85 vs Moog::new)."""
    rng = np.random.default_rng(4)
    x = (rng.random((1, 300), dtype=np.float32) - 0.5).astype(np.float32)
    a = O.moog_hz(2000.0, 0.4)
    a.set_sample_rate(48000.0)
    ctl = np.concatenate([x, np.full((1, 300), 2000.0, np.float32), np.full((1, 300), 0.4, np.float32)])
    b = O.moog()
    b.set_sample_rate(48000.0)
    assert np.array_equal(a.render_blocks(x), b.render_blocks(ctl))


def test_wavesynth_check_wave_and_tables():
    """tests/test_basic.rs:188-216: saw/square/triangle via Wave::render equal the tick path within 1e-4;
    table layout as derived in SURVEY.md section 7 (40 tables, 41 024 floats)."""
    for kind in ("saw", "square", "triangle"):
        check_wave(lambda k=kind: O.dc(110.0) >> O.wavesynth(k))
        check_wave(lambda k=kind: O.dc(1760.0) >> O.wavesynth(k))
    p, w = O.make_wavetable_arrays("saw")
    assert len(p) == 40 and sum(len(x) for x in w) == 41024
    assert len(w[0]) == 8192 and len(w[-1]) == 32 and abs(max(np.abs(x).max() for x in w) - 1.0) < 1e-6
    # a saw at 100 Hz has the 1/n spectrum (bandlimited below 20 kHz)
    g = O.dc(100.0) >> O.wavesynth("saw")
    y = O.wave_render(48000.0, 1.0, g)[0].astype(np.float64)
    spec = np.abs(np.fft.rfft(y * np.hanning(len(y))))
    h1, h2, h3 = spec[100], spec[200], spec[300]
    assert abs(h1 / h2 - 2.0) < 0.05 and abs(h1 / h3 - 3.0) < 0.05


def test_adsr_live_shape():
    """adsr.rs:21-70: nothing before the first low->high gate transition; attack to 1, decay to sustain, release to 0."""
    sr = 48000.0
    gate = np.zeros((1, 48000), dtype=np.float32)
    gate[0, 100:24000] = 1.0
    e = O.adsr_live(0.01, 0.1, 0.6, 0.2)
    e.set_sample_rate(sr)
    y = e.render_blocks(gate)[0]
    assert np.all(y[:100] == 0.0)
    assert abs(y[100 + 480 + 150] - 1.0) < 0.25 and y.max() <= 1.0 + 1e-6      # peak around the end of the attack
    assert np.all(np.abs(y[12000:23000] - 0.6) < 1e-3)                         # sustain
    assert y[24000 + 9600 + 400] == 0.0 or abs(y[24000 + 9600 + 400]) < 1e-6   # released after 0.2 s
    # a gate that is high from the very first sample never attacks (release_start starts at -1, adsr.rs:30,37-43)
    e2 = O.adsr_live(0.01, 0.1, 0.6, 0.2)
    e2.set_sample_rate(sr)
    assert not e2.render_blocks(np.ones((1, 4000), dtype=np.float32)).any()


def test_pan_weights_exact():
    """pan.rs:13-17: centre pan gives cos(pi/4) on both sides (libm cosf/sinf), hard left/right give (1, ~0)/(~0, 1)."""
    x = np.ones((1, 8), dtype=np.float32)
    c = O.pan(0.0).render_blocks(x)
    assert abs(c[0, 0] - np.float32(np.sqrt(0.5))) < 1e-7 and abs(c[1, 0] - np.float32(np.sqrt(0.5))) < 1e-7
    l = O.pan(-1.0).render_blocks(x)
    assert l[0, 0] == 1.0 and abs(l[1, 0]) < 1e-7
    assert np.array_equal(O.pan(5.0).render_blocks(x), O.pan(1.0).render_blocks(x))  # clamp11


def test_nonlinear_biquads_and_polyblep_check_wave():
    """tests/test_basic.rs:219-237: dbell_hz/dhighpass_hz/dresonator_hz/dlowpass_hz/fbell_hz/flowpass_hz/fresonator_hz/
    fhighpass_hz with Tanh/Softsign/Atan/Clip shapes, ramp; :308-312 PolyBLEP oscillators."""
    cases = [
        lambda: O.noise() >> O.nlbiquad(True, 1, "bell", "tanh", 1.0, 0, 1000.0, 10.0, 2.0),
        lambda: O.noise() >> O.nlbiquad(True, 1, "highpass", "softsign", 1.0, 0, 2000.0, 2.0),
        lambda: O.noise() >> O.nlbiquad(True, 1, "resonator", "tanh", 0.5, 0, 1000.0, 10.0),
        lambda: O.noise() >> O.nlbiquad(True, 1, "lowpass", "softsign", 0.5, 0, 2000.0, 2.0),
        lambda: O.noise() >> O.nlbiquad(False, 1, "bell", "atan", 1.0, 0, 500.0, 50.0, 0.5),
        lambda: O.noise() >> O.nlbiquad(False, 1, "lowpass", "clip", 1.0, 0, 2000.0, 2.0),
        lambda: O.noise() >> O.nlbiquad(False, 1, "resonator", "atan", 0.5, 0, 500.0, 50.0),
        lambda: O.noise() >> O.nlbiquad(False, 1, "highpass", "softsign", 0.2, 0, 2000.0, 2.0),
        lambda: O.dc(440.0) >> O.ramp(),
        lambda: O.dc(110.0) >> O.poly_saw(),
        lambda: O.dc(220.0) >> O.poly_square(),
        lambda: O.dc(220.0, 0.3) >> O.poly_pulse(),
        lambda: O.noise() >> O.shape("tanh", 2.0),
        lambda: O.noise() >> O.shape("atan", 1.0),      # Shaper::process uses wide atan, tick uses libm atanf
        lambda: O.noise() >> O.shape("softsign", 3.0),
        lambda: O.noise() >> O.shape("soft_crush", 4.0),
    ]
    for make in cases:
        check_wave(make)


def test_atan_restatements_and_polyblep_saw_shape():
    L = O.lib()
    xs = np.linspace(-40.0, 40.0, 20001).astype(np.float32)
    ref = np.arctan(xs.astype(np.float64))
    got = np.array([L.o_math_atanf(float(x)) for x in xs], dtype=np.float64)
    ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(got - ref)[xs != 0] / ulp[xs != 0]) < 1.0          # musl atanf: < 1 ulp
    gw = np.array([L.o_math_wide_atanf(float(x)) for x in xs], dtype=np.float64)
    assert np.max(np.abs(gw - ref)) < 3e-7                                  # vectorclass atan_f
    y = O.wave_render(48000.0, 0.1, O.dc(100.0) >> O.poly_saw())[0]
    assert y.min() > -1.05 and y.max() < 1.05
    d = np.diff(y)
    assert np.sum(d < -1.0) in (9, 10, 11)                                  # ten downward resets in 0.1 s at 100 Hz


def test_taps_check_wave_and_allnest_allpass():
    """tests/test_basic.rs:347-352 (multitap / multitap_linear through check_wave) and test_flow.rs:251-283 (allnest)."""
    check_wave(lambda: (O.noise() | O.dc(0.004)) >> O.tap(0.001, 0.01))
    check_wave(lambda: (O.noise() | O.dc(0.004)) >> O.tap_linear(0.0, 0.01))
    for make in (lambda: O.allnest_c(0.5, O.pass_()), lambda: O.allnest_c(0.6, O.tick()),
                 lambda: O.allnest_c(-0.6, O.allpass_hz(3000.0, 3.0)), lambda: O.allnest_c(0.4, O.delay(0.001))):
        n = make()
        n.set_sample_rate(44100.0)
        x = np.zeros((1, 0x8000), dtype=np.float32)
        x[0, 0] = 1.0
        mag = np.abs(np.fft.rfft(n.render_blocks(x)[0].astype(np.float64)))[1:]
        assert np.all(np.abs(mag - 1.0) <= 1e-5)
    # a tap at an (almost) integer delay reproduces the input (Catmull-Rom passes through its knots); (96/sr)*sr is not
    # exactly 96 in f32, hence the small tolerance
    sr = 48000.0
    rng = np.random.default_rng(5)
    x = (rng.random((1, 600), dtype=np.float32) - 0.5).astype(np.float32)
    t = O.tap(0.0, 0.01)
    t.set_sample_rate(sr)
    y = t.render_ticks(np.concatenate([x, np.full((1, 600), 96.0 / sr, np.float32)]))
    assert np.max(np.abs(y[0, 96:] - x[0, :-96])) < 1e-5


# ---- SURVEY 8(f) row 1, second half: Rez, Follow, AFollow; and Mls (8a row a16 sibling) -------------------------
def test_mls_sequences_are_maximum_length():
    """Pins the transcription of MLS_POLY (noise.rs:23-55): every table entry must generate a cycle of exactly
    2^n - 1 states (that is what 'maximum length sequence' means), and one period holds 2^(n-1) ones."""
    for n in range(1, 25):
        assert O.lib().o_mls_period(n) == (1 << n) - 1, n
    for n in (25, 28, 31):  # the long ones: a few seconds of C
        assert O.lib().o_mls_period(n) == (1 << n) - 1, n
    node = O.mls_bits(10)
    node.set_seed(5)
    y = node.render_ticks(length=1023)[0]
    assert set(np.unique(y)) == {-1.0, 1.0}
    assert int((y > 0).sum()) == 512          # balance property of an MLS
    assert np.array_equal(node.render_ticks(length=1023)[0], y)  # period 1023


def test_follow_halfway_response():
    """follow(t) reaches halfway to a new value after t seconds (follow.rs:12-24: accurate to 0.5 %)."""
    sr = 48000.0
    for rt in (0.001, 0.01, 0.1):
        n = O.follow(rt)
        n.set_sample_rate(sr)
        x = np.ones((1, int(rt * sr * 3)), dtype=np.float32)
        x[0, 0] = 0.0                       # the first sample is taken over as is (coeff_now = 1, follow.rs:93,104-109)
        y = n.render_ticks(x)[0]
        assert y[0] == 0.0
        assert abs(y[int(round(rt * sr))] - 0.5) < 0.02, (rt, y[int(round(rt * sr))])
        assert np.all(np.diff(y) >= 0)      # three one-poles in series: monotone step response


def test_afollow_attack_release_asymmetry():
    sr = 48000.0
    n = O.afollow(0.001, 0.05)
    n.set_sample_rate(sr)
    x = np.concatenate([np.zeros(1), np.ones(4800), np.zeros(4800)]).astype(np.float32)[None, :]
    y = n.render_ticks(x)[0]
    up = np.argmax(y[1:4801] >= 0.5)            # halfway up after ~ attack time
    down = np.argmax(y[4801:] <= 0.5)           # halfway down after ~ release time
    assert abs(up - 48) <= 3 and abs(down - 2400) <= 60, (up, down)
    m = O.afollow(0.01, 0.01)                   # equal times behave like follow() after the first sample
    f = O.follow(0.01)
    m.set_sample_rate(sr); f.set_sample_rate(sr)
    z = np.random.default_rng(3).random((1, 500), dtype=np.float32)
    assert np.allclose(m.render_ticks(z), f.render_ticks(z), atol=2e-6)


def test_rez_lowpass_bandpass_and_inputs_variant():
    sr = 48000.0
    lp, bp = O.lowrez_hz(1000.0, 0.5), O.bandrez_hz(1000.0, 0.5)
    lp.set_sample_rate(sr); bp.set_sample_rate(sr)
    dc = np.ones((1, 4000), dtype=np.float32) * 0.25
    assert abs(lp.render_ticks(dc)[0, -1] - 0.25) < 1e-3       # lowpass passes DC
    assert abs(bp.render_ticks(dc)[0, -1]) < 1e-3              # bandpass blocks it
    # Rez<U3> with constant cutoff / q inputs == Rez<U1> (rez.rs:68-75 only re-derives on change)
    fixed, var = O.lowrez_hz(700.0, 0.3), O.lowrez()
    fixed.set_sample_rate(sr); var.set_sample_rate(sr)
    x = (np.random.default_rng(4).random((1, 300), dtype=np.float32) * 2 - 1).astype(np.float32)
    xin = np.concatenate([x, np.full_like(x, 700.0), np.full_like(x, 0.3)])
    assert np.array_equal(fixed.render_ticks(x), var.render_ticks(xin))


# ---- Oversampler (oversample.rs) --------------------------------------------------------------
def test_oversample_passband_and_tick_process_identity():
    sr = 48000.0
    n = O.oversample(O.pass_())
    n.set_sample_rate(sr)
    dc = np.ones((1, 600), dtype=np.float32)
    y = n.render_ticks(dc)[0]
    assert abs(y[-1] - 1.0) < 2e-3                       # interpolator gain 2 x decimator gain 1 x zero-stuffing 1/2
    t = np.arange(4800) / sr
    for f, lo, hi in ((1000.0, 0.98, 1.02), (10000.0, 0.95, 1.05)):
        n.reset()
        x = np.sin(2 * np.pi * f * t).astype(np.float32)[None, :]
        y = n.render_ticks(x)[0, 600:]
        assert lo < np.sqrt(2 * np.mean(y.astype(np.float64) ** 2)) < hi, f
    # process == tick bit for bit when the enclosed node's process is its tick (even block sizes)
    a, b = O.oversample(O.lowpass_hz(3000.0, 0.7) >> O.shape("tanh")), O.oversample(O.lowpass_hz(3000.0, 0.7) >> O.shape("tanh"))
    a.set_sample_rate(sr); b.set_sample_rate(sr)
    x = (np.random.default_rng(9).random((1, 64 * 4 + 30), dtype=np.float32) * 2 - 1).astype(np.float32)
    assert np.array_equal(a.render_ticks(x), b.render_blocks(x))
    # odd tail: the reference's process never writes the last sample of an odd block (oversample.rs:184 `size / 2`)
    c = O.oversample(O.pass_())
    c.set_sample_rate(sr)
    z = c.render_blocks(np.ones((1, 64 + 5), dtype=np.float32))[0]
    assert z[-1] == 0.0 and z[-2] != 0.0


def test_oversampled_fm_generator():
    """README.md:1631: oversample(sine_hz(f) * f * m + f >> sine()) -- a generator; inner node runs at 2 x sr."""
    sr = 48000.0
    f, m = 440.0, 2.0
    g = O.oversample(O.sine_hz(f) * f * m + f >> O.sine())
    g.set_sample_rate(sr)
    g.set_seed(3)
    y = g.render_ticks(length=4800)[0]
    assert 0.6 < np.max(np.abs(y[600:])) <= 1.05
    h = O.oversample(O.sine_hz(f) * f * m + f >> O.sine())
    h.set_sample_rate(sr)
    h.set_seed(3)
    z = h.render_blocks(length=4800)[0]
    # tick vs process differ only by Sine's own tick / process arithmetic (unwrapped f32 phase inside a block), which at
    # the inner 96 kHz rate and modulated frequencies up to 1.3 kHz stays below 1e-3
    assert np.max(np.abs(y - z)) < 1e-3


def test_dsf_is_the_partial_sum_it_claims_to_be():
    """Dsf (oscillator.rs:105-187): closed form of sum_{i<=n} r^i sin(f + i d) (Moorer 1976), n = floor(22050 / f / spacing)."""
    sr, f0 = 48000.0, 441.0
    for node, spacing, r in ((O.dsf_saw_r(0.5), 1.0, 0.5), (O.dsf_square_r(0.7), 2.0, 0.7)):
        node.set_sample_rate(sr)
        node.phase(0.125)                                   # Setting::phase (oscillator.rs:192)
        T = 300
        y = node.render_ticks(np.full((1, T), f0, dtype=np.float32))[0]
        n = np.floor(22050.0 / f0 / spacing)
        k = np.arange(0, int(n) + 1)
        ph = (0.125 + f0 / sr * np.arange(1, T + 1)) % 1.0  # tick advances the phase before evaluating (:176-178)
        want = np.array([(r ** k * np.sin(p * 2 * np.pi + k * p * 2 * np.pi * spacing)).sum() for p in ph])
        assert np.max(np.abs(want - y)) < 2e-3, (spacing, np.max(np.abs(want - y)))
    two = O.dsf_saw()
    two.set_sample_rate(sr); two.set_seed(9)
    one = O.dsf_saw_r(0.3)
    one.set_sample_rate(sr); one.set_seed(9)
    x = np.stack([np.full(200, 300.0), np.full(200, 0.3)]).astype(np.float32)
    assert np.array_equal(two.render_ticks(x), one.render_ticks(x[:1]))   # roughness input == fixed roughness


def test_pluck_is_tuned_and_decays():
    """Karplus-Strong (oscillator.rs:215-317): period = sample_rate / frequency (loop delay + 1 damping + allpass), and the
    loop gain makes the level fall by gain_per_second per second."""
    sr, f, gps = 48000.0, 220.0, 0.5
    exc = np.random.default_rng(21).uniform(-1, 1, 4096).astype(np.float32)
    n = O.pluck(f, gps, 0.1, exc)
    n.set_sample_rate(sr)
    y = n.render_ticks(np.zeros((1, 48000), dtype=np.float32))[0].astype(np.float64)
    seg = y[2000:2000 + 8192]
    ac = np.correlate(seg, seg, "full")[len(seg) - 1:]
    lag = 100 + int(np.argmax(ac[100:400]))
    assert abs(lag - sr / f) <= 1, lag
    spec = np.abs(np.fft.rfft(y[:32768] * np.hanning(32768)))
    peak = np.argmax(spec[50:]) + 50
    h = peak * sr / 32768 / f                                 # the strongest partial is a harmonic of f:
    assert abs(h - round(h)) < 0.02, h                        # the allpass does the fine tuning
    e1, e2 = np.sqrt(np.mean(y[4800:9600] ** 2)), np.sqrt(np.mean(y[4800 + 24000:9600 + 24000] ** 2))
    assert e2 < e1 * 0.8                                      # decays (gain 0.5/s plus high-frequency damping)
    n.reset()                                                 # reset -> the line is re-initialised from the same excitation
    assert np.array_equal(n.render_ticks(np.zeros((1, 500), dtype=np.float32))[0], y[:500].astype(np.float32))


def test_envelope_samples_its_closure_and_tick_matches_process():
    """Envelope (envelope.rs:17-179): lfo(|t| exp(-t)) follows exp(-t) within the 2 ms linear-interpolation error, the
    segment grid is jittered deterministically from the hash, tick == process (check_wave's 1e-4 bar; here exact-ish)."""
    sr = 48000.0
    f = lambda t: O.m_expf(-t)
    a, b = O.lfo(f), O.lfo(f)
    for n in (a, b):
        n.set_sample_rate(sr)
        n.set_seed(77)
    T = 64 * 40 + 13
    y = a.render_ticks(length=T)[0]
    z = b.render_blocks(length=T)[0]
    t = np.arange(T) / sr
    assert np.max(np.abs(y - np.exp(-t))) < 1e-5
    assert np.max(np.abs(y - z)) < 1e-5   # tick accumulates t per sample, process per chunk: different f32 rounding of t
    two = O.envelope(lambda t: (O.m_sinf(t * np.float32(2.0) * np.float32(6.2831855)), O.m_cosf(t * np.float32(2.0) * np.float32(6.2831855))), outputs=2)
    two.set_sample_rate(sr)
    w = two.render_blocks(length=4800)
    assert w.shape == (2, 4800)
    assert np.max(np.abs(w[0] ** 2 + w[1] ** 2 - 1.0)) < 1e-3


def test_resample_speed_one_is_a_one_sample_delay_and_speed_two_skips():
    """Resample (resample.rs:205-315): Catmull-Rom through the knots -- at speed 1 the output is the generator read from
    its third sample on (consumer starts at 1.0 and is advanced before the read); at speed 2 every second sample; at
    speed 0.5 every other output is a knot."""
    sr = 48000.0
    gen = lambda: O.constant(330.0) >> O.sine().phase(0.25)   # explicit phase: independent of the ping hash
    inner = gen()
    inner.set_sample_rate(sr)
    ref = inner.render_ticks(length=600)[0]
    for speed, idx in ((1.0, np.arange(2, 202)), (2.0, np.arange(3, 403, 2))):
        r = O.resample(gen())
        r.set_sample_rate(sr)
        y = r.render_ticks(np.full((1, 200), speed, dtype=np.float32))[0]
        assert np.array_equal(y, ref[idx]), speed
    h = O.resample(gen())
    h.set_sample_rate(sr)
    y = h.render_ticks(np.full((1, 300), 0.5, dtype=np.float32))[0]
    assert np.array_equal(y[1::2][:100], ref[2:102])            # consumer 2.0, 3.0, ...: knots
    mid = 0.5 * (ref[1:101] + ref[2:102])                       # consumer 1.5, 2.5, ...: between two knots
    assert np.max(np.abs(y[0::2][:100] - mid)) < 2e-3           # cubic vs linear midpoint of a 330 Hz sine
