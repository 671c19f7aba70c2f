"""Known-answer tests of the oracle's filters against the reference's closed-form frequency responses -- CPU only.

Procedure and tolerances are the reference's own (tests/test_flow.rs:18-80 `test_response`: warm up 16 384 zeros,
unit impulse, 32 768 samples, real FFT, compare bins 10 Hz..22 kHz with |x-y| <= 2e-4 * max(1,|x|,|y|);
tests/test_flow.rs:251-283 `test_allpass`: |H| = 1 within 1e-5).  The analytic responses are restated from
svf.rs:315-322,364-371,413-420,462-469,511-519,561-568,613-622,667-679,724-741, biquad.rs:119-128,
fir.rs:72-88 and delay.rs:53-64,126-138.
"""
import numpy as np
import pytest

import oracle as O

SR = O.DEFAULT_SR
LENGTH = 0x8000


def measured_response(node):
    node.reset()
    node.set_sample_rate(SR)
    x = np.zeros((1, LENGTH // 2 + LENGTH), dtype=np.float32)
    x[0, LENGTH // 2] = 1.0
    y = node.render_ticks(x)[0, LENGTH // 2:]
    return np.fft.rfft(y.astype(np.float64))


def check(node, analytic, tol=2.0e-4):
    spec = measured_response(node)
    f = 10.0
    worst = 0.0
    while f <= 22000.0:
        i = int(round(f * LENGTH / SR))
        if i >= len(spec):
            break
        fi = i / LENGTH * SR
        rep, mea = analytic(fi), spec[i]
        err = abs(rep - mea) / max(1.0, abs(rep), abs(mea))
        worst = max(worst, err)
        f += 10.0 if f < 1000.0 else 100.0
    assert worst <= tol, worst


def svf_response(mode, cutoff, q, gain=1.0):
    g = np.tan(np.pi * cutoff / SR)
    k = 1.0 / q
    a = np.sqrt(gain)
    sa = np.sqrt(a)

    def H(f):
        z = np.exp(1j * f * 2 * np.pi / SR)
        den = (z - 1) ** 2 + g * g * (1 + z) ** 2 + g * k * (z * z - 1)
        if mode == "lowpass":
            return g * g * (1 + z) ** 2 / den
        if mode == "highpass":
            return (z - 1) ** 2 / den
        if mode == "bandpass":
            return g * (z * z - 1) / den
        if mode == "notch":
            return ((z - 1) ** 2 + g * g * (1 + z) ** 2) / den
        if mode == "peak":
            return -((1 + g + (g - 1) * z) * (-1 + g + z + g * z)) / den
        if mode == "allpass":
            return ((z - 1) ** 2 + g * g * (1 + z) ** 2 + g * (k - k * z * z)) / den
        if mode == "bell":
            return (g * k * (z * z - 1) + a * (g * (1 + z) * ((a * a - 1) * k / a * (z - 1)) + ((z - 1) ** 2 + g * g * (1 + z) ** 2))) / \
                   (g * k * (z * z - 1) + a * ((z - 1) ** 2 + g * g * (z + 1) ** 2))
        if mode == "lowshelf":
            return (a * (z - 1) ** 2 + g * g * a * a * (z + 1) ** 2 + sa * g * a * k * (z * z - 1)) / \
                   (a * (z - 1) ** 2 + g * g * (1 + z) ** 2 + sa * g * k * (z * z - 1))
        if mode == "highshelf":
            return (sa * g * (1 + z) * (-(a - 1) * a * k * (z - 1) + sa * g * (1 - a * a) * (1 + z))
                    + a * a * ((z - 1) ** 2 + a * g * g * (1 + z) ** 2 + sa * g * k * (z * z - 1))) / \
                   ((z - 1) ** 2 + a * g * g * (1 + z) ** 2 + sa * g * k * (z * z - 1))
        raise KeyError(mode)

    return H


def biquad_response(c):
    a1, a2, b0, b1, b2 = [float(v) for v in c]

    def H(f):
        z1 = np.exp(-2j * np.pi * f / SR)
        return (b0 + b1 * z1 + b2 * z1 * z1) / (1 + a1 * z1 + a2 * z1 * z1)

    return H


# the FixedSvf cases of tests/test_flow.rs:85-94
@pytest.mark.parametrize("mode,args", [
    ("bell", (500.0, 1.0, 2.0)), ("lowshelf", (2000.0, 10.0, 5.0)), ("highshelf", (2000.0, 10.0, 5.0)),
    ("peak", (5000.0, 1.0)), ("allpass", (500.0, 5.0)), ("notch", (1000.0, 1.0)), ("lowpass", (50.0, 1.0)),
    ("highpass", (5000.0, 1.0)), ("bandpass", (100.0, 1.0)),
])
def test_fixed_svf_responses(mode, args):
    node = O._fsvf(mode, *args)
    check(node, svf_response(mode, *args))


def test_bell_times_half():  # test_flow.rs:85 `bell_hz(500, 1, 2) * 0.5`
    H = svf_response("bell", 500.0, 1.0, 2.0)
    check(O.bell_hz(500.0, 1.0, 2.0) * 0.5, lambda f: 0.5 * H(f))


def test_svf_with_constant_parameter_inputs_equals_fixed():
    """Svf<LowpassMode> driven by constant (cutoff, q) inputs has the FixedSvf response (svf.rs:299-313)."""
    node = (O.pass_() | O.dc(1200.0, 0.8)) >> O.svf("lowpass")
    check(node, svf_response("lowpass", 1200.0, 0.8))


@pytest.mark.parametrize("coefs", [(0.0, 0.17149, 0.29287, 0.58574, 0.29287),
                                   (0.033717, 0.171773, 1.059253, -0.035714, 0.181952)])
def test_raw_biquad(coefs):  # test_flow.rs:165-166
    check(O.biquad(*coefs), biquad_response(np.float32(coefs)))


def test_biquad_family():  # test_flow.rs:106-110: resonator_hz, butterpass_hz
    check(O.resonator_hz(300.0, 20.0), biquad_response(O.biquad_coefs("resonator", SR, 300.0, 20.0)))
    for f in (200.0, 1000.0):
        check(O.butterpass_hz(f), biquad_response(O.biquad_coefs("butter", SR, f)))
    # cookbook constructors: unity gain at DC (lowpass) / Nyquist (highpass), bell gain at centre
    lp = biquad_response(O.biquad_coefs("lowpass", SR, 1000.0, 0.7))
    hp = biquad_response(O.biquad_coefs("highpass", SR, 1000.0, 0.7))
    bell = biquad_response(O.biquad_coefs("bell", SR, 1000.0, 2.0, 4.0))
    assert abs(abs(lp(0.0)) - 1) < 1e-4 and abs(abs(hp(SR / 2)) - 1) < 1e-4 and abs(abs(bell(1000.0)) - 4.0) < 2e-3
    bt = biquad_response(O.biquad_coefs("butter", SR, 1000.0))
    assert abs(abs(bt(1000.0)) - np.sqrt(0.5)) < 1e-4   # -3 dB at the cutoff


def test_biquad_bank_lane_3():  # test_flow.rs:171-177
    bank = O.biquad_bank()
    O.set_biquad_bank(bank, 3, (0.05, 0.1, 0.3, 0.1, 0.15))
    bank.set_sample_rate(SR)
    x = np.zeros((8, LENGTH), dtype=np.float32)
    x[3, 0] = 1.0
    y = bank.render_ticks(x)
    H = biquad_response(np.float32((0.05, 0.1, 0.3, 0.1, 0.15)))
    spec = np.fft.rfft(y[3].astype(np.float64))
    for f in (10.0, 100.0, 1000.0, 5000.0, 20000.0):
        i = int(round(f * LENGTH / SR))
        assert abs(spec[i] - H(i / LENGTH * SR)) <= 2e-4 * max(1, abs(spec[i]))
    assert not np.any(y[[0, 1, 2, 4, 5, 6, 7]])  # other lanes have zero coefficients and zero input


@pytest.mark.parametrize("w", [(0.5, 0.5), (0.25, 0.5, 0.25), (0.4, 0.3, 0.2, 0.1)])
def test_fir(w):  # test_flow.rs:158-160; response fir.rs:72-88
    def H(f):
        z1 = np.exp(-2j * np.pi * f / SR)
        return sum(w[len(w) - 1 - i] * z1 ** i for i in range(len(w)))
    check(O.fir(*w), H)


def test_delays():  # test_flow.rs:98-100,111-112
    check(O.delay(0.0), lambda f: 1.0)
    n = round(0.0001 * SR)
    check(O.delay(0.0001), lambda f: np.exp(-2j * np.pi * n * f / SR))
    check(O.tick(), lambda f: np.exp(-2j * np.pi * f / SR))
    check(O.pass_() * 0.25 + O.tick() * 0.5, None) if False else None


@pytest.mark.parametrize("make", [lambda: O.pass_(), lambda: O.tick(), lambda: O.delay(0.0001), lambda: O.delay(0.001),
                                  lambda: O.allpass_hz(1000.0, 1.0), lambda: O.allpass_hz(2000.0, 2.0)])
def test_allpass_property(make):  # test_flow.rs:251-283
    node = make()
    node.set_sample_rate(SR)
    x = np.zeros((1, LENGTH), dtype=np.float32)
    x[0, 0] = 1.0
    y = node.render_blocks(x)[0]
    mag = np.abs(np.fft.rfft(y.astype(np.float64)))[1:]
    assert np.all(np.abs(mag - 1.0) <= 1e-5)


def test_one_pole_family_responses():
    """test_flow.rs:101-104,112-115: lowpole_hz, dcblock, highpole; closed forms from filter.rs:79-92 (Lowpole),
    :162-175 (DCBlock), :417-431 (Highpole), :338-349 (Allpole); allpole is allpass (test_flow.rs:256-257)."""
    for fc in (1000.0, 10000.0):
        c = float(np.float32(np.exp(-2 * np.pi * fc / SR)))
        check(O.lowpole_hz(fc), lambda f, c=c: (1 - c) / (1 - c * np.exp(-2j * np.pi * f / SR)), tol=3e-4)
    c = float(np.float32(np.exp(-2 * np.pi * 5000.0 / SR)))
    check(O.highpole_hz(5000.0), lambda f: c * (1 - np.exp(-2j * np.pi * f / SR)) / (1 - c * np.exp(-2j * np.pi * f / SR)), tol=3e-4)
    cd = float(np.float32(1.0 - 2 * np.pi / SR * 100.0))
    check(O.dcblock_hz(100.0), lambda f: (1 - np.exp(-2j * np.pi * f / SR)) / (1 - cd * np.exp(-2j * np.pi * f / SR)), tol=3e-4)
    for d in (0.5, 0.8):
        n = O.allpole_delay(d)
        x = np.zeros((1, LENGTH), dtype=np.float32)
        x[0, 0] = 1.0
        mag = np.abs(np.fft.rfft(n.render_blocks(x)[0].astype(np.float64)))[1:]
        assert np.all(np.abs(mag - 1.0) <= 1e-5)
