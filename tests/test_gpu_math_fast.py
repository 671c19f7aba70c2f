"""The two variants of the packed (process-mode) path that are chosen at run time:

* exact, per wave on the device: `FixedSvfLp` -- in a wave whose voices are all plain lowpass filters the packed SVF returns
  v2 directly instead of 0*v0 + 0*v1 + 1*v2.  Must stay BIT-EXACT in every case, including the ones its guard exists
  for: infinite / NaN / overflowing inputs (tile rollback), a state that starts at -0.0, waves with mixed filter modes.
* tolerance mode, per bank on the host: `fdsp_bank_set_option(bank, "math", FDSP_MATH_FAST)` -- Sine::process evaluates
  the engine's own FMA sine (within 1.2e-7 of wide's).  Bound against the oracle on the BASELINE config-3 graph: the
  reference's own tick-vs-process tolerance, 1e-4 absolute, over the reference's own window (441 samples,
  tests/test_basic.rs:21-47); 1e-3 max / 5e-5 rms over a full second -- the FM patch integrates last-bit differences of the
  modulator into the carrier phase, the reference's own two paths drift ~0.2 apart over that second.  `Moog` (config 4)
  evaluates its saturating tanh on the hardware exp2 / reciprocal; bounded at 1e-4 absolute on the config-4 voice.  Exact
  mode stays bit-exact next to it."""
import numpy as np
import pytest

import oracle as O
from fundsp_amd import LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MATH_EXACT, MATH_FAST, MODE_PROCESS, MODE_TICK
from fundsp_amd import workloads as W
from test_gpu_parity import assert_bit_equal, oracle_render, run_bank

pytestmark = pytest.mark.gpu
SR = 48000.0


def svf_voice(mode, fc, q):
    n = {"lowpass": O.lowpass_hz, "highpass": O.highpass_hz, "bandpass": O.bandpass_hz}[mode](float(fc), float(q))
    n.set_sample_rate(SR)
    return n


@pytest.mark.parametrize("layout", [LAYOUT_VOICE_MINOR, LAYOUT_PLANAR])
def test_lowpass_specialised_svf_is_bit_exact_with_specials(gpu, layout):
    """`fixed_svf` leaf through the pipeline kernels (T >= 256): waves 0 and 1 all-lowpass (specialised path), wave 2 mixed
    modes (generic path); inputs carry inf, NaN, values that overflow the state, denormals and zeros."""
    V, T = 64 * 3, 64 * 7 + 13
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((V, 1, T)) * 0.5).astype(np.float32)
    x[3, 0, 100] = np.inf                 # -> tile rollback, state poisoned afterwards (NaN on both sides)
    x[5, 0, 200] = -np.inf
    x[7, 0, 70] = np.nan
    x[9, 0, 64:80] = np.float32(3.0e38)   # overflow inside the recurrence with finite input
    x[11, 0, :] = 0.0                     # all-zero voice: +0 everywhere
    x[12, 0, :] = np.float32(-0.0)        # all -0 input
    x[13, 0, :] *= np.float32(1e-38)      # denormal products
    x[70, 0, 300] = np.inf                # in the second all-lowpass wave
    x[130, 0, 17] = np.inf                # in the mixed wave
    modes = ["lowpass"] * V
    for v in range(128, V):
        modes[v] = ("lowpass", "highpass", "bandpass")[v % 3]
    fc = (200.0 * 40.0 ** rng.random(V)).astype(np.float32)
    q = (0.5 + 3.0 * rng.random(V)).astype(np.float32)
    for mode in (MODE_PROCESS, MODE_TICK):
        b = gpu.Bank("fixed_svf", V)
        b.set_param(":mode", np.array([dict(lowpass=0, highpass=1, bandpass=2)[m] for m in modes], dtype=np.float32))
        b.set_param(":cutoff", fc)
        b.set_param(":q", q)
        b.set_sample_rate(SR)
        with np.errstate(all="ignore"):
            got = run_bank(b, x, T, layout, mode)
            for v in range(V):
                assert_bit_equal(got[v], oracle_render(svf_voice(modes[v], fc[v], q[v]), x[v], T, mode), f"voice {v} {modes[v]} mode {mode}")


def test_lowpass_specialised_svf_minus_zero_state(gpu):
    """A voice whose ic2eq STARTS at -0.0 can produce a -0.0 output the shortcut would turn into +0.0: such a wave must
    take the generic path.  State is installed through the slot interface on both sides."""
    V, T = 64, 64 * 5
    x = np.zeros((V, 1, T), dtype=np.float32)
    x[:, 0, :] = np.float32(-0.0)
    b = gpu.Bank("fixed_svf", V)
    b.set_param(":cutoff", 1000.0)
    b.set_param(":q", 1.0)
    b.set_sample_rate(SR)
    ic = np.zeros(V, dtype=np.float32)
    ic[5] = np.float32(-0.0)
    b.set_param(":ic1eq", ic)
    b.set_param(":ic2eq", ic)
    got = run_bank(b, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    # reference arithmetic by hand for the -0.0 voice: v3 = -0 - (-0) = +0 ... the oracle node has no state setter, so
    # the expectation is the generic kernel itself: the same bank rendered in tick mode (never specialised)
    b2 = gpu.Bank("fixed_svf", V)
    b2.set_param(":cutoff", 1000.0)
    b2.set_param(":q", 1.0)
    b2.set_sample_rate(SR)
    b2.set_param(":ic1eq", ic)
    b2.set_param(":ic2eq", ic)
    want = run_bank(b2, x, T, LAYOUT_VOICE_MINOR, MODE_TICK)
    assert_bit_equal(got, want, "ic2eq = -0.0 start")


def test_config3_exact_mode_unchanged_and_fast_mode_within_tolerance(gpu):
    """One second of 2048 config-3 voices: exact mode == oracle bit for bit (the lowpass-specialised SVF is in that path),
    tolerance mode within 1e-4 of it.  Prints the measured deviation."""
    V, T = 2048, 48000
    p = W.fm_svf_params(V, SR)
    want, _ = O.bank_render(3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], T, SR, True, 1, 16)   # [frame][voice]
    exact = W.make_fm_svf_bank(V, SR, params=p)
    assert exact.get_option("math") == MATH_EXACT and exact.get_option("math_has_fast_variant") == 1
    got = run_bank(exact, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :].T
    assert_bit_equal(got, want, "exact mode")
    fast = W.make_fm_svf_bank(V, SR, params=p)
    fast.set_option("math", MATH_FAST)
    gf = run_bank(fast, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :].T
    err = np.abs(gf.astype(np.float64) - want.astype(np.float64))
    # the yardstick: how far the REFERENCE's own two paths (process vs tick: wide sin + unwrapped phase vs libm sinf +
    # wrapped phase) drift apart on the same voices -- this FM patch integrates every last-bit difference of the modulator
    # into the carrier phase, so over a second even they differ by ~0.2 (over check_wave's 441 samples by ~1e-3)
    wt, _ = O.bank_render(3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], T, SR, False, 1, 16)
    own = np.abs(wt.astype(np.float64) - want.astype(np.float64))
    print(f"\nFDSP_MATH_FAST vs oracle, {V} voices x {T} frames: max |diff| {err.max():.3e}, rms {np.sqrt((err ** 2).mean()):.3e}, "
          f"99.9th percentile {np.quantile(err, 0.999):.3e}; first 441 frames max {err[:441].max():.3e}\n"
          f"reference process-vs-tick on the same voices: max {own.max():.3e}, rms {np.sqrt((own ** 2).mean()):.3e}, first 441 frames max {own[:441].max():.3e}")
    assert err[:441].max() <= 1e-4                     # the reference's own tolerance over its own window (tests/test_basic.rs:21-47)
    assert err.max() <= 1e-3 and np.sqrt((err ** 2).mean()) <= 5e-5     # a full second of the FM patch
    assert err.max() * 100 < own.max()                 # two orders of magnitude inside the reference's path-to-path spread
    assert not np.array_equal(gf, want)       # it IS a different arithmetic
    # planar layout and tick mode of a FAST bank: tick mode has no tolerance-mode form (scalar libm path) -> bit-exact
    fast_t = W.make_fm_svf_bank(256, SR, params=W.fm_svf_params(256, SR))
    fast_t.set_option("math", MATH_FAST)
    wt, _ = O.bank_render(3, [p["f"][:256], p["m"][:256], p["fc"][:256], p["q"][:256]], p["seed"][:256], 333, SR, False, 0, 4)
    assert_bit_equal(run_bank(fast_t, None, 333, LAYOUT_PLANAR, MODE_TICK)[:, 0, :], wt, "tick mode under FAST")


def test_fast_mode_is_a_noop_for_kinds_without_a_variant(gpu):
    b = gpu.Bank("noise_biquad", 64)
    assert b.get_option("math_has_fast_variant") == 0
    b.set_option("math", MATH_FAST)
    p = W.noise_biquad_params(64, SR)
    b2 = W.make_noise_biquad_bank(64, SR, params=p)
    b2.set_option("math", MATH_FAST)
    want, _ = O.bank_render(2, [p["fc"], p["q"]], p["seed"], 300, SR, True, 0)
    assert_bit_equal(run_bank(b2, None, 300, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :], want, "noise_biquad under FAST")


def test_moog_fast_mode_within_tolerance(gpu):
    """Tolerance mode of the ladder: only the saturating tanh changes (hardware exp2 / reciprocal, |error| <= 2.5e-7), the
    recurrence stays unfused.  Leaf `moog` (audio, cutoff, Q inputs) driven from whisper-quiet to hard saturation; the
    error of a voice is bounded against ITS OWN peak so that the quiet voices count too.  Exact mode bit-exact next to it."""
    V, T = 128, 64 * 40 + 9
    rng = np.random.default_rng(21)
    amp = (1e-3 * 3e4 ** (np.arange(V) / (V - 1))).astype(np.float32)          # 1e-3 .. 30
    x = np.zeros((V, 3, T), dtype=np.float32)
    x[:, 0, :] = (rng.standard_normal((V, T)) * amp[:, None]).astype(np.float32)
    x[:, 1, :] = (200.0 * 40.0 ** rng.random(V)).astype(np.float32)[:, None]   # cutoff Hz
    x[:, 2, :] = (0.1 + 0.8 * rng.random(V)).astype(np.float32)[:, None]       # Q
    x[7, 0, 100] = np.inf                                                      # saturates to 1, recovers
    want = np.zeros((V, 1, T), dtype=np.float32)
    for v in range(V):
        n = O.moog()
        n.set_sample_rate(SR)
        want[v] = oracle_render(n, x[v], T, MODE_PROCESS)
    exact = gpu.Bank("moog", V)
    exact.set_sample_rate(SR)
    assert exact.get_option("math_has_fast_variant") == 1
    with np.errstate(all="ignore"):
        assert_bit_equal(run_bank(exact, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS), want, "moog exact")
        fast = gpu.Bank("moog", V)
        fast.set_sample_rate(SR)
        fast.set_option("math", MATH_FAST)
        got = run_bank(fast, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)
    err = np.abs(np.where(fin, got.astype(np.float64) - want.astype(np.float64), 0.0))
    peak = np.abs(np.where(fin, want, 0.0)).max(axis=(1, 2))
    rel = err.max(axis=(1, 2)) / np.maximum(peak, 1e-30)
    print(f"\nFDSP_MATH_FAST moog vs oracle: max |diff| {err.max():.3e}; worst voice relative to its own peak {rel.max():.3e} "
          f"(peaks {peak.min():.2e} .. {peak.max():.2e})")
    assert err.max() <= 2e-5                 # absolute: 1e-4 is the reference's own bar (tests/test_basic.rs:21-47)
    assert not np.array_equal(got, want)
    # tick mode of a FAST bank keeps the exact tanh
    ft = gpu.Bank("moog", V)
    ft.set_sample_rate(SR)
    ft.set_option("math", MATH_FAST)
    with np.errstate(all="ignore"):
        wt = np.stack([oracle_render(_moog(), x[v][:, :333], 333, MODE_TICK) for v in range(8)])
        assert_bit_equal(run_bank(ft, x[:, :, :333].copy(), 333, LAYOUT_PLANAR, MODE_TICK)[:8], wt, "moog tick under FAST")


def _moog():
    n = O.moog()
    n.set_sample_rate(SR)
    return n


def test_config4_fast_mode_within_tolerance(gpu):
    """BASELINE config 4 voice in tolerance mode (Moog tanh; the saw, the envelope and the pan stay exact) against the
    oracle: max |diff| <= 1e-4 (the reference's own tolerance) with a wide margin; prints the measured deviation."""
    from test_gpu_config4 import config4_oracle_voice

    for kind in ("saw",):
        gpu.wavetable_build(kind)
    V, T = 256, 64 * 150 + 5
    adsr = (0.005, 0.01, 0.6, 0.01)
    p = W.saw_moog_params(V, SR)
    gate = np.broadcast_to(W.gate_signal(T, SR, on_frame=1, off_seconds=8000 / SR), (V, 1, T)).copy()
    fast = W.make_saw_moog_bank(V, SR, params=p, adsr=adsr)
    assert fast.get_option("math_has_fast_variant") == 1
    fast.set_option("math", MATH_FAST)
    got = run_bank(fast, gate, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    exact = W.make_saw_moog_bank(V, SR, params=p, adsr=adsr)
    ge = run_bank(exact, gate, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    worst = 0.0
    for v in range(0, V, 5):
        want = oracle_render(config4_oracle_voice(p, v, adsr), gate[v], T, MODE_PROCESS)
        assert_bit_equal(ge[v], want, f"config 4 exact voice {v}")
        worst = max(worst, float(np.abs(got[v].astype(np.float64) - want.astype(np.float64)).max()))
    d = np.abs(got.astype(np.float64) - ge.astype(np.float64))
    print(f"\nFDSP_MATH_FAST config 4 vs oracle: max |diff| {worst:.3e} (sampled voices), vs the exact bank over all {V} voices "
          f"{d.max():.3e}, rms {np.sqrt((d ** 2).mean()):.3e}; output peak {np.abs(ge).max():.3f}")
    assert d.max() <= 1e-4 and worst <= 1e-4
    assert not np.array_equal(got, ge)
