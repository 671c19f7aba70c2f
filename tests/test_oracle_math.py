"""Pins the oracle's restated libm / wide / hashing substrate (oracle/o_math.h) -- CPU only.

The `libm` and `wide` crate sources are not under /root/reference, so bit-level parity for the transcendentals is
UNPINNED; these tests bound the restatement the way the algorithms' authors document them (musl: sinf/cosf < 0.501
ulp, tanf < 0.8 ulp, expf/expm1f < 1 ulp, tanhf < 2.5 ulp) -- a mistyped polynomial constant shows up as a multi-ulp
error -- and pin the integer paths exactly, including the [derived] check values of SURVEY.md section 8(a).
"""
import numpy as np
import pytest

import oracle as O


def max_ulp(fn, ref, xs):
    got = np.array([fn(float(x)) for x in xs], dtype=np.float32).astype(np.float64)
    r = ref(xs.astype(np.float64))
    ulp = np.spacing(np.abs(r).astype(np.float32)).astype(np.float64)
    return float(np.max(np.abs(got - r) / ulp))


def test_libm_restatement_accuracy():
    L = O.lib()
    xs = np.linspace(-7.2, 7.2, 60001).astype(np.float32)          # covers every explicit quadrant case up to 9pi/4
    assert max_ulp(L.o_math_sinf, np.sin, xs) < 0.51
    assert max_ulp(L.o_math_cosf, np.cos, xs) < 0.51
    big = np.linspace(-3000.0, 3000.0, 20001).astype(np.float32)   # __rem_pio2f medium path
    assert max_ulp(L.o_math_sinf, np.sin, big) < 0.51
    assert max_ulp(L.o_math_cosf, np.cos, big) < 0.51
    xt = np.linspace(-1.55, 1.55, 40001).astype(np.float32)
    assert max_ulp(L.o_math_tanf, np.tan, xt) < 0.81
    xh = np.linspace(-12.0, 12.0, 40001).astype(np.float32)
    assert max_ulp(L.o_math_tanhf, np.tanh, xh) < 2.5
    assert max_ulp(L.o_math_expf, np.exp, xh) < 1.0
    assert max_ulp(L.o_math_expm1f, np.expm1, xh) < 1.0


def test_libm_payne_hanek_range():
    """|x| >= 2^28*pi/2: musl's __rem_pio2_large (restated in o_math.h) keeps sinf/cosf/tanf inside their documented
    error bounds over every exponent up to FLT_MAX -- the reference's libm returns real values there, not NaN."""
    import math
    L = O.lib()
    rng = np.random.default_rng(5)
    bits = rng.integers(0x4DC90FDB, 0x7F800000, size=40000, dtype=np.uint32) | (rng.integers(0, 2, size=40000, dtype=np.uint32) << 31)
    xs = bits.astype(np.uint32).view(np.float32)
    for fn, ref, bound in ((L.o_math_sinf, math.sin, 0.51), (L.o_math_cosf, math.cos, 0.51), (L.o_math_tanf, math.tan, 0.81)):
        got = np.array([fn(float(x)) for x in xs], dtype=np.float32).astype(np.float64)
        r = np.array([ref(float(x)) for x in xs])       # glibc's double sin/cos/tan reduce huge arguments exactly
        ulp = np.spacing(np.abs(r).astype(np.float32)).astype(np.float64)
        assert float(np.max(np.abs(got - r) / ulp)) < bound
    assert L.o_math_sinf(1e30) == np.float32(math.sin(float(np.float32(1e30))))
    assert L.o_math_sinf(-1e30) == -L.o_math_sinf(1e30)


def test_libm_special_cases():
    L = O.lib()
    assert L.o_math_sinf(0.0) == 0.0 and np.signbit(np.float32(L.o_math_sinf(-0.0)))
    assert L.o_math_cosf(0.0) == 1.0
    assert L.o_math_sinf(1e-5) == np.float32(1e-5)   # |x| < 2^-12 returns x
    assert np.isnan(L.o_math_sinf(float("inf")))
    assert L.o_math_tanhf(20.0) == 1.0 and L.o_math_tanhf(-20.0) == -1.0
    assert L.o_math_expm1f(-100.0) == -1.0


def test_wide_sin_restatement():
    """f32x8::sin (vectorclass sincos_f): absolute error ~1 ulp of 1.0 over the range Sine::process feeds it
    (unwrapped phase of one 64-sample block: up to 2*pi*64*f/sr ~ 170 rad), and sign/quadrant symmetry."""
    L = O.lib()
    xs = np.linspace(-200.0, 200.0, 100001).astype(np.float32)
    got = np.array([L.o_math_wide_sinf(float(x)) for x in xs], dtype=np.float64)
    assert np.max(np.abs(got - np.sin(xs.astype(np.float64)))) < 2.0e-7
    for x in (0.3, 1.7, 3.0, 5.5, 100.25):
        assert L.o_math_wide_sinf(-x) == -L.o_math_wide_sinf(x)
    assert L.o_math_wide_sinf(0.0) == 0.0


def test_tick_and_process_sine_agree_like_the_reference_requires():
    """tests/test_basic.rs:21-47 (check_wave): 441 samples via Wave::render (process path) equal the per-sample
    path within 1e-4 -- this bounds |wide::sin - libm::sinf| and the unwrapped-phase drift together."""
    for f in (110.0, 220.0, 440.0, 880.0):  # the frequencies the reference exercises (test_basic.rs:171-187)
        g = O.sine_hz(f)
        w = O.wave_render(44100.0, 441 / 44100.0, g)
        g.reset()
        t = g.render_ticks(length=441)
        assert np.max(np.abs(w - t)) <= 1e-4
    # Property of the REFERENCE, documented here: at high frequencies the block path's unwrapped f32 phase
    # (up to ~27 cycles inside a 64-sample block) loses ~5 bits against the wrapped tick path, so the two reference
    # paths drift apart beyond 1e-4 within 441 samples.  This is why the engine mirrors each path separately.
    g = O.sine_hz(15000.0)
    w = O.wave_render(44100.0, 441 / 44100.0, g)
    g.reset()
    assert 1e-4 < np.max(np.abs(w - g.render_ticks(length=441))) < 2e-3


def test_integer_hashes_exact():
    L = O.lib()
    # SplitMix64 finaliser / degski hash: independent pure-python restatement (math.rs:569-599)
    M = (1 << 64) - 1

    def rnd1(x):
        x ^= 0x5555555555555555
        x = x * 0x9E3779B97F4A7C15 & M
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9 & M
        x = (x ^ (x >> 27)) * 0x94D049BB133111EB & M
        x ^= x >> 31
        return (x >> 11) * (1.0 / (1 << 53))

    def hash1(x):
        x ^= 0x5555555555555555
        x = x * 0x517CC1B727220A95 & M
        x = (x ^ (x >> 32)) * 0xD6E8FEB86659FD93 & M
        x = (x ^ (x >> 32)) * 0xD6E8FEB86659FD93 & M
        return x ^ (x >> 32)

    for x in (0, 1, 2, 12345, M, 0x5555555555555555):
        assert L.o_math_rnd1(x) == rnd1(x)
        assert L.o_math_hash1(x) == hash1(x)
    assert 0.0 <= L.o_math_rnd1(7) < 1.0
    # AttoHash::hash = (state.rotl(5) ^ data) * 0x517cc1b727220a95  (math.rs:649-658)
    assert L.o_math_atto(1, 2) == (((1 << 5) ^ 2) * 0x517CC1B727220A95) & M
    assert L.o_math_atto(1 << 63, 0) == (((1 << 63) << 5 | (1 << 63) >> 59) & M) * 0x517CC1B727220A95 & M


def test_ping_hashes_match_survey_derived_values():
    """SURVEY.md 8(a) [derived] check values, computed there by hand from math.rs:569-576,649-658 and the ping order
    of audionode.rs:156-161,1459-1461,1286-1288 -- an independent derivation of the same quantities."""
    L = O.lib()
    g = O.sine_hz(440.0)
    s = g.children[1]
    assert L.o_sine_hash(s.ptr) == 15420871073424133422 and abs(L.o_sine_phase(s.ptr) - 0.5081898) < 1e-7
    g = O.sine_hz(440.0) >> O.lowpass_hz(1000.0, 1.0)
    s = g.children[0].children[1]
    assert L.o_sine_hash(s.ptr) == 9345624126978454524 and abs(L.o_sine_phase(s.ptr) - 0.6899407) < 1e-7
    f, m = 440.0, 2.0
    g = (O.sine_hz(f) * f * m + f) >> O.sine() >> O.lowpass_hz(1000.0, 1.0)
    car = g.children[0].children[1]
    mod = g.children[0].children[0].children[0].children[0].children[0].children[1]
    assert L.o_sine_hash(mod.ptr) == 13918874322061213918 and abs(L.o_sine_phase(mod.ptr) - 0.45837957) < 1e-7
    assert L.o_sine_hash(car.ptr) == 1679009996693057617 and abs(L.o_sine_phase(car.ptr) - 0.18989392) < 1e-7


def test_noise_is_integer_exact_and_uniform():
    n = O.noise().seed(1)
    x = n.render_ticks(length=1 << 16)[0]
    assert x.min() >= -1.0 and x.max() <= 1.0
    assert abs(float(x.mean())) < 0.01 and abs(float(x.std()) - 1 / np.sqrt(3)) < 0.01
    # process() hashes the same counter sequence as tick() (noise.rs:197-218)
    a = O.noise().seed(99).render_blocks(length=1000)
    b = O.noise().seed(99).render_ticks(length=1000)
    assert np.array_equal(a, b)


def test_powf_restatement_accuracy():
    """libm 0.2 powf (FreeBSD e_powf.c) restated in o_math.h: documented error < 1 ulp; special cases of pow()."""
    rng = np.random.default_rng(17)
    L = O.lib()
    worst = 0.0
    xs = np.concatenate([rng.uniform(1e-4, 0.9999, 20000), rng.uniform(0.5, 3.0, 5000), np.exp(rng.uniform(-30, 30, 5000))]).astype(np.float32)
    ys = np.concatenate([np.floor(rng.uniform(1, 500, 20000)), rng.uniform(-20, 20, 5000), rng.uniform(-3, 3, 5000)]).astype(np.float32)
    for x, y in zip(xs, ys):
        got = L.o_math_powf(float(x), float(y))
        want = np.float64(x) ** np.float64(y)
        if want == 0.0 or not np.isfinite(want) or want < 1e-37 or want > 1e38:
            continue
        ulp = np.spacing(np.float32(want))
        worst = max(worst, abs(np.float64(got) - want) / ulp)
    assert worst < 1.0, worst
    assert L.o_math_powf(2.0, 0.0) == 1.0 and L.o_math_powf(float("nan"), 0.0) == 1.0 and L.o_math_powf(1.0, float("nan")) == 1.0
    assert L.o_math_powf(0.5, 1.0) == 0.5 and L.o_math_powf(3.0, 2.0) == 9.0 and L.o_math_powf(4.0, 0.5) == 2.0
    assert L.o_math_powf(-2.0, 3.0) == -8.0 and L.o_math_powf(-2.0, 2.0) == 4.0 and np.isnan(L.o_math_powf(-2.0, 0.5))
    assert L.o_math_powf(0.5, float("inf")) == 0.0 and L.o_math_powf(2.0, float("inf")) == float("inf")
    assert L.o_math_powf(10.0, 50.0) == float("inf") and L.o_math_powf(10.0, -50.0) == 0.0
    assert L.o_math_powf(0.5, 140.0) == np.float32(2.0 ** -140)        # subnormal result through scalbnf


def test_exhaustive_ulp_sweep_program_and_its_committed_result():
    """tests/host/ulp_sweep.c bounds EVERY restated libm / wide function over all 2^32 f32 inputs against double precision
    (VERDICT r02 Next 1b: the sampled ranges above left tanf beyond +-1.55, expf / tanhf beyond +-12 and both atan forms
    unbounded).  The full pass takes ~6 minutes on 8 cores and is committed as profiles/r03_math_ulp_exhaustive.txt; the
    suite re-runs the same program on every 1021st bit pattern (all binades, ~1 s) and checks the committed full pass."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "host", "_build", "ulp_sweep")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-pthread", "-I", os.path.join(root, "oracle"),
                           "-o", exe, os.path.join(root, "tests", "host", "ulp_sweep.c"), "-lm"])
    r = subprocess.run([exe, "4", "1021"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "all 14 functions inside their bounds" in r.stdout
    full = open(os.path.join(root, "profiles", "r03_math_ulp_exhaustive.txt")).read()
    assert "ALL 2^32 f32 bit patterns (stride 1)" in full and "all 14 functions inside their bounds" in full
    for name in ("sinf", "cosf", "tanf", "expf", "expm1f", "tanhf", "atanf", "powf(x,x)", "wide_sin", "wide_atan"):
        line = next(l for l in full.splitlines() if l.startswith(name + " "))
        assert line.endswith(": OK") and "class mismatches 0" in line, line
