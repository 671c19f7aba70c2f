"""GPU parity, second batch of leaf rows (a9, a13, a14): waveshapers (both Shape::shape and the f32x8
Shape::simd path of Shaper::process), Ramp / PolyBLEP oscillators, Rossler / Lorenz, and the biquads with nonlinear
feedback / state shaping.  Bit-exact vs the oracle."""
import numpy as np
import pytest

import oracle as O
from fundsp_amd import LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MODE_PROCESS, MODE_TICK
from test_gpu_parity import assert_bit_equal, noise_input, oracle_render, run_bank

pytestmark = pytest.mark.gpu
SR = 48000.0
MODES = [MODE_PROCESS, MODE_TICK]
SHAPE_CASES = [("clip", 2.0, 0.0), ("clip_to", -0.3, 0.6), ("tanh", 1.5, 0.0), ("atan", 0.8, 0.0), ("softsign", 3.0, 0.0),
               ("crush", 8.0, 0.0), ("soft_crush", 6.0, 0.0), ("adaptive_tanh", 1.0, 0.05)]


def set_shape(bank, prefix, kind, p0, p1):
    bank.set_param(f"{prefix}:shape", float(O.SHAPES[kind]))
    bank.set_param(f"{prefix}:shape_p0", p0)
    bank.set_param(f"{prefix}:shape_p1", p1)
    if kind == "adaptive_tanh":
        bank.set_param(f"{prefix}:shape_smoothing", float(np.float32(O.lib().o_adaptive_smoothing(p1, SR))))


@pytest.mark.parametrize("case", SHAPE_CASES, ids=[c[0] for c in SHAPE_CASES])
@pytest.mark.parametrize("mode", MODES)
def test_shaper(gpu, case, mode):
    kind, p0, p1 = case
    V, T = 64, 64 * 2 + 13
    x = noise_input(V, 1, T, seed=31) * 2.0
    x[0, 0, :16] = np.array([0.5, -0.5, 1.5, -1.5, 2.5, 0.0, -0.0, 1.0, 0.0625, 0.1875, -0.0625, 0.3125, 7.0, -7.0, 1e-8, 3.0],
                            dtype=np.float32) / np.float32(p0 if kind in ("crush", "soft_crush") else 1.0)  # exact ties
    b = gpu.Bank("shape", V)
    set_shape(b, "", kind, p0, p1)
    b.set_sample_rate(SR)
    b.reset() if kind == "adaptive_tanh" else None
    got = run_bank(b, x, T, LAYOUT_VOICE_MINOR, mode)
    for v in (0, 1, 17, 63):
        n = O.shape(kind, p0, p1)
        n.set_sample_rate(SR)
        n.reset() if kind == "adaptive_tanh" else None
        assert_bit_equal(got[v], oracle_render(n, x[v], T, mode), f"shape {kind} voice {v}")


@pytest.mark.parametrize("case", [c for c in SHAPE_CASES if c[0] not in ("tanh", "adaptive_tanh")], ids=lambda c: "adaptive_" + c[0])
@pytest.mark.parametrize("mode", MODES)
def test_shaper_adaptive_any_inner(gpu, case, mode):
    """Shaper<Adaptive<S>> for every plain shape S (shape.rs:162-201): level follower, then S::shape on input / sqrt(level);
    Adaptive has no simd form of its own, so the process path applies S's SCALAR shape lane by lane (Crush rounds half
    away from zero there, not half to even as Shaper<Crush>::process does)."""
    inner, p0, p1 = case
    timescale = 0.03
    V, T = 64, 64 * 3 + 5
    x = noise_input(V, 1, T, seed=32) * 2.0
    x[2] *= np.float32(1e-3)
    x[3, 0, 40:] = 0.0                     # level decays towards the 1e-6 floor
    b = gpu.Bank("shape", V)
    b.set_param(":shape", float(O.SHAPES["adaptive_" + inner]))
    b.set_param(":shape_p0", p0)
    b.set_param(":shape_p1", p1)
    b.set_param(":shape_smoothing", float(np.float32(O.lib().o_adaptive_smoothing(timescale, SR))))
    b.set_sample_rate(SR)
    b.reset()
    got = run_bank(b, x, T, LAYOUT_VOICE_MINOR, mode)
    for v in (0, 2, 3, 63):
        n = O.shape_adaptive(inner, p0, p1, timescale)
        n.set_sample_rate(SR)
        n.reset()
        assert_bit_equal(got[v], oracle_render(n, x[v], T, mode), f"adaptive<{inner}> voice {v}")


@pytest.mark.parametrize("kind", ["ramp", "poly_saw", "poly_square", "poly_pulse"])
def test_phase_oscillators(gpu, kind):
    V, T = 64, 300
    rng = np.random.default_rng(2)
    b = gpu.Bank(kind, V)
    ni = b.inputs()
    x = np.zeros((V, ni, T), dtype=np.float32)
    x[:, 0, :] = (20.0 * 500.0 ** rng.random(V)).astype(np.float32)[:, None]
    x[3, 0, :] = -220.0
    x[4, 0, :] = np.linspace(100, 8000, T)
    if ni == 2:
        x[:, 1, :] = (0.1 + 0.8 * rng.random(V)).astype(np.float32)[:, None]
    b.set_sample_rate(SR)
    b.set_seed(np.arange(V, dtype=np.uint64) + 11)
    got = run_bank(b, x, T, LAYOUT_PLANAR, MODE_PROCESS)
    for v in range(0, V, 5):
        n = O.Node(O.lib().o_phase_osc(O.OSCS[kind]))
        n.set_sample_rate(SR)
        n.set_seed(v + 11)
        assert_bit_equal(got[v], n.render_blocks(x[v]), f"{kind} voice {v}")


@pytest.mark.parametrize("lorenz", [False, True])
def test_chaotic_oscillators(gpu, lorenz):
    V, T = 64, 2000
    b = gpu.Bank("lorenz" if lorenz else "rossler", V)
    b.set_sample_rate(SR)
    b.set_seed(np.arange(V, dtype=np.uint64) * 5 + 3)
    x = np.full((V, 1, T), 440.0, dtype=np.float32)
    x[:, 0, :] *= np.linspace(0.25, 4.0, V, dtype=np.float32)[:, None]
    got = run_bank(b, x, T, LAYOUT_VOICE_MINOR, MODE_TICK)
    for v in (0, 9, 33, 63):  # chaotic: any arithmetic difference would explode within 2000 samples
        n = O.lorenz() if lorenz else O.rossler()
        n.set_sample_rate(SR)
        n.set_seed(v * 5 + 3)
        assert_bit_equal(got[v], n.render_ticks(x[v]), f"chaos voice {v}")


@pytest.mark.parametrize("dirty", [False, True])
def test_fixed_nonlinear_biquads(gpu, dirty):
    """dbell_hz / dhighpass_hz / dresonator_hz / dlowpass_hz / fbell_hz / flowpass_hz / fresonator_hz / fhighpass_hz with the
    shapes the reference tests them with (tests/test_basic.rs:219-234) and more."""
    modes = ["resonator", "lowpass", "highpass", "bell"]
    V, T = 64, 400
    rng = np.random.default_rng(6)
    mode_i = np.arange(V) % 4
    shape_i = (np.arange(V) // 4) % len(SHAPE_CASES)
    center = (200.0 * 20.0 ** rng.random(V)).astype(np.float32)
    q = (0.7 + 10 * rng.random(V)).astype(np.float32)
    gain = (0.5 + 2 * rng.random(V)).astype(np.float32)
    b = gpu.Bank("dbiquad_hz" if dirty else "fbiquad_hz", V)
    b.set_param(":mode", np.array([O.BQ_KINDS[modes[m]] for m in mode_i], dtype=np.float32))
    b.set_param(":center", center); b.set_param(":q", q); b.set_param(":gain", gain)
    for which in ([0, 1] if dirty else [0]):
        b.set_param(f"{which}:shape", np.array([O.SHAPES[SHAPE_CASES[s][0]] for s in shape_i], dtype=np.float32))
        b.set_param(f"{which}:shape_p0", np.array([SHAPE_CASES[s][1] for s in shape_i], dtype=np.float32))
        b.set_param(f"{which}:shape_p1", np.array([SHAPE_CASES[s][2] for s in shape_i], dtype=np.float32))
        b.set_param(f"{which}:shape_smoothing", np.array(
            [np.float32(O.lib().o_adaptive_smoothing(SHAPE_CASES[s][2], SR)) if SHAPE_CASES[s][0] == "adaptive_tanh" else 0.0
             for s in shape_i], dtype=np.float32))
    b.set_sample_rate(SR)
    x = noise_input(V, 1, T, seed=41)
    got = run_bank(b, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    for v in range(V):
        kind, p0, p1 = SHAPE_CASES[shape_i[v]]
        n = O.nlbiquad(dirty, 1, modes[mode_i[v]], kind, p0, p1, float(center[v]), float(q[v]), float(gain[v]))
        n.set_sample_rate(SR)
        assert_bit_equal(got[v], n.render_blocks(x[v]), f"nlbiquad dirty={dirty} voice {v} {modes[mode_i[v]]} {kind}")


@pytest.mark.parametrize("kind,dirty,ni", [("fbiquad3", False, 3), ("fbiquad4", False, 4), ("dbiquad3", True, 3), ("dbiquad4", True, 4)])
def test_modulated_nonlinear_biquads(gpu, kind, dirty, ni):
    V, T = 64, 256
    rng = np.random.default_rng(9)
    x = noise_input(V, ni, T, seed=43)
    hold = 32
    x[:, 1, :] = np.repeat(300 + 3000 * rng.random((V, T // hold)), hold, axis=1)
    x[:, 2, :] = np.repeat(0.7 + 5 * rng.random((V, T // hold)), hold, axis=1)
    if ni == 4:
        x[:, 3, :] = np.repeat(0.5 + 2 * rng.random((V, T // hold)), hold, axis=1)
    mode = "bell" if ni == 4 else "lowpass"
    b = gpu.Bank(kind, V)
    b.set_param(":mode", float(O.BQ_KINDS[mode]))
    for which in ([0, 1] if dirty else [0]):
        set_shape(b, str(which), "softsign", 0.5, 0.0)
    b.set_sample_rate(SR)
    got = run_bank(b, x, T, LAYOUT_PLANAR, MODE_PROCESS)
    for v in range(0, V, 7):
        n = O.nlbiquad(dirty, ni, mode, "softsign", 0.5, 0.0)
        n.set_sample_rate(SR)
        assert_bit_equal(got[v], n.render_blocks(x[v]), f"{kind} voice {v}")


@pytest.mark.parametrize("kind,make", [
    ("lowpole_hz", lambda p: O.lowpole_hz(p)), ("highpole_hz", lambda p: O.highpole_hz(p)),
    ("dcblock_hz", lambda p: O.dcblock_hz(p)), ("allpole_delay", lambda p: O.allpole_delay(p)),
])
def test_one_pole_filters(gpu, kind, make):
    V, T = 64, 500
    rng = np.random.default_rng(61)
    p = (20.0 * 500.0 ** rng.random(V)).astype(np.float32) if kind != "allpole_delay" else (0.05 + 2 * rng.random(V)).astype(np.float32)
    b = gpu.Bank(kind, V)
    b.set_param(":delay" if kind == "allpole_delay" else ":cutoff", p)
    b.set_sample_rate(SR)
    x = noise_input(V, 1, T, seed=62)
    got = run_bank(b, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    for v in range(0, V, 5):
        n = make(float(p[v]))
        n.set_sample_rate(SR)
        assert_bit_equal(got[v], n.render_blocks(x[v]), f"{kind} voice {v}")


@pytest.mark.parametrize("kind,make", [("lowpole", O.lowpole), ("highpole", O.highpole), ("allpole", O.allpole),
                                       ("pinkpass", O.pinkpass), ("morph", O.morph)])
def test_one_pole_modulated_pink_morph(gpu, kind, make):
    V, T = 64, 320
    rng = np.random.default_rng(63)
    b = gpu.Bank(kind, V)
    ni = b.inputs()
    x = noise_input(V, ni, T, seed=64)
    if ni >= 2:
        x[:, 1, :] = np.repeat((0.2 + 3 * rng.random((V, T // 32))) if kind == "allpole" else (100 + 5000 * rng.random((V, T // 32))), 32, axis=1)
    if ni == 4:
        x[:, 2, :] = np.repeat(0.5 + 4 * rng.random((V, T // 32)), 32, axis=1)
        x[:, 3, :] = np.repeat(-1 + 2 * rng.random((V, T // 32)), 32, axis=1)
    b.set_sample_rate(SR)
    got = run_bank(b, x, T, LAYOUT_PLANAR, MODE_PROCESS)
    for v in range(0, V, 9):
        n = make()
        n.set_sample_rate(SR)
        assert_bit_equal(got[v], n.render_blocks(x[v]), f"{kind} voice {v}")
