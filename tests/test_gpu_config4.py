"""GPU parity for the BASELINE config-4 path: WaveSynth (saw), adsr_live (EnvelopeIn), Panner, the 3-input Moog,
and the fused voice `((dc(f) >> saw() | dc(fc) | dc(q)) >> moog()) * adsr_live(..) >> pan(p)`.  Bit-exact vs oracle.
The wavetables are the ENGINE'S OWN (fdsp_wavetable_build, the path bench.py and users run): its make_wave restatement
is bit-identical to the oracle's, so nothing is uploaded from the oracle here."""
import numpy as np
import pytest

import oracle as O
from mix_order import mix_order_reference
from fundsp_amd import LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MODE_PROCESS, MODE_TICK
from fundsp_amd import workloads as W
from test_gpu_parity import assert_bit_equal, noise_input, oracle_render, run_bank

pytestmark = pytest.mark.gpu
SR = 48000.0
MODES = [MODE_PROCESS, MODE_TICK]


@pytest.fixture(scope="module")
def tables(gpu):
    for kind in ("saw", "square", "triangle"):
        gpu.wavetable_build(kind)     # engine-built tables: no upload from the oracle
    return True


def test_builtin_table_generator_matches_oracle_tables(gpu):
    """Engine-side Wavetable::new / make_wave vs the oracle's restatement (oracle/o_wavetable.c): same table layout
    (40 tables, 41 024 floats for saw: SURVEY.md section 7) and IDENTICAL bits, as installed on the device."""
    for kind in ("saw", "square", "triangle", "organ", "soft_saw", "hammond"):   # wavetable.rs:493-623
        gpu.wavetable_build(kind)
        p, waves = gpu.wavetable_get(kind)
        op, owaves = O.make_wavetable_arrays(kind)
        assert len(p) == len(op) == 40 and np.array_equal(p, op)
        assert [len(w) for w in waves] == [len(w) for w in owaves]
        assert sum(len(w) for w in waves) == 41024
        for a, b in zip(waves, owaves):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("kind", ["saw", "square", "triangle"])
@pytest.mark.parametrize("mode", MODES)
def test_wavesynth_leaf(gpu, tables, kind, mode):
    V, T = 80, 64 * 3 + 21
    rng = np.random.default_rng(5)
    x = np.zeros((V, 1, T), dtype=np.float32)
    f = (20.0 * 900.0 ** rng.random(V)).astype(np.float32)
    x[:, 0, :] = f[:, None]
    x[1] = -x[1]                                             # negative frequency (abs() for the table choice)
    x[2, 0, :] = np.linspace(30.0, 12000.0, T)               # sweep across table boundaries (binary search path)
    x[3, 0, :] = 440.0 + 400.0 * np.sin(np.arange(T) / 7.0)  # vibrato: lane-0 table choice differs from per-sample
    x[4] = 25000.0                                           # above the last table pitch
    x[5] = 5.0                                               # below the first table pitch
    b = gpu.Bank(kind, V)
    b.set_sample_rate(SR)
    b.set_seed(np.arange(V, dtype=np.uint64) + 7)
    got = run_bank(b, x, T, LAYOUT_VOICE_MINOR if mode == MODE_PROCESS else LAYOUT_PLANAR, mode)
    for v in range(V):
        n = O.wavesynth(kind)
        n.set_sample_rate(SR)
        n.set_seed(v + 7)
        assert_bit_equal(got[v], oracle_render(n, x[v], T, mode), f"{kind} voice {v}")


@pytest.mark.parametrize("layout", [LAYOUT_VOICE_MINOR, LAYOUT_PLANAR])
@pytest.mark.parametrize("mode", MODES)
def test_adsr_live_leaf(gpu, layout, mode):
    V, T = 70, 64 * 40 + 9
    rng = np.random.default_rng(8)
    gate = np.zeros((V, 1, T), dtype=np.float32)
    for v in range(V):
        on, off = int(rng.integers(0, 50)), int(rng.integers(300, 2000))
        gate[v, 0, on + 1:off] = 1.0
        if v % 3 == 0:
            gate[v, 0, off + 200:off + 400] = 0.5       # re-trigger during the release
    a = (0.001 + 0.01 * rng.random(V)).astype(np.float32)
    d = (0.002 + 0.02 * rng.random(V)).astype(np.float32)
    s = (0.2 + 0.7 * rng.random(V)).astype(np.float32)
    r = (0.002 + 0.02 * rng.random(V)).astype(np.float32)
    b = gpu.Bank("adsr_live", V)
    for name, val in (("attack", a), ("decay", d), ("sustain", s), ("release", r)):
        b.set_param(f":{name}", val)
    b.set_sample_rate(SR)
    b.set_seed(np.arange(V, dtype=np.uint64) * 3 + 1)
    got = run_bank(b, gate, T, layout, mode)
    for v in range(0, V, 3):
        n = O.adsr_live(float(a[v]), float(d[v]), float(s[v]), float(r[v]))
        n.set_sample_rate(SR)
        n.set_seed(v * 3 + 1)
        assert_bit_equal(got[v], oracle_render(n, gate[v], T, mode), f"adsr_live voice {v}")
    assert got.max() > 0.5  # the envelopes actually opened


def test_pan_leaf(gpu):
    V, T = 64, 100
    pans = np.linspace(-1.3, 1.3, V).astype(np.float32)  # beyond +-1: clamp11
    b = gpu.Bank("pan", V)
    b.set_param(":pan", pans)
    x = noise_input(V, 1, T, seed=4)
    got = run_bank(b, x, T, LAYOUT_PLANAR, MODE_PROCESS)
    for v in (0, 1, 31, 32, 63):
        assert_bit_equal(got[v], O.pan(float(pans[v])).render_blocks(x[v]), f"pan voice {v}")


def config4_oracle_voice(p, v, adsr):
    g = (((O.dc(float(p["f"][v])) >> O.saw()) | O.dc(float(p["fc"][v])) | O.dc(float(p["q"][v]))) >> O.moog()) \
        * O.adsr_live(*adsr) >> O.pan(float(p["pan"][v]))
    g.set_sample_rate(SR)
    g.set_seed(int(p["seed"][v]))
    return g


@pytest.mark.parametrize("layout", [LAYOUT_VOICE_MINOR, LAYOUT_PLANAR])
@pytest.mark.parametrize("mode", MODES)
def test_config4_fused_voice(gpu, tables, layout, mode):
    V, T = 130, 64 * 30 + 7
    adsr = (0.005, 0.01, 0.6, 0.01)
    p = W.saw_moog_params(V, SR)
    b = W.make_saw_moog_bank(V, SR, params=p, adsr=adsr)
    gate = np.broadcast_to(W.gate_signal(T, SR, on_frame=1, off_seconds=1200 / SR), (V, 1, T)).copy()
    got = run_bank(b, gate, T, layout, mode)
    assert got.shape == (V, 2, T)
    for v in range(0, V, 9):
        want = oracle_render(config4_oracle_voice(p, v, adsr), gate[v], T, mode)
        assert_bit_equal(got[v], want, f"config 4 voice {v}")
    assert np.abs(got).max() > 0.05


def test_sum_voices(gpu):
    import torch

    C, T, V = 2, 33, 700
    rng = np.random.default_rng(3)
    x = (rng.random((C, T, V), dtype=np.float32) - 0.5).astype(np.float32)
    got = gpu.sum_voices(torch.from_numpy(x).cuda()).cpu().numpy()
    assert_bit_equal(got, mix_order_reference(x), "sum_voices")   # the mix-down's fixed order (include/fundsp_hip.h)


@pytest.mark.parametrize("kind", ["saw", "triangle"])
def test_wavesynth_pipeline_kernel_table_crossings(gpu, tables, kind):
    """WaveSynth through the PIPELINE kernel (long launch: loader wave, packed two-frame interpolation with the taps
    gathered one pair ahead): voices across the whole pitch range, voices that cross table boundaries in both directions
    inside one launch (the gather-ahead's prediction must miss and re-gather), voices exactly on table pitches, negative
    frequencies -- bit-exact against the oracle."""
    V, T = 96, 64 * 12 + 5
    rng = np.random.default_rng(11)
    x = np.zeros((V, 1, T), dtype=np.float32)
    f = (40.0 * 60.0 ** rng.random(V)).astype(np.float32)          # 40 .. 2400 Hz
    x[:, 0, :] = f[:, None]
    x[0, 0, :] = np.linspace(60.0, 200.0, T)                       # crosses table boundaries upwards
    x[1, 0, :] = np.linspace(300.0, 50.0, T)                       # ... and downwards
    x[2, 0, :] = 100.0 + 30.0 * np.sin(np.arange(T) / 11.0)        # wobbles across a boundary
    for k, pitch in enumerate(20.0 * 2.0 ** (np.arange(7, 13) / 4.0)):   # exactly on the pitches 67 .. 160 Hz
        x[3 + k] = np.float32(pitch)
    x[10] = -x[0]                                                  # negative frequencies, same crossings
    b = gpu.Bank(kind, V)
    b.set_sample_rate(SR)
    b.set_seed(np.arange(V, dtype=np.uint64) + 3)
    got = run_bank(b, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    assert b.get_option("last_kernel") == 2, "the launch must take the pipeline kernel"
    for v in list(range(12)) + list(range(12, V, 7)):
        n = O.wavesynth(kind)
        n.set_sample_rate(SR)
        n.set_seed(v + 3)
        assert_bit_equal(got[v], oracle_render(n, x[v], T, MODE_PROCESS), f"{kind} voice {v}")
