"""Bit-exact GPU parity of the nodes added after the first round-1 sweep: Rez (rez.rs), Follow / AFollow (follow.rs),
Mls (noise.rs) -- SURVEY.md 8(f) row 1 and 8(a) a16."""
import numpy as np
import pytest

import oracle as O
from fundsp_amd import LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MODE_PROCESS, MODE_TICK
from test_gpu_parity import assert_bit_equal, noise_input, oracle_render, run_bank

pytestmark = pytest.mark.gpu
SR = 48000.0
MODES = [MODE_PROCESS, MODE_TICK]


@pytest.mark.parametrize("bandpass", [0.0, 1.0])
@pytest.mark.parametrize("mode", MODES)
def test_rez_fixed(gpu, bandpass, mode):
    V, T = 96, 64 * 3 + 21
    rng = np.random.default_rng(61)
    fc = (100.0 + 6000.0 * rng.random(V)).astype(np.float32)
    q = (0.05 + 0.9 * rng.random(V)).astype(np.float32)
    b = gpu.Bank("rez_hz", V)
    b.set_param(":bandpass", bandpass)
    b.set_param(":cutoff", fc)
    b.set_param(":q", q)
    b.set_sample_rate(SR)
    x = noise_input(V, 1, T, seed=62)
    for layout in (LAYOUT_VOICE_MINOR, LAYOUT_PLANAR):
        b.reset()
        got = run_bank(b, x, T, layout, mode)
        for v in (0, 1, 31, 64, 95):
            n = O.bandrez_hz(float(fc[v]), float(q[v])) if bandpass else O.lowrez_hz(float(fc[v]), float(q[v]))
            n.set_sample_rate(SR)
            assert_bit_equal(got[v], oracle_render(n, x[v], T, mode), f"rez_hz voice {v}")


def test_rez_with_inputs(gpu):
    V, T = 64, 64 * 2 + 40
    x = noise_input(V, 3, T, seed=63)
    x[:, 1] = 200.0 + 3000.0 * np.abs(x[:, 1])
    x[:, 1, 50:90] = x[:, 1, 49:50]           # stretches of constant cutoff / q: coefficients must NOT be re-derived
    x[:, 2] = 0.1 + 0.8 * np.abs(x[:, 2])
    x[:, 2, 50:90] = x[:, 2, 49:50]
    b = gpu.Bank("rez", V)
    b.set_param(":bandpass", 1.0)
    b.set_sample_rate(SR)
    got = run_bank(b, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    for v in (0, 7, 63):
        n = O.bandrez()
        n.set_sample_rate(SR)
        assert_bit_equal(got[v], oracle_render(n, x[v], T, MODE_PROCESS), f"rez voice {v}")


@pytest.mark.parametrize("mode", MODES)
def test_follow_and_afollow(gpu, mode):
    V, T = 70, 64 * 5 + 9
    rng = np.random.default_rng(64)
    rt = (0.0005 + 0.05 * rng.random(V)).astype(np.float32)
    rel = (0.0005 + 0.05 * rng.random(V)).astype(np.float32)
    x = np.repeat(noise_input(V, 1, (T + 31) // 32, seed=65), 32, axis=2)[:, :, :T].copy()  # stepped control signal
    b = gpu.Bank("follow", V)
    b.set_param(":response_time", rt)
    b.set_sample_rate(SR)
    got = np.concatenate([run_bank(b, x, T, LAYOUT_VOICE_MINOR, mode), run_bank(b, x, T, LAYOUT_VOICE_MINOR, mode)], axis=-1)
    a = gpu.Bank("afollow", V)
    a.set_param(":attack_time", rt)
    a.set_param(":release_time", rel)
    a.set_sample_rate(SR)
    gota = run_bank(a, x, T, LAYOUT_PLANAR, mode)
    xx = np.concatenate([x, x], axis=-1)
    for v in (0, 3, 64, 69):
        n = O.follow(float(rt[v]))
        n.set_sample_rate(SR)
        assert_bit_equal(got[v], oracle_render(n, xx[v], 2 * T, mode), f"follow voice {v}")
        m = O.afollow(float(rt[v]), float(rel[v]))
        m.set_sample_rate(SR)
        assert_bit_equal(gota[v], oracle_render(m, x[v], T, mode), f"afollow voice {v}")


@pytest.mark.parametrize("bits", [5, 10, 29, 31])
def test_mls(gpu, bits):
    V, T = 64, 64 * 3 + 11
    b = gpu.Bank("mls", V)
    b.set_param(":bits", float(bits))
    b.set_sample_rate(SR)                                   # refreshes the feedback polynomial for the new width
    seeds = np.arange(V, dtype=np.uint64) * 977 + 5
    b.set_seed(seeds)
    got = run_bank(b, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    assert set(np.unique(got)) <= {-1.0, 1.0}
    for v in (0, 1, 63):
        n = O.mls_bits(bits)
        n.set_seed(int(seeds[v]))
        assert_bit_equal(got[v], oracle_render(n, None, T, MODE_PROCESS), f"mls({bits}) voice {v}")


# ---- Oversampler ---------------------------------------------------------------------------------
@pytest.mark.parametrize("layout", [LAYOUT_VOICE_MINOR, LAYOUT_PLANAR])
@pytest.mark.parametrize("mode", MODES)
def test_oversample_fm(gpu, layout, mode):
    """oversample(sine_hz(f) * f * m + f >> sine())  (README.md:1631): per-voice f, m; tick and process walks.
    T is even per block (the reference leaves the last sample of an odd block unwritten; covered below)."""
    from fundsp_amd import workloads as W

    V, T = 70, 64 * 3 + 22
    p = W.fm_svf_params(V, SR)
    b = gpu.Bank("oversample_fm", V)
    b.set_param("0.0.0.0.0.0:value[0]", p["f"])        # Constant inside sine_hz
    b.set_param("0.0.0.0:scalar", p["f"])              # * f
    b.set_param("0.0.0:scalar", p["m"])                # * m
    b.set_param("0.0:scalar", p["f"])                  # + f
    b.set_sample_rate(SR)
    seeds = np.arange(V, dtype=np.uint64) + 11
    b.set_seed(seeds)
    got = np.concatenate([run_bank(b, None, T, layout, mode), run_bank(b, None, T, layout, mode)], axis=-1)
    for v in (0, 5, 64, 69):
        f, m = float(p["f"][v]), float(p["m"][v])
        n = O.oversample(O.sine_hz(f) * f * m + f >> O.sine())
        n.set_sample_rate(SR)
        n.set_seed(int(seeds[v]))
        want = np.concatenate([oracle_render(n, None, T, mode), oracle_render(n, None, T, mode)], axis=-1)
        assert_bit_equal(got[v], want, f"oversample_fm voice {v}")


@pytest.mark.parametrize("T", [64 * 2 + 40, 64 + 5, 7])
@pytest.mark.parametrize("mode", MODES)
def test_oversample_shape(gpu, mode, T):
    """oversample(shape(Tanh(2))): 1 in, 1 out; odd block tails follow the reference's process (last sample unwritten -> 0)."""
    V = 64
    x = noise_input(V, 1, T, seed=71)
    b = gpu.Bank("oversample_shape", V)
    b.set_param("0:shape", float(O.SHAPES["tanh"]))
    b.set_param("0:shape_p0", 2.0)
    b.set_sample_rate(SR)
    got = run_bank(b, x, T, LAYOUT_VOICE_MINOR, mode)
    for v in (0, 33, 63):
        n = O.oversample(O.shape("tanh", 2.0))
        n.set_sample_rate(SR)
        assert_bit_equal(got[v], oracle_render(n, x[v], T, mode), f"oversample_shape voice {v} T={T}")


# ---- Dsf (oscillator.rs:103-208): needs libm powf, restated in fd_math.hpp / o_math.h ---------------------------------
@pytest.mark.parametrize("spacing", [1.0, 2.0])
@pytest.mark.parametrize("mode", MODES)
def test_dsf(gpu, spacing, mode):
    V, T = 96, 64 * 2 + 19
    rng = np.random.default_rng(81)
    rough = (0.05 + 0.9 * rng.random(V)).astype(np.float32)
    rough[:3] = [0.0, 1.0, 0.5]                                  # clamped to 0.0001 / 0.9999 by set_roughness
    x = np.zeros((V, 2, T), dtype=np.float32)
    x[:, 0] = (30.0 + 8000.0 * rng.random((V, 1)) ** 2) * (1.0 + 0.2 * rng.random((V, T)))   # wandering frequency
    x[5, 0, :] = 23000.0                                         # n = 0: pow(r, 1) special case
    x[6, 0, :] = 11025.0                                         # n = 1 or 2: pow(r, 2) special case
    x[:, 1] = np.clip(rough[:, None] + 0.3 * (rng.random((V, T)) - 0.5), -0.2, 1.2)
    b1 = gpu.Bank("dsf_r", V)
    b1.set_param(":harmonic_spacing", spacing)
    b1.set_param(":roughness", rough)
    b1.set_sample_rate(SR)
    b1.set_seed(np.arange(V, dtype=np.uint64) + 3)
    g1 = run_bank(b1, x[:, :1], T, LAYOUT_VOICE_MINOR, mode)
    b2 = gpu.Bank("dsf", V)
    b2.set_param(":harmonic_spacing", spacing)
    b2.set_sample_rate(SR)
    b2.set_seed(np.arange(V, dtype=np.uint64) + 3)
    g2 = run_bank(b2, x, T, LAYOUT_PLANAR, mode)
    for v in (0, 1, 2, 5, 6, 40, 95):
        n1 = O.dsf_saw_r(float(rough[v])) if spacing == 1.0 else O.dsf_square_r(float(rough[v]))
        n1.set_sample_rate(SR)
        n1.set_seed(v + 3)
        assert_bit_equal(g1[v], oracle_render(n1, x[v, :1], T, mode), f"dsf_r voice {v}")
        n2 = O.dsf_saw() if spacing == 1.0 else O.dsf_square()
        n2.set_sample_rate(SR)
        n2.set_seed(v + 3)
        assert_bit_equal(g2[v], oracle_render(n2, x[v], T, mode), f"dsf voice {v}")


# ---- Pluck (oscillator.rs:210-317): the funutd excitation stream is uploaded, everything after it is on the device ----
@pytest.mark.parametrize("mode", MODES)
def test_pluck(gpu, mode):
    V, T = 70, 64 * 12 + 5
    rng = np.random.default_rng(83)
    freq = (60.0 + 2000.0 * rng.random(V) ** 2).astype(np.float32)
    gps = (0.1 + 0.85 * rng.random(V)).astype(np.float32)
    damp = rng.random(V).astype(np.float32)
    exc = rng.uniform(-1, 1, (V, 1024)).astype(np.float32)     # >= sample_rate / 60 Hz - 1 samples
    x = np.zeros((V, 1, T), dtype=np.float32)
    x[:, 0, 300:340] = (rng.random((V, 40)) - 0.5).astype(np.float32)   # "extra string excitation" on the input
    b = gpu.Bank("pluck", V, ring_frames=1024)
    b.set_param(":frequency", freq)
    b.set_param(":gain_per_second", gps)
    b.set_param(":high_frequency_damping", damp)
    b.set_ring(0, exc)
    b.set_sample_rate(SR)
    got = np.concatenate([run_bank(b, x, T, LAYOUT_VOICE_MINOR, mode), run_bank(b, x, T, LAYOUT_PLANAR, mode)], axis=-1)
    b.reset()                                                   # reset re-initialises the line from the same stream
    again = run_bank(b, x, T, LAYOUT_VOICE_MINOR, mode)
    assert_bit_equal(again, got[:, :, :T], "pluck after reset")
    xx = np.concatenate([x, x], axis=-1)
    for v in (0, 1, 33, 64, 69):
        n = O.pluck(float(freq[v]), float(gps[v]), float(damp[v]), exc[v])
        n.set_sample_rate(SR)
        assert_bit_equal(got[v], oracle_render(n, xx[v], 2 * T, mode), f"pluck voice {v}")
    assert np.abs(got).max() > 0.05


# ---- Envelope / lfo (envelope.rs:17-179): closures as device functors ---------------------------------------------------
@pytest.mark.parametrize("sr", [48000.0, 5000.0])  # 5 kHz: several ~2 ms segments end inside one 64-sample block
@pytest.mark.parametrize("mode", MODES)
def test_lfo_functors(gpu, mode, sr):
    SR = sr
    V, T = 70, 64 * 20 + 9
    rng = np.random.default_rng(85)
    a = (0.2 + rng.random(V)).astype(np.float32)
    k = (0.5 + 40.0 * rng.random(V)).astype(np.float32)
    b = gpu.Bank("lfo_exp", V)
    b.set_param("0:a", a)
    b.set_param("0:k", k)
    b.set_sample_rate(SR)
    seeds = np.arange(V, dtype=np.uint64) * 5 + 1
    b.set_seed(seeds)
    got = np.concatenate([run_bank(b, None, T, LAYOUT_VOICE_MINOR, mode), run_bank(b, None, T, LAYOUT_PLANAR, mode)], axis=-1)
    hz = (0.2 + 15.0 * rng.random(V)).astype(np.float32)
    s = gpu.Bank("lfo_sine_hz", V)
    s.set_param("0:hz", hz)
    s.set_param("0:lo", 0.25)
    s.set_param("0:hi", 2.0)
    s.set_sample_rate(SR)
    s.set_seed(seeds)
    gots = run_bank(s, None, T, LAYOUT_VOICE_MINOR, mode)
    TAU = np.float32(6.2831855)
    for v in (0, 9, 64, 69):
        av, kv, hv = a[v], k[v], hz[v]
        n = O.lfo(lambda t: av * O.m_expf(-t * kv))
        n.set_sample_rate(SR)
        n.set_seed(int(seeds[v]))
        want = np.concatenate([oracle_render(n, None, T, mode), oracle_render(n, None, T, mode)], axis=-1)  # same block partition
        assert_bit_equal(got[v], want, f"lfo_exp voice {v}")

        def sine(t):
            u = O.m_sinf(t * hv * TAU) * np.float32(0.5) + np.float32(0.5)
            return np.float32(0.25) * (np.float32(1.0) - u) + np.float32(2.0) * u
        m = O.lfo(sine)
        m.set_sample_rate(SR)
        m.set_seed(int(seeds[v]))
        assert_bit_equal(gots[v], oracle_render(m, None, T, mode), f"lfo_sine_hz voice {v}")


def test_jit_envelope_with_user_functor(gpu):
    """envelope(|t| (sin_hz(r, t), cos_hz(r, t))) * noise stack: the closure arrives as C++ source with the graph."""
    from fundsp_amd import graph as G

    src = """
struct EnvCircle {  // |t| (sin_hz(rate, t), cos_hz(rate, t))
    static constexpr int OUT = 2;
    float rate;
    template <class V> FD_HD void visit(V& v) { v.f(rate, PARAM, "rate"); }
    FD_HD void init() { rate = 1.0f; }
    FD_HD void eval(float t, float* out) const {
        out[0] = sinf_musl(t * rate * F32_TAU);
        out[1] = cosf_musl(t * rate * F32_TAU);
    }
};
"""
    V, T = 64, 64 * 9 + 30
    rate = np.linspace(0.5, 20.0, V).astype(np.float32)
    g = G.envelope("EnvCircle", src, outputs=2, rate=rate) * (G.noise() | G.noise())
    b = gpu.Bank.from_graph(g, V, sample_rate=SR)
    b.set_seed(np.arange(V, dtype=np.uint64) + 2)
    got = run_bank(b, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    TAU = np.float32(6.2831855)
    for v in (0, 31, 63):
        r = rate[v]
        n = O.envelope(lambda t: (O.m_sinf(t * r * TAU), O.m_cosf(t * r * TAU)), outputs=2) * (O.noise() | O.noise())
        n.set_sample_rate(SR)
        n.set_seed(v + 2)
        assert_bit_equal(got[v], oracle_render(n, None, T, MODE_PROCESS), f"jit envelope voice {v}")


# ---- Resample (resample.rs:205-315) ---------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", MODES)
def test_resample_fm(gpu, mode):
    from fundsp_amd import workloads as W

    V, T = 70, 64 * 4 + 27
    rng = np.random.default_rng(87)
    p = W.fm_svf_params(V, SR)
    speed = (0.25 + 2.5 * rng.random((V, 1, T))).astype(np.float32)
    speed[0] = 1.0                                              # original speed
    speed[1, 0, 50:90] = 0.0                                    # frozen: no inner samples consumed
    speed[2, 0, :] = -1.0                                       # negative speeds clamp to 0 (max(0.0, input))
    speed[3, 0, 100] = 37.5                                     # a jump: many inner ticks for one output sample
    b = gpu.Bank("resample_fm", V)
    b.set_param("0.0.0.0.0.0:value[0]", p["f"])
    b.set_param("0.0.0.0:scalar", p["f"])
    b.set_param("0.0.0:scalar", p["m"])
    b.set_param("0.0:scalar", p["f"])
    b.set_sample_rate(SR)
    seeds = np.arange(V, dtype=np.uint64) + 21
    b.set_seed(seeds)
    got = np.concatenate([run_bank(b, speed, T, LAYOUT_VOICE_MINOR, mode), run_bank(b, speed, T, LAYOUT_PLANAR, mode)], axis=-1)
    xx = np.concatenate([speed, speed], axis=-1)
    for v in (0, 1, 2, 3, 40, 69):
        f, m = float(p["f"][v]), float(p["m"][v])
        n = O.resample(O.sine_hz(f) * f * m + f >> O.sine())
        n.set_sample_rate(SR)
        n.set_seed(int(seeds[v]))
        assert_bit_equal(got[v], oracle_render(n, xx[v], 2 * T, mode), f"resample_fm voice {v}")


@pytest.mark.parametrize("mode", MODES)
def test_lfo2_envelope_in(gpu, mode):
    """lfo2(|t, speed| exp(-t * speed)) (prelude32.rs:623): EnvelopeIn with a stateless closure of (t, input)."""
    V, T = 70, 64 * 12 + 31
    rng = np.random.default_rng(89)
    x = np.repeat((1.0 + 30.0 * rng.random((V, 1, T // 50 + 1))).astype(np.float32), 50, axis=2)[:, :, :T].copy()
    b = gpu.Bank("lfo2_exp", V)
    b.set_sample_rate(SR)
    seeds = np.arange(V, dtype=np.uint64) * 3 + 2
    b.set_seed(seeds)
    got = run_bank(b, x, T, LAYOUT_VOICE_MINOR, mode)
    gotp = None
    for v in (0, 17, 64, 69):
        n = O.lfo2(lambda t, s: O.m_expf(-t * s))
        n.set_sample_rate(SR)
        n.set_seed(int(seeds[v]))
        assert_bit_equal(got[v], oracle_render(n, x[v], T, mode), f"lfo2_exp voice {v}")


def test_limiter_long_windows_walk_the_tree_incrementally(gpu):
    """Limiter windows update their reduce tree incrementally (path and sibling values in registers / slots, the nodes the path leaves written when it
    leaves them: fd_nodes.hpp tree_set_inc).  Per-voice attack times -- windows of 192 .. 2 400 frames, so the lanes of a wave sit at different tree
    indices in trees of different heights (two specialisations of the walk in one wave) --, several laps of every window over three launches, a clone
    taken in mid-stream, a reset: every voice bit-equal to the oracle's ReduceBuffer walk."""
    from fundsp_amd import graph as GR

    V = 70
    attack = np.linspace(0.004, 0.05, V).astype(np.float32)      # 192 .. 2 400 frames at 48 kHz
    g = GR.pass_() * 2.5 >> GR.limiter(attack, 0.05)
    b = gpu.Bank.from_graph(g, V, ring_frames=8192, sample_rate=SR)
    chunks = (3000, 64 * 31 + 5, 2500)
    x = noise_input(V, 1, sum(chunks), seed=11)
    x[:, :, 4000:4100] *= 6.0   # a burst the look-ahead has to catch

    def render(bank):
        outs, at = [], 0
        for n in chunks:
            outs.append(run_bank(bank, x[:, :, at:at + n], n, LAYOUT_VOICE_MINOR, MODE_PROCESS))
            at += n
        return np.concatenate(outs, axis=2)

    first = run_bank(b, x[:, :, :chunks[0]], chunks[0], LAYOUT_VOICE_MINOR, MODE_PROCESS)
    twin = b.clone()
    rest_b = [run_bank(b, x[:, :, 3000:3000 + chunks[1]], chunks[1], LAYOUT_VOICE_MINOR, MODE_PROCESS)]
    rest_t = [run_bank(twin, x[:, :, 3000:3000 + chunks[1]], chunks[1], LAYOUT_VOICE_MINOR, MODE_TICK)]   # (the limiter has no process override: tick == process)
    at = 3000 + chunks[1]
    rest_b.append(run_bank(b, x[:, :, at:], chunks[2], LAYOUT_VOICE_MINOR, MODE_PROCESS))
    rest_t.append(run_bank(twin, x[:, :, at:], chunks[2], LAYOUT_PLANAR, MODE_PROCESS))
    got = np.concatenate([first] + rest_b, axis=2)
    assert_bit_equal(np.concatenate([first] + rest_t, axis=2), got, "the clone continues like the original (other executor / layout)")
    for v in (0, 5, 6, 33, 69):
        n = O.pass_() * 2.5 >> O.limiter(float(attack[v]), 0.05)
        n.set_sample_rate(SR)
        want = np.concatenate([n.render_blocks(x[v][:, a:a + k], block=64) for a, k in ((0, chunks[0]), (3000, chunks[1]), (at, chunks[2]))], axis=1)
        assert_bit_equal(got[v], want, f"voice {v} (window {int(round(SR * float(attack[v])))} frames)")
    b.reset()
    again = render(b)
    n = O.pass_() * 2.5 >> O.limiter(float(attack[40]), 0.05)   # (reset clears the window and the delay line; the follower keeps its state, dynamics.rs:184-186)
    n.set_sample_rate(SR)
    spans = ((0, chunks[0]), (3000, chunks[1]), (at, chunks[2]))
    for a, k in spans:
        n.render_blocks(x[40][:, a:a + k], block=64)
    n.reset()
    want = np.concatenate([n.render_blocks(x[40][:, a:a + k], block=64) for a, k in spans], axis=1)
    assert_bit_equal(again[40], want, "after reset")


def test_oversampled_stateful_inner_node_over_odd_launch_lengths(gpu):
    """Oversampler::process hands its inner node a block of `size` samples per pass (oversample.rs:191-195): with an odd `size` that is one zero-input
    sample more than the pass interpolated, and a STATEFUL inner node (a filter, an oscillator) advances by it.  Launches whose last block is odd
    (275 = 4 x 64 + 19), launches of 1, 19 and 7 frames, then a whole block: every one continues the oracle's state.  (Found by the wider-pool
    fuzzer; tests/host/check_oversample_odd.hip is the same check on the CPU.)"""
    import oracle as O
    from fundsp_amd import graph as GR
    from test_gpu_parity import run_bank as run_bank_

    V = 5
    seeds = np.arange(V, dtype=np.uint64) * 13 + 5
    graphs = {
        "noise >> oversample(lowpass_hz)": lambda m: m.noise() >> m.oversample(m.lowpass_hz(3000.0, 1.0)),
        "oversample(sine_hz * 0.5 >> highpole_hz)": lambda m: m.oversample(m.sine_hz(440.0) * 0.5 >> m.highpole_hz(200.0)),
    }
    for name, mk in graphs.items():
        for layout in (LAYOUT_VOICE_MINOR, LAYOUT_PLANAR):
            b = gpu.Bank.from_graph(mk(GR), V, sample_rate=SR)
            b.set_seed(seeds)
            nodes = []
            for v in (0, V - 1):
                n = mk(O)
                n.set_sample_rate(SR)
                n.set_seed(int(seeds[v]))
                nodes.append((v, n))
            for frames in (275, 1, 19, 7, 64, 1, 63, 128):
                got = run_bank_(b, None, frames, layout, MODE_PROCESS)
                for v, n in nodes:
                    want = n.render_blocks(None, length=frames, block=64)
                    even = frames - (frames % 64 % 2)   # the odd last outer sample of a block is never written by the reference
                    assert_bit_equal(got[v][:, :even], want[:, :even], f"{name} layout {layout} launch of {frames} frames, instance {v}")
