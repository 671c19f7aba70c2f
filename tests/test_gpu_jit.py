"""Run-time compiled voice graphs (fdsp_graph_compile, hiprtc): arbitrary FunDSP graph notation -> one fused kernel.
The same graph expression is built twice -- with the engine's host notation (fundsp_amd.graph) and with the oracle's
(tests/oracle.py) -- and the outputs must match bit for bit; a JIT kind must also equal the same graph compiled ahead
of time."""
import numpy as np
import pytest

import oracle as O
from fundsp_amd import LAYOUT_PLANAR, LAYOUT_VOICE_MINOR, MODE_PROCESS, MODE_TICK
from fundsp_amd import graph as GR
from fundsp_amd import workloads as W
from test_gpu_parity import assert_bit_equal, noise_input, oracle_render, run_bank

pytestmark = pytest.mark.gpu
SR = 48000.0


def test_jit_equals_aot_config3(gpu):
    V, T = 300, 64 * 5 + 9
    p = W.fm_svf_params(V, SR)
    f, m, fc, q = p["f"], p["m"], p["fc"], p["q"]
    g = GR.sine_hz(f) * f * m + f >> GR.sine() >> GR.lowpass_hz(fc, q)          # README.md:98-103, per-voice arrays
    assert g.type == "Pipe<Pipe<Unop<Unop<Unop<Pipe<Constant<1>,Sine>,UMulScalar>,UMulScalar>,UAddScalar>,Sine>,FixedSvf>"
    jit = gpu.Bank.from_graph(g, V, sample_rate=SR)
    jit.set_seed(p["seed"])
    aot = W.make_fm_svf_bank(V, SR, params=p)
    for layout in (LAYOUT_VOICE_MINOR, LAYOUT_PLANAR):
        for mode in (MODE_PROCESS, MODE_TICK):
            jit.reset(); aot.reset()
            jit.set_seed(p["seed"]); aot.set_seed(p["seed"])
            assert_bit_equal(run_bank(jit, None, T, layout, mode), run_bank(aot, None, T, layout, mode), "jit == aot")
    # same kind name -> cached, no recompilation
    g8 = GR.sine_hz(100.0) * 100.0 * 2.0 + 100.0 >> GR.sine() >> GR.lowpass_hz(500.0, 1.0)   # same TYPE, scalar params
    again = gpu.Bank.from_graph(g8, 8, sample_rate=SR)
    assert again.kind == jit.kind


GRAPHS = {
    # name: (builder taking a notation module, inputs, ring_frames)
    "noise_moog": (lambda m: m.noise() >> m.moog_hz(1500.0, 0.4), 0, 0),
    "fm_pair_shaped": (lambda m: (m.sine_hz(110.0) + m.sine_hz(220.0) * 0.5) >> m.shape("tanh", 2.0), 0, 0),
    "modulated_svf": (lambda m: (m.noise() | m.sine_hz(0.7) * 800.0 + 1000.0 | m.dc(2.0)) >> m.lowpass(), 0, 0),
    "stack_binop_sub": (lambda m: (m.noise() | m.noise()) >> (m.lowpole_hz(500.0) | m.highpole_hz(2000.0)) >> (m.pass_() - m.pass_()), 0, 0),
    "comb_allpass_chain": (lambda m: m.noise() >> m.allnest_c(0.5, m.delay(0.002)) >> m.dcblock_hz(20.0) >> m.peak_hz(3000.0, 2.0) * 0.25, 0, 128),
    "saw_filter_env": (lambda m: (m.saw_hz(82.4) >> m.lowpass_hz(900.0, 3.0)) * (m.pass_() >> m.adsr_live(0.002, 0.01, 0.5, 0.005)), 1, 0),
    "chorus_tap": (lambda m: (m.pass_() | m.sine_hz(1.3) * 0.001 + 0.003) >> m.tap(0.001, 0.005), 1, 512),
    "pulse_resonator": (lambda m: (m.dc(140.0, 0.3) >> m.poly_pulse()) >> m.resonator_hz(700.0, 40.0) >> m.pan(-0.4), 0, 0),
    # Bus `&`, Branch `^`, Thru `!`, Sink, routing leaves, the N-fold closure forms, Declick, the composed opcodes
    "bus_branch_thru": (lambda m: (m.sine_hz(440.0) & m.sine_hz(220.0)) >> (m.pass_() ^ m.lowpole_hz(100.0)) >> (~m.sink() | m.pass_()), 0, 0),
    "split_join": (lambda m: (m.noise() | m.noise()) >> m.multisplit(2, 3) >> m.multijoin(2, 3) >> m.reverse(2) >> m.join(2) >> m.split(3) >> m.join(3), 0, 0),
    "busi_sines": (lambda m: m.busi(4, lambda i: m.sine_hz(100.0 * (i + 1))) * 0.25, 0, 0),
    "stacki_sumi": (lambda m: m.stacki(3, lambda i: m.sine_hz(100.0 * (i + 1))) >> m.sumi(3, lambda i: m.lowpole_hz(100.0 + i)), 0, 0),
    "branchf_filters": (lambda m: m.branchf(3, lambda t: m.lowpass_hz(500.0 + 1500.0 * float(t), 1.0)) >> m.join(3), 1, 0),
    "pipei_poles": (lambda m: m.noise() >> m.pipei(4, lambda i: m.lowpole_hz(1000.0 + 100.0 * i)), 0, 0),
    "busf_resonators": (lambda m: m.busf(5, lambda t: (m.noise() | m.dc(200.0 + 900.0 * float(t), 20.0)) >> ~m.resonator() >> m.resonator()), 0, 0),
    "impulse_declick": (lambda m: (m.impulse() + m.noise()) >> m.declick_s(0.004), 0, 0),
    "svf_q_forms": (lambda m: (m.pass_() | m.sine_hz(2.0) * 300.0 + 1000.0) >> (m.lowpass_q(2.0) ^ m.bell_q(1.5, 2.0)) >> (m.pass_() - m.pass_()), 1, 0),
    "brown_pink": (lambda m: m.brown() & m.pink(), 0, 0),
    "nl_biquads": (lambda m: m.fresonator_hz(m.Tanh(1.0), 500.0, 2.0) >> m.dlowpass_hz(m.Softsign(0.9), 800.0, 1.0) >> m.clip_to(-0.5, 0.5), 1, 0),
    # Feedback / Feedback2 with FrameId and FrameHadamard (feedback.rs); the enclosed nodes run their tick arithmetic
    "feedback_echo": (lambda m: m.feedback(m.delay(0.001) * 0.9) >> m.feedback2(m.delay(0.0007), m.lowpole_hz(1500.0) * 0.8), 1, 64),
    "fdn4": (lambda m: m.split(4) >> m.fdn(m.stacki(4, lambda i: m.delay(0.0005 * (i + 1)) >> m.fir(0.3, 0.4, 0.2))) >> m.join(4), 1, 128),
    "fdn2_loop_filters": (lambda m: (m.pass_() | m.noise() * 0.01) >> m.fdn2(m.stacki(2, lambda i: m.delay(0.0011 * (i + 1))), m.stacki(2, lambda i: m.lowpole_hz(3000.0) * 0.7)) >> m.join(2), 1, 128),
    "feedback_denormal_decay": (lambda m: m.impulse() >> m.feedback(m.tick() * 0.5) >> m.lowpole_hz(5000.0), 0, 0),
    # wavetable family: PulseWave (WaveSynth<U2> + PhaseSynth), organ / soft saw / hammond tables
    "pulse_wave": (lambda m: (m.sine_hz(3.0) * 50.0 + 220.0 | m.sine_hz(0.7) * 0.3 + 0.5) >> m.pulse() * 0.2, 0, 0),
    "organ_family": (lambda m: m.organ_hz(110.0) + m.soft_saw_hz(220.0) * 0.5 + (m.sine_hz(5.0) * 3.0 + 55.0 >> m.hammond()), 0, 0),
    "multitap_allnest_panner": (lambda m: (m.pass_() | m.sine_hz(0.9) * 0.001 + 0.003 | m.sine_hz(1.7) * 0.0005 + 0.002) >> m.multitap(2, 0.001, 0.005)
                                >> (m.pass_() | m.sine_hz(3.0) * 0.6) >> m.allnest(m.delay(0.0013)) >> m.multitick(1)
                                >> (m.pass_() | m.sine_hz(0.5)) >> m.panner(), 1, 512),
    "multitap_linear3": (lambda m: (m.pass_() | m.dc(0.001, 0.0021, 0.0034)) >> m.multitap_linear(3, 0.0005, 0.004) >> m.join(1), 1, 256),
    # dynamics: look-ahead limiter (reduce tree + delay rings), meters, a Shared value as a parameter
    "limiter_mono": (lambda m: m.pass_() * 3.0 >> m.limiter(0.002, 0.02), 1, 512),
    "limiter_stereo": (lambda m: (m.pass_() * 2.0 | m.noise() * (m.sine_hz(3.0) + 1.0)) >> m.limiter_stereo(0.001, 0.01), 1, 256),
    "meters": (lambda m: (m.pass_() * m.var(0.7)) >> (m.meter("peak", 0.01) ^ m.meter("rms", 0.02) ^ m.meter("sample") ^ m.monitor("rms", 0.005)), 1, 0),
    "moog_q_thru_cut": (lambda m: (m.pass_() | m.dc(800.0)) >> m.moog_q(0.5) >> m.clip() >> m.split(2) >> ~(m.sink() | m.sink()) >> m.join(2), 1, 0),
}


@pytest.mark.parametrize("name", list(GRAPHS))
def test_jit_graph_matches_oracle(gpu, name):
    build, ni, ring = GRAPHS[name]
    g = build(GR)
    V, T = 70, 64 * 9 + 11
    for kind in GR.uses_wavetables(g):   # the oracle's numpy-built tables, so both sides read identical table bits
        t = O.Wavetable.get(kind)
        offs = np.concatenate([[0], np.cumsum(t.lengths)])
        gpu.wavetable_upload(kind, t.pitches, [t.data[offs[i]:offs[i + 1]] for i in range(len(t.lengths))])
    seeds = np.arange(V, dtype=np.uint64) * 7919 + 13
    x = None
    if ni:
        x = noise_input(V, ni, T, seed=77)
        if name in ("saw_filter_env", "saw_dc2_moog_env"):
            x[:, 0, :] = 0.0
            x[:, 0, 3:400] = 1.0  # gate
    for mode in (MODE_PROCESS, MODE_TICK):
        # a fresh bank per mode: reset() deliberately leaves adsr_live's closure state alone (envelope.rs:293-298)
        b = gpu.Bank.from_graph(g, V, ring_frames=ring, sample_rate=SR)
        assert b.inputs() == ni and b.outputs() == g.nout
        b.set_seed(seeds)
        got = run_bank(b, x, T, LAYOUT_VOICE_MINOR if mode == MODE_PROCESS else LAYOUT_PLANAR, mode)
        for v in (0, 1, 35, 69):
            n = build(O)
            n.set_sample_rate(SR)
            n.set_seed(int(seeds[v]))
            assert_bit_equal(got[v], oracle_render(n, None if x is None else x[v], T, mode), f"{name} voice {v} mode {mode}")


# A stage cut right behind a Stack that ends in Constants: those channels do not travel through the hand-over tiles, the consumer
# stage loads the Constant slots itself (fd_device.hpp ConstTail) -- here a Constant<2> tail in front of a ladder, under a Binop tail that reads
# the graph's input (the nested one-by-one tail is config 4's own kind, tests/test_gpu_config4.py).  One graph only: a heavy kind
# takes hiprtc ~100 s to compile.  (Kept apart from GRAPHS: that table is also the inventory of the golden fixtures and of the Rust front door.)
CUT_GRAPHS = {
    "saw_dc2_moog_env": (lambda m: ((m.saw_hz(61.7) | m.dc(1200.0, 0.3)) >> m.moog()) * (m.pass_() >> m.adsr_live(0.003, 0.02, 0.6, 0.01)), 1, 0),
}


@pytest.mark.parametrize("name", list(CUT_GRAPHS))
def test_jit_constants_behind_a_cut(gpu, name):
    GRAPHS[name] = CUT_GRAPHS[name]
    try:
        test_jit_graph_matches_oracle(gpu, name)
    finally:
        del GRAPHS[name]


def test_jit_map_and_shape_fn_closures(gpu):
    """map(|i| ..) (audionode.rs:1330, prelude32.rs:332) and shape_fn(|x| ..) (shape.rs:35): the Rust closure arrives as
    a C++ functor with the graph; the oracle plays it through a callback with the same f32 arithmetic."""
    src = """
struct MidSide {  // map(|i: &Frame<f32, U2>| (i[0] + i[1], i[0] - i[1], i[0] * i[1]))
    static FD_HD void f(const float* in, float* out) { out[0] = in[0] + in[1]; out[1] = in[0] - in[1]; out[2] = in[0] * in[1]; }
};
struct SoftFold {  // shape_fn(|x| x / (1.0 + x * x) + tanh(x))
    static FD_HD float f(float x) { return x / (1.0f + x * x) + tanhf_musl(x); }
};
"""
    f32 = np.float32
    g = (GR.noise() | GR.sine_hz(330.0)) >> GR.map_("MidSide", src, 2, 3) >> (GR.shape_fn("SoftFold", src) | GR.pass_() | GR.lowpole_hz(900.0))
    n0 = lambda: ((O.noise() | O.sine_hz(330.0)) >> O.map_(lambda i: (i[0] + i[1], i[0] - i[1], i[0] * i[1]), 2, 3)
                  >> (O.shape_fn(lambda x: x / (f32(1.0) + x * x) + f32(O.lib().o_math_tanhf(float(x)))) | O.pass_() | O.lowpole_hz(900.0)))
    V, T = 40, 64 * 3 + 13
    seeds = np.arange(V, dtype=np.uint64) * 31 + 5
    for mode in (MODE_PROCESS, MODE_TICK):
        b = gpu.Bank.from_graph(g, V, sample_rate=SR)
        b.set_seed(seeds)
        got = run_bank(b, None, T, LAYOUT_VOICE_MINOR, mode)
        for v in (0, 39):
            n = n0()
            n.set_sample_rate(SR)
            n.set_seed(int(seeds[v]))
            assert_bit_equal(got[v], oracle_render(n, None, T, mode), f"map/shape_fn voice {v} mode {mode}")


def test_jit_mixer_varfn_biquad_bank_envelope3(gpu):
    """rotate(angle, gain) = Mixer<U2, U2> (pan.rs:95, prelude32.rs:2432), var_fn(&shared, f) (shared.rs:136),
    biquad_bank() with per-lane coefficients (biquad_bank.rs), lfo3(|t, x, y| ..) (prelude32.rs:697)."""
    f32 = np.float32
    src = """
struct Detune { static FD_HD void f(float v, float* out) { out[0] = v * 0.99f; out[1] = v * 1.01f; } };
struct Env3 {  // |t, x, y| exp(-t * x) * y
    static constexpr int IN = 2, OUT = 1;
    template <class V> FD_HD void visit(V&) {}
    FD_HD void init() {}
    FD_HD void eval(float t, const float* in, float* out) const { out[0] = expf_musl(-t * in[0]) * in[1]; }
};
"""
    coefs = [O.biquad_coefs("lowpass", SR, 300.0 * (i + 1), 0.7 + 0.2 * i) for i in range(8)]
    V, T = 20, 64 * 4 + 31
    base = np.linspace(110.0, 440.0, V).astype(np.float32)
    g = (GR.var_fn(base, "Detune", src, 2) >> (GR.sine() | GR.sine()) >> GR.rotate(0.6, 0.8) >> GR.multisplit(2, 4)
         >> GR.biquad_bank(coefs) >> GR.multijoin(2, 4)
         >> (GR.pass_() * (GR.dc(3.0, 0.5) >> GR.lfo3("Env3", src)) | GR.pass_()))
    def make(v):
        return (O.var_fn(float(base[v]), lambda x: (x * f32(0.99), x * f32(1.01)), 2) >> (O.sine() | O.sine())
                >> O.rotate(0.6, 0.8) >> O.multisplit(2, 4) >> O.biquad_bank_coefs(coefs) >> O.multijoin(2, 4)
                >> (O.pass_() * (O.dc(3.0, 0.5) >> O.lfo3(lambda t, x, y: O.m_expf(-t * x) * y)) | O.pass_()))
    seeds = np.arange(V, dtype=np.uint64) + 7
    for mode in (MODE_PROCESS, MODE_TICK):
        b = gpu.Bank.from_graph(g, V, sample_rate=SR)
        b.set_seed(seeds)
        got = run_bank(b, None, T, LAYOUT_VOICE_MINOR, mode)
        for v in (0, V - 1):
            n = make(v)
            n.set_sample_rate(SR)
            n.set_seed(int(seeds[v]))
            assert_bit_equal(got[v], oracle_render(n, None, T, mode), f"mixer/var_fn/bank voice {v} mode {mode}")


def test_jit_sampler_voices(gpu):
    """playwave_at(wave, channel, start, end, loop) (wave.rs:739-797) as a bank of sampler heads over one shared Wave
    in HBM: per-voice start / end / loop points, and resample(playwave(..)) for a per-voice, per-sample playback speed
    (resample.rs:205-315)."""
    V, T, L = 48, 64 * 9 + 7, 700
    rng = np.random.default_rng(41)
    wave = (rng.random((2, L), dtype=np.float32) * 2 - 1).astype(np.float32)
    gpu.wave_upload(3, wave)
    start = rng.integers(0, 300, V).astype(np.uint32)
    end = (start + rng.integers(1, 400, V)).astype(np.uint32)
    loop = np.where(np.arange(V) % 3 == 0, 0xFFFFFFFF, start + (end - start) // 2).astype(np.uint32)   # a third plays once
    chan = (np.arange(V) % 2).astype(np.uint32)
    g = GR.playwave_at(3, chan, start, end, loop) * 0.5
    for mode in (MODE_PROCESS, MODE_TICK):
        b = gpu.Bank.from_graph(g, V, sample_rate=SR)
        got = run_bank(b, None, T, LAYOUT_VOICE_MINOR, mode)
        for v in (0, 1, 2, 17, V - 1):
            n = O.playwave_at(wave, int(chan[v]), int(start[v]), int(end[v]), None if loop[v] == 0xFFFFFFFF else int(loop[v])) * 0.5
            assert_bit_equal(got[v], oracle_render(n, None, T, mode), f"sampler voice {v} mode {mode}")
    # variable speed: the speed input drives Resample's cubic read of the playing wave
    speed = (0.3 + 1.7 * rng.random((V, 1, T))).astype(np.float32)
    g2 = GR.resample(GR.playwave(3, 0, L, loop_point=10))
    b = gpu.Bank.from_graph(g2, V, sample_rate=SR)
    got = run_bank(b, speed, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    for v in (0, V - 1):
        n = O.resample(O.playwave(wave, 0, loop_point=10))
        assert_bit_equal(got[v], oracle_render(n, speed[v], T, MODE_PROCESS), f"resampled sampler voice {v}")


def test_jit_hold_with_uploaded_rnd_stream(gpu):
    """hold_hz(f, variability) (noise.rs:242-322, prelude32.rs:831): sample-and-hold whose hold lengths come from
    funutd's Rnd -- the draws are uploaded per voice (as for Pluck), everything else runs on the device in f64 time."""
    V, T, ND = 40, 64 * 30 + 11, 64
    rng = np.random.default_rng(31)
    draws = rng.random((V, ND))                                   # stands in for Rnd::from_u64(hash).f64()
    g = GR.noise() >> GR.hold_hz(700.0, 0.5)
    seeds = np.arange(V, dtype=np.uint64) + 3
    for mode in (MODE_PROCESS, MODE_TICK):
        b = gpu.Bank.from_graph(g, V, ring_frames=2 * ND, sample_rate=SR)
        b.set_seed(seeds)
        b.set_ring(0, GR.hold_stream(draws))
        got = run_bank(b, None, T, LAYOUT_VOICE_MINOR, mode)
        for v in (0, 17, V - 1):
            n = O.noise() >> O.hold_hz(700.0, 0.5, draws[v])
            n.set_sample_rate(SR)
            n.set_seed(int(seeds[v]))
            want = oracle_render(n, None, T, mode)
            assert len(np.unique(want)) > 20                      # it does hold and re-sample
            assert_bit_equal(got[v], want, f"hold voice {v} mode {mode}")
        b.reset()                                                  # the stream restarts with the generator (:281-285)
        again = run_bank(b, None, 256, LAYOUT_VOICE_MINOR, mode)
        assert_bit_equal(again, got[:, :, :256], "hold after reset")


def test_jit_flanger_and_phaser(gpu):
    """flanger(..) / phaser(..) (prelude.rs:2719-2753): Bus + Feedback2 / Feedback around taps, a tanh shaper, ten
    pass-through allpoles; the modulation closure is a functor on the engine side and a callback in the oracle."""
    f32 = np.float32
    TAU = f32(6.2831855)
    hz = 0.7
    sin01 = lambda t: O.m_sinf(t * f32(hz) * TAU) * f32(0.5) + f32(0.5)                        # EnvSineHz::eval
    V, T = 33, 64 * 7 + 5
    x = noise_input(V, 1, T, seed=9)
    cases = {
        "flanger": (GR.flanger(0.6, 0.002, 0.006, "EnvSineHz", hz=hz, lo=0.002, hi=0.006),
                    lambda: O.flanger(0.6, 0.002, 0.006, lambda t: f32(0.002) * (f32(1.0) - sin01(t)) + f32(0.006) * sin01(t)), 512),
        "phaser": (GR.phaser(0.5, "EnvSineHz", hz=hz, lo=0.0, hi=1.0),
                   lambda: O.phaser(0.5, lambda t: f32(0.0) * (f32(1.0) - sin01(t)) + f32(1.0) * sin01(t)), 0),
    }
    seeds = np.arange(V, dtype=np.uint64) + 100
    for name, (g, make, ring) in cases.items():
        for mode in (MODE_PROCESS, MODE_TICK):
            b = gpu.Bank.from_graph(g, V, ring_frames=ring, sample_rate=SR)
            b.set_seed(seeds)
            got = run_bank(b, x, T, LAYOUT_VOICE_MINOR, mode)
            for v in (0, 32):
                n = make()
                n.set_sample_rate(SR)
                n.set_seed(int(seeds[v]))
                assert_bit_equal(got[v], oracle_render(n, x[v], T, mode), f"{name} voice {v} mode {mode}")


def test_jit_reverb4_stereo(gpu):
    """reverb4_stereo(room_size, time) (prelude.rs:1873-1941) as a run-time compiled graph: two 16-line Hadamard FDNs in
    series (stacki of delay >> fir inside fdn), multisplit / multijoin, a sumf of 16 panners -- 32 delay rings per voice,
    rendered long enough for both feedback loops to close several times."""
    V, T = 6, 64 * 200 + 17
    g = GR.reverb4_stereo(20.0, 2.0)
    assert g.rings == 32 and (g.nin, g.nout) == (2, 2)
    x = noise_input(V, 2, T, seed=21)
    x[:, :, 2000:] = 0.0                                           # a burst, then the tail
    b = gpu.Bank.from_graph(g, V, ring_frames=8192, sample_rate=SR, fdn_kernel=False)   # (the run-time compiled graph, not the dedicated kernel)
    assert b.kind.startswith("jit_")
    got = run_bank(b, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    for v in (0, V - 1):
        n = O.reverb4_stereo(20.0, 2.0)
        n.set_sample_rate(SR)
        want = oracle_render(n, x[v], T, MODE_PROCESS)
        assert np.abs(want[:, 9000:]).max() > 1e-4                 # the tail is alive
        assert_bit_equal(got[v], want, f"reverb4_stereo voice {v}")


def test_jit_reverb3_stereo(gpu):
    """reverb3_stereo(time, diffusion, filter) (reverb.rs:152-279): the allpass-loop reverb as one node -- 76 delay rings
    per voice walked serially, the loop filter a template parameter with per-voice cutoff."""
    V, T = 5, 64 * 300 + 9
    cut = np.linspace(900.0, 4000.0, V).astype(np.float32)
    g = GR.reverb3_stereo(2.0, 0.6, lambda: GR.lowpole_hz(cut))
    assert g.rings == 76 and (g.nin, g.nout) == (2, 2)
    x = noise_input(V, 2, T, seed=23)
    x[:, :, 3000:] = 0.0
    b = gpu.Bank.from_graph(g, V, ring_frames=2048, sample_rate=SR)
    got = run_bank(b, x, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    for v in (0, V - 1):
        n = O.reverb3_stereo(2.0, 0.6, lambda: O.lowpole_hz(float(cut[v])))
        n.set_sample_rate(SR)
        want = oracle_render(n, x[v], T, MODE_PROCESS)
        assert np.abs(want[:, 15000:]).max() > 1e-4                # past the first trip through all eight blocks
        assert_bit_equal(got[v], want, f"reverb3_stereo voice {v}")
    # reset() leaves the input diffusers' lines alone (reverb.rs:211-224): second render after reset == the oracle's
    b.reset()
    got2 = run_bank(b, np.ascontiguousarray(x[:, :, :64 * 20]), 64 * 20, LAYOUT_VOICE_MINOR, MODE_PROCESS)
    n.reset()
    assert_bit_equal(got2[V - 1], oracle_render(n, x[V - 1][:, :64 * 20], 64 * 20, MODE_PROCESS), "reverb3 after reset")


def test_jit_type_errors_are_reported(gpu):
    rc = gpu.lib().fdsp_graph_compile(b"bad_graph", b"Pipe<Sine,Stack<Sine,Sine>>")   # 1 output into 2 inputs
    assert rc < 0 and "Pipe arity mismatch" in gpu.lib().fdsp_last_error().decode()
    with pytest.raises(TypeError):
        GR.sine() >> (GR.sine() | GR.sine())                                             # caught on the host as well


def test_jit_pipeline_kernel_matches_single_wave(gpu):
    """Run-time compiled graphs take the pipeline kernel too (loader wave for inputs, two compute stages where the chain
    allows): same samples as their single-wave kernel."""
    V, T = 200, 64 * 6 + 21
    x = noise_input(V, 1, T, seed=5)
    for build, ni in ((lambda m: m.lowpass_hz(1200.0, 2.0) >> m.shape("tanh", 3.0) >> m.highpole_hz(200.0), 1),
                      (lambda m: m.noise() >> m.moog_hz(1500.0, 0.4), 0)):
        outs = []
        for flag in (0, 1):
            assert gpu.lib().fdsp_set_option(b"pipe_split", flag) == 0
            b = gpu.Bank.from_graph(build(GR), V, sample_rate=SR)
            b.set_seed(np.arange(V, dtype=np.uint64))
            outs.append([run_bank(b, x if ni else None, T, LAYOUT_VOICE_MINOR, mode) for mode in (MODE_PROCESS, MODE_TICK)])
        gpu.lib().fdsp_set_option(b"pipe_split", 1)
        for a, c in zip(*outs):
            assert_bit_equal(a, c, "jit pipeline vs single wave")


@pytest.mark.parametrize("name", ["butterpass_hz", "resonator_hz", "biquad"])
def test_jit_fixed_biquad_holders_cut_at_the_seam(gpu, name):
    """butterpass_hz(f) / resonator_hz(c, q) / biquad(..) behind a generator: the held DF1 biquad is a chain of two stages (feed-forward half |
    recurrence, fd_device.hpp HeldBiquadSeg), so `noise() >> filter` has three chain stages.  Every plan renders the oracle's samples: the
    best plan, exactly two stages, exactly three (noise | feed-forward half | recurrence), the single-wave kernel; ragged launch, state carry."""
    V, T = 64 * 3 + 11, 64 * 7 + 5
    build = {"butterpass_hz": lambda m: m.noise() >> m.butterpass_hz(900.0),
             "resonator_hz": lambda m: m.noise() >> m.resonator_hz(700.0, 120.0),
             "biquad": lambda m: m.noise() >> m.biquad(-1.2, 0.5, 0.2, 0.3, 0.1)}[name]
    want = []
    for v in range(0, V, 13):
        n = build(O)
        n.set_sample_rate(SR)
        n.set_seed(v)
        want.append(np.concatenate([n.render_blocks(length=T), n.render_blocks(length=64)], axis=1))
    outs = []
    for flag in (1, 2, 3, 0):
        assert gpu.lib().fdsp_set_option(b"pipe_split", flag) == 0
        try:
            b = gpu.Bank.from_graph(build(GR), V, sample_rate=SR)
            b.set_seed(np.arange(V, dtype=np.uint64))
            a = run_bank(b, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)
            c = run_bank(b, None, 64, LAYOUT_VOICE_MINOR, MODE_PROCESS)
            outs.append(np.concatenate([a, c], axis=2))
            if flag in (0, 2, 3):
                assert b.get_option("last_kernel") == (1 if flag == 0 else 2), (name, flag)
        finally:
            gpu.lib().fdsp_set_option(b"pipe_split", 1)
    for k, v in enumerate(range(0, V, 13)):
        assert_bit_equal(outs[0][v], want[k], f"{name} voice {v} vs oracle")
    for o in outs[1:]:
        assert_bit_equal(o, outs[0], f"{name}: every stage plan renders the same samples")


def test_jit_small_banks_take_the_time_split_kernels(gpu):
    """Run-time compiled three-stage generator chains on banks that leave most SIMDs idle (<= 2 voice groups per CU) take the three-way
    time-split kernels like the ahead-of-time kinds do (their module is compiled on the first such launch): the FM voice of BASELINE
    config 3 and `noise() >> biquad` (config 2's shape), one group per CU and two; bit-equal to the stage pipeline, to the ahead-of-time
    kind and to the oracle; a ragged launch falls back to the pipeline and continues the same state."""
    V, T = 64 * 10 + 37, 64 * 6
    p = W.fm_svf_params(V, SR)
    g = GR.sine_hz(p["f"]) * p["f"] * p["m"] + p["f"] >> GR.sine() >> GR.lowpass_hz(p["fc"], p["q"])
    want, _ = O.bank_render(3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], 2 * T + 13, SR, True, 0, 8)
    outs = {}
    for split in (1, 0):
        b = gpu.Bank.from_graph(g, V, sample_rate=SR)
        b.set_seed(p["seed"])
        b.set_option("time_split", split)
        a = run_bank(b, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :]
        assert b.get_option("last_kernel") == (4 if split else 2)
        c = run_bank(b, None, T, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :]
        d = run_bank(b, None, 13, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :]
        assert b.get_option("last_kernel") in (1, 2)
        outs[split] = np.concatenate([a, c, d], axis=1)
    assert_bit_equal(outs[1], want, "run-time compiled FM voice, time split vs oracle")
    assert_bit_equal(outs[0], want, "run-time compiled FM voice, pipeline vs oracle")
    # ... and the fused mix-down of the same small bank takes the time-split kernels too (MIX_PAN: every voice panned and summed in the launch)
    import torch
    from fundsp_amd import MIX_PAN

    pan = np.linspace(-1, 1, V).astype(np.float32)
    bm = gpu.Bank.from_graph(g, V, sample_rate=SR)
    bm.set_seed(p["seed"])
    bm.set_pan(pan)
    mix = bm.process_mix(T, mix=MIX_PAN)
    assert bm.get_option("last_kernel") == 4
    vo = torch.from_numpy(np.ascontiguousarray(outs[1][:, :T].T)).cuda()                  # [T][V] voice-out of the first launch
    assert_bit_equal(mix.cpu().numpy(), gpu.mix_stereo(vo, torch.from_numpy(pan).cuda()).cpu().numpy(), "run-time compiled FM voice: time-split fused mix vs mix_stereo(voice-out)")
    # two voice groups per CU (302 groups on 256 CUs) and the noise >> biquad shape
    V2 = 64 * 301 + 5
    nb = GR.noise() >> GR.biquad(-1.2, 0.5, 0.2, 0.3, 0.1)
    res = []
    for split in (1, 0):
        b = gpu.Bank.from_graph(nb, V2, sample_rate=SR)
        b.set_seed(np.arange(V2, dtype=np.uint64))
        b.set_option("time_split", split)
        res.append(run_bank(b, None, 64 * 3, LAYOUT_VOICE_MINOR, MODE_PROCESS)[:, 0, :])
        assert b.get_option("last_kernel") == 4 if split else b.get_option("last_kernel") in (1, 2)   # (a light graph: single wave below four blocks)
    assert_bit_equal(res[0], res[1], "noise >> biquad, 302 voice groups: time split == the other kernel families")
    n = O.noise() >> O.biquad(-1.2, 0.5, 0.2, 0.3, 0.1)
    n.set_sample_rate(SR)
    n.set_seed(V2 - 1)
    assert_bit_equal(res[0][V2 - 1], n.render_blocks(length=64 * 3)[0], "noise >> biquad last voice vs oracle")
