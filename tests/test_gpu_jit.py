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
}


@pytest.mark.parametrize("name", list(GRAPHS))
def test_jit_graph_matches_oracle(gpu, name):
    build, ni, ring = GRAPHS[name]
    g = build(GR)
    V, T = 70, 64 * 9 + 11
    if "saw" in name:
        t = O.Wavetable.get("saw")
        offs = np.concatenate([[0], np.cumsum(t.lengths)])
        gpu.wavetable_upload("saw", t.pitches, [t.data[offs[i]:offs[i + 1]] for i in range(len(t.lengths))])
    seeds = np.arange(V, dtype=np.uint64) * 7919 + 13
    x = None
    if ni:
        x = noise_input(V, ni, T, seed=77)
        if name == "saw_filter_env":
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
