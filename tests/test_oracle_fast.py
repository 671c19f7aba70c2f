"""oracle/o_fast.c -- the monomorphised SIMD form of the config-3 process() path that bench.py times as `cpu_baseline`
(VERDICT r02 Weak 7 / Next 7) -- must render exactly what the generic tree-walking oracle renders: bit for bit, for every
block shape (full SIMD items, remainders, a ragged last block), voice-major and voice-minor output, several threads."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle as O
from fundsp_amd import workloads as W

SR = 48000.0


@pytest.mark.parametrize("frames", [64, 1, 7, 8, 13, 64 * 3 + 37, 441])
def test_fast_equals_tree_walk_bit_for_bit(frames):
    V = 24
    p = W.fm_svf_params(V, SR)
    args = (3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], frames, SR, True)
    for layout in (0, 1):
        want, _ = O.bank_render(*args, layout, 1)
        got, _ = O.bank_render(*args, layout, 3, fast=True)
        assert got.shape == want.shape
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"frames={frames} layout={layout}"


def test_fast_special_phases_and_huge_modulation():
    """Modulation indices that push the carrier phase through wide's round_int saturation and its q > 2^25 overflow rule,
    negative frequencies, denormal frequencies: the 8-lane sine is o_wide_sinf lane for lane there too."""
    V, frames = 16, 64 * 2 + 5
    p = W.fm_svf_params(V, SR)
    p["m"][1] = np.float32(3.0e6)
    p["m"][2] = np.float32(1.0e12)
    p["m"][3] = np.float32(3.0e30)
    p["f"][4] = np.float32(-440.0)
    p["f"][5] = np.float32(1e-41)
    p["f"][6] = np.float32(2.0e9)
    args = (3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], frames, SR, True)
    with np.errstate(all="ignore"):
        want, _ = O.bank_render(*args, 0, 1)
        got, _ = O.bank_render(*args, 0, 2, fast=True)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_native_build_is_bit_identical_too():
    """bench.py times the -O3 -march=native build (oracle/Makefile `native`): same bits as the portable build."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "native"])
    L = C.CDLL(os.path.join(root, "oracle", "_native", "libfundsp_oracle_native.so"))
    L.o_bank_render.restype = C.c_double
    L.o_bank_render.argtypes = [C.POINTER(O.BankJob), C.POINTER(C.c_float)]
    L.o_fast_simd_flavour.restype = C.c_char_p
    V, frames = 32, 64 * 4 + 3
    p = W.fm_svf_params(V, SR)
    args = (3, [p["f"], p["m"], p["fc"], p["q"]], p["seed"], frames, SR, True)
    want, _ = O.bank_render(*args, 1, 1)
    got, _ = O.bank_render(*args, 1, 4, lib=L, fast=True)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), L.o_fast_simd_flavour().decode()


# ---- configs 4 and 5: the monomorphised process() of the cpu_baseline legs == the tree walk == the graph the tests build ------------
ADSR = (0.005, 0.01, 0.6, 0.01)


def _c4_python_voice(p, v, adsr, var):
    gate = O.var(0.0) if var else None
    env = (gate >> O.adsr_live(*adsr)) if var else O.adsr_live(*adsr)
    g = (((O.dc(float(p["f"][v])) >> O.saw()) | O.dc(float(p["fc"][v])) | O.dc(float(p["q"][v]))) >> O.moog()) * env >> O.pan(float(p["pan"][v]))
    g.set_sample_rate(SR)
    g.set_seed(int(p["seed"][v]))
    return g, gate


@pytest.mark.parametrize("frames", [64 * 20 + 7, 64, 5])
def test_config4_stream_gate_fast_equals_tree_walk_and_the_python_graph(frames):
    V = 12
    p = W.saw_moog_params(V, SR)
    gate = W.gate_signal(frames, SR, on_frame=1, off_seconds=700 / SR)
    want, _ = O.c4_bank_render(p, ADSR, frames, SR, gate=gate, fast=False)
    got, _ = O.c4_bank_render(p, ADSR, frames, SR, gate=gate, threads=3, fast=True)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    for v in (0, 5, V - 1):
        g, _ = _c4_python_voice(p, v, ADSR, False)
        assert np.array_equal(g.render_blocks(gate[None, :]).view(np.uint32), want[v].view(np.uint32)), f"voice {v}"
    assert frames <= 64 or np.abs(want).max() > 0.05      # (the envelope's first ~2 ms segment is silent: the closure saw the gate low at t = 0)


def test_config4_var_gate_fast_equals_tree_walk_and_the_python_graph():
    V = 10
    plan = [(0.0, 64), (1.0, 64 * 9), (0.0, 64 * 3 + 7), (0.5, 64 * 5), (0.0, 64 * 6 + 3)]
    frames = sum(n for _, n in plan)
    p = W.saw_moog_params(V, SR)
    want, _ = O.c4_bank_render(p, ADSR, frames, SR, plan=plan, fast=False)
    got, _ = O.c4_bank_render(p, ADSR, frames, SR, plan=plan, threads=2, fast=True)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    for v in (0, 4, V - 1):
        g, gate = _c4_python_voice(p, v, ADSR, True)
        parts = []
        for value, n in plan:
            gate.set_value(value)
            parts.append(g.render_blocks(length=n))
        assert np.array_equal(np.concatenate(parts, axis=1).view(np.uint32), want[v].view(np.uint32)), f"voice {v}"
    assert np.abs(want).max() > 0.05


def test_var_process_is_a_block_constant_splat():
    """Var::process (shared.rs:122-125) reads the variable once per block; Var::tick per sample: same values while nobody writes"""
    n = O.var(0.25)
    a = n.render_blocks(length=77)
    n.set_value(-3.0)
    b = n.render_ticks(length=9)
    assert np.all(a == np.float32(0.25)) and np.all(b == np.float32(-3.0))


@pytest.mark.parametrize("frames", [64 * 30 + 11, 64])
def test_config5_block_form_equals_tree_walk(frames):
    rng = np.random.default_rng(5)
    x = (rng.random((2, frames), dtype=np.float32) * 2 - 1).astype(np.float32)
    x[:, frames // 2:] = 0.0            # a silent tail: the feedback decays into the flush-to-zero range
    want, _ = O.reverb_bank_render(3, x, SR, fast=False)
    got, _ = O.reverb_bank_render(3, x, SR, threads=2, fast=True)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    n = O.reverb_stereo(10.0, 2.0, 0.5)
    n.set_sample_rate(SR)
    assert np.array_equal(n.render_blocks(x).view(np.uint32), want[0].view(np.uint32))
    assert frames <= 64 or np.abs(want).max() > 0.01       # (the shortest delay line is longer than one block)


@pytest.mark.parametrize("V,frames", [(24, 64 * 5 + 13), (13, 64), (8, 7)])
def test_config2_biquad_bank_f32x8_equals_the_scalar_voices(V, frames):
    """The cpu_baseline leg of config 2 is the reference's own SIMD form, BiquadBank<f32x8> (eight voices per instruction,
    biquad_bank.rs:73-84): lane k of bank j == voice 8 j + k of the scalar restatement, bit for bit; ragged last bank, both layouts."""
    p = W.noise_biquad_params(V, SR)
    args = (2, [p["fc"], p["q"]], p["seed"], frames, SR, True)
    for layout in (0, 1):
        want, _ = O.bank_render(*args, layout, 1)
        got, _ = O.bank_render(*args, layout, 3, fast=True)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"V={V} frames={frames} layout={layout}"


def test_graph_bank_driver_equals_the_node_render():
    """o_graph_bank_render (the threaded CPU leg of bench.py's reverb3_stereo / fdn16 entries) renders what the same graph renders through the Node
    interface: every instance, three threads, a ragged length."""
    import oracle as O

    rng = np.random.default_rng(9)
    T = 64 * 30 + 7
    x2 = (rng.random((2, T), dtype=np.float32) * 2 - 1).astype(np.float32)
    out, secs = O.graph_bank_render("reverb3", (2.0, 0.5, 8000.0), 5, x2, threads=3)
    fast, _ = O.graph_bank_render("reverb3", (2.0, 0.5, 8000.0), 5, x2, threads=2, fast=True)      # the monomorphised block form
    assert np.array_equal(fast.view(np.uint32), out.view(np.uint32))
    n = O.reverb3_stereo(2.0, 0.5, lambda: O.lowpole_hz(8000.0))
    n.set_sample_rate(48000.0)
    want = n.render_blocks(x2)
    assert secs > 0 and all(np.array_equal(out[i].view(np.uint32), want.view(np.uint32)) for i in range(5))
    d = [float(np.float32(0.01 + 0.00125 * i)) for i in range(16)]
    out, _ = O.graph_bank_render("fdn16", d + [0.2, 0.4, 0.2], 4, x2[:1], threads=2)
    fast, _ = O.graph_bank_render("fdn16", d + [0.2, 0.4, 0.2], 4, x2[:1], threads=3, fast=True)
    assert np.array_equal(fast.view(np.uint32), out.view(np.uint32))
    n = O.split(16) >> O.fdn(O.stacki(16, lambda i: O.delay(np.float32(d[i])) >> O.fir(0.2, 0.4, 0.2))) >> O.join(16)
    n.set_sample_rate(48000.0)
    want = n.render_blocks(x2[:1])
    assert all(np.array_equal(out[i].view(np.uint32), want.view(np.uint32)) for i in range(4))
