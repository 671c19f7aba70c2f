"""Host-side checks of the product library (no GPU needed): libfundsp_hip.so loads, exports every symbol
include/fundsp_hip.h declares, introspects its voice-graph kinds, and its host-callable coefficient constructors
agree bit-for-bit with the oracle.  No compute kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def F():
    import fundsp_amd

    fundsp_amd.lib()
    return fundsp_amd


def test_exports_every_declared_symbol(F):
    header = open(os.path.join(ROOT, "include", "fundsp_hip.h")).read()
    declared = set(re.findall(r"\b(fdsp_[a-z0-9_]+)\s*\(", header))
    declared -= {"fdsp_bank"}  # the opaque struct tag
    assert len(declared) >= 30
    raw = C.CDLL(F._lib.SO_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), f"{name} is declared in include/fundsp_hip.h but not exported"
    assert declared == set(F._lib.SYMBOLS), declared ^ set(F._lib.SYMBOLS)


def test_kinds_and_slots(F):
    kinds = F.kinds()
    for k in ("sine", "noise", "fixed_svf", "svf3", "svf4", "biquad", "biquad_bank", "moog", "moog_hz", "fir3",
              "sine_hz_lowpass_hz", "noise_biquad", "fm_svf"):
        assert k in kinds
    slots = dict(F.kind_slots("fm_svf"))
    from fundsp_amd import workloads as W

    for name in W.FM_SLOTS.values():
        assert name in slots, name
    assert slots["1:cutoff"] == 0 and slots["1:a1"] == 1 and slots["1:ic1eq"] == 2  # param / coef / state
    assert F.lib().fdsp_kind_by_name(b"nope") == -1
    L = F.lib()
    k = L.fdsp_kind_by_name(b"svf4")
    assert L.fdsp_kind_inputs(k) == 4 and L.fdsp_kind_outputs(k) == 1


def test_host_coefficient_constructors_match_oracle(F):
    rng = np.random.default_rng(0)
    for _ in range(300):
        f = float(np.float32(20.0 * 1000.0 ** rng.random()))
        q = float(np.float32(0.3 + 9 * rng.random()))
        g = float(np.float32(0.2 + 4 * rng.random()))
        sr = float(rng.choice([44100.0, 48000.0, 96000.0]))
        f = min(f, 0.49 * sr)
        for kind in O.BQ_KINDS:
            assert np.array_equal(F.biquad_coefs(kind, sr, f, q, g).view(np.uint32),
                                  O.biquad_coefs(kind, sr, f, q, g).view(np.uint32)), kind
        for mode in O.SVF_MODES:
            assert np.array_equal(F.svf_coefs(mode, sr, f, q, g).view(np.uint32),
                                  O.svf_coefs(mode, sr, f, q, g).view(np.uint32)), mode
    L = F.lib()
    for x in (0, 1, 77, 2**63 + 5):
        assert L.fdsp_rnd1(x) == O.lib().o_math_rnd1(x)
        assert L.fdsp_hash1(x) == O.lib().o_math_hash1(x)


def test_no_cpu_fallback(F):
    """Without a HIP device the product path must fail loudly, never compute on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device error path cannot be exercised")
    with pytest.raises(F.FdspError) as e:
        F.Bank("fm_svf", 64)
    assert e.value.code == F._lib.EDEVICE and "no CPU fallback" in str(e.value)


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under fundsp_amd/ or include/ may import, include or link it."""
    bad = []
    for base in ("fundsp_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"fundsp_oracle|o_math\.h|libfundsp_oracle|import oracle|from oracle", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_graph_notation_lowers_to_reference_types_and_compiles(F):
    """The host graph notation builds the combinator TYPE FunDSP's operators would build (combinator.rs:289-488, Rust
    precedence `*` > `+` > `>>` > `|`), and hiprtc accepts it (compile only: no device needed)."""
    from fundsp_amd import graph as G

    g = G.sine_hz(440.0) * 440.0 * 2.0 + 440.0 >> G.sine() >> G.lowpass_hz(1000.0, 1.0)
    assert g.type == "Pipe<Pipe<Unop<Unop<Unop<Pipe<Constant<1>,Sine>,UMulScalar>,UMulScalar>,UAddScalar>,Sine>,FixedSvf>"
    assert (g.nin, g.nout) == (0, 1)
    names = [n for n, _, _ in g.slot_values()]
    assert names[:4] == ["0.0.0.0.0.0:value[0]", "0.0.0.0:scalar", "0.0.0:scalar", "0.0:scalar"]
    aot = dict(F.kind_slots("fm_svf"))
    assert all(n in aot for n in names)                      # same slot addressing as the ahead-of-time kind
    c4 = ((G.dc(110.0) >> G.saw() | G.dc(800.0) | G.dc(0.3)) >> G.moog()) * G.adsr_live(.01, .1, .6, .2) >> G.pan(0.2)
    assert c4.type == ("Pipe<Binop<OpMul,Pipe<Stack<Stack<Pipe<Constant<1>,WaveSynth<0>>,Constant<1>>,Constant<1>>,Moog<3>>,"
                       "AdsrLive>,Panner>") and (c4.nin, c4.nout) == (1, 2)
    L = F.lib()
    assert L.fdsp_graph_check(g.type.encode()) == 0
    assert L.fdsp_graph_check(b"Pipe<Noise,AllNest<Delay>>") == 0
    assert L.fdsp_graph_check(b"Pipe<Sine,Stack<Sine,Sine>>") < 0 and b"arity mismatch" in L.fdsp_last_error()
    with pytest.raises(TypeError):
        G.sine() >> (G.sine() | G.sine())


def test_rust_shim_declares_the_c_abi():
    """rust_shim/src/lib.rs (the reference-side binding, source only) stays in sync with include/fundsp_hip.h: every
    `pub fn fdsp_*` of its extern block exists in the header with the same number of parameters."""
    header = open(os.path.join(ROOT, "include", "fundsp_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    shim = open(os.path.join(ROOT, "rust_shim", "src", "lib.rs")).read()
    shim = re.sub(r"//[^\n]*", "", shim)
    decls = re.findall(r"pub fn (fdsp_[a-z0-9_]+)\s*\(([^)]*)\)", shim, flags=re.S)
    assert len(decls) >= 25
    for name, args in decls:
        m = re.search(r"\b" + name + r"\s*\(([^)]*)\)", header, flags=re.S)
        assert m, f"{name} is bound by rust_shim but not declared in include/fundsp_hip.h"
        n_rust = len([a for a in args.split(",") if a.strip()])
        c_args = m.group(1).strip()
        n_c = 0 if c_args in ("", "void") else len([a for a in c_args.split(",") if a.strip()])
        assert n_rust == n_c, f"{name}: {n_rust} parameters in rust_shim, {n_c} in the header"
    bound = {name for name, _ in decls}
    # the device-resident entries behind HipBank::render_device / render_mix / set_pan / allocate_mix
    for name in ("fdsp_bank_process", "fdsp_bank_process_mix", "fdsp_bank_set_pan", "fdsp_bank_mix_reserve", "fdsp_bank_synchronize", "fdsp_sum_voices"):
        assert name in bound, f"{name} is not bound by rust_shim"
    for method in ("pub fn render_device", "pub fn render_mix", "pub fn set_pan", "pub fn allocate_mix", "pub fn take_error"):
        assert method in shim, f"HipBank lacks `{method}`"
    # the infallible trait methods keep the engine's return code instead of dropping it
    assert shim.count("self.last_error = Some(last_error())") >= 4


def test_second_module_of_run_time_compiled_graphs_builds():
    """fdsp_graph_check compiles BOTH modules of a run-time compiled graph with hiprtc (no device needed): the main one and the lazily built
    second one (fused mix-down, time-split kernels).  Shapes that exercise its guards: a three-stage generator chain (time-split kernels
    instantiated), one with three outputs behind it (time-split mix-down refused by its guard, pipeline mix-down fits), four outputs (no mix
    tile fits: empty bodies), a graph with an input and rings (no pipeline plan for the mix), a feedback graph (flush-to-zero flags)."""
    import fundsp_amd

    L = fundsp_amd.lib()
    fm = "Pipe<Pipe<Unop<Pipe<Constant<1>,Sine>,UAddScalar>,Sine>,FixedSvf>"
    for expr in (fm, "Pipe<Noise,Biquad>", "Pipe<Noise,ButterLowpass<1>>",
                 f"Stack<Stack<{fm},{fm}>,{fm}>", f"Stack<Stack<{fm},{fm}>,Stack<{fm},{fm}>>",
                 "Pipe<Pass,FixedSvf>",
                 "Pipe<Pipe<Unop<Pipe<Constant<1>,Sine>,UAddScalar>,Sine>,Panner>"):      # a stereo generator chain: the time-split MIX_SUM kernels
        assert L.fdsp_graph_check(expr.encode()) == 0, (expr, L.fdsp_last_error())
