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
