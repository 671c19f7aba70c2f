"""The kernel templates only run-time compiled kinds instantiate -- the branch-major wide-sum bodies (fd_device.hpp render_body_wide /
render_body_wide_chain) and nodes no ahead-of-time kind uses (Limiter's incremental reduce tree) -- never meet the compiler when the library is
built: hiprtc compiles them on the GPU box.  This cross-compiles them here (hipcc, gfx950, no GPU) with the flags fd_jit.hip hands hiprtc, so a header
edit that breaks them fails the CPU suite, and holds their register budgets (no spills: a spilling wide kernel is the defect the branch-major form
exists to remove)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fundsp_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "--cuda-device-only", "-S", "-I" + CSRC]
SIG = ("(float* __restrict__ slots, size_t stride, size_t V, const float* __restrict__ in, float* __restrict__ out, size_t T, size_t fstride, "
       "const void* aux, float* ring, uint32_t cap)")
ARGS = "(slots, stride, V, in, out, T, fstride, aux, ring, cap)"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")


def compile_kernels(tmp_path, type_expr, extra=()):
    src = ['#include "fd_device.hpp"', f"namespace fd {{ using JitG = {type_expr}; }}", "using fd::JitG;",
           "constexpr int JIT_WPB0 = fd::RenderGeom<JitG, 0>::WPB, JIT_WPB1 = fd::RenderGeom<JitG, 1>::WPB;"]
    for m in (0, 1):
        for l in (0, 1):
            src.append(f'extern "C" __global__ __launch_bounds__(64 * JIT_WPB{l}) void jit_render_{m}{l}{SIG} {{ '
                       f"fd::render_body<JitG, {m}, {l}, JIT_WPB{l}>{ARGS}; }}")
            src.append(f'extern "C" __global__ __launch_bounds__(64 * fd::WideChain<JitG>::W) void jit_wide_{m}{l}{SIG} {{ '
                       f"fd::render_body_wide_chain<JitG, {m}, {l}>{ARGS}; }}")
    f = tmp_path / "k.hip"
    f.write_text("\n".join(src) + "\n")
    out = tmp_path / "k.s"
    r = subprocess.run([HIPCC] + FLAGS + list(extra) + [str(f), "-o", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    asm = out.read_text()
    meta = {}
    for name, block in re.findall(r"\.amdhsa_kernel (jit_\w+)\n(.*?)\.end_amdhsa_kernel", asm, re.S):
        lds = int(re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", block).group(1))
        note = asm[asm.index(f".name:           {name}"):]   # the kernel's metadata entry: .name, .private_segment.., .sgpr.., .vgpr_count, .vgpr_spill_count
        meta[name] = dict(vgpr=int(re.search(r"\.vgpr_count:\s+(\d+)", note).group(1)), spill=int(re.search(r"\.vgpr_spill_count:\s+(\d+)", note).group(1)), lds=lds)
    return meta


ILP = ("-mllvm", "-amdgpu-sched-strategy=max-ilp")   # what fd_jit.hip adds for wide sums of plain feed-forward branches


def test_wide_sum_of_generators_compiles_without_spills(tmp_path):
    """the reference's own `sine` bench type: sumi::<U100>(sine_hz(..))"""
    meta = compile_kernels(tmp_path, "Reduce<100, Pipe<Constant<1>, Sine>, OpAdd>", ILP)
    assert set(meta) == {f"jit_{k}_{m}{l}" for k in ("render", "wide") for m in (0, 1) for l in (0, 1)}
    for name, k in meta.items():
        assert k["spill"] == 0, (name, k)
    assert meta["jit_wide_00"]["lds"] == 8 * 64 * 64 * 4 and meta["jit_wide_00"]["vgpr"] <= 256     # 8 waves per voice group: half the register file each
    assert meta["jit_render_00"]["lds"] == 4 * 64 * 64 * 4


def test_wide_sums_with_inputs_and_stereo_branches_compile_without_spills(tmp_path):
    for t in ("MultiBus<20, Pipe<Binop<OpMul, MultiPass<1>, Constant<1>>, Sine>>",       # busi(|i| mul(i + 1) >> sine()): a shared input
              "Reduce<8, FixedSvf, OpAdd>",                                                 # eight filters on their own inputs
              "MultiBus<10, Pipe<Resonator<1>, Panner>>"):                                  # 1 in, 2 out
        meta = compile_kernels(tmp_path, t)
        for name, k in meta.items():
            assert k["spill"] == 0, (t, name, k)
        assert meta["jit_wide_00"]["lds"] in (4 * 64 * 64 * 4, 4 * 2 * 64 * 64 * 4), (t, meta["jit_wide_00"])   # four-wave chains


def test_wide_sum_with_a_tail_compiles_without_spills(tmp_path):
    """sumi(..) * gain >> lowpass_hz(..) >> pan(..): the sum at the head of a Pipe / Unop spine, the rest of the graph walked as its tail (mono sum, stereo tile)"""
    meta = compile_kernels(tmp_path, "Pipe<Pipe<Unop<Reduce<12, Pipe<Constant<1>, Sine>, OpAdd>, UMulScalar>, FixedSvf>, Panner>", ILP)
    for name, k in meta.items():
        assert k["spill"] == 0, (name, k)
    assert meta["jit_wide_00"]["lds"] == (8 + 1) * 64 * 64 * 4   # a chain of eight mono tiles + the one tile only the tail's second channel needs


def test_limiter_graph_compiles_without_scratch(tmp_path):
    """noise() >> limiter(..): the incremental reduce tree keeps 2 x 20 path / sibling values in registers"""
    meta = compile_kernels(tmp_path, "Pipe<Noise, Limiter<1>>")
    for name in ("jit_render_00", "jit_render_01", "jit_render_10", "jit_render_11"):
        assert meta[name]["spill"] == 0, (name, meta[name])
