"""Run-time compiled kinds must come out of the compiler the library was built with (include/fundsp_hip.h fdsp_jit_compiler, fd_jit.hip "which
compiler compiles the graphs").  In a Python process PyTorch's wheel has already loaded ITS bundled libhiprtc / libamd_comgr (another ROCm release,
same sonames) when the library arrives, and plain linkage hands the library that copy.  How it was found: a fuzzer rendered `(noise() ^ impulse()) + c`
into PLANAR rows of an odd length (the single-wave planar kernel; aligned rows take the planar pipeline) and the impulse's first frame was missing --
the bundled compiler allocates the Impulse's value and a dead word of the node in front of it to one register in exactly that kernel variant
(profiles/r06_jit_compiler_miscompile.txt holds both compilers' code).  The library now isolates its own ROCm's hiprtc in a link-map namespace."""
import numpy as np
import pytest

import oracle as O
from fundsp_amd import LAYOUT_PLANAR, MODE_PROCESS, MODE_TICK
from fundsp_amd import graph as GR
from test_gpu_parity import assert_bit_equal, oracle_render

SR = 48000.0
GRAPHS = {
    "(mls ^ impulse) + c": lambda m: (m.mls() ^ m.impulse()) + (-0.1115),
    "(noise | impulse) + c": lambda m: (m.noise() | m.impulse()) + (-0.1115),
    "-(noise ^ impulse)": lambda m: -(m.noise() ^ m.impulse()),
    "(noise ^ impulse) * c": lambda m: (m.noise() ^ m.impulse()) * 0.5,
    "1.3 - (mls * c | -impulse)": lambda m: 1.3249953985214233 - ((m.mls() >> m.mul(-1.0605363845825195)) | -m.impulse()),   # the second graph the fuzzer tripped on
}


def test_the_library_names_its_compiler():
    """no GPU needed: the compiler the library will use is this ROCm's, isolated when the process holds another one"""
    import fundsp_amd as F

    who = F.lib().fdsp_jit_compiler().decode()
    assert who.startswith(("linked: ", "isolated: ")) and "libhiprtc" in who, who
    if "torch/lib/libhiprtc" in who:
        assert who.startswith("isolated: ") and "torch" not in who.split(" (the process's own is")[0], who
    assert F.lib().fdsp_graph_check(b"Unop<Branch<Mls, Impulse<1>>, UAddScalar>") == 0   # ... and it compiles (no device involved)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(GRAPHS))
def test_single_wave_planar_kernel_of_an_impulse_behind_a_hashed_generator(gpu, name):
    """planar rows of odd lengths, one-block launches, "pipe_split" 0: every route into the single-wave planar kernel, both executors"""
    import torch

    V = 6
    seeds = np.arange(V, dtype=np.uint64) * 17 + 3
    for frames, stride, split in ((275, 275, 1), (8, 8, 1), (13, 13, 1), (275, 384, 0), (581, 581, 1)):
        for mode in (MODE_PROCESS, MODE_TICK):
            b = gpu.Bank.from_graph(GRAPHS[name](GR), V, sample_rate=SR)
            b.set_option("pipe_split", split)
            b.set_seed(seeds)
            out = b.process(frames, layout=LAYOUT_PLANAR, frame_stride=stride, mode=mode)
            torch.cuda.synchronize()
            assert b.get_option("last_kernel") == 1
            got = out.cpu().numpy()[:, :, :frames]
            for v in (0, V - 1):
                n = GRAPHS[name](O)
                n.set_sample_rate(SR)
                n.set_seed(int(seeds[v]))
                assert_bit_equal(got[v], oracle_render(n, None, frames, mode), f"{name} frames {frames} stride {stride} pipe_split {split} mode {mode} instance {v}")
