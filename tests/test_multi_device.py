"""One process, several threads, several GPUs, and the in-library RCCL all-reduce of the stereo mix-down -- through the C
ABI from a plain C++ host (tests/host/test_multi_device.cpp).  Host mode here (no device: entry points, argument checks,
FDSP_EDEVICE from two threads); device mode on the GPU box (threads x devices x RCCL; a one-GPU box runs two threads on
device 0 and a one-rank communicator)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "test_multi_device.cpp")
BIN = os.path.join(ROOT, "tests", "host", "_build", "test_multi_device")


def build():
    lib = os.path.join(ROOT, "fundsp_amd", "libfundsp_hip.so")
    assert os.path.exists(lib), "build the HIP engine first (__graft_entry__.build())"
    deps = [SRC, os.path.join(ROOT, "include", "fundsp_hip.h"), lib]
    if os.path.exists(BIN) and all(os.path.getmtime(d) <= os.path.getmtime(BIN) for d in deps):
        return BIN
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    subprocess.check_call([
        "g++", "-std=c++17", "-O1", "-Wall", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
        "-I/opt/rocm/include", SRC, "-o", BIN, "-L" + os.path.join(ROOT, "fundsp_amd"), "-lfundsp_hip", "-L/opt/rocm/lib",
        "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "fundsp_amd"), "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return BIN


def run(mode):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([build(), mode], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failure(s)" in r.stdout
    return r.stdout


def test_multi_device_entry_points_without_a_device():
    run("--host")


@pytest.mark.gpu
def test_threads_devices_and_rccl_mix_allreduce():
    print(run("--gpu"))


@pytest.mark.gpu
def test_comm_from_python_single_rank(gpu):
    """fundsp_amd.Comm: rank communicator of one rank through the unique-id path, all-reduce on the side stream ordered
    behind the mix kernel, identity result; a second render overlaps it."""
    import numpy as np
    import torch

    from fundsp_amd import workloads as W

    V, T = 4096, 4096
    b = W.make_fm_svf_bank(V, 48000.0)
    comm = gpu.Comm.rank(gpu.Comm.unique_id(), 1, 0)
    assert comm.ranks() == 1
    out = b.process(T)
    mix = gpu.mix_stereo(out[0])
    want = mix.clone()
    torch.cuda.synchronize()
    for _ in range(3):
        out = b.process(T)                 # next render ...
        mix = gpu.mix_stereo(out[0])
        ref = mix.clone()
        comm.allreduce(mix)                # ... collective on the side stream, the render stream is not blocked
        out2 = b.process(T)
        comm.wait(stream="current")
        torch.cuda.synchronize()
        assert torch.equal(mix, ref)
    assert not torch.equal(want, mix)      # different seconds of audio
    comm.close()
