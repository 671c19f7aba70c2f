import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build them once, the way __graft_entry__.build()
    does, before any test needs libfundsp_hip.so or the oracle.  (The product itself never builds or falls back.)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "fundsp_amd", "libfundsp_hip.so")):
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(root, "fundsp_amd", "csrc"), "-j", str(min(8, os.cpu_count() or 1))])


@pytest.fixture(scope="session")
def oracle():
    import oracle as O

    O.lib()
    return O


def has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """The HIP engine on cuda:0.  GPU tests must fail loudly (not skip) when the extension is missing."""
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import fundsp_amd

    fundsp_amd.lib()  # raises if libfundsp_hip.so is missing: no fallback
    torch.cuda.set_device(0)
    return fundsp_amd
