import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAS_GPU = _has_gpu()


@pytest.fixture(scope="session")
def root():
    return ROOT


@pytest.fixture(scope="session")
def kat_bytes():
    with open(os.path.join(ROOT, "tests", "golden", "kat_simple", "proof"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def hip_lib_path():
    p = os.path.join(ROOT, "luminair_amd", "csrc", "libluminair_hip.so")
    if not os.path.exists(p):
        import __graft_entry__
        __graft_entry__.build()
    return p


@pytest.fixture(scope="session")
def gpu_prover(hip_lib_path):
    import luminair_amd
    return luminair_amd.Prover(0)
