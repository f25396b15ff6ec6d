import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "sanitize: the emulation suite under AddressSanitizer + UBSan / ThreadSanitizer builds of "
                                       "the host code (minutes; runs only when asked for: -m sanitize)")
    # LMN_EMU_SANITIZER=asan|tsan (set by tests/test_sanitizers.py for its pytest subprocesses, which it starts with the
    # sanitizer runtime preloaded): every test that loads tests/emu/libluminair_emu.so gets the sanitizer build instead
    san = os.environ.get("LMN_EMU_SANITIZER")
    if san:
        from luminair_amd import backend
        plain = os.path.join(ROOT, "tests", "emu", "libluminair_emu.so")
        built = os.path.join(ROOT, "tests", "emu", "libluminair_emu_%s.so" % san)
        orig = backend.Library.__init__

        def redirected(self, path=None):
            if path and os.path.abspath(path) == plain:
                path = built
            orig(self, path)
        backend.Library.__init__ = redirected


def pytest_collection_modifyitems(config, items):
    # the sanitizer runs take minutes and need nothing the plain CPU suite does not already cover functionally: they run
    # when the marker is asked for by name, and are skipped (visibly) otherwise
    if "sanitize" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="sanitizer builds: run with -m sanitize")
    for item in items:
        if "sanitize" in item.keywords:
            item.add_marker(skip)


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
    import __graft_entry__
    __graft_entry__.ensure_built()     # missing, or built by another hipcc than this box's (csrc/toolchain.stamp): rebuild
    return p


@pytest.fixture(scope="session")
def gpu_prover(hip_lib_path):
    import luminair_amd
    return luminair_amd.Prover(0)
