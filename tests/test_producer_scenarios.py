"""The reference's producer edge cases on the test-only emulation build (CPU); tests/test_gpu_parity.py runs the same
scenarios on the MI355X.  See tests/producer_scenarios.py."""
import os
import subprocess

import pytest

from luminair_amd import backend

import producer_scenarios as ps


@pytest.fixture(scope="module")
def lib(root):
    so = os.path.join(root, "tests", "emu", "libluminair_emu.so")
    if not os.path.exists(so):
        subprocess.run([os.path.join(root, "tests", "emu", "build_emu.sh")], check=True, capture_output=True)
    return backend.Library(so)


@pytest.mark.parametrize("build", ps.EXPANSIONS, ids=lambda f: f.__name__)
def test_expansion_scenarios(lib, build):
    ps.run_scenario(lib, build, 42 + ps.EXPANSIONS.index(build))


# the 32x32 shapes and the wide LUT cases are left to the GPU run (the emulation runs one fibre per GPU thread)
_CPU_OPS = [f for f in ps.OPS if "32x32" not in f.__name__]


@pytest.mark.parametrize("build", _CPU_OPS, ids=lambda f: f.__name__)
def test_op_shape_matrix(lib, build):
    ps.run_scenario(lib, build, 7)
