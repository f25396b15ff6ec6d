"""The C-ABI library loads and exports every symbol include/luminair_hip.h declares (no compute
without a GPU) and refuses to run without a HIP device."""
import os
import re

import pytest

from conftest import HAS_GPU


def test_header_symbols_are_exported(root, hip_lib_path):
    import ctypes
    from luminair_amd import backend
    hdr = open(os.path.join(root, "include", "luminair_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(lmn_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    lib = ctypes.CDLL(hip_lib_path)
    for name in declared:
        assert hasattr(lib, name), "symbol %s declared in the header but not exported" % name
    assert sorted(backend.EXPORTS) == declared
    blib = backend.Library(hip_lib_path)
    assert [blib.kind_columns(k) for k in range(18)] == [15, 16, 13, 12, 1, 14, 15, 13, 16, 12, 1, 12, 1, 22, 1, 7, 11, 0]
    cfg = blib.default_config()
    assert (cfg.pow_bits, cfg.log_blowup, cfg.log_last_layer, cfg.n_queries) == (5, 1, 0, 3)


@pytest.mark.skipif(HAS_GPU, reason="checks the no-GPU failure mode")
def test_no_device_fails_loudly(hip_lib_path):
    import luminair_amd
    with pytest.raises(luminair_amd.LuminairError) as e:
        luminair_amd.Prover(0)
    assert e.value.variant == "NoDevice"


def test_product_sources_do_not_touch_the_oracle(root):
    """The product path must never import/link anything under oracle/ or tests/emu."""
    pkg = os.path.join(root, "luminair_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "libluminair_emu" not in src, f


def test_settings_bincode_matches_the_reference_fixture(root):
    """`ui/demo/public/settings` (KAT era: a single `None` for `lookups.sin`) round-trips through the mirror."""
    import luminair_amd
    data = open(os.path.join(root, "tests", "golden", "kat_simple", "settings"), "rb").read()
    assert luminair_amd.CircuitSettings().to_bincode(kat_era=True) == data
    assert luminair_amd.CircuitSettings.from_bincode(data).lookups is None
    assert luminair_amd.CircuitSettings().to_bincode() == bytes(4)
    with pytest.raises(luminair_amd.LuminairError):
        luminair_amd.CircuitSettings.from_bincode(b"\x01")
