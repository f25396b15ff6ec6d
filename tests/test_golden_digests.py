"""Committed golden digests (tests/golden/proof_digests.json, made by tests/golden/make_golden.py)."""
import hashlib
import importlib.util
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)
GOLD = json.load(open(os.path.join(HERE, "golden", "proof_digests.json")))


@pytest.mark.parametrize("name", sorted(GOLD))
def test_oracle_reproduces_golden_digest(name):
    assert mg.digest(name) == GOLD[name]


def test_kat_digest_is_the_reference_file(kat_bytes):
    assert GOLD["simple_example/kat"]["sha256"] == hashlib.sha256(kat_bytes).hexdigest()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(GOLD))
def test_gpu_reproduces_golden_digest(name, hip_lib_path):
    import luminair_amd
    from luminair_amd import backend
    variant = mg.CASES[name][1]
    tabs, luts = mg.tables_and_luts(name)
    prover = luminair_amd.Prover(0, protocol_variant=backend.VARIANT_PINNED if variant == "PINNED" else backend.VARIANT_KAT)
    b = prover.prove(luminair_amd.LuminairPie.from_tables(tabs), luminair_amd.CircuitSettings(luts)).to_bincode()
    assert {"sha256": hashlib.sha256(b).hexdigest(), "len": len(b)} == GOLD[name]
