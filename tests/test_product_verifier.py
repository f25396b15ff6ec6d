"""`lmn_verify` (host-side verifier of the product, crates/verifiers/rust/src/verifier.rs:21-143) against
the reference's known-answer proof, oracle-made proofs and the oracle verifier on tampered proofs.
Host code only: runs through the test-only emulation build here and through the HIP library on the GPU box."""
import os
import subprocess

import numpy as np
import pytest

from luminair_amd import backend, synthetic as syn
from oracle.channel import ProtocolVariant
from oracle.proof import from_bincode, to_bincode
from oracle.prover import prove
from oracle.verifier import verify as oracle_verify


@pytest.fixture(scope="module")
def lib(root):
    so = os.path.join(root, "tests", "emu", "libluminair_emu.so")
    if not os.path.exists(so):
        subprocess.run([os.path.join(root, "tests", "emu", "build_emu.sh")], check=True, capture_output=True)
    return backend.Library(so)


def test_product_verifier_accepts_the_reference_kat(lib, kat_bytes):
    lib.verify(kat_bytes, backend.VARIANT_KAT)
    with pytest.raises(backend.LuminairBackendError):
        lib.verify(kat_bytes, backend.VARIANT_PINNED)        # other claim layout / encodings
    with pytest.raises(backend.LuminairBackendError) as e:
        lib.verify(kat_bytes[:-3], backend.VARIANT_KAT)
    assert e.value.code == backend.ERR_SERIALIZATION


@pytest.mark.parametrize("tabs,variant", [
    (syn.chain_graph(300, 3), 0), (syn.config3_mixed(10, 9, 9, 8), 0), (syn.linear_layer(20, 7, 2, True), 0),
    (syn.config2_graph_faithful(100, 3), 1), (syn.less_than_graph(100, 3), 1),
])
def test_product_verifier_accepts_oracle_proofs(lib, tabs, variant):
    b = to_bincode(prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=ProtocolVariant(variant)))
    lib.verify(b, variant)


def test_product_verifier_rejects_unbalanced_logup(lib):
    tabs = syn.config2_graph_faithful(32, 3)[:1]
    b = to_bincode(prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=ProtocolVariant.PINNED))
    with pytest.raises(backend.LuminairBackendError) as e:
        lib.verify(b, backend.VARIANT_PINNED)
    assert e.value.code == backend.ERR_INVALID_LOGUP


def test_product_verifier_agrees_with_oracle_verifier_on_tampered_proofs(lib, kat_bytes):
    """Flip one bit at a time over the proof body (skipping the PCS-config words, which both verifiers
    merely range-check): accept/reject decisions must agree, and every flip must be rejected."""
    n_rej = 0
    for off in list(range(0, 46, 5)) + list(range(70, len(kat_bytes), 41)):
        b = bytearray(kat_bytes)
        b[off] ^= 0x04
        try:
            lib.verify(bytes(b), backend.VARIANT_KAT)
            prod_ok = True
        except backend.LuminairBackendError:
            prod_ok = False
        try:
            oracle_verify(from_bincode(bytes(b), 8))
            ora_ok = True
        except Exception:
            ora_ok = False
        assert prod_ok == ora_ok, off
        n_rej += not prod_ok
    assert n_rej > 100


def test_product_verifier_survives_fuzzed_proofs(hip_lib_path, kat_bytes):
    """Bit flips, truncations, overwritten length fields and inserted bytes: always a clean rejection
    (LuminairBackendError), never a crash, never an acceptance."""
    import random
    from luminair_amd import backend
    lib = backend.Library(hip_lib_path)
    rnd = random.Random(7)
    for it in range(800):
        b = bytearray(kat_bytes)
        mode = it % 4
        if mode == 0:
            for _ in range(rnd.randint(1, 4)):
                b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        elif mode == 1:
            b = b[:rnd.randrange(len(b))]
        elif mode == 2:
            i = rnd.randrange(len(b) - 8)
            b[i:i + 8] = rnd.getrandbits(64).to_bytes(8, "little")
        else:
            i = rnd.randrange(len(b))
            b[i:i] = bytes(rnd.getrandbits(8) for _ in range(rnd.randint(1, 64)))
        with pytest.raises(backend.LuminairBackendError):
            lib.verify(bytes(b), rnd.choice([0, 0, 0, 1]))
