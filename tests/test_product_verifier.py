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
    (syn.config2_graph_faithful(100, 3), backend.VARIANT_PINNED), (syn.less_than_graph(100, 3), backend.VARIANT_PINNED),
    (syn.chain_graph(100, 5), backend.PV_MIX_U64_HASHED | backend.PV_POW_PREFIXED | backend.PV_MUL_ONE_SLOT | backend.PV_RECIP_NEG),
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


def test_verifier_owns_the_security_parameters(lib, kat_bytes):
    """pow_bits / n_queries / log_last_layer are the verifier's, not the proof's.  Rewriting the
    config words of a valid proof (a 'downgrade' a forger would ship together with a matching cheap proof) is
    rejected by the product verifier and by the oracle verifier alike; a proof honestly made for another config is
    accepted only by a verifier that expects that config."""
    import struct
    off = (8 + 2 * 4) + (8 + 2 * 16)          # 8 claim slots (2 Some(u32)) + 8 interaction slots (2 Some(QM31)) of the KAT
    assert struct.unpack_from("<IIIQ", kat_bytes, off) == (5, 1, 0, 3)
    for field, val in ((0, 0), (0, 4), (2, 1), (3, 1), (3, 2)):
        b = bytearray(kat_bytes)
        if field == 3:
            struct.pack_into("<Q", b, off + 12, val)
        else:
            struct.pack_into("<I", b, off + 4 * field, val)
        with pytest.raises(backend.LuminairBackendError) as e:
            lib.verify(bytes(b), backend.VARIANT_KAT)
        assert e.value.code == backend.ERR_VERIFICATION
        with pytest.raises(Exception):
            oracle_verify(from_bincode(bytes(b), 8))
    # a proof made for 2 queries / 3 pow bits: rejected by the default verifier, accepted by one configured for it
    cfg = lib.default_config()
    cfg.n_queries, cfg.pow_bits = 2, 3
    tabs = syn.chain_graph(40, 2)
    from oracle.prover import PcsConfig
    p = to_bincode(prove([(k, r.astype(np.uint64)) for k, r in tabs], PcsConfig(pow_bits=3, n_queries=2)))
    with pytest.raises(backend.LuminairBackendError):
        lib.verify(p, backend.VARIANT_KAT)
    lib.verify(p, config=cfg)
    oracle_verify(from_bincode(p, 8), config=(3, 1, 0, 2))
    with pytest.raises(Exception):
        oracle_verify(from_bincode(p, 8))


def test_verifier_rejects_non_canonical_encodings(lib, kat_bytes):
    """field words >= 2^31-1, Option tags other than 0/1 and absurd claim sizes are
    serialization errors before any arithmetic happens."""
    import struct
    P = (1 << 31) - 1
    # first interaction claim's first coordinate: value v and v + P would be the same field element
    off = 8 * 1 + 2 * 4 + 1                        # claims: 8 tags + 2 u32; then the first Some tag of the interaction claims
    assert kat_bytes[off - 1] == 1
    v = struct.unpack_from("<I", kat_bytes, off)[0]
    for bad in (v + P, P, 0xffffffff):
        if bad > 0xffffffff:
            continue
        b = bytearray(kat_bytes)
        struct.pack_into("<I", b, off, bad)
        with pytest.raises(backend.LuminairBackendError) as e:
            lib.verify(bytes(b), backend.VARIANT_KAT)
        assert e.value.code == backend.ERR_SERIALIZATION
        with pytest.raises(ValueError):
            from_bincode(bytes(b), 8)
    b = bytearray(kat_bytes)
    b[0] = 2                                        # Option tag of the first claim slot
    with pytest.raises(backend.LuminairBackendError) as e:
        lib.verify(bytes(b), backend.VARIANT_KAT)
    assert e.value.code == backend.ERR_SERIALIZATION
    b = bytearray(kat_bytes)
    struct.pack_into("<I", b, 1, 0x80000004)        # claim log_size that a signed cast would turn negative
    with pytest.raises(backend.LuminairBackendError) as e:
        lib.verify(bytes(b), backend.VARIANT_KAT)
    assert e.value.code == backend.ERR_SERIALIZATION


def test_verifier_cross_checks_settings(lib):
    """`settings` is no longer ignored - lookups announced by the settings must be the ones the
    proof's claim carries, LUT sizes included."""
    import ctypes as C
    tabs, luts = syn.activation_graph(30, 4, names=("sin",))
    p = to_bincode(prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=ProtocolVariant.PINNED, luts=luts))
    c0 = np.ascontiguousarray(luts["sin"][0], dtype=np.uint32)
    c1 = np.ascontiguousarray(luts["sin"][1], dtype=np.uint32)
    log = len(c0).bit_length() - 1
    arr = (backend.LmnLut * 1)(backend.LmnLut(0, log, c0.ctypes.data, c1.ctypes.data))
    lib.verify(p, backend.VARIANT_PINNED, settings=backend.LmnSettings(1, 1, arr))
    lib.verify(p, backend.VARIANT_PINNED, settings=backend.LmnSettings(0, 0, None))
    with pytest.raises(backend.LuminairBackendError):                       # settings announce exp2, proof has sin
        lib.verify(p, backend.VARIANT_PINNED, settings=backend.LmnSettings(2, 0, None))
    arr2 = (backend.LmnLut * 1)(backend.LmnLut(0, log + 1, c0.ctypes.data, c1.ctypes.data))
    with pytest.raises(backend.LuminairBackendError):                       # LUT of another size
        lib.verify(p, backend.VARIANT_PINNED, settings=backend.LmnSettings(1, 1, arr2))
