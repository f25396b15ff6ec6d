"""The plain-C restatement (oracle/c) against the numpy restatement and the reference's KAT."""
import numpy as np
import pytest

from luminair_amd import synthetic as syn
from oracle.channel import ProtocolVariant
from oracle.proof import to_bincode
from oracle.prover import prove


@pytest.fixture(scope="module")
def ck():
    from oracle.cbackend import CKernels
    return CKernels()


def test_c_oracle_reproduces_kat(ck, kat_bytes):
    assert to_bincode(prove(syn.simple_example(), kernels=ck)) == kat_bytes


@pytest.mark.parametrize("name,tabs,variant", [
    ("chain-300", syn.chain_graph(300, 3), ProtocolVariant.KAT),
    ("single-row", syn.chain_graph(1, 6), ProtocolVariant.KAT),
    ("mixed-sizes", [(0, syn.chain_graph(64, 4)[0][1]), (1, syn.chain_graph(500, 5)[1][1])], ProtocolVariant.KAT),
    ("add-2^13", syn.config2_add_only(1 << 13, 6), ProtocolVariant.KAT),
    ("2b-inputs", syn.config2_graph_faithful(200, 7), ProtocolVariant.PINNED),
    ("linear-layer", syn.linear_layer(8, 16, 1), ProtocolVariant.KAT),
    ("linear-layer+max", syn.linear_layer(20, 7, 2, True), ProtocolVariant.KAT),
    ("less-than+range-check-lut", syn.less_than_graph(100, 3), ProtocolVariant.PINNED),
    ("sqrt+rem", syn.sqrt_rem_graph(50, 4), ProtocolVariant.PINNED),
])
def test_c_oracle_equals_numpy_oracle(ck, name, tabs, variant):
    a = to_bincode(prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=variant))
    b = to_bincode(prove(tabs, variant=variant, kernels=ck))
    assert a == b, name


def test_c_oracle_equals_numpy_oracle_on_lut_activations(ck):
    tabs, luts = syn.activation_graph(70, 5)
    a = to_bincode(prove(tabs, variant=ProtocolVariant.PINNED, luts=luts))
    assert a == to_bincode(prove(tabs, variant=ProtocolVariant.PINNED, kernels=ck, luts=luts))


def test_c_oracle_config4_black_scholes_shape_verifies(ck):
    from oracle.verifier import verify
    tabs, luts = syn.config4_black_scholes_shape()
    assert [k for k, _ in tabs] == [0, 1, 2, 5, 9, 10]
    verify(prove(tabs, variant=ProtocolVariant.PINNED, kernels=ck, luts=luts), ProtocolVariant.PINNED)


def test_c_kernels_individually(ck):
    from oracle import fft
    from oracle.field import P, QM31
    from oracle.merkle import MerkleTree
    rng = np.random.default_rng(9)
    ev = rng.integers(0, P, size=(3, 1 << 9), dtype=np.uint64)
    co = ck.interpolate_cols(list(ev))
    assert np.array_equal(np.stack(co), fft.interpolate(ev).astype(np.uint32))
    lde = ck.lde(co, [9, 9, 9], 1)
    assert np.array_equal(np.stack(lde), fft.evaluate(np.stack(co).astype(np.uint64), 10).astype(np.uint32))
    cols = [rng.integers(0, P, size=1 << k, dtype=np.uint64).astype(np.uint32) for k in (8, 8, 5, 8, 2)]
    assert ck.merkle(cols).root() == MerkleTree(cols).root()
    pt = (QM31(*[int(v) for v in rng.integers(0, P, size=4)]), QM31(*[int(v) for v in rng.integers(0, P, size=4)]))
    c = rng.integers(0, P, size=1 << 11, dtype=np.uint64)
    assert ck.eval_at_point(c, pt) == fft.eval_at_point(c, pt)


def test_scalar_build_of_the_c_oracle_gives_the_same_bytes(ck, kat_bytes):
    """liboracle_kernels_scalar.so (no lanes, vectoriser off: cpu_baseline's "port-scalar") against the 16-lane build,
    at sizes where the chunked batch inversions and the row-at-a-time tails both run."""
    from oracle.cbackend import CKernels
    ks = CKernels(scalar=True)
    assert ks.lib._name.endswith("liboracle_kernels_scalar.so")
    assert to_bincode(prove(syn.simple_example(), kernels=ks)) == kat_bytes
    for tabs, variant in ((syn.chain_graph(3000, 3), ProtocolVariant.KAT), (syn.sqrt_rem_graph(700, 4), ProtocolVariant.PINNED)):
        assert to_bincode(prove(tabs, variant=variant, kernels=ks)) == to_bincode(prove(tabs, variant=variant, kernels=ck))
