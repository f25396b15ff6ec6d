"""The oracle against the reference's only golden vector: ui/demo/public/proof."""
import hashlib
import os

import numpy as np
import pytest

from luminair_amd import synthetic as syn
from oracle import air
from oracle.blake2s import blake2s, blake2s_ref, blake2s_words_vec
from oracle.channel import Blake2sChannel, ProtocolVariant
from oracle.circle import CanonicCoset, bit_reverse_index
from oracle.fft import evaluate, interpolate, eval_at_point, domain_twiddles
from oracle.field import P, QM31, q_inv, q_mul
from oracle.proof import from_bincode, to_bincode
from oracle.prover import PcsConfig, prove
from oracle.verifier import VerificationError, verify

Q = QM31


def _tables():
    return [(k, r.astype(np.uint64)) for k, r in syn.simple_example()]


def test_kat_parses_and_round_trips(kat_bytes):
    p = from_bincode(kat_bytes, 8)
    assert to_bincode(p) == kat_bytes
    assert p.claim[:2] == [4, 4] and all(c is None for c in p.claim[2:])
    s = p.proof
    assert (s.pow_bits, s.log_blowup, s.log_last_layer, s.n_queries) == (5, 1, 0, 3)
    assert [len(t) for t in s.sampled_values] == [0, 31, 24, 4]
    assert [len(q) for q in s.queried_values] == [0, 93, 72, 12]
    assert s.proof_of_work == 2
    assert s.commitments[0] == hashlib.blake2s(b"").digest()


def test_oracle_reproduces_kat_bytes(kat_bytes):
    """Starting only from the three input tensors a, b, w of examples/simple."""
    proof, tr = prove(_tables(), PcsConfig(), ProtocolVariant.KAT, want_trace=True)
    assert to_bincode(proof) == kat_bytes
    # golden transcript values, SURVEY.md Appendix A.11
    assert tr.digests["root0"].hex() == "cf8e32a943ba475e63f0f7f85e3a7b84fee897eabb375881ed0fe78392e86d70"
    assert tr.digests["claims"].hex() == "d8803f759deb749d933ff3cf9654658a6bbf86cdca506be85a38a9628b3764a4"
    assert tr.digests["root1"].hex() == "806b65457e2d76fe1224c9594d6e7f0448ef789713b538901bf41cd9a3376dac"
    assert tr.digests["root2"].hex() == "33edc03702a000872ed04955d8953d72b3f8137211c8fb907a5956165f5b98db"
    assert tr.digests["root3"].hex() == "bbc612e8f69ab60b3619ab23e4501dd48d2c7522f9c830d04c4f7afa94b7300f"
    assert tr.digests["before_pow"].hex() == "27aa41607f1c1ac3d9dcdc2fe7a56be2d87e822e300e7c2f11f50d31154bbdf4"
    assert tr.z == Q(1354497678, 30677172, 1144715120, 1609814521)
    assert tr.alpha_rel == Q(929076832, 1279522411, 452464722, 918734119)
    assert tr.claimed_sums[0] == Q(423011912, 209621612, 1072831704, 712041665)
    assert tr.claimed_sums[1] == -tr.claimed_sums[0]
    assert tr.composition_alpha == Q(137293579, 733405986, 1348299213, 871775106)
    assert tr.oods_point[0] == Q(685080583, 1702087524, 118247749, 701520535)
    assert tr.oods_point[1] == Q(1210140318, 727558780, 1264925987, 52624792)
    assert tr.quotient_alpha == Q(1457221513, 1917982884, 99696584, 484777430)
    assert tr.fri_alphas[0] == Q(112743498, 2140550112, 1054552672, 2071504381)
    assert tr.fri_alphas[1:] == [Q(1709842450, 1027040626, 1897302991, 1020927303),
                                 Q(1290836927, 1415190035, 1381382497, 1862474939),
                                 Q(341792504, 130513683, 1399479259, 1780709486),
                                 Q(1274099128, 1912777712, 117659290, 1260210290)]
    assert tr.queries == [20, 25, 40]
    assert [r.hex()[:8] for r in tr.roots] == ["69217a30", "a3e53a74", "3d4916a3", "a18f65e6"]
    assert proof.proof.last_layer_coeffs == [Q(2000869715, 121772074, 453148178, 1304667972)]


def test_kat_verifies_and_tampering_is_rejected(kat_bytes):
    verify(from_bincode(kat_bytes, 8))
    for off in (200, 1000, 2000, 3000, 4000, 4700):
        b = bytearray(kat_bytes)
        b[off] ^= 1
        with pytest.raises((VerificationError, ValueError, ZeroDivisionError, AssertionError, KeyError)):
            verify(from_bincode(bytes(b), 8))


def test_pow_nonce_and_trailing_zeros(kat_bytes):
    """A.11: nonce 2 gives 5 trailing zeros; nonces 0, 1, 3 give 2, 1, 1."""
    _, tr = prove(_tables(), want_trace=True)
    base = Blake2sChannel()
    base.digest = tr.digests["before_pow"]
    tz = []
    for nonce in range(4):
        c = base.clone()
        c.mix_u64(nonce)
        tz.append(c.trailing_zeros())
    assert tz == [2, 1, 5, 1]


def test_blake2s_restatement_matches_hashlib():
    rng = np.random.default_rng(0)
    for n in (0, 1, 31, 32, 63, 64, 65, 127, 128, 129, 500):
        d = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        assert blake2s_ref(d) == hashlib.blake2s(d).digest()
    for w in (1, 4, 15, 16, 17, 20, 31, 32, 33):
        words = rng.integers(0, 2 ** 32, size=(7, w), dtype=np.uint32)
        got = blake2s_words_vec(words)
        for i in range(7):
            assert got[i].astype("<u4").tobytes() == hashlib.blake2s(words[i].astype("<u4").tobytes()).digest()


def test_field_and_fft_contract():
    rng = np.random.default_rng(1)
    a = rng.integers(0, P, size=(50, 4), dtype=np.uint64)
    one = np.zeros((50, 4), dtype=np.uint64)
    one[:, 0] = 1
    assert np.array_equal(q_mul(a, q_inv(a)), one)
    n = 6
    c = rng.integers(0, P, size=1 << n, dtype=np.uint64)
    ev = evaluate(c, n)
    assert np.array_equal(interpolate(ev), c)
    # evaluation contract: ev[s] = poly(at(bit_reverse(s)))
    dom = CanonicCoset(n).circle_domain()
    for s in (0, 1, 2, 17, 63):
        x, y = dom.at(bit_reverse_index(s, n))
        assert eval_at_point(c, (QM31(x), QM31(y))) == QM31(int(ev[s]))
    # LDE = same polynomial on the next canonic domain
    ev2 = evaluate(c, n + 1)
    dom2 = CanonicCoset(n + 1).circle_domain()
    for s in (0, 5, 100, 127):
        x, y = dom2.at(bit_reverse_index(s, n + 1))
        assert eval_at_point(c, (QM31(x), QM31(y))) == QM31(int(ev2[s]))
    # circle twiddles are derivable from the first line layer: [y, -y, -x, x]
    tws, _ = domain_twiddles(7)
    t1 = tws[1]
    exp = np.stack([t1[1::2], (P - t1[1::2]) % P, (P - t1[0::2]) % P, t1[0::2]], axis=1).reshape(-1)
    assert np.array_equal(exp.astype(np.uint64), tws[0])


@pytest.mark.parametrize("n_rows,seed", [(1, 0), (5, 1), (16, 2), (17, 3), (100, 4), (1000, 5)])
def test_oracle_proofs_verify_ragged_sizes(n_rows, seed):
    """Ragged / padded tables (write_trace pads to max(next_pow2, 16)) prove and verify."""
    tabs = syn.chain_graph(n_rows, seed)
    proof = prove([(k, r.astype(np.uint64)) for k, r in tabs])
    verify(from_bincode(to_bincode(proof), 8))


def test_oracle_rejects_bad_trace_and_empty_table():
    from oracle.prover import ProvingError
    tabs = syn.config2_add_only(64, 9)
    bad = tabs[0][1].copy()
    bad[3, 11] = (int(bad[3, 11]) + 1) % P   # out != lhs + rhs
    with pytest.raises(ProvingError):
        prove([(0, bad.astype(np.uint64))])
    with pytest.raises(ProvingError):
        prove([(0, np.zeros((0, 15), dtype=np.uint64))])


def test_unbalanced_logup_fails_verification_only():
    """prove() never checks the logup sum — only verify() does (verifier.rs:97-99)."""
    tabs = syn.config2_graph_faithful(32, 3)[:1]   # Add consumes inputs nobody yields
    proof = prove([(k, r.astype(np.uint64)) for k, r in tabs])
    with pytest.raises(VerificationError, match="InvalidLogUp"):
        verify(proof)


def test_oracle_pinned_variant_proves_and_verifies():
    """LuminAIR-HEAD layout (17 claim slots, Inputs component) — parity unpinned, self-consistent."""
    tabs = syn.config2_graph_faithful(100, 3)
    proof = prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=ProtocolVariant.PINNED)
    assert len(proof.claim) == 17 and proof.claim[0] == 7 and proof.claim[15] == 8
    verify(from_bincode(to_bincode(proof), 17), ProtocolVariant.PINNED)
    with pytest.raises(Exception):
        verify(from_bincode(to_bincode(proof), 17), ProtocolVariant.KAT)   # different transcript encodings


def test_oracle_linear_layer_components_verify():
    """Mul + SumReduce + Add (+ MaxReduce) — the linear-layer lowering of BASELINE config 5; the
    SumReduce / MaxReduce / Contiguous constraint forms are fully visible in the reference."""
    for tabs in (syn.linear_layer(8, 16, 1), syn.linear_layer(20, 7, 2, True)):
        proof = prove([(k, r.astype(np.uint64)) for k, r in tabs])
        verify(from_bincode(to_bincode(proof), 8))
    tabs = [(0, syn.add_rows([5, 6, 7], [1, 2, 3], node=2, lhs_id=0, rhs_id=1, mults=(0, 0, 1))),
            (16, syn.contiguous_rows([6, 8, 10], node=3, input_id=2, input_mult=-1, out_mult=0))]
    proof = prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=ProtocolVariant.PINNED)
    verify(from_bincode(to_bincode(proof), 17), ProtocolVariant.PINNED)
    # a wrong running sum must be caught by the AIR
    from oracle.prover import ProvingError
    bad = syn.linear_layer(8, 16, 1)
    rows = bad[2][1].copy()
    rows[5, 10] = (int(rows[5, 10]) + 1) % P     # next_acc != acc + input
    with pytest.raises(ProvingError):
        prove([(bad[0][0], bad[0][1].astype(np.uint64)), (bad[1][0], bad[1][1].astype(np.uint64)),
               (bad[2][0], rows.astype(np.uint64))])


def test_oracle_less_than_with_range_check_lut():
    """LessThan + RangeCheckLookup (PINNED variant): tree 0 holds the preprocessed 8-bit LUT column,
    the four limb relations use RangeCheckLookupElements, logup sums cancel."""
    from oracle.prover import ProvingError
    tabs = syn.less_than_graph(100, 3)
    proof = prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=ProtocolVariant.PINNED)
    assert [c for c in proof.claim if c is not None] == [7, 7, 8, 9]
    assert [len(t) for t in proof.proof.sampled_values] == [1, 15 + 22 + 1 + 7, 12 + 28 + 4 + 4, 4]
    assert proof.proof.commitments[0] != hashlib.blake2s(b"").digest()
    verify(from_bincode(to_bincode(proof), 17), ProtocolVariant.PINNED)
    # a limb outside its recomposition is caught by the AIR
    rows = tabs[1][1].copy()
    rows[3, 14] = (int(rows[3, 14]) + 1) % P
    with pytest.raises(ProvingError):
        prove([(tabs[0][0], tabs[0][1].astype(np.uint64)), (13, rows.astype(np.uint64)),
               (tabs[2][0], tabs[2][1].astype(np.uint64)), (tabs[3][0], tabs[3][1].astype(np.uint64))],
              variant=ProtocolVariant.PINNED)
    # wrong LUT multiplicities prove fine but fail the verifier's logup check
    bad = tabs[2][1].copy()
    bad[7, 0] += 1
    p2 = prove([(tabs[0][0], tabs[0][1].astype(np.uint64)), (tabs[1][0], tabs[1][1].astype(np.uint64)),
                (14, bad.astype(np.uint64)), (tabs[3][0], tabs[3][1].astype(np.uint64))], variant=ProtocolVariant.PINNED)
    with pytest.raises(VerificationError, match="InvalidLogUp"):
        verify(p2, ProtocolVariant.PINNED)
    # range-check relations need the HEAD relation draws
    with pytest.raises(ProvingError):
        prove([(k, r.astype(np.uint64)) for k, r in tabs[:2]], variant=ProtocolVariant.KAT)


def test_oracle_lut_activations():
    """Sin / Exp2 / Log2 with their lookup components (sin/component.rs:50-122, lookups/sin/component.rs:40-59):
    tree 0 holds the three two-column LUTs (different sizes, size-descending order), the LUT relations are
    width 2 with their own element sets, logup sums cancel against Inputs."""
    from oracle.prover import ProvingError
    tabs, luts = syn.activation_graph(40, 3)
    assert [k for k, _ in tabs] == [3, 4, 9, 10, 11, 12, 15]
    proof = prove(tabs, variant=ProtocolVariant.PINNED, luts=luts)
    assert [(k, c) for k, c in enumerate(proof.claim) if c is not None] == \
        [(3, 6), (4, 16), (9, 6), (10, 15), (11, 6), (12, 14), (15, 7)]
    assert [len(t) for t in proof.proof.sampled_values] == [6, 3 * 12 + 3 + 7, 3 * 12 + 3 * 4 + 4, 4]
    verify(from_bincode(to_bincode(proof), 17), ProtocolVariant.PINNED)
    # an output that is not the LUT's value for its input: the AIR holds (no local constraint ties out to
    # input), the logup sum does not cancel
    rows = tabs[0][1].copy()
    rows[5, 8] ^= 1
    p2 = prove([(3, rows)] + tabs[1:], variant=ProtocolVariant.PINNED, luts=luts)
    with pytest.raises(VerificationError, match="InvalidLogUp"):
        verify(p2, ProtocolVariant.PINNED)
    # the LUT columns are inputs of the proof: omitted or mis-sized ones are rejected
    with pytest.raises(ProvingError):
        prove(tabs, variant=ProtocolVariant.PINNED, luts={k: v for k, v in luts.items() if k != "exp2"})
    with pytest.raises(ProvingError):
        prove(tabs, variant=ProtocolVariant.PINNED, luts=dict(luts, sin=(luts["sin"][0][:1 << 15], luts["sin"][1][:1 << 15])))
    # KAT era: a single LUT relation draw (sin); exp2 needs the HEAD draws
    lo, hi = -2 * 4096, 2 * 4096
    a = np.random.default_rng(4).integers(lo, hi + 1, size=20)
    rows, counts = syn.unary_lut_rows("sin", a, lo, mults=(0, 0))
    lut = syn.make_lut("sin", lo, hi)
    kat_tabs = [(3, rows), (4, syn.lut_lookup_rows(counts, len(lut[0])))]
    verify(prove(kat_tabs, luts={"sin": lut}), ProtocolVariant.KAT)
    with pytest.raises(ProvingError):
        prove([t for t in tabs if t[0] in (9, 10)], luts=luts)
