"""The protocol-flag search kit (luminair_amd/pinning.py, tools/pin_variant.py): pins which of the LMN_PV_* flag
combinations (include/luminair_hip.h) a proof was made with, through the product's host-only verifier.

(a) the reference's own known-answer proof -> exactly the KAT combination, uniquely;
(b) oracle proofs made under random flag combinations -> each combination is recovered;
(c) with the trace tables: the first proof field where the prover's proof under another combination differs.
Host code only (emulation build here, the HIP library on the GPU box)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from luminair_amd import backend as B, pinning, synthetic as syn
from oracle.channel import ProtocolVariant
from oracle.proof import to_bincode
from oracle.prover import prove


@pytest.fixture(scope="module")
def lib(root):
    so = os.path.join(root, "tests", "emu", "libluminair_emu.so")
    if not os.path.exists(so):
        subprocess.run([os.path.join(root, "tests", "emu", "build_emu.sh")], check=True, capture_output=True)
    return B.Library(so)


def test_kat_bytes_pin_the_kat_combination_uniquely(lib, kat_bytes):
    res = pinning.search(lib, kat_bytes, keep_steps=True)
    assert res.accepted == [B.VARIANT_KAT] and res.unique and res.flags == 0
    assert res.kinds == [0, 1]                                   # Add, Mul
    # everything the proof depends on is determined; the rest is reported as not determinable from this proof
    assert set(res.determined) == {B.PV_CLAIM17, B.PV_MIX_U64_HASHED, B.PV_DRAW_CTR_U32, B.PV_POW_PREFIXED, B.PV_MUL_ONE_SLOT}
    assert not any(res.determined.values()) and not res.undetermined
    assert B.PV_LUT_DRAWS4 in res.irrelevant_bits and B.PV_RECIP_NEG in res.irrelevant_bits
    # the claim does not even parse under the 17-slot layout
    t17 = [t for t in res.trials if t.flags & B.PV_CLAIM17]
    assert t17 and all(not (t.passed & B.CHECK_PARSE) for t in t17)
    # every single-flag neighbour of the KAT combination fails a check that flag feeds
    by_flags = {t.flags: t for t in res.trials}
    assert by_flags[B.PV_POW_PREFIXED].failed & B.CHECK_POW
    assert by_flags[B.PV_MUL_ONE_SLOT].failed == B.CHECK_OODS       # transcript intact: only the composition identity breaks
    assert by_flags[B.PV_MUL_ONE_SLOT].passed & B.CHECK_TREE_DECOMMIT and by_flags[B.PV_MUL_ONE_SLOT].passed & B.CHECK_FRI_FOLDS
    assert by_flags[B.PV_MIX_U64_HASHED].failed & B.CHECK_OODS
    assert "UNIQUE: protocol_variant = 0x0000" in pinning.format_report(res)


def test_replayed_digests_are_the_golden_transcript_values(lib, kat_bytes):
    """`lmn_verify_diagnose` reports the channel digest after every mix: equal to the oracle prover's own transcript
    (SURVEY.md Appendix A.11 values, pinned in tests/test_oracle_kat.py)."""
    _, tr = prove([(k, r.astype(np.uint64)) for k, r in syn.simple_example()], want_trace=True)
    rc, rep = lib.diagnose(kat_bytes, B.VARIANT_KAT)
    assert rc == B.LMN_OK and rep.checks_passed == B.CHECK_ALL and rep.checks_failed == 0
    got = {(B.STEP_NAMES[rep.steps[i].step], rep.steps[i].index): bytes(rep.steps[i].digest) for i in range(rep.n_steps)}
    assert got[("root_preprocessed", 0)] == tr.digests["root0"]
    assert got[("claim", 0)] == tr.digests["claims"]
    assert got[("root_main", 0)] == tr.digests["root1"]
    assert got[("root_interaction", 0)] == tr.digests["root2"]
    assert got[("root_composition", 0)] == tr.digests["root3"]
    assert got[("sampled_values", 0)] == tr.digests["sampled"]
    assert got[("fri_last_layer", 0)] == tr.digests["before_pow"]
    assert rep.n_steps == 10 + len(tr.fri_roots) - 1


def _random_combinations(n, seed, claim17):
    rng = np.random.default_rng(seed)
    bits = [B.PV_MIX_U64_HASHED, B.PV_DRAW_CTR_U32, B.PV_POW_PREFIXED]
    bits += ([B.PV_SQRT_TWO_SLOTS, B.PV_SQRT_NEG, B.PV_REM_TWO_SLOTS, B.PV_REM_NEG] if claim17
             else [B.PV_MUL_ONE_SLOT, B.PV_RECIP_TWO_SLOTS, B.PV_RECIP_NEG])
    out = set()
    while len(out) < n:
        f = (B.PV_CLAIM17 if claim17 else 0) | sum(b for b in bits if rng.integers(2))
        out.add(int(f))
    return sorted(out)


@pytest.mark.parametrize("flags", _random_combinations(4, 11, False) + _random_combinations(4, 12, True))
def test_random_flag_combinations_are_recovered(lib, flags):
    """An oracle proof made under a random combination: the search accepts that combination and no other (over
    the flags the proof's components depend on)."""
    if flags & B.PV_CLAIM17:
        tabs = syn.sqrt_rem_graph(40, 4)         # Sqrt, Rem, Inputs: kinds without a KAT-era claim slot
    else:
        tabs = syn.chain_graph(50, 5)            # Add, Mul, Recip
    proof = to_bincode(prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=ProtocolVariant(flags)))
    res = pinning.search(lib, proof)
    assert res.accepted == [flags], (hex(flags), [hex(f) for f in res.accepted])
    assert res.unique
    lib.verify(proof, flags)


def test_lut_draw_count_is_undetermined_without_lut_consumers_and_determined_with(lib):
    """The number of LUT relation draws changes nothing a proof without LUT relations shows (draws do not alter the
    digest): reported as not determinable; a LessThan proof (range-check relation = 4th LUT draw) pins it."""
    tabs = syn.config2_graph_faithful(40, 3)                      # Add + Inputs
    proof = to_bincode(prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=ProtocolVariant.PINNED))
    res = pinning.search(lib, proof)
    assert B.PV_LUT_DRAWS4 in res.irrelevant_bits
    assert res.accepted == [B.VARIANT_PINNED & ~B.PV_LUT_DRAWS4] and res.unique
    tabs = syn.less_than_graph(40, 3)
    proof = to_bincode(prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=ProtocolVariant.PINNED))
    res = pinning.search(lib, proof)
    assert res.accepted == [B.VARIANT_PINNED] and res.unique and res.determined[B.PV_LUT_DRAWS4] is True


def test_diagnosis_when_nothing_accepts(lib, kat_bytes):
    res = pinning.search(lib, kat_bytes[:-7])
    assert not res.accepted and "do not parse" in res.diagnosis()
    # a proof whose composition identity fails under every restated form, transcript intact: what a build of the
    # reference with another constraint form would look like.  Made here by giving the oracle a Mul form of its own.
    from oracle import air
    orig = air.MUL.local

    def other_form(c):
        cons = orig(c)
        cons[2] = cons[1] * 3            # a second eval_fixed_mul slot that is not identically zero (vanishes on a valid trace)
        return cons
    air.MUL.local = other_form
    try:
        tabs = syn.chain_graph(50, 5)
        proof = to_bincode(prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=ProtocolVariant.MIX_U64_HASHED))
    finally:
        air.MUL.local = orig
    res = pinning.search(lib, proof)
    assert not res.accepted
    d = res.diagnosis()
    assert "composition identity at the OODS point fails" in d and "mix_u64_hashed" in d and "FRI folds too" in d
    best = max(res.trials, key=lambda t: t.score)
    assert best.failed == B.CHECK_OODS and best.flags & B.PV_TRANSCRIPT_MASK == B.PV_MIX_U64_HASHED


@pytest.mark.parametrize("flip,where", [
    (B.PV_POW_PREFIXED, "proof_of_work"), (B.PV_MIX_U64_HASHED, "interaction_claim"),
    (B.PV_DRAW_CTR_U32, "interaction_claim"), (B.PV_MUL_ONE_SLOT, "commitments[3]"), (0, None)])
def test_first_divergence_names_the_step(lib, flip, where):
    """With the tables: prove under another combination and name the first differing field in transcript order."""
    tabs = syn.chain_graph(40, 9)
    theirs = to_bincode(prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=ProtocolVariant(B.PV_DRAW_CTR_U32)))
    cfg = lib.default_config()
    cfg.protocol_variant = B.PV_DRAW_CTR_U32 ^ flip
    ctx = B.Context(0, cfg, lib)
    ours = ctx.prove_tables([(k, r, len(r)) for k, r in tabs])
    ctx.close()
    got = pinning.first_divergence(ours, theirs, False)
    assert (got is None) if where is None else got.startswith(where), got


def test_pin_variant_cli(root, tmp_path, kat_bytes):
    """tools/pin_variant.py end to end: the KAT file, then an oracle-made proof with its table dumps."""
    so = os.path.join(root, "tests", "emu", "libluminair_emu.so")
    tool = os.path.join(root, "tools", "pin_variant.py")
    r = subprocess.run([sys.executable, tool, os.path.join(root, "tests", "golden", "kat_simple", "proof"), "--library", so,
                        "--digests"], capture_output=True, text=True)
    assert r.returncode == 0 and "UNIQUE: protocol_variant = 0x0000" in r.stdout and "root_composition" in r.stdout, r.stdout + r.stderr
    flags = B.PV_MIX_U64_HASHED | B.PV_POW_PREFIXED | B.PV_RECIP_NEG
    tabs = syn.chain_graph(30, 2)
    proof = to_bincode(prove([(k, r.astype(np.uint64)) for k, r in tabs], variant=ProtocolVariant(flags)))
    (tmp_path / "proof.bin").write_bytes(proof)
    for k, rows in tabs:
        rows.astype("<u4").tofile(str(tmp_path / ("table_%d.bin" % k)))
    rep = tmp_path / "report.json"
    r = subprocess.run([sys.executable, tool, str(tmp_path / "proof.bin"), "--library", so, "--tables", str(tmp_path),
                        "--report", str(rep)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "UNIQUE: protocol_variant = 0x%04x" % flags in r.stdout and "byte-identical to the given proof" in r.stdout
    import json
    out = json.loads(rep.read_text())
    assert out["accepted"] == [flags] and out["unique"] and out["divergence"] == {"0x%04x" % flags: "byte-identical to the given proof"}
