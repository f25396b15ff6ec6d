"""Pin the protocol a proof was made with by exhaustive search over the LMN_PV_* flags (include/luminair_hip.h).

The reference holds one known-answer proof, made by an older build than its sources at HEAD, and
the arithmetic of the path lives in un-vendored dependencies: which encodings HEAD really uses cannot be read off
`/root/reference`.  Every observable difference is one independent flag; this module tries every combination through
the product's host-only verifier (`lmn_verify_diagnose`, milliseconds per call, no GPU) and reports

* the accepting combinations, as determined / undetermined flags (a flag no component of the proof depends on - e.g.
  the number of LUT relation draws in a proof without LUT components - cannot be determined from that proof);
* when nothing accepts: how far the best combinations get.  The verifier's checks depend on different parts of the
  protocol (transcript encodings / Merkle hashing / constraint forms), so the pattern of passed checks says which part
  disagrees;
* given the trace tables the proof was made from: the first field, in transcript order, where the prover's own proof
  under a flag combination differs from the given one (`first_divergence`).

What a maintainer runs on the reference's side to produce the inputs: INTEGRATION.md "Pinning the protocol of a build".
Anchors: SURVEY.md Appendix A.3 (both columns); /root/reference/crates/air/src/lib.rs:30-104;
/root/reference/crates/verifiers/rust/src/verifier.rs:21-143.
"""
from __future__ import annotations

import itertools
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

from . import backend as B

TRANSCRIPT_BITS = (B.PV_CLAIM17, B.PV_LUT_DRAWS4, B.PV_MIX_U64_HASHED, B.PV_DRAW_CTR_U32, B.PV_POW_PREFIXED)
# constraint-form bits and the component kind (TraceTable order, pie.rs:31-66) each one concerns
FORM_BITS = {B.PV_MUL_ONE_SLOT: 1, B.PV_RECIP_TWO_SLOTS: 2, B.PV_RECIP_NEG: 2, B.PV_SQRT_TWO_SLOTS: 7, B.PV_SQRT_NEG: 7,
             B.PV_REM_TWO_SLOTS: 8, B.PV_REM_NEG: 8}
# kinds whose relations use a LUT element set (the only consumers of the extra draws of LUT_DRAWS4)
LUT_KINDS = (3, 4, 9, 10, 11, 12, 13, 14)

CHECK_ORDER = (B.CHECK_PARSE, B.CHECK_SHAPE, B.CHECK_LOGUP_SUM, B.CHECK_OODS, B.CHECK_POW, B.CHECK_TREE_DECOMMIT,
               B.CHECK_FRI_DECOMMIT, B.CHECK_FRI_FOLDS)


def flag_names(flags: int) -> List[str]:
    return [n for b, n in B.PV_NAMES.items() if flags & b]


def check_names(mask: int) -> List[str]:
    return [B.CHECK_NAMES[c] for c in CHECK_ORDER if mask & c]


def claim_kinds(proof: bytes, n_slots: int) -> Optional[List[int]]:
    """Component kinds present in a proof's claim if its first bytes parse as `n_slots` Option<Claim> (1 tag byte +
    u32 log_size) followed by `n_slots` Option<InteractionClaim> (1 tag byte + 4 x u32); None if they do not."""
    o, kinds = 0, []
    try:
        for k in range(n_slots):
            tag = proof[o]
            o += 1
            if tag > 1:
                return None
            if tag:
                (ls,) = struct.unpack_from("<I", proof, o)
                o += 4
                if not 4 <= ls <= 26:
                    return None
                kinds.append(k)
        for k in range(n_slots):
            tag = proof[o]
            o += 1
            if tag > 1 or bool(tag) != (k in kinds):
                return None
            o += 16 * tag
    except (IndexError, struct.error):
        return None
    return kinds or None


@dataclass
class Trial:
    flags: int
    rc: int
    run: int
    passed: int
    failed: int
    message: str
    steps: List[Tuple[str, int, str]] = field(default_factory=list)

    @property
    def accepted(self) -> bool:
        return self.rc == B.LMN_OK and self.failed == 0 and self.passed == B.CHECK_ALL

    @property
    def score(self) -> int:
        return bin(self.passed).count("1")


@dataclass
class PinResult:
    trials: List[Trial]
    accepted: List[int]
    searched_bits: List[int]          # flags that were varied (relevant to this proof)
    irrelevant_bits: List[int]        # flags no component of the proof depends on: left at 0, undeterminable
    determined: Dict[int, bool]       # flag -> value, same in every accepting combination
    undetermined: List[int]           # searched flags whose value does not change the verdict
    kinds: List[int]

    @property
    def unique(self) -> bool:
        """Exactly one accepting combination over the flags this proof depends on."""
        return len(self.accepted) > 0 and not self.undetermined and len(self.accepted) == 1

    @property
    def flags(self) -> Optional[int]:
        return self.accepted[0] if self.accepted else None

    def diagnosis(self) -> str:
        """What the pattern of passed checks says when no combination accepts."""
        if self.accepted:
            return "accepted"
        best = sorted(self.trials, key=lambda t: -t.score)
        if not best or best[0].score == 0 or all(not (t.passed & B.CHECK_PARSE) for t in self.trials):
            return "the bytes do not parse as a LuminairProof under either claim layout (bincode 1.3, 8 or 17 Option slots)"
        transcript_ok = [t for t in self.trials if (t.passed & (B.CHECK_TREE_DECOMMIT | B.CHECK_FRI_DECOMMIT | B.CHECK_POW))
                         == (B.CHECK_TREE_DECOMMIT | B.CHECK_FRI_DECOMMIT | B.CHECK_POW)]
        if transcript_ok:
            tb = sorted({t.flags & B.PV_TRANSCRIPT_MASK for t in transcript_ok})
            folds = any(t.passed & B.CHECK_FRI_FOLDS for t in transcript_ok)
            return ("transcript encodings and Merkle hashing are consistent under transcript flags %s (proof of work and every "
                    "decommitment at the drawn queries hold%s) but the composition identity at the OODS point fails for every "
                    "constraint-form combination: a constraint form or logup relation of a component in the claim (kinds %s) "
                    "differs from the restated one"
                    % ([flag_names(f) or ["none"] for f in tb], ", FRI folds too" if folds else "; FRI folds fail as well: "
                       "quotient / fold arithmetic differs", self.kinds))
        oods_ok = [t for t in self.trials if t.passed & B.CHECK_OODS]
        pow_ok = [t for t in self.trials if t.passed & B.CHECK_POW]
        if oods_ok:
            tb = sorted({t.flags & B.PV_TRANSCRIPT_MASK & ~B.PV_POW_PREFIXED for t in oods_ok})
            return ("the composition identity holds under transcript flags %s (claim mix, relation draws and constraint forms "
                    "agree) but %s: the encodings used after the OODS draw (mix_felts of the sampled values, FRI roots, proof of "
                    "work, query draw) or the Merkle hashing differ"
                    % ([flag_names(f) or ["none"] for f in tb],
                       "no combination passes the proof of work" if not [t for t in oods_ok if t.passed & B.CHECK_POW]
                       else "the decommitments at the drawn queries fail"))
        if pow_ok:
            return ("only the proof-of-work check passes for some combinations (5 bits of evidence: expect 1 in 32 by chance); "
                    "no combination reproduces the composition identity: the transcript encodings before the OODS draw differ "
                    "from every restated form")
        return "no check beyond parsing passes under any combination: the channel encodings differ from every restated form"


def search(lib: B.Library, proof: bytes, config: Optional[B.LmnConfig] = None, settings=None,
           keep_steps: bool = False) -> PinResult:
    """Try every combination of the protocol flags the proof can depend on."""
    trials: List[Trial] = []
    kinds_by_layout = {c17: claim_kinds(proof, 17 if c17 else 8) for c17 in (0, 1)}
    kinds_all = sorted({k for ks in kinds_by_layout.values() if ks for k in ks})
    searched, irrelevant = set(), set()
    for c17 in (0, 1):
        kinds = kinds_by_layout[c17]
        if kinds is None:
            # one diagnose call so that the report shows the parse failure
            rc, rep = lib.diagnose(proof, B.PV_CLAIM17 if c17 else 0, config, settings)
            trials.append(Trial(B.PV_CLAIM17 if c17 else 0, rc, rep.checks_run, rep.checks_passed, rep.checks_failed,
                                rep.first_failure.decode(errors="replace")))
            continue
        bits = [B.PV_MIX_U64_HASHED, B.PV_DRAW_CTR_U32, B.PV_POW_PREFIXED]
        if any(k in LUT_KINDS for k in kinds):
            bits.append(B.PV_LUT_DRAWS4)
        else:
            irrelevant.add(B.PV_LUT_DRAWS4)
        for b, kind in FORM_BITS.items():
            (bits.append(b) if kind in kinds else irrelevant.add(b))
        searched.update(bits)
        searched.add(B.PV_CLAIM17)
        for values in itertools.product((0, 1), repeat=len(bits)):
            flags = (B.PV_CLAIM17 if c17 else 0) | sum(b for b, v in zip(bits, values) if v)
            rc, rep = lib.diagnose(proof, flags, config, settings)
            steps = []
            if keep_steps:
                steps = [(B.STEP_NAMES[rep.steps[i].step], int(rep.steps[i].index), bytes(rep.steps[i].digest).hex())
                         for i in range(rep.n_steps)]
            trials.append(Trial(flags, rc, rep.checks_run, rep.checks_passed, rep.checks_failed,
                                rep.first_failure.decode(errors="replace"), steps))
    irrelevant -= searched
    accepted = sorted(t.flags for t in trials if t.accepted)
    acc = set(accepted)
    determined, undetermined = {}, []
    for b in sorted(searched):
        if not accepted:
            break
        vals = {bool(f & b) for f in accepted}
        if len(vals) == 1:
            determined[b] = vals.pop()
        elif all((f ^ b) in acc for f in accepted):
            undetermined.append(b)
        else:
            determined[b] = None   # correlated with another flag: listed through `accepted`
    return PinResult(trials, accepted, sorted(searched), sorted(irrelevant), determined, undetermined, kinds_all)


# ---------------------------------------------------------------------------------------------- proof comparison
def _fields_in_transcript_order(d: dict) -> List[Tuple[str, object]]:
    """Proof fields in the order the transcript consumes / the prover produces them."""
    p = d["proof"]
    out: List[Tuple[str, object]] = [("commitments[0] (preprocessed root: LUT columns / rounding)", p["commitments"][0]),
                                     ("claim (log sizes)", d["claim"]),
                                     ("commitments[1] (main trace root: rows, padding, column order)", p["commitments"][1]),
                                     ("interaction_claim (claimed sums: claim mix + relation draws + logup)", d["interaction_claim"]),
                                     ("commitments[2] (interaction trace root)", p["commitments"][2]),
                                     ("commitments[3] (composition root: composition randomness draw + constraint forms)", p["commitments"][3])]
    for t, tree in enumerate(p["sampled_values"]):
        out.append(("sampled_values[%d] (OODS point draw)" % t, tree))
    fp = p["fri_proof"]
    out.append(("fri first layer commitment (mix_felts of sampled values + quotient randomness)", fp["first_layer"]["commitment"]))
    for i, l in enumerate(fp["inner_layers"]):
        out.append(("fri inner layer %d commitment (fold randomness draw)" % i, l["commitment"]))
    out.append(("fri last layer", fp["last_layer_poly"]))
    out.append(("proof_of_work (nonce: proof-of-work form / mix_u64)", p["proof_of_work"]))
    out.append(("queried_values (query draw)", p["queried_values"]))
    out.append(("decommitments", p["decommitments"]))
    out.append(("fri first layer witness", fp["first_layer"]["fri_witness"]))
    out.append(("fri inner layer witnesses", [l["fri_witness"] for l in fp["inner_layers"]]))
    return out


def first_divergence(ours: bytes, theirs: bytes, claim17: bool) -> Optional[str]:
    """Name of the first proof field (transcript order) where two proofs differ; None if byte-identical."""
    if ours == theirs:
        return None
    from .pie import LuminairProof
    a = LuminairProof(ours).to_dict(kat_era=not claim17)
    b = LuminairProof(theirs).to_dict(kat_era=not claim17)
    for (name, va), (_, vb) in zip(_fields_in_transcript_order(a), _fields_in_transcript_order(b)):
        if va != vb:
            return name
    return "config / shape"


def format_report(res: PinResult, top: int = 6) -> str:
    lines = []
    n_acc = len(res.accepted)
    lines.append("component kinds in the claim: %s" % res.kinds)
    lines.append("flags searched: %s" % [B.PV_NAMES[b] for b in res.searched_bits])
    if res.irrelevant_bits:
        lines.append("flags this proof cannot determine (no component depends on them): %s"
                     % [B.PV_NAMES[b] for b in res.irrelevant_bits])
    lines.append("combinations tried: %d, accepted: %d" % (len(res.trials), n_acc))
    if n_acc:
        for b, v in res.determined.items():
            lines.append("  %-16s = %s" % (B.PV_NAMES[b], "correlated (see list)" if v is None else int(v)))
        for b in res.undetermined:
            lines.append("  %-16s   undetermined (verdict does not depend on it)" % B.PV_NAMES[b])
        if B.PV_POW_PREFIXED in res.undetermined:
            lines.append("  note: the proof-of-work form is decided by ONE check worth pow_bits bits (5 by default): a nonce found "
                         "under one form passes the other with probability 2^-pow_bits, as it did here.  Run the tool on a second "
                         "proof, or with --tables (the prover's own nonce differs between the two forms).")
        lines.append("accepting protocol_variant values: %s" % ", ".join("0x%04x %s" % (f, flag_names(f) or ["KAT"]) for f in res.accepted))
        if res.unique:
            f = res.accepted[0]
            lines.append("UNIQUE: protocol_variant = 0x%04x%s" % (f, "  (= LMN_VARIANT_KAT)" if f == 0 else
                                                                 "  (= LMN_VARIANT_PINNED)" if f == B.VARIANT_PINNED else ""))
    else:
        lines.append("NO combination accepts.  " + res.diagnosis())
        lines.append("best combinations:")
        for t in sorted(res.trials, key=lambda t: (-t.score, t.flags))[:top]:
            lines.append("  0x%04x %-60s passed %s; first failure: %s" % (t.flags, flag_names(t.flags) or ["KAT"],
                                                                         check_names(t.passed), t.message))
    return "\n".join(lines)
