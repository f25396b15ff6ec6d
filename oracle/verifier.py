"""CPU restatement of `verify(proof, settings)` (oracle; test infrastructure only).

Follows `crates/verifiers/rust/src/verifier.rs:21-143` (replay commitments, `log_sum_valid`
`crates/air/src/utils.rs:29-57`, then `stwo::core::verifier::verify`), with stwo's verifier
restated from SURVEY.md Appendix A.3-A.8 / Appendix B.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import air
from .air import COMPONENTS
from .blake2s import blake2s
from .channel import Blake2sChannel, ProtocolVariant
from .circle import (CanonicCoset, Coset, LineDomain, bit_reverse_index, point_of_index, qp_add, qp_from_m,
                     subgroup_gen_index, ORDER)
from .field import P, QM31, ONE, ZERO, m_inv
from .merkle import verify_decommitment
from .proof import LuminairProof


class VerificationError(Exception):
    pass


def eval_composition_at_point(instances, sampled_values, oods, elems, comp_alpha: QM31) -> QM31:
    """Σ_k c_k(oods)/Z_k(oods) * alpha^(N-1-k) from the sampled mask values (A.7)."""
    acc = ZERO
    for ci in instances:
        comp = ci.comp
        main = [sampled_values[1][i][0] for i in range(*ci.main_span)]
        cons = list(comp.local(main))
        n_rel = len(comp.relations)
        prev = ZERO
        shift = ci.claimed_sum / QM31((1 << ci.log_size) % P)
        pre = [sampled_values[0][i][0] for i in ci.pre_idx]
        for j, rel in enumerate(comp.relations):
            cols = [sampled_values[2][ci.inter_span[0] + 4 * j + t] for t in range(4)]
            z, alpha_rel = elems[rel.elems]
            src = pre if rel.pre else main
            den = src[rel.val] - z
            if rel.id is not None:
                den = den + alpha_rel * src[rel.id]
            num = -main[rel.mult] if rel.neg else main[rel.mult]
            if j < n_rel - 1:
                cur = QM31.from_partial_evals([c[0] for c in cols])
                diff = cur - prev
            else:
                prev_row = QM31.from_partial_evals([c[0] for c in cols])
                cur = QM31.from_partial_evals([c[1] for c in cols])
                diff = cur - prev_row - prev + shift
            cons.append(diff * den - num)
            prev = cur
        x = oods[0]
        for _ in range(ci.log_size - 1):
            x = x * x * 2 - 1
        zinv = x.inverse()
        # kernel-slot values -> the protocol's constraint list (constraint-form bits: slots added / dropped, signs)
        n_proto, proto_index, neg = air.constraint_layout(comp, ci.flags)
        proto = [ZERO] * n_proto
        for c, pi, ng in zip(cons, proto_index, neg):
            if pi is not None:
                proto[pi] = -c if ng else c
        for c in proto:
            acc = acc * comp_alpha + c * zinv
    return acc


def _preprocessed_ids(claim):
    """Tree-0 layout implied by the claim: [(column id, log size)] of the LUT columns the present lookup
    components read (a LUT column has the log size of its lookup component), PreProcessedTrace order."""
    size = {}
    for kind, ls in enumerate(claim):
        if ls is not None:
            for pc in COMPONENTS[kind].pre_cols:
                size[pc] = ls
    pre_ids = [(cid, size[cid]) for cid in air.PREPROCESSED_ORDER if cid in size]
    pre_ids.sort(key=lambda pc: -pc[1])
    return pre_ids


def _instances_from_claim(claim, iclaim, flags=0):
    from .prover import ComponentInstance
    pre_ids = _preprocessed_ids(claim)
    inst = []
    m_off = i_off = 0
    for kind, ls in enumerate(claim):
        if ls is None:
            continue
        comp = COMPONENTS[kind]
        ni = 4 * len(comp.relations)
        inst.append(ComponentInstance(comp, ls, (m_off, m_off + comp.n_cols), (i_off, i_off + ni), iclaim[kind],
                                      tuple([cid for cid, _ in pre_ids].index(pc) for pc in comp.pre_cols), int(flags)))
        m_off += comp.n_cols
        i_off += ni
    return inst


#: PcsConfig::default() (crates/verifiers/rust/src/verifier.rs:36): (pow_bits, log_blowup, log_last_layer, n_queries)
DEFAULT_PCS_CONFIG = (5, 1, 0, 3)


def verify(proof: LuminairProof, variant: ProtocolVariant = ProtocolVariant.KAT, config=DEFAULT_PCS_CONFIG) -> None:
    """Raises VerificationError on failure.  `config` is the VERIFIER's PcsConfig: the reference builds it itself
    and never reads it from the proof, so a proof announcing other security parameters is rejected."""
    from .prover import (draw_queries, fold_positions, quotient_batches)
    s = proof.proof
    lb = s.log_blowup
    channel = Blake2sChannel(variant)
    inst = _instances_from_claim(proof.claim, proof.interaction_claim, int(variant))
    if not inst:
        raise VerificationError("empty claim")
    if len(s.commitments) != 4:
        raise VerificationError("expected 4 commitments")
    if (s.pow_bits, s.log_blowup, s.log_last_layer, s.n_queries) != tuple(config):
        raise VerificationError("proof was made for a different PCS config than the verifier's")
    if not (0 < s.n_queries <= 1024) or s.log_last_layer > 10 or not (1 <= s.log_blowup <= 3):
        raise VerificationError("bad PCS config")
    if s.last_layer_log_size != s.log_last_layer:
        raise VerificationError("last layer degree bound")
    if any(ls is not None and not (4 <= ls <= 26) for ls in proof.claim):
        raise VerificationError("bad log_size")
    channel.mix_root(s.commitments[0])
    for ls in proof.claim:
        if ls is not None:
            channel.mix_u64(ls)
    channel.mix_root(s.commitments[1])
    z, alpha_rel = channel.draw_felts(2)
    n_lut_rel = 4 if int(variant) & ProtocolVariant.LUT_DRAWS4 else 1
    from .prover import relation_elements
    lut_draws = [tuple(channel.draw_felts(2)) for _ in range(n_lut_rel)]
    elems = relation_elements((z, alpha_rel), lut_draws)
    # log_sum_valid (verifier.rs:97-99)
    tot = ZERO
    for c in proof.interaction_claim:
        if c is not None:
            tot = tot + c
    if not tot.is_zero():
        raise VerificationError("InvalidLogUp")
    for c in proof.interaction_claim:
        if c is not None:
            channel.mix_felts([c])
    channel.mix_root(s.commitments[2])
    comp_alpha = channel.draw_felt()
    channel.mix_root(s.commitments[3])
    t = channel.draw_felt()
    t2 = t * t
    inv = (t2 + 1).inverse()
    oods = ((ONE - t2) * inv, t.double() * inv)

    # column log sizes per tree
    main_sizes, inter_sizes = [], []
    for ci in inst:
        main_sizes += [ci.log_size] * ci.comp.n_cols
        inter_sizes += [ci.log_size] * (4 * len(ci.comp.relations))
    comp_log = max(ci.log_size for ci in inst) + 1
    pre_ids = _preprocessed_ids(proof.claim)
    tree_sizes = [[ls for _, ls in pre_ids], main_sizes, inter_sizes, [comp_log] * 4]
    # sample points
    pts = [[[oods]] * len(pre_ids), [[oods]] * len(main_sizes), [], [[oods]] * 4]
    for ci in inst:
        step = qp_from_m(point_of_index(-subgroup_gen_index(ci.log_size) % ORDER))
        prev_pt = qp_add(oods, step)
        n_i = 4 * len(ci.comp.relations)
        pts[2] += [[oods]] * (n_i - 4) + [[prev_pt, oods]] * 4
    sv = s.sampled_values
    for ti in range(4):
        if len(sv[ti]) != len(tree_sizes[ti]) or any(len(c) != len(p) for c, p in zip(sv[ti], pts[ti])):
            raise VerificationError("sampled values shape")
    # OODS composition identity
    lhs = QM31.from_partial_evals([sv[3][k][0] for k in range(4)])
    rhs = eval_composition_at_point(inst, sv, oods, elems, comp_alpha)
    if lhs != rhs:
        raise VerificationError("OodsNotMatching")
    channel.mix_felts([v for ts in sv for col in ts for v in col])
    quot_alpha = channel.draw_felt()

    # FRI commit phase replay
    channel.mix_root(s.first_layer.commitment)
    lde_sizes = sorted({ls + lb for ts in tree_sizes for ls in ts}, reverse=True)
    max_log = lde_sizes[0]
    alphas = [channel.draw_felt()]
    for l in s.inner_layers:
        channel.mix_root(l.commitment)
        alphas.append(channel.draw_felt())
    n_inner = max_log - 1 - (s.log_last_layer + lb)
    if len(s.inner_layers) != n_inner:
        raise VerificationError("inner layer count")
    if len(s.last_layer_coeffs) != 1 << s.log_last_layer:
        raise VerificationError("last layer degree")
    channel.mix_felts(s.last_layer_coeffs)
    # PoW
    if not channel.verify_pow_nonce(s.pow_bits, s.proof_of_work):
        raise VerificationError("ProofOfWork")
    channel.mix_u64(s.proof_of_work)
    queries = draw_queries(channel, max_log, s.n_queries)
    pos_by_log = {ls: fold_positions(queries, max_log - ls) for ls in lde_sizes}

    # trace decommitments
    for ti in range(4):
        qmap = {ls + lb: pos_by_log[ls + lb] for ls in set(tree_sizes[ti])}
        ok = verify_decommitment(s.commitments[ti], [ls + lb for ls in tree_sizes[ti]], qmap,
                                 s.queried_values[ti], s.decommitments[ti].hash_witness,
                                 s.decommitments[ti].column_witness)
        if not ok:
            raise VerificationError("Merkle tree %d" % ti)

    # quotient values at the queried positions (fri_answers)
    flat_sizes, flat_samples, col_reader = [], [], []
    for ti in range(4):
        # queried_values[ti]: layer by layer (size desc), node-ascending, columns of that layer in order
        sizes = [ls + lb for ls in tree_sizes[ti]]
        offs, off = {}, 0
        for ls in sorted(set(sizes), reverse=True):
            ncol = sum(1 for x in sizes if x == ls)
            offs[ls] = (off, ncol)
            off += ncol * len(pos_by_log[ls])
        seen_in_size = {}
        for cidx, ls in enumerate(sizes):
            j = seen_in_size.get(ls, 0)
            seen_in_size[ls] = j + 1
            base, ncol = offs[ls]
            col_reader.append((ti, base, ncol, j))
            flat_sizes.append(ls)
            flat_samples.append([(p, v) for p, v in zip(pts[ti][cidx], sv[ti][cidx])])
    quot_at = {}  # log -> {position: QM31}
    for ls in lde_sizes:
        idx = [i for i, x in enumerate(flat_sizes) if x == ls]
        batches = quotient_batches([flat_samples[i] for i in idx])
        dom = CanonicCoset(ls).circle_domain()
        vals = {}
        for qi, pos in enumerate(pos_by_log[ls]):
            px, py = dom.at(bit_reverse_index(pos, ls))
            fvals = []
            for i in idx:
                ti, base, ncol, j = col_reader[i]
                fvals.append(s.queried_values[ti][base + qi * ncol + j])
            acc = ZERO
            for (pt, cols_vals) in batches:
                alpha = ONE
                num = ZERO
                for (ci_, val) in cols_vals:
                    alpha = alpha * quot_alpha
                    a = val.conj() - val
                    cc = pt[1].conj() - pt[1]
                    b = val * cc - a * pt[1]
                    num = num + alpha * (cc * fvals[ci_] - (a * py + b))
                prx, pix = pt[0].v[0:2], pt[0].v[2:4]
                pry, piy = pt[1].v[0:2], pt[1].v[2:4]
                dx = ((prx[0] - px) % P, prx[1])
                dy = ((pry[0] - py) % P, pry[1])
                den = ((dx[0] * piy[0] - dx[1] * piy[1]) - (dy[0] * pix[0] - dy[1] * pix[1]),
                       (dx[0] * piy[1] + dx[1] * piy[0]) - (dy[0] * pix[1] + dy[1] * pix[0]))
                n = m_inv(den[0] * den[0] + den[1] * den[1])
                di = QM31(den[0] * n, -den[1] * n, 0, 0)
                acc = acc * (quot_alpha ** len(cols_vals)) + num * di
            vals[pos] = acc
        quot_at[ls] = vals

    # FRI first layer: rebuild sibling pairs, check Merkle, fold into line
    fw = iter(s.first_layer.fri_witness)
    dec_by_log, first_vals, queried_first = {}, {}, []
    try:
        for ls in lde_sizes:
            dpos, vals = [], {}
            for pos in pos_by_log[ls]:
                pass
            qp = pos_by_log[ls]
            i = 0
            while i < len(qp):
                start = (qp[i] >> 1) << 1
                subset = []
                while i < len(qp) and (qp[i] >> 1) << 1 == start:
                    subset.append(qp[i])
                    i += 1
                for pos in (start, start + 1):
                    dpos.append(pos)
                    vals[pos] = quot_at[ls][pos] if pos in subset else next(fw)
            dec_by_log[ls] = dpos
            first_vals[ls] = vals
    except StopIteration:
        raise VerificationError("first layer witness too short")
    if any(True for _ in fw):
        raise VerificationError("first layer witness too long")
    col_sizes = [ls for ls in lde_sizes for _ in range(4)]
    qv = []
    for ls in lde_sizes:  # layer by layer, node ascending, 4 coords
        for pos in dec_by_log[ls]:
            qv += list(first_vals[ls][pos].v)
    if not verify_decommitment(s.first_layer.commitment, col_sizes, dec_by_log, qv,
                               s.first_layer.decommitment.hash_witness, s.first_layer.decommitment.column_witness):
        raise VerificationError("FRI first layer Merkle")

    def fold_circle(vals, ls, alpha):
        dom = CanonicCoset(ls).circle_domain()
        out = {}
        for pos in sorted(vals):
            if pos & 1:
                continue
            a, b = vals[pos], vals[pos + 1]
            y = dom.at(bit_reverse_index(pos, ls))[1]
            out[pos >> 1] = (a + b) + alpha * ((a - b) * m_inv(y))
        return out

    layer_log = max_log - 1
    cur = fold_circle(first_vals[max_log], max_log, alphas[0])   # dst = 0*a^2 + fold
    line_dom = LineDomain(Coset.half_odds(layer_log))
    sizes_left = [ls for ls in lde_sizes if ls != max_log]
    lq = fold_positions(queries, 1)
    for li, l in enumerate(s.inner_layers):
        # sparse evaluation: need sibling pairs from witness
        fw = iter(l.fri_witness)
        dpos, vals = [], {}
        i = 0
        try:
            while i < len(lq):
                start = (lq[i] >> 1) << 1
                subset = []
                while i < len(lq) and (lq[i] >> 1) << 1 == start:
                    subset.append(lq[i])
                    i += 1
                for pos in (start, start + 1):
                    dpos.append(pos)
                    vals[pos] = cur[pos] if pos in subset else next(fw)
        except StopIteration:
            raise VerificationError("inner layer witness too short")
        if any(True for _ in fw):
            raise VerificationError("inner layer witness too long")
        qv = []
        for pos in dpos:
            qv += list(vals[pos].v)
        if not verify_decommitment(l.commitment, [layer_log] * 4, {layer_log: dpos}, qv,
                                   l.decommitment.hash_witness, l.decommitment.column_witness):
            raise VerificationError("FRI inner layer %d Merkle" % li)
        alpha = alphas[li + 1]
        nxt = {}
        for pos in dpos:
            if pos & 1:
                continue
            a, b = vals[pos], vals[pos + 1]
            x = line_dom.at(bit_reverse_index(pos, layer_log))
            nxt[pos >> 1] = (a + b) + alpha * ((a - b) * m_inv(x))
        line_dom = line_dom.double()
        layer_log -= 1
        lq = fold_positions(lq, 1)
        for ls in list(sizes_left):
            if ls - 1 == layer_log:
                fc = fold_circle(first_vals[ls], ls, alpha)
                for pos in nxt:
                    nxt[pos] = nxt[pos] * (alpha * alpha) + fc[pos]
                sizes_left.remove(ls)
        cur = nxt
    if sizes_left:
        raise VerificationError("unconsumed FRI columns")
    # last layer: evaluate the polynomial at the remaining positions
    for pos, v in cur.items():
        x = line_dom.at(bit_reverse_index(pos, layer_log))
        acc, xp = ZERO, x
        # ordered coefficients in the basis 1, x, pi(x), x*pi(x), ...
        val = ZERO
        for j, cj in enumerate(s.last_layer_coeffs):
            term, xx, jj = cj, x, j
            while jj:
                if jj & 1:
                    term = term * xx
                xx = (2 * xx * xx - 1) % P
                jj >>= 1
            val = val + term
        if val != v:
            raise VerificationError("FRI last layer")
