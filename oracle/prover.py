"""CPU restatement of `prove(pie, settings)` (oracle; test infrastructure only).

Follows `crates/prover/src/prover.rs:28-319` step by step; everything stwo does behind
`tree_builder.commit` (`:59,179,298`) and `stwo::prover::prove` (`:312`) is restated from
SURVEY.md Appendix A (stwo @0790eba4 is un-vendored): commitment (A.4/A.5), logup (A.6),
composition (A.7), OODS sampling + FRI quotients + FRI + PoW + decommitment (A.8), wire
format (A.9).  Pinned bit-for-bit against `tests/golden/kat_simple/proof` in the KAT variant.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import air
from .air import COMPONENTS, Component, MV
from .channel import Blake2sChannel, ProtocolVariant
from .circle import (CanonicCoset, Coset, LineDomain, bit_reverse_indices, coset_order_storage_indices,
                     coset_vanishing_x, point_of_index, qp_add, qp_from_m, subgroup_gen_index, ORDER)
from .fft import evaluate, interpolate, eval_at_point, line_interpolate
from .field import (P, U64, QM31, ONE, ZERO, m_inv_vec, m_mul, m_sub, q_add, q_const, q_from_m, q_inv, q_mul,
                    q_mul_c, q_mul_m, q_sub, q_to_scalar, c_inv, c_mul, m_add, m_neg)
from .merkle import MerkleTree
from .proof import Decommitment, FriLayerProof, LuminairProof, StarkProof


class ProvingError(Exception):
    """Mirrors `LuminairError` variants (`crates/utils/src/lib.rs:5-34`) by message."""


@dataclass
class PcsConfig:
    """`PcsConfig::default()` (`prover.rs:36`), KAT-confirmed: 5 / 1 / 0 / 3."""
    pow_bits: int = 5
    log_blowup: int = 1
    log_last_layer: int = 0
    n_queries: int = 3


# ----------------------------------------------------------------------------- commitment scheme
class CommittedTree:
    def __init__(self, coeffs: List[np.ndarray], log_blowup: int, K=None):
        K = K or NUMPY_KERNELS
        self.coeffs = list(coeffs)
        self.log_sizes = [len(c).bit_length() - 1 for c in self.coeffs]
        self.evals = K.lde(self.coeffs, self.log_sizes, log_blowup)
        self.merkle = K.merkle(self.evals)

    def root(self) -> bytes:
        return self.merkle.root()


def commit_evals(cols: Sequence[np.ndarray], log_blowup: int, K=None) -> CommittedTree:
    K = K or NUMPY_KERNELS
    return CommittedTree(K.interpolate_cols(cols), log_blowup, K)


# ----------------------------------------------------------------------------- logup
def combine(z: QM31, alpha: QM31, v0: np.ndarray, v1: Optional[np.ndarray]) -> np.ndarray:
    """`Relation::combine`: sum_i alpha^i * v_i - z (A.6); width 2 (NodeElements: value, tensor id)
    or width 1 (RangeCheckLookupElements: value only, v1 = None)."""
    acc = q_from_m(v0)
    if v1 is not None:
        acc = q_add(acc, q_mul_m(q_const(alpha, v1.shape), v1))
    return q_sub(acc, q_const(z, v0.shape))


def rel_operands(rel, main_cols, pre_cols):
    """(value vector, second value vector or None, numerator vector) of one relation entry."""
    src = pre_cols if rel.pre else main_cols
    val = src[rel.val]
    idv = src[rel.id] if rel.id is not None else None
    mult = main_cols[rel.mult]
    if rel.neg:
        mult = (U64(P) - np.asarray(mult, dtype=U64)) % U64(P)
    return val, idv, mult


def gen_interaction_trace(comp: Component, main_cols: np.ndarray, elems, pre_cols=()):
    """-> (list of 4k base columns, claimed_sum).  Restates `write_interaction_trace`
    (`add/witness.rs:126-167`) + stwo `LogupTraceGenerator::{write_frac,finalize_col,finalize_last}`.
    elems[i] = (z, alpha) of relation element set i; pre_cols = the component's preprocessed columns
    on the trace domain."""
    n = main_cols.shape[1]
    log_size = n.bit_length() - 1
    S = np.zeros((n, 4), dtype=U64)
    ext_cols = []
    for rel in comp.relations:
        z, alpha = elems[rel.elems]
        val, idv, mult = rel_operands(rel, main_cols, pre_cols)
        den = combine(z, alpha, val, idv)
        frac = q_mul_m(q_inv(den), mult)
        S = q_add(S, frac)
        ext_cols.append(S)
    last = ext_cols[-1]
    claimed = QM31(*[int(last[:, k].sum() % P) for k in range(4)])
    shift = claimed / QM31(n % P)
    shifted = q_sub(last, q_const(shift, (n,)))
    order = coset_order_storage_indices(log_size)  # storage index of coset position i
    pref = np.cumsum(shifted[order].astype(object), axis=0) % P  # exact: python ints
    T = np.zeros_like(last)
    T[order] = pref.astype(U64)
    ext_cols[-1] = T
    base_cols = []
    for e in ext_cols:
        for k in range(4):
            base_cols.append(np.ascontiguousarray(e[:, k]))
    return base_cols, claimed


# ----------------------------------------------------------------------------- composition
@dataclass
class ComponentInstance:
    comp: Component
    log_size: int
    main_span: Tuple[int, int]    # [start, end) in tree 1
    inter_span: Tuple[int, int]   # [start, end) in tree 2
    claimed_sum: QM31
    pre_idx: Tuple[int, ...] = ()  # tree-0 column indices of the component's preprocessed columns
    flags: int = 0                 # protocol flags (constraint-form bits decide the coefficient of every constraint slot)


def prev_row_indices(log_size: int, eval_log: int) -> np.ndarray:
    """For each storage index s of the eval domain (log `eval_log`), the storage index of the
    point p_s - subgroup_gen(log_size)*G  (mask offset -1 of a log_size trace; A.2/A.6)."""
    size = 1 << eval_log
    order = coset_order_storage_indices(eval_log)
    pos = np.empty(size, dtype=np.int64)
    pos[order] = np.arange(size)
    back = 1 << (eval_log - log_size)
    return order[(pos - back) % size]


def eval_component_constraints_on_domain(ci: ComponentInstance, main_e: np.ndarray, inter_e: np.ndarray,
                                         elems, coeff_powers: List[QM31], eval_log: int, pre_e=()):
    """Σ_k c_k * coeff_powers[k] / Z on the eval domain -> (E, 4) array."""
    comp = ci.comp
    E = 1 << eval_log
    acc = np.zeros((E, 4), dtype=U64)
    cols = [MV(main_e[i]) for i in range(comp.n_cols)]
    local = comp.local(cols)
    k = 0
    for c in local:
        acc = q_add(acc, q_mul_m(q_const(coeff_powers[k], (E,)), c.a))
        k += 1
    n_rel = len(comp.relations)
    prev = np.zeros((E, 4), dtype=U64)
    shift = ci.claimed_sum / QM31((1 << ci.log_size) % P)
    for j, rel in enumerate(comp.relations):
        cur = np.stack([inter_e[4 * j + t] for t in range(4)], axis=-1)
        z, alpha_rel = elems[rel.elems]
        val, idv, mult = rel_operands(rel, main_e, pre_e)
        den = combine(z, alpha_rel, val, idv)
        num = q_from_m(mult)
        if j < n_rel - 1:
            diff = q_sub(cur, prev)
        else:
            pr = prev_row_indices(ci.log_size, eval_log)
            diff = q_add(q_sub(q_sub(cur, cur[pr]), prev), q_const(shift, (E,)))
        c = q_sub(q_mul(diff, den), num)
        acc = q_add(acc, q_mul(c, q_const(coeff_powers[k], (E,))))
        prev = cur
        k += 1
    xs, _ = CanonicCoset(eval_log).circle_domain().points_bitrev()
    zinv = m_inv_vec(coset_vanishing_x(xs, ci.log_size))
    return q_mul_m(acc, zinv)


# ----------------------------------------------------------------------------- quotients
def quotient_batches(samples: List[List[Tuple[Tuple[QM31, QM31], QM31]]]):
    """Group (column, value) by sample point in first-appearance order (`ColumnSampleBatch::new_vec`)."""
    batches, index = [], {}
    for ci, col_samples in enumerate(samples):
        for (pt, val) in col_samples:
            key = (pt[0].v, pt[1].v)
            if key not in index:
                index[key] = len(batches)
                batches.append((pt, []))
            batches[index[key]][1].append((ci, val))
    return batches


def accumulate_quotients(log_size: int, columns: List[np.ndarray], samples, random_coeff: QM31) -> np.ndarray:
    """FRI quotient secure-column over the LDE domain of `log_size` (A.8) -> (2^log_size, 4)."""
    L = 1 << log_size
    xs, ys = CanonicCoset(log_size).circle_domain().points_bitrev()
    batches = quotient_batches(samples)
    acc = np.zeros((L, 4), dtype=U64)
    for (pt, cols_vals) in batches:
        px, py = pt
        alpha = ONE
        num = np.zeros((L, 4), dtype=U64)
        for (ci, val) in cols_vals:
            alpha = alpha * random_coeff
            a = val.conj() - val
            c = py.conj() - py
            b = val * c - a * py
            a, b, c = alpha * a, alpha * b, alpha * c
            value = q_mul_m(q_const(c, (L,)), columns[ci])
            linear = q_add(q_mul_m(q_const(a, (L,)), ys), q_const(b, (L,)))
            num = q_add(num, q_sub(value, linear))
        batch_coeff = random_coeff ** len(cols_vals)
        prx, pix = np.array(px.v[0:2], dtype=U64), np.array(px.v[2:4], dtype=U64)
        pry, piy = np.array(py.v[0:2], dtype=U64), np.array(py.v[2:4], dtype=U64)
        zeros = np.zeros(L, dtype=U64)
        dx = (prx[None, :] + P - np.stack([xs, zeros], axis=-1)) % U64(P)
        dy = (pry[None, :] + P - np.stack([ys, zeros], axis=-1)) % U64(P)
        den = (c_mul(dx, np.broadcast_to(piy, (L, 2))) + P - c_mul(dy, np.broadcast_to(pix, (L, 2)))) % U64(P)
        den_inv = c_inv(den)
        acc = q_add(q_mul(acc, q_const(batch_coeff, (L,))), q_mul_c(num, den_inv))
    return acc


# ----------------------------------------------------------------------------- FRI
def fold_circle_into_line(dst: np.ndarray, src: np.ndarray, alpha: QM31, log_size: int) -> np.ndarray:
    _, ys = CanonicCoset(log_size).circle_domain().points_bitrev()
    yinv = m_inv_vec(ys[0::2])
    a, b = src[0::2], src[1::2]
    f0 = q_add(a, b)
    f1 = q_mul_m(q_sub(a, b), yinv)
    n = f0.shape[0]
    fp = q_add(f0, q_mul(f1, q_const(alpha, (n,))))
    return q_add(q_mul(dst, q_const(alpha * alpha, (n,))), fp)


def fold_line(vals: np.ndarray, alpha: QM31, domain: LineDomain) -> np.ndarray:
    xs = domain.xs_bitrev()
    xinv = m_inv_vec(xs[0::2])
    a, b = vals[0::2], vals[1::2]
    f0 = q_add(a, b)
    f1 = q_mul_m(q_sub(a, b), xinv)
    n = f0.shape[0]
    return q_add(f0, q_mul(f1, q_const(alpha, (n,))))


def secure_merkle(cols_by_coord: List[np.ndarray]) -> MerkleTree:
    """Merkle tree over secure columns: each (L,4) array contributes 4 base columns."""
    base = []
    for c in cols_by_coord:
        for k in range(4):
            base.append(np.ascontiguousarray(c[:, k]).astype(np.uint32))
    return MerkleTree(base)


def fold_positions(positions: List[int], n: int) -> List[int]:
    out = []
    for p in positions:
        q = p >> n
        if not out or out[-1] != q:
            out.append(q)
    return out


def decommit_positions_and_witness(col: np.ndarray, query_positions: List[int], fold_step: int, K=None):
    """`compute_decommitment_positions_and_witness_evals` (A.8)."""
    dec, wit = [], []
    i = 0
    while i < len(query_positions):
        start = (query_positions[i] >> fold_step) << fold_step
        subset = []
        while i < len(query_positions) and (query_positions[i] >> fold_step) << fold_step == start:
            subset.append(query_positions[i])
            i += 1
        for pos in range(start, start + (1 << fold_step)):
            dec.append(pos)
            if pos in subset:
                continue
            wit.append((K or NUMPY_KERNELS).secure_at(col, pos))
    return dec, wit


def draw_queries(channel: Blake2sChannel, log_domain_size: int, n_queries: int) -> List[int]:
    """`Queries::generate` (A.3 'queries'): n_queries draws masked to the domain, deduped + sorted."""
    qs, cnt = set(), 0
    mask = (1 << log_domain_size) - 1
    while True:
        b = channel.draw_random_bytes()
        for i in range(8):
            qs.add(int.from_bytes(b[4 * i:4 * i + 4], "little") & mask)
            cnt += 1
            if cnt == n_queries:
                return sorted(qs)


def relation_elements(node, lut_draws):
    """Element sets indexed by air.ELEMS_*: node, range_check, sin, exp2, log2.  HEAD draws sin, exp2,
    log2, range_check after NodeElements; the KAT era drew a single LUT relation (sin)."""
    if len(lut_draws) == 4:
        return [node, lut_draws[3], lut_draws[0], lut_draws[1], lut_draws[2]]
    return [node, None, lut_draws[0], None, None]


# ----------------------------------------------------------------------------- kernel sets
class NumpyKernels:
    """The per-row work of the prover as vectorised numpy (the KAT-pinned restatement).
    `oracle/cbackend.py` provides the same interface over the plain-C restatement."""
    name = "numpy"

    def interpolate_cols(self, cols):
        return [interpolate(c) for c in cols]

    def lde(self, coeffs, log_sizes, log_blowup):
        return [evaluate(c, ls + log_blowup) for c, ls in zip(coeffs, log_sizes)]

    def merkle(self, cols):
        return MerkleTree(cols)

    def gen_interaction_trace(self, comp, cols, elems, pre_cols=()):
        return gen_interaction_trace(comp, cols, elems, pre_cols)

    def composition(self, instances, tree0, tree1, tree2, elems, powers, n_total):
        sub: Dict[int, np.ndarray] = {}
        k0 = 0
        for ci in instances:
            e = ci.log_size + 1
            cp, nc = air.component_coeffs(ci.comp, ci.flags, powers, n_total, k0)
            k0 += nc
            main_e = np.stack([evaluate(tree1.coeffs[i], e) for i in range(*ci.main_span)])
            inter_e = np.stack([evaluate(tree2.coeffs[i], e) for i in range(*ci.inter_span)])
            pre_e = [evaluate(tree0.coeffs[i], e) for i in ci.pre_idx]
            val = eval_component_constraints_on_domain(ci, main_e, inter_e, elems, cp, e, pre_e)
            sub[e] = q_add(sub[e], val) if e in sub else val
        cur = None  # coefficient form, (4, 2^e)
        for e in sorted(sub):
            vals = sub[e]
            if cur is not None:
                vals = q_add(vals, evaluate(cur, e).T)
            cur = interpolate(np.ascontiguousarray(vals.T))
        return [cur[k] for k in range(4)]

    def eval_at_point(self, coeffs, pt):
        return eval_at_point(coeffs, pt)

    def accumulate_quotients(self, log_size, columns, samples, alpha):
        return accumulate_quotients(log_size, columns, samples, alpha)

    def secure_merkle(self, cols):
        return secure_merkle(cols)

    def fold_circle_into_line(self, dst, src, alpha, log_size):
        if dst is None:
            dst = np.zeros((1 << (log_size - 1), 4), dtype=U64)
        return fold_circle_into_line(dst, src, alpha, log_size)

    def fold_line(self, vals, alpha, domain):
        return fold_line(vals, alpha, domain)

    def secure_len(self, col):
        return col.shape[0]

    def secure_at(self, col, pos):
        return q_to_scalar(col[pos])


NUMPY_KERNELS = NumpyKernels()


# ----------------------------------------------------------------------------- prove
@dataclass
class ProverTrace:
    """Intermediate values kept for stage-by-stage parity tests against the HIP path."""
    roots: List[bytes] = field(default_factory=list)
    z: Optional[QM31] = None
    alpha_rel: Optional[QM31] = None
    claimed_sums: List[QM31] = field(default_factory=list)
    composition_alpha: Optional[QM31] = None
    oods_point: Optional[Tuple[QM31, QM31]] = None
    quotient_alpha: Optional[QM31] = None
    fri_alphas: List[QM31] = field(default_factory=list)
    fri_roots: List[bytes] = field(default_factory=list)
    queries: List[int] = field(default_factory=list)
    digests: Dict[str, bytes] = field(default_factory=dict)
    trees: List[CommittedTree] = field(default_factory=list)
    quotients: Dict[int, np.ndarray] = field(default_factory=dict)


def claim_slots(variant: ProtocolVariant) -> int:
    return air.N_KINDS if int(variant) & ProtocolVariant.CLAIM17 else air.N_KINDS_KAT


def prove(tables: Sequence[Tuple[int, np.ndarray]], config: PcsConfig = PcsConfig(),
          variant: ProtocolVariant = ProtocolVariant.KAT, want_trace: bool = False, kernels=None, luts=None):
    """tables: [(kind, AoS rows (n_rows, n_cols) of canonical M31)] in pie order.

    Returns LuminairProof (and a ProverTrace if want_trace)."""
    K = kernels or NUMPY_KERNELS
    tr = ProverTrace()
    channel = Blake2sChannel(variant)
    n_slots = claim_slots(variant)
    lb = config.log_blowup

    # PHASE 0: preprocessed trace (prover.rs:54-59): the LUT columns the present components use,
    # sorted by log size descending (PreProcessedTrace::new); empty for LUT-free graphs
    used = set()
    for kind, _ in tables:
        if kind in COMPONENTS:
            used.update(COMPONENTS[kind].pre_cols)
    pre_ids = [cid for cid in air.PREPROCESSED_ORDER if cid in used]
    try:
        pre_by_id = {cid: air.preprocessed_column(cid, luts) for cid in pre_ids}
    except ValueError as e:
        raise ProvingError(str(e)) from e
    pre_ids.sort(key=lambda cid: -len(pre_by_id[cid]))          # PreProcessedTrace::new: stable, size desc
    pre_evals = [pre_by_id[cid] for cid in pre_ids]
    tree0 = commit_evals(pre_evals, lb, K) if pre_evals else CommittedTree([], lb, K)
    channel.mix_root(tree0.root())
    tr.digests["root0"] = channel.digest

    # PHASE 1: main trace (pie order), prover.rs:70-179
    seen = {}
    main_cols: List[np.ndarray] = []
    pie_spans = {}
    for kind, rows in tables:
        if kind not in COMPONENTS:
            raise ProvingError("unsupported component kind %d" % kind)
        if kind in seen:
            raise ProvingError("duplicate table kind %d" % kind)
        if kind >= n_slots:
            raise ProvingError("component kind %d has no claim slot in this protocol variant" % kind)
        comp = COMPONENTS[kind]
        try:
            cols = K.pad_table(comp, rows) if hasattr(K, "pad_table") else air.pad_table(comp, rows)
        except ValueError as e:
            raise ProvingError("TraceError(EmptyTrace)") from e
        seen[kind] = (comp, cols)
        pie_spans[kind] = (len(main_cols), len(main_cols) + comp.n_cols)
        main_cols.extend(list(cols))
    if not seen:
        raise ProvingError("no trace tables")
    claim: List[Optional[int]] = [None] * n_slots
    for kind, (comp, cols) in seen.items():
        claim[kind] = cols.shape[1].bit_length() - 1
    for kind in range(n_slots):  # LuminairClaim::mix_into, struct order (lib.rs:52-104)
        if claim[kind] is not None:
            channel.mix_u64(claim[kind])
    tr.digests["claims"] = channel.digest
    tree1 = commit_evals(main_cols, lb, K)
    channel.mix_root(tree1.root())
    tr.digests["root1"] = channel.digest

    # PHASE 2: interaction trace, prover.rs:186-298
    z, alpha_rel = channel.draw_felts(2)          # NodeElements (relation!(NodeElements, 2))
    # LookupElements::draw (lookups/mod.rs:44-51): KAT era 1 LUT relation; HEAD: sin, exp2, log2, range_check
    n_lut_rel = 4 if int(variant) & ProtocolVariant.LUT_DRAWS4 else 1
    lut_draws = [tuple(channel.draw_felts(2)) for _ in range(n_lut_rel)]
    elems = relation_elements((z, alpha_rel), lut_draws)
    tr.z, tr.alpha_rel = z, alpha_rel
    inter_cols: List[np.ndarray] = []
    iclaim: List[Optional[QM31]] = [None] * n_slots
    instances: List[ComponentInstance] = []
    main_off = 0
    for kind in range(n_slots):                   # fixed struct order
        if claim[kind] is None:
            continue
        comp, cols = seen[kind]
        if any(elems[r.elems] is None for r in comp.relations):
            raise ProvingError("component needs relation elements this protocol variant does not draw")
        pre_idx = tuple(pre_ids.index(pc) for pc in comp.pre_cols)
        if any(len(pre_evals[i]) != cols.shape[1] for i in pre_idx):
            raise ProvingError("lookup table rows must match the LUT column size")
        base_cols, claimed = K.gen_interaction_trace(comp, cols, elems, [pre_evals[i] for i in pre_idx])
        iclaim[kind] = claimed
        # TraceLocationAllocator hands out spans in component (struct) order
        instances.append(ComponentInstance(comp, claim[kind], (main_off, main_off + comp.n_cols),
                                           (len(inter_cols), len(inter_cols) + len(base_cols)), claimed, pre_idx, int(variant)))
        main_off += comp.n_cols
        inter_cols.extend(base_cols)
    for kind in range(n_slots):
        if iclaim[kind] is not None:
            channel.mix_felts([iclaim[kind]])
            tr.claimed_sums.append(iclaim[kind])
    tree2 = commit_evals(inter_cols, lb, K)
    channel.mix_root(tree2.root())
    tr.digests["root2"] = channel.digest

    # stwo::prover::prove
    comp_alpha = channel.draw_felt()
    tr.composition_alpha = comp_alpha
    n_total = sum(air.constraint_layout(ci.comp, ci.flags)[0] for ci in instances)
    powers = [ONE]
    for _ in range(n_total - 1):
        powers.append(powers[-1] * comp_alpha)
    # composition: per eval-domain size accumulation (DomainEvaluationAccumulator)
    comp_coeffs = K.composition(instances, tree0, tree1, tree2, elems, powers, n_total)
    tree3 = CommittedTree(comp_coeffs, lb, K)
    channel.mix_root(tree3.root())
    tr.digests["root3"] = channel.digest
    trees = [tree0, tree1, tree2, tree3]
    tr.trees = trees
    tr.roots = [t.root() for t in trees]

    # OODS point
    t = channel.draw_felt()
    t2 = t * t
    inv = (t2 + 1).inverse()
    oods = ((ONE - t2) * inv, t.double() * inv)
    tr.oods_point = oods

    # mask points: offset 0 everywhere, [-1, 0] on the last logup column of each component
    sample_points: List[List[List[Tuple[QM31, QM31]]]] = [[[oods]] * len(pre_ids), [None] * len(main_cols),
                                                          [None] * len(inter_cols), [[oods]] * 4]
    for ci in instances:
        for i in range(*ci.main_span):
            sample_points[1][i] = [oods]
        step = qp_from_m(point_of_index(-subgroup_gen_index(ci.log_size) % ORDER))
        prev_pt = qp_add(oods, step)
        for i in range(*ci.inter_span):
            last4 = i >= ci.inter_span[1] - 4
            sample_points[2][i] = [prev_pt, oods] if last4 else [oods]
    if any(p is None for p in sample_points[1]):
        raise ProvingError("ConstraintsNotSatisfied")  # span mismatch (pie order != component order)

    samples = []   # per tree, per column: [(point, value)]
    for ti, tree in enumerate(trees):
        ts = []
        for cidx, coeffs in enumerate(tree.coeffs):
            ts.append([(pt, K.eval_at_point(coeffs, pt)) for pt in sample_points[ti][cidx]])
        samples.append(ts)
    sampled_values = [[[v for (_, v) in col] for col in ts] for ts in samples]
    channel.mix_felts([v for ts in sampled_values for col in ts for v in col])
    tr.digests["sampled"] = channel.digest

    # sanity: composition OODS eval must match the constraints at the sampled values
    from .verifier import eval_composition_at_point
    lhs = QM31.from_partial_evals([sampled_values[3][k][0] for k in range(4)])
    rhs = eval_composition_at_point(instances, sampled_values, oods, elems, comp_alpha)
    if lhs != rhs:
        raise ProvingError("ProverError(ConstraintsNotSatisfied)")

    # FRI quotients, grouped by LDE log size (descending)
    quot_alpha = channel.draw_felt()
    tr.quotient_alpha = quot_alpha
    flat_cols, flat_samples = [], []
    for ti, tree in enumerate(trees):
        for cidx in range(len(tree.coeffs)):
            flat_cols.append(tree.evals[cidx])
            flat_samples.append(samples[ti][cidx])
    sizes = sorted({len(c).bit_length() - 1 for c in flat_cols}, reverse=True)
    quotients = []
    for ls in sizes:
        idx = [i for i, c in enumerate(flat_cols) if len(c) == 1 << ls]
        qv = K.accumulate_quotients(ls, [flat_cols[i] for i in idx], [flat_samples[i] for i in idx], quot_alpha)
        quotients.append((ls, qv))
        tr.quotients[ls] = qv

    # FRI commit
    first_tree = K.secure_merkle([q for _, q in quotients])
    channel.mix_root(first_tree.root())
    tr.fri_roots.append(first_tree.root())
    folding_alpha = channel.draw_felt()
    tr.fri_alphas.append(folding_alpha)
    qi = 0
    ls0, q0 = quotients[0]
    layer_log = ls0 - 1
    layer = K.fold_circle_into_line(None, q0, folding_alpha, ls0)
    line_dom = LineDomain(Coset.half_odds(layer_log))
    qi = 1
    inner = []
    last_size = 1 << (config.log_last_layer + lb)
    if K.secure_len(layer) < last_size:
        # stwo's commit_last_layer asserts len == last_layer_domain_size: the reference panics here
        raise ProvingError("FRI: first line layer smaller than the last layer (largest table < 2^log_last_layer rows)")
    while K.secure_len(layer) > last_size:
        mt = K.secure_merkle([layer])
        channel.mix_root(mt.root())
        tr.fri_roots.append(mt.root())
        folding_alpha = channel.draw_felt()
        tr.fri_alphas.append(folding_alpha)
        inner.append((layer, mt, layer_log))
        layer = K.fold_line(layer, folding_alpha, line_dom)
        line_dom = line_dom.double()
        layer_log -= 1
        while qi < len(quotients) and quotients[qi][0] - 1 == layer_log:
            layer = K.fold_circle_into_line(layer, quotients[qi][1], folding_alpha, quotients[qi][0])
            qi += 1
    if qi != len(quotients):
        raise ProvingError("FRI: unconsumed columns")
    coeffs = line_interpolate([K.secure_at(layer, i) for i in range(K.secure_len(layer))], line_dom)
    bound = 1 << config.log_last_layer
    if any(not c.is_zero() for c in coeffs[bound:]):
        raise ProvingError("FRI: invalid degree")
    last_coeffs = coeffs[:bound]
    channel.mix_felts(last_coeffs)
    tr.digests["before_pow"] = channel.digest

    # proof of work
    nonce = channel.grind(config.pow_bits)
    channel.mix_u64(nonce)

    # queries + decommitment
    queries = draw_queries(channel, ls0, config.n_queries)
    tr.queries = queries
    positions_by_log = {ls: fold_positions(queries, ls0 - ls) for ls in sizes}
    # first FRI layer
    fw, dec_by_log = [], {}
    for ls, qv in quotients:
        dpos, wit = decommit_positions_and_witness(qv, positions_by_log[ls], 1, K)
        dec_by_log[ls] = dpos
        fw.extend(wit)
    _, hw, cw = first_tree.decommit(dec_by_log)
    first_proof = FriLayerProof(fw, Decommitment(hw, cw), first_tree.root())
    inner_proofs = []
    lq = fold_positions(queries, 1)
    for (vals, mt, llog) in inner:
        dpos, wit = decommit_positions_and_witness(vals, lq, 1, K)
        _, hw, cw = mt.decommit({llog: dpos})
        inner_proofs.append(FriLayerProof(wit, Decommitment(hw, cw), mt.root()))
        lq = fold_positions(lq, 1)
    # trace trees
    decommitments, queried_values = [], []
    for tree in trees:
        qmap = {}
        for ls in set(tree.log_sizes):
            qmap[ls + lb] = positions_by_log[ls + lb]
        qv, hw, cw = tree.merkle.decommit(qmap)
        queried_values.append(qv)
        decommitments.append(Decommitment(hw, cw))

    proof = LuminairProof(claim, iclaim, StarkProof(
        config.pow_bits, lb, config.log_last_layer, config.n_queries, [t.root() for t in trees], sampled_values,
        decommitments, queried_values, nonce, first_proof, inner_proofs, last_coeffs, config.log_last_layer))
    return (proof, tr) if want_trace else proof
