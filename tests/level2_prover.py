"""A kernel set for `oracle.prover.prove` that does ALL per-row / per-coefficient work through the level-2 C ABI on
device handles (`lmn_col_*` / `lmn_tree_*`, include/luminair_hip.h) - what a Rust `HipBackend` implementing stwo's
`Backend` traits would call (/root/reference/crates/prover/src/prover.rs:38-46,312,
/root/reference/crates/air/src/components/mod.rs:122,530, /root/reference/crates/air/src/utils.rs:112-128).

The host side (Fiat-Shamir channel, claim bookkeeping, query drawing, decommitment order, proof container) is the
oracle's: in the reference that part is stwo's host code, which stays in place when `SimdBackend` is swapped out.
`lmn_prove` is never called.  Test infrastructure (tests/test_level2_only_prove.py, tests/test_gpu_parity.py)."""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from oracle import air
from oracle.field import QM31
from oracle.merkle import MerkleTree


class DCol:
    """One column that lives in HBM: column `j` of the batch handle `h` (`lmn_col`)."""

    def __init__(self, h, j: int):
        self.h, self.j = h, j

    def __len__(self):
        return 1 << self.h.log_size


class DSecure:
    """A secure (QM31) column in HBM: an `lmn_col` with 4 coordinate columns."""

    def __init__(self, h):
        self.h = h
        self._host = None

    def __len__(self):
        return 1 << self.h.log_size

    def host(self) -> np.ndarray:
        if self._host is None:
            self._host = self.h.to_cpu()
        return self._host


class DeviceMerkle(MerkleTree):
    """`lmn_tree`: hashed on the device by `lmn_col_commit`; the (tiny) decommitment walk of the oracle reads the
    layers through `lmn_tree_layer_to_cpu` and the queried column values through `lmn_col_to_cpu`."""

    def __init__(self, ctx, handles: List):
        self.tree = ctx.commit(handles)
        cols = []
        for h in handles:
            cols.extend(list(h.to_cpu()))
        self.columns = [np.ascontiguousarray(c, dtype=np.uint32) for c in cols]
        order = sorted(range(len(self.columns)), key=lambda i: -len(self.columns[i]))
        self.sorted_columns = [self.columns[i] for i in order]
        self.layers = [np.frombuffer(b"".join(self.tree.layer(k)), dtype="<u4").reshape(1 << k, 8)
                       for k in range(self.tree.log_size + 1)]
        assert self.layers[0][0].astype("<u4").tobytes() == self.tree.root()


def _groups(cols):
    """maximal runs of DCols that are the consecutive columns of one handle -> [(handle, first, n)]"""
    out = []
    for c in cols:
        if out and out[-1][0] is c.h and out[-1][1] + out[-1][2] == c.j:
            out[-1][2] += 1
        else:
            out.append([c.h, c.j, 1])
    return out


def _whole_handles(cols):
    hs = []
    for h, first, n in _groups(cols):
        assert first == 0 and n == h.ncols, "a batch handle is always used whole"
        hs.append(h)
    return hs


def _q(words) -> QM31:
    return QM31(*[int(w) for w in words])


class Level2Kernels:
    name = "level-2 C ABI on device handles"

    def __init__(self, ctx):
        self.ctx = ctx
        self.calls = {}

    def _count(self, what):
        self.calls[what] = self.calls.get(what, 0) + 1

    # ---- PolyOps
    def interpolate_cols(self, cols):
        out, i = [], 0
        while i < len(cols):
            if isinstance(cols[i], DCol):                       # already resident (interaction trace): in place
                h = cols[i].h
                assert all(isinstance(c, DCol) and c.h is h and c.j == k for k, c in enumerate(cols[i:i + h.ncols]))
                h.interpolate()
                out += [DCol(h, k) for k in range(h.ncols)]
                i += h.ncols
            else:                                               # host trace columns: one upload per run of equal sizes
                j = i
                while j < len(cols) and not isinstance(cols[j], DCol) and len(cols[j]) == len(cols[i]):
                    j += 1
                h = self.ctx.col_from_cpu(np.stack([np.asarray(c, dtype=np.uint32) for c in cols[i:j]]))
                h.interpolate()
                out += [DCol(h, k) for k in range(j - i)]
                i = j
            self._count("interpolate")
        return out

    def lde(self, coeffs, log_sizes, log_blowup):
        out = []
        for h in _whole_handles(coeffs):
            e = h.evaluate(h.log_size + log_blowup)
            self._count("evaluate")
            out += [DCol(e, k) for k in range(e.ncols)]
        return out

    def eval_at_point(self, coeffs: DCol, pt):
        self._count("eval_at_point")
        return _q(coeffs.h.eval_at_point(coeffs.j, list(pt[0].v) + list(pt[1].v)))

    # ---- MerkleOps
    def merkle(self, cols):
        if not cols:
            return MerkleTree([])                               # the empty tree: root = blake2s("")
        self._count("commit")
        return DeviceMerkle(self.ctx, _whole_handles(cols))

    def secure_merkle(self, cols: List[DSecure]):
        self._count("commit")
        return DeviceMerkle(self.ctx, [c.h for c in cols])

    # ---- constraint framework on the backend's columns
    @staticmethod
    def _elems(elems):
        return {i: (e[0].v, e[1].v) for i, e in enumerate(elems) if e is not None}

    def gen_interaction_trace(self, comp, cols, elems, pre_cols=()):
        main = self.ctx.col_from_cpu(np.stack([np.asarray(c, dtype=np.uint32) for c in cols]))
        pre = self.ctx.col_from_cpu(np.stack([np.asarray(c, dtype=np.uint32) for c in pre_cols])) if len(pre_cols) else None
        inter, claimed = self.ctx.col_logup(comp.kind, main, pre, self._elems(elems))
        self._count("logup")
        main.free()
        if pre is not None:
            pre.free()
        return [DCol(inter, k) for k in range(inter.ncols)], _q(claimed)

    def composition(self, instances, tree0, tree1, tree2, elems, powers, n_total):
        acc = {}
        k0 = 0
        for ci in instances:
            e = ci.log_size + 1
            cpq, nc = air.component_coeffs(ci.comp, ci.flags, powers, n_total, k0)
            coeffs = [c.v for c in cpq]
            k0 += nc
            if e not in acc:
                acc[e] = self.ctx.col_zeros(4, e)
            m0 = tree1.evals[ci.main_span[0]]
            i0 = tree2.evals[ci.inter_span[0]]
            main = m0.h.view(m0.j, ci.main_span[1] - ci.main_span[0])
            inter = i0.h.view(i0.j, ci.inter_span[1] - ci.inter_span[0])
            pre = None
            if ci.pre_idx:
                p0 = tree0.evals[ci.pre_idx[0]]
                assert all(tree0.evals[i].h is p0.h and tree0.evals[i].j == p0.j + k for k, i in enumerate(ci.pre_idx))
                pre = p0.h.view(p0.j, len(ci.pre_idx))
            self.ctx.col_composition(ci.comp.kind, main, inter, pre, self._elems(elems), ci.claimed_sum.v, coeffs, acc[e])
            self._count("composition")
            for v in (main, inter, pre):
                if v is not None:
                    v.free()
        cur = None   # DomainEvaluationAccumulator::finalize: fold smaller sizes into larger ones, coefficient form
        for e in sorted(acc):
            vals = acc[e]
            if cur is not None:
                ext = cur.evaluate(e)
                vals.accumulate(ext)
                ext.free()
                cur.free()
            cur = vals.interpolate()
        return [DCol(cur, k) for k in range(4)]

    # ---- QuotientOps / FriOps
    def accumulate_quotients(self, log_size, columns, samples, alpha):
        points, index, flat = [], {}, []
        for ci, col_samples in enumerate(samples):
            for (pt, val) in col_samples:
                key = (pt[0].v, pt[1].v)
                if key not in index:
                    index[key] = len(points)
                    points.append(list(pt[0].v) + list(pt[1].v))
                flat.append((ci, index[key], val.v))
        self._count("accumulate_quotients")
        return DSecure(self.ctx.col_accumulate_quotients(_whole_handles(columns), flat, points, alpha.v))

    def fold_circle_into_line(self, dst: Optional[DSecure], src: DSecure, alpha, log_size):
        if dst is None:
            dst = DSecure(self.ctx.col_zeros(4, log_size - 1))
        dst.h.fold_circle_into_line(src.h, alpha.v)
        dst._host = None
        self._count("fold_circle_into_line")
        return dst

    def fold_line(self, vals: DSecure, alpha, domain):
        self._count("fold_line")
        return DSecure(vals.h.fold_line(alpha.v))

    def secure_len(self, col: DSecure):
        return len(col)

    def secure_at(self, col: DSecure, pos):
        return _q(col.host()[:, pos])


def prove_with_level2_only(ctx, tables, variant=None, luts=None):
    """-> (proof bincode bytes, op-call counts)"""
    from oracle.channel import ProtocolVariant
    from oracle.proof import to_bincode
    from oracle.prover import prove
    K = Level2Kernels(ctx)
    proof = prove([(k, np.asarray(r).astype(np.uint64)) for k, r in tables],
                  variant=variant if variant is not None else ProtocolVariant.KAT, kernels=K, luts=luts)
    return to_bincode(proof), K.calls
