"""Kernel set over the plain-C restatement (oracle/c/stark_kernels.c) — test infrastructure only.

Same interface as `oracle.prover.NumpyKernels`; columns are uint32 arrays and secure columns are
coordinate-major (4, L) uint32 arrays.  Every C function is cross-checked against the numpy
restatement (which is pinned on the reference's known-answer proof) in tests/test_oracle_c.py.
Used (a) to compare full-size GPU proofs byte-for-byte and (b) as bench.py's `cpu_baseline`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List

import numpy as np

from . import air
from .circle import CanonicCoset, LineDomain, coset_order_storage_indices, coset_vanishing_x
from .fft import domain_twiddles, point_mappings
from .field import P, QM31, ONE, m_inv_vec
from .merkle import MerkleTree
from .prover import prev_row_indices, quotient_batches

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "c", "liboracle_kernels.so")
U32 = np.uint32


def build(scalar: bool = False):
    """Path of the kernel library (built on demand): the 16-lane build, or with scalar=True / ORACLE_KERNELS=scalar
    the same source compiled without lanes and without the vectoriser (cpu_baseline's "port-scalar")."""
    so = _SO
    if scalar or os.environ.get("ORACLE_KERNELS") == "scalar":
        so = os.path.join(_HERE, "c", "liboracle_kernels_scalar.so")
    src = os.path.join(_HERE, "c", "stark_kernels.c")
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.run(["make", "-C", os.path.join(_HERE, "c"), "all"], check=True, capture_output=True)
    return so


def _q4(q: QM31):
    return (C.c_uint32 * 4)(*q.v)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _rows_as_block(rows):
    """2-D array over a list of equal-length rows: without a copy when the rows are consecutive rows of one array
    (what `lde` / `interpolate_cols` hand out), else stacked."""
    base = rows[0].base
    if isinstance(base, np.ndarray) and base.ndim == 2 and base.dtype == U32 and base.flags["C_CONTIGUOUS"] and \
            all(r.base is base and r.dtype == U32 for r in rows):
        stride = base.strides[0]
        i0 = (rows[0].ctypes.data - base.ctypes.data) // stride
        if all(r.ctypes.data == base.ctypes.data + (i0 + k) * stride and r.shape == (base.shape[1],) for k, r in enumerate(rows)):
            return base[i0:i0 + len(rows)]
    return np.ascontiguousarray(np.stack([np.asarray(r, dtype=U32) for r in rows]))


class CKernels:
    name = "c"

    def __init__(self, threads: int = 0, scalar: bool = False):
        self.lib = C.CDLL(build(scalar))
        if threads:
            os.environ["OMP_NUM_THREADS"] = str(threads)
        self._tw = {}

    def pad_table(self, comp, rows):
        """`air.pad_table` (AoS rows -> padded SoA columns) without the 64-bit detour: (n_cols, 2^log_size) uint32."""
        rows = np.ascontiguousarray(np.asarray(rows).reshape(-1, comp.n_cols))
        if rows.dtype != U32:
            if rows.size and int(rows.max()) >= P:
                raise ValueError("non-canonical M31 word")
            rows = rows.astype(U32)
        n = rows.shape[0]
        if n == 0:
            raise ValueError("EmptyTrace")  # TraceError::EmptyTrace, add/witness.rs:39-41
        size = max(1 << (n - 1).bit_length(), 16)
        out = np.empty((comp.n_cols, size), dtype=U32)
        pad = np.array(comp.padding, dtype=U32)
        self.lib.orc_transpose_pad(_ptr(rows), C.c_long(n), C.c_int(comp.n_cols), C.c_long(size), _ptr(pad), _ptr(out))
        return out

    # ---- twiddles (computed by the numpy restatement of the domain; uploaded as uint32 tables)
    def _twiddles(self, log_n, inverse):
        key = (log_n, inverse)
        if key not in self._tw:
            tws, itws = domain_twiddles(log_n)
            arrs = [np.ascontiguousarray(t, dtype=U32) for t in (itws if inverse else tws)]
            for a, b in zip(arrs, tws):   # forward/inverse tables must be element-wise inverses
                assert not inverse or int(a[0]) * int(b[0]) % P == 1
            ptrs = (C.c_void_p * log_n)(*[a.ctypes.data for a in arrs])
            self._tw[key] = (arrs, ptrs)
        return self._tw[key][1]

    def _fft(self, data: np.ndarray, log_n: int, inverse: bool):
        ncols = data.shape[0] if data.ndim == 2 else 1
        self.lib.orc_circle_fft(_ptr(data), C.c_long(ncols), C.c_int(log_n), self._twiddles(log_n, inverse),
                                C.c_int(1 if inverse else 0))

    def interpolate_cols(self, cols):
        cols = [np.asarray(c) for c in cols]
        out = []
        i = 0
        while i < len(cols):      # batch runs of equal size
            j = i
            while j < len(cols) and len(cols[j]) == len(cols[i]):
                j += 1
            block = np.array(_rows_as_block(cols[i:j]) if cols[i].dtype == U32 else np.stack(cols[i:j]), dtype=U32, order="C")
            self._fft(block, len(cols[i]).bit_length() - 1, True)
            out.extend(list(block))
            i = j
        return out

    def _evaluate(self, coeffs2d: np.ndarray, log_size: int) -> np.ndarray:
        n, m = coeffs2d.shape
        block = np.zeros((n, 1 << log_size), dtype=U32)
        block[:, :m] = coeffs2d
        self._fft(block, log_size, False)
        return block

    def lde(self, coeffs, log_sizes, log_blowup):
        out = []
        i = 0
        while i < len(coeffs):
            j = i
            while j < len(coeffs) and log_sizes[j] == log_sizes[i]:
                j += 1
            block = self._evaluate(np.stack([np.asarray(c, dtype=U32) for c in coeffs[i:j]]), log_sizes[i] + log_blowup)
            out.extend(list(block))
            i = j
        return out

    def _hash_rows(self, words: np.ndarray) -> np.ndarray:
        words = np.ascontiguousarray(words, dtype=U32)
        n, w = words.shape
        out = np.empty((n, 8), dtype=U32)
        self.lib.orc_blake2s_rows(_ptr(words), C.c_long(n), C.c_int(w), _ptr(out))
        return out

    def _merkle_layer(self, prev, cols, size):
        out = np.empty((size, 8), dtype=U32)
        cptr = (C.c_void_p * max(len(cols), 1))(*[c.ctypes.data for c in cols])
        self.lib.orc_merkle_layer(_ptr(prev) if prev is not None else None, cptr, C.c_int(len(cols)), C.c_long(size),
                                  _ptr(out))
        return out

    def merkle(self, cols):
        return MerkleTree(cols, layer_fn=self._merkle_layer)

    def secure_merkle(self, cols):
        base = []
        for c in cols:
            base.extend([c[k] for k in range(4)])
        return MerkleTree(base, layer_fn=self._merkle_layer)

    def _inv(self, v):
        v = np.ascontiguousarray(v, dtype=U32)
        out = np.empty_like(v)
        self.lib.orc_batch_inverse(_ptr(v), C.c_long(len(v)), _ptr(out))
        return out

    def _domain(self, log_size):
        key = ("dom", log_size)
        if key not in self._tw:
            xs, ys = CanonicCoset(log_size).circle_domain().points_bitrev()
            self._tw[key] = (np.ascontiguousarray(xs, dtype=U32), np.ascontiguousarray(ys, dtype=U32))
        return self._tw[key]

    def _rel_args(self, comp, main2d, pre_list, elems):
        """ctypes argument arrays for the relations of `comp` over the given column arrays."""
        k = len(comp.relations)
        PK = C.c_void_p * k
        keep = [np.ascontiguousarray(p, dtype=U32) for p in pre_list]
        val = PK(*[(keep[r.val] if r.pre else main2d[r.val]).ctypes.data for r in comp.relations])
        idp = PK(*[((keep[r.id] if r.pre else main2d[r.id]).ctypes.data if r.id is not None else None)
                   for r in comp.relations])
        mult = PK(*[main2d[r.mult].ctypes.data for r in comp.relations])
        zs = np.array([elems[r.elems][0].v for r in comp.relations], dtype=U32).reshape(-1)
        als = np.array([elems[r.elems][1].v for r in comp.relations], dtype=U32).reshape(-1)
        neg = (C.c_int * k)(*[1 if r.neg else 0 for r in comp.relations])
        return k, val, idp, mult, zs, als, neg, keep

    def gen_interaction_trace(self, comp, cols, elems, pre_cols=()):
        cols = np.ascontiguousarray(cols, dtype=U32)
        n = cols.shape[1]
        k, val, idp, mult, zs, als, neg, keep = self._rel_args(comp, cols, pre_cols, elems)
        out = np.empty((4 * k, n), dtype=U32)
        claimed = (C.c_uint32 * 4)()
        self.lib.orc_logup_columns(val, idp, mult, C.c_int(k), C.c_long(n), _ptr(zs), _ptr(als), neg, _ptr(out),
                                   claimed)
        cl = QM31(*claimed)
        shift = cl / QM31(n % P)
        okey = ("order", n)
        if okey not in self._tw:   # a function of the size alone, like the twiddles
            self._tw[okey] = np.ascontiguousarray(coset_order_storage_indices(n.bit_length() - 1), dtype=np.int64)
        order = self._tw[okey]
        last = out[4 * (k - 1):]
        self.lib.orc_logup_prefix(_ptr(last), C.c_long(n), _ptr(order), _q4(shift))
        return [out[i] for i in range(4 * k)], cl

    def composition(self, instances, tree0, tree1, tree2, elems, powers, n_total):
        sub = {}
        k0 = 0
        for ci in instances:
            comp = ci.comp
            e = ci.log_size + 1
            E = 1 << e
            cpq, nc = air.component_coeffs(comp, ci.flags, powers, n_total, k0)
            cp = np.array([c.v for c in cpq], dtype=U32)
            k0 += nc
            # the evaluation domain of a component is its committed LDE at blow-up 2 (stwo reuses the committed
            # evaluations when the sizes agree); other blow-ups evaluate the coefficients again
            def on_eval_domain(tree, idx):
                if all(len(tree.evals[i]) == E for i in idx):
                    return _rows_as_block([tree.evals[i] for i in idx])
                return self._evaluate(np.stack([np.asarray(tree.coeffs[i], dtype=U32) for i in idx]), e)
            main_e = on_eval_domain(tree1, range(*ci.main_span))
            inter_e = on_eval_domain(tree2, range(*ci.inter_span))
            pre_e = [on_eval_domain(tree0, [i])[0] for i in ci.pre_idx]
            pkey = ("prev", ci.log_size, e)
            if pkey not in self._tw:
                self._tw[pkey] = np.ascontiguousarray(prev_row_indices(ci.log_size, e), dtype=np.int64)
            prev = self._tw[pkey]
            zkey = ("zinv", e, ci.log_size)
            if zkey not in self._tw:
                xs, _ = self._domain(e)
                self._tw[zkey] = self._inv(coset_vanishing_x(xs.astype(np.uint64), ci.log_size))
            zinv = self._tw[zkey]
            nrel, val, idp, mult, zs, als, neg, keep = self._rel_args(comp, main_e, pre_e, elems)
            shift = ci.claimed_sum / QM31((1 << ci.log_size) % P)
            first = e not in sub
            if first:
                sub[e] = np.zeros((4, E), dtype=U32)
            self.lib.orc_composition(C.c_int(comp.kind), C.c_int(comp.n_cols), C.c_int(nrel), val, idp, mult, _ptr(zs),
                                     _ptr(als), neg, _ptr(main_e), _ptr(inter_e), C.c_long(E), _ptr(prev), _q4(shift),
                                     _ptr(cp), _ptr(zinv), _ptr(sub[e]), C.c_int(0 if first else 1))
        cur = None
        for e in sorted(sub):
            vals = sub[e]
            if cur is not None:
                ext = self._evaluate(cur, e)
                vals = ((vals.astype(np.uint64) + ext) % P).astype(U32)
            vals = np.ascontiguousarray(vals)
            self._fft(vals, e, True)
            cur = vals
        return [cur[k] for k in range(4)]

    def eval_at_point(self, coeffs, pt):
        coeffs = np.ascontiguousarray(coeffs, dtype=U32)
        n = len(coeffs).bit_length() - 1
        maps = np.array([m.v for m in point_mappings(pt[0], pt[1], n)], dtype=U32).reshape(-1, 4)
        out = (C.c_uint32 * 4)()
        self.lib.orc_eval_at_point(_ptr(coeffs), C.c_int(n), _ptr(maps), out)
        return QM31(*out)

    def accumulate_quotients(self, log_size, columns, samples, alpha):
        L = 1 << log_size
        cols = [np.ascontiguousarray(c, dtype=U32) for c in columns]
        batches = quotient_batches(samples)
        bstart, col_idx, la, lb, lc, pts, bc = [0], [], [], [], [], [], []
        for (pt, cols_vals) in batches:
            px, py = pt
            a_pow = ONE
            for (ci, val) in cols_vals:
                a_pow = a_pow * alpha
                a = val.conj() - val
                c = py.conj() - py
                b = val * c - a * py
                col_idx.append(ci)
                la.append((a_pow * a).v)
                lb.append((a_pow * b).v)
                lc.append((a_pow * c).v)
            bstart.append(len(col_idx))
            pts.append(px.v[0:2] + px.v[2:4] + py.v[0:2] + py.v[2:4])
            bc.append((alpha ** len(cols_vals)).v)
        xs, ys = self._domain(log_size)
        out = np.empty((4, L), dtype=U32)
        cptr = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
        arr = lambda v, t: np.ascontiguousarray(np.array(v, dtype=t))
        bs, cidx = arr(bstart, np.int32), arr(col_idx, np.int32)
        la_, lb_, lc_, pts_, bc_ = arr(la, U32), arr(lb, U32), arr(lc, U32), arr(pts, U32), arr(bc, U32)
        self.lib.orc_quotients(cptr, C.c_long(L), C.c_int(len(batches)), _ptr(bs), _ptr(cidx), _ptr(la_), _ptr(lb_),
                               _ptr(lc_), _ptr(pts_), _ptr(bc_), _ptr(xs), _ptr(ys), _ptr(out))
        return out

    def fold_circle_into_line(self, dst, src, alpha, log_size):
        key = ("iy", log_size)
        if key not in self._tw:
            self._tw[key] = self._inv(self._domain(log_size)[1][0::2])
        itw = self._tw[key]
        n = 1 << (log_size - 1)
        acc = 1
        if dst is None:
            dst = np.zeros((4, n), dtype=U32)
            acc = 0
        else:
            dst = np.ascontiguousarray(dst).copy()
        self.lib.orc_fold(_ptr(dst), _ptr(np.ascontiguousarray(src)), C.c_long(2 * n), _ptr(itw), _q4(alpha), C.c_int(acc))
        return dst

    def fold_line(self, vals, alpha, domain: LineDomain):
        key = ("ix", domain.log_size, domain.coset.initial_index)
        if key not in self._tw:
            self._tw[key] = self._inv(domain.xs_bitrev()[0::2])
        itw = self._tw[key]
        L = vals.shape[1]
        dst = np.empty((4, L // 2), dtype=U32)
        self.lib.orc_fold(_ptr(dst), _ptr(np.ascontiguousarray(vals)), C.c_long(L), _ptr(itw), _q4(alpha), C.c_int(0))
        return dst

    def secure_len(self, col):
        return col.shape[1]

    def secure_at(self, col, pos):
        return QM31(int(col[0, pos]), int(col[1, pos]), int(col[2, pos]), int(col[3, pos]))
