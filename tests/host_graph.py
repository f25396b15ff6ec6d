"""numpy mirror of `DeviceGraph.gen_trace` for tests: walks the same node list on the host, evaluates every op on
`Fixed<12>` integers and builds each kind's trace table with the row generators of luminair_amd/synthetic.py
(themselves checked against the reference's `process_trace` layouts and, for Add / Mul, against the KAT).
Returns what the device-side producer must emit: {kind: rows}, {node_id: values}."""
from __future__ import annotations

import numpy as np

from luminair_amd import synthetic as syn
from luminair_amd.pie import TraceTableKind as K

_LUT_NAME = {int(K.Sin): ("sin", int(K.SinLookup)), int(K.Exp2): ("exp2", int(K.Exp2Lookup)),
             int(K.Log2): ("log2", int(K.Log2Lookup))}


def read_view(vals, v) -> np.ndarray:
    """the view's elements in row-major order of its shape"""
    base = vals[v.base.node_id].reshape(-1)
    idx = np.full(v.shape, v.offset, dtype=np.int64)
    for ax, (d, st) in enumerate(zip(v.shape, v.strides)):
        shp = [1] * len(v.shape)
        shp[ax] = d
        idx = idx + (np.arange(d, dtype=np.int64) * st).reshape(shp)
    return base[idx.reshape(-1)]


def host_tables(g):
    vals, tabs = {}, {}
    lut_counts, rc_counts = {}, np.zeros(256, dtype=np.int64)
    S = syn.SCALE
    for n in g.nodes:
        t, kind = n.out, n.kind
        om = 0 if t.is_output else t.consumers
        ids = [v.base.node_id for v in n.inputs]
        if kind == int(K.Inputs):
            out = n.host.astype(np.int64).reshape(-1)
            rows = syn.inputs_rows(out, t.node_id, om)
        elif kind in (int(K.Add), int(K.Mul), int(K.Rem), int(K.LessThan)):
            a, b = read_view(vals, n.inputs[0]), read_view(vals, n.inputs[1])
            if kind == int(K.Add):
                out, rows = a + b, syn.add_rows(a, b, t.node_id, ids[0], ids[1], (-1, -1, om))
            elif kind == int(K.Mul):
                out, rows = (a * b) >> 12, syn.mul_rows(a, b, t.node_id, ids[0], ids[1], (-1, -1, om))
            elif kind == int(K.Rem):
                out, rows = a % b, syn.rem_rows(a, b, t.node_id, ids[0], ids[1], (-1, -1, om))
            else:
                out = np.where(a < b, S, 0)
                rows, c = syn.less_than_rows(a, b, t.node_id, ids[0], ids[1], (-1, -1, om))
                rc_counts += c
        elif kind == int(K.Recip):
            a = read_view(vals, n.inputs[0])
            out, rows = (S * S) // a, syn.recip_rows(a, t.node_id, ids[0], (-1, om))
        elif kind == int(K.Sqrt):
            a = read_view(vals, n.inputs[0])
            out = np.array([int(np.floor(np.sqrt(float(x * S)))) for x in a], dtype=np.int64)
            out = np.where(out * out > a * S, out - 1, out)
            out = np.where((out + 1) * (out + 1) <= a * S, out + 1, out)
            rows = syn.sqrt_rows(a, t.node_id, ids[0], (-1, om))
        elif kind == int(K.Contiguous):
            v = n.inputs[0]
            out = read_view(vals, v)
            if v.expansion == 1:
                rows = syn.contiguous_rows_ref(vals[v.base.node_id], out, t.node_id, ids[0], -1, om)
            else:
                rows = syn.contiguous_rows(out, t.node_id, ids[0], -1, om)
        elif kind in (int(K.SumReduce), int(K.MaxReduce)):
            a = n.inputs[0].base
            x = vals[a.node_id].reshape(a.shape)
            x2 = np.moveaxis(x, n.axis, -1).reshape(-1, a.shape[n.axis])      # (front*back, dim), output order
            if kind == int(K.SumReduce):
                out, rows = x2.sum(axis=1), syn.sum_reduce_rows(x2, t.node_id, ids[0], -1, om)
            else:
                out, rows = x2.max(axis=1), syn.max_reduce_rows(x2, t.node_id, ids[0], -1, om)
        elif kind in _LUT_NAME:
            name, lk = _LUT_NAME[kind]
            lo, hi, _ = g.luts[name]
            a = read_view(vals, n.inputs[0])
            rows, c = syn.unary_lut_rows(name, a, lo, t.node_id, ids[0], (-1, om))
            out = np.rint(syn._LUT_FN[name](a / S) * S).astype(np.int64)
            ranges = g.lut_ranges.get(name) or [(lo, hi)]
            n_vals = sum(b - a_ + 1 for a_, b in ranges)
            cnt = lut_counts.setdefault(lk, np.zeros(n_vals, dtype=np.int64))
            # LookupLayout::find_index (preprocessed.rs:60-77): values of earlier ranges first
            base = 0
            for a_, b in ranges:
                sel = a[(a >= a_) & (a <= b)]
                cnt[base:base + (b - a_ + 1)] += np.bincount(sel - a_, minlength=b - a_ + 1)
                base += b - a_ + 1
        else:
            raise ValueError("host mirror: kind %d" % kind)
        vals[t.node_id] = np.asarray(out, dtype=np.int64).reshape(-1)
        tabs.setdefault(kind, []).append(rows)
    tables = {k: np.concatenate(v) for k, v in tabs.items()}
    for lk, cnt in lut_counts.items():
        name = {v[1]: v[0] for v in _LUT_NAME.values()}[lk]
        tables[lk] = syn.lut_lookup_rows(cnt, len(g.luts[name][2][0]))
    if int(K.LessThan) in tabs:
        tables[int(K.RangeCheckLookup)] = syn.range_check_lookup_rows(rc_counts)
    return tables, vals
