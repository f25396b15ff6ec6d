"""Synthetic `LuminairPie` generators for the BASELINE.json configs.

They emit exactly the rows `gen_trace` would record for the named graphs
(`crates/graph/src/op/prim.rs:967-1013` Add, `:1090-1139` Mul, `:388-431` Recip,
`:52-88` Inputs): AoS uint32 rows in `Column::index()` order, values = `Fixed<12>::to_m31()`.
Seeds are numpy PCG64 seeds; the default seed 42 follows the reference tests
(`crates/graph/src/tests/mod.rs:202-214`).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

P = (1 << 31) - 1
SCALE = 1 << 12
KIND_ADD, KIND_MUL, KIND_RECIP, KIND_INPUTS = 0, 1, 2, 15
KIND_SUM_REDUCE, KIND_MAX_REDUCE, KIND_CONTIGUOUS = 5, 6, 16
KIND_LESS_THAN, KIND_RANGE_CHECK_LOOKUP = 13, 14
KIND_SQRT, KIND_REM = 7, 8


def to_m31(v: np.ndarray) -> np.ndarray:
    """Fixed<12>::to_m31: negatives map to P - |v|."""
    v = np.asarray(v, dtype=np.int64)
    return np.where(v >= 0, v, P + v).astype(np.uint32)


def _ids(n, node, a, b=None):
    idx = np.arange(n, dtype=np.int64)
    last = (idx == n - 1).astype(np.int64)
    cols = [np.full(n, node), np.full(n, a)]
    if b is not None:
        cols.append(np.full(n, b))
    cols += [idx, last, np.full(n, node), np.full(n, a)]
    if b is not None:
        cols.append(np.full(n, b))
    cols.append(idx + 1)
    return cols


def add_rows(lhs, rhs, node=2, lhs_id=0, rhs_id=1, mults=(0, 0, 0)) -> np.ndarray:
    lhs, rhs = np.asarray(lhs, np.int64), np.asarray(rhs, np.int64)
    n = len(lhs)
    out = lhs + rhs
    cols = _ids(n, node, lhs_id, rhs_id) + [to_m31(lhs), to_m31(rhs), to_m31(out)]
    cols += [np.full(n, m % P) for m in mults]
    return np.stack([np.asarray(c, dtype=np.int64) % P for c in cols], axis=1).astype(np.uint32)


def mul_rows(lhs, rhs, node=2, lhs_id=0, rhs_id=1, mults=(0, 0, 0)) -> np.ndarray:
    """Operands >= 0 (the sign convention of numerair's rem is unverified)."""
    lhs, rhs = np.asarray(lhs, np.int64), np.asarray(rhs, np.int64)
    n = len(lhs)
    prod = lhs * rhs
    out, rem = prod >> 12, prod & (SCALE - 1)
    cols = _ids(n, node, lhs_id, rhs_id) + [to_m31(lhs), to_m31(rhs), to_m31(out), to_m31(rem)]
    cols += [np.full(n, m % P) for m in mults]
    return np.stack([np.asarray(c, dtype=np.int64) % P for c in cols], axis=1).astype(np.uint32)


def recip_rows(inp, node=2, input_id=0, mults=(0, 0)) -> np.ndarray:
    inp = np.asarray(inp, np.int64)
    n = len(inp)
    out = (SCALE * SCALE) // inp
    rem = SCALE * SCALE - inp * out
    cols = _ids(n, node, input_id) + [to_m31(inp), to_m31(out), to_m31(rem), np.full(n, SCALE)]
    cols += [np.full(n, m % P) for m in mults]
    return np.stack([np.asarray(c, dtype=np.int64) % P for c in cols], axis=1).astype(np.uint32)


def inputs_rows(vals, node, multiplicity) -> np.ndarray:
    vals = np.asarray(vals, np.int64)
    n = len(vals)
    idx = np.arange(n, dtype=np.int64)
    last = (idx == n - 1).astype(np.int64)
    cols = [np.full(n, node), idx, last, np.full(n, node), idx + 1, to_m31(vals), np.full(n, multiplicity % P)]
    return np.stack([np.asarray(c, dtype=np.int64) % P for c in cols], axis=1).astype(np.uint32)


def _reduce_common(x, node, input_id):
    """x: (n_out, dim) int64.  One row per INPUT element, idx = output index (prim.rs:1486-1510)."""
    n_out, dim = x.shape
    idx = np.repeat(np.arange(n_out, dtype=np.int64), dim)
    last = (idx == n_out - 1).astype(np.int64)
    step_last = np.tile((np.arange(dim) == dim - 1).astype(np.int64), n_out)
    n = n_out * dim
    head = [np.full(n, node), np.full(n, input_id), idx, last, np.full(n, node), np.full(n, input_id), idx + 1]
    return head, step_last


def sum_reduce_rows(x, node=2, input_id=0, input_mult=-1, out_mult=0) -> np.ndarray:
    """`LuminairSumReduce::process_trace` (crates/graph/src/op/prim.rs:1514-1565)."""
    x = np.asarray(x, np.int64)
    head, step_last = _reduce_common(x, node, input_id)
    next_acc = np.cumsum(x, axis=1)
    acc = next_acc - x
    out = (next_acc * (np.arange(x.shape[1]) == x.shape[1] - 1)).reshape(-1)
    cols = head + [to_m31(x.reshape(-1)), to_m31(out), to_m31(acc.reshape(-1)), to_m31(next_acc.reshape(-1)),
                   step_last, np.full(x.size, input_mult % P), (out_mult * step_last) % P]
    return np.stack([np.asarray(c, dtype=np.int64) % P for c in cols], axis=1).astype(np.uint32)


def max_reduce_rows(x, node=2, input_id=0, input_mult=-1, out_mult=0) -> np.ndarray:
    """`LuminairMaxReduce::process_trace` (crates/graph/src/op/prim.rs:1591-1734): running max starts at
    the first element; is_max marks rows whose input becomes the new running max."""
    x = np.asarray(x, np.int64)
    head, step_last = _reduce_common(x, node, input_id)
    n_out, dim = x.shape
    next_max = np.maximum.accumulate(x, axis=1)
    mx = np.concatenate([x[:, :1], next_max[:, :-1]], axis=1)     # max_val before this step
    is_max = (x > mx).astype(np.int64)       # strict, as in prim.rs:1638-1642
    out = (next_max * (np.arange(dim) == dim - 1)).reshape(-1)
    cols = head + [to_m31(x.reshape(-1)), to_m31(out), to_m31(mx.reshape(-1)), to_m31(next_max.reshape(-1)),
                   step_last, is_max.reshape(-1), np.full(x.size, input_mult % P), (out_mult * step_last) % P]
    return np.stack([np.asarray(c, dtype=np.int64) % P for c in cols], axis=1).astype(np.uint32)


def contiguous_rows(x, node=2, input_id=0, input_mult=-1, out_mult=0) -> np.ndarray:
    """`LuminairContiguous::process_trace` (crates/graph/src/op/prim.rs:229-301): out = input."""
    x = np.asarray(x, np.int64).reshape(-1)
    n = len(x)
    idx = np.arange(n, dtype=np.int64)
    cols = [np.full(n, node), np.full(n, input_id), idx, (idx == n - 1).astype(np.int64), np.full(n, node),
            np.full(n, input_id), idx + 1, to_m31(x), to_m31(x), np.full(n, input_mult % P), np.full(n, out_mult % P)]
    return np.stack([np.asarray(c, dtype=np.int64) % P for c in cols], axis=1).astype(np.uint32)


def contiguous_rows_ref(phys, out, node=2, input_id=0, input_mult=-1, out_mult=0) -> np.ndarray:
    """`LuminairContiguous::process_trace` exactly as the reference iterates (prim.rs:253-296, zip_longest of the input
    BUFFER and the output): row idx = (idx-th buffer element or 0, idx-th output element, wrapping past the output's
    end), max(len(phys), len(out)) rows, is_last_idx on the buffer's last element."""
    phys, out = np.asarray(phys, np.int64).reshape(-1), np.asarray(out, np.int64).reshape(-1)
    n = max(len(phys), len(out))
    idx = np.arange(n, dtype=np.int64)
    inp = np.concatenate([phys, np.zeros(n - len(phys), np.int64)])
    cols = [np.full(n, node), np.full(n, input_id), idx, (idx == len(phys) - 1).astype(np.int64), np.full(n, node),
            np.full(n, input_id), idx + 1, to_m31(inp), to_m31(out[idx % len(out)]), np.full(n, input_mult % P),
            np.full(n, out_mult % P)]
    return np.stack([np.asarray(c, dtype=np.int64) % P for c in cols], axis=1).astype(np.uint32)


def sqrt_rows(inp, node=2, input_id=0, mults=(0, 0)) -> np.ndarray:
    """Fixed-point sqrt rows (`crates/graph/src/op/prim.rs:573-660`): out = floor(sqrt(input*scale)),
    rem = input*scale - out^2 (the natural identity; numerair's exact form is unpinned)."""
    inp = np.asarray(inp, np.int64)
    n = len(inp)
    out = np.floor(np.sqrt((inp * SCALE).astype(np.float64))).astype(np.int64)
    out = np.where(out * out > inp * SCALE, out - 1, out)
    out = np.where((out + 1) * (out + 1) <= inp * SCALE, out + 1, out)
    rem = inp * SCALE - out * out
    cols = _ids(n, node, input_id) + [to_m31(inp), to_m31(out), to_m31(rem), np.full(n, SCALE)]
    cols += [np.full(n, m % P) for m in mults]
    return np.stack([np.asarray(c, dtype=np.int64) % P for c in cols], axis=1).astype(np.uint32)


def rem_rows(lhs, rhs, node=2, lhs_id=0, rhs_id=1, mults=(0, 0, 0)) -> np.ndarray:
    """Remainder rows (`crates/graph/src/op/prim.rs:1323-1421`): lhs = rhs*quotient + rem, operands > 0;
    the out relation carries `rem`."""
    lhs, rhs = np.asarray(lhs, np.int64), np.asarray(rhs, np.int64)
    n = len(lhs)
    quo, rem = lhs // rhs, lhs % rhs
    cols = _ids(n, node, lhs_id, rhs_id) + [to_m31(lhs), to_m31(rhs), to_m31(rem), to_m31(quo)]
    cols += [np.full(n, m % P) for m in mults]
    return np.stack([np.asarray(c, dtype=np.int64) % P for c in cols], axis=1).astype(np.uint32)


def sqrt_rem_graph(n: int, seed: int = 42) -> List[Tuple[int, np.ndarray]]:
    """s = sqrt(a); r = s % m with fresh inputs a, m (PINNED variant: Sqrt, Rem, Inputs tables)."""
    rng = np.random.default_rng(seed)
    a = rng.integers(1, 1 << 20, size=n)
    m = rng.integers(1, 4096, size=n)
    sq = sqrt_rows(a, node=2, input_id=0, mults=(-1, 1))
    s_out = sq[:, 8].astype(np.int64)
    rm = rem_rows(s_out, m, node=3, lhs_id=2, rhs_id=1, mults=(-1, -1, 0))
    inp = np.concatenate([inputs_rows(a, 0, 1), inputs_rows(m, 1, 1)])
    return [(KIND_SQRT, sq), (KIND_REM, rm), (KIND_INPUTS, inp)]


_LUT_FN = {"sin": np.sin, "exp2": np.exp2, "log2": np.log2}
_LUT_KINDS = {"sin": (3, 4), "exp2": (9, 10), "log2": (11, 12)}   # (component kind, lookup kind)


def make_lut(name: str, lo: int, hi: int):
    """LUT columns for one value range, the way `SinPreProcessed::gen_column` lays them out
    (crates/air/src/preprocessed.rs:351-383): the fixed-point values lo..=hi ascending in column 0,
    round(f(v/scale)*scale) in column 1, zero rows up to the next power of two (at least 16 rows).
    Returns (col0, col1) as uint32 M31 words."""
    v = np.arange(lo, hi + 1, dtype=np.int64)
    out = np.rint(_LUT_FN[name](v / SCALE) * SCALE).astype(np.int64)
    log = max(4, int(len(v) - 1).bit_length())
    pad = (1 << log) - len(v)
    col0 = np.concatenate([to_m31(v), np.zeros(pad, np.int64)]).astype(np.uint32)
    col1 = np.concatenate([to_m31(out), np.zeros(pad, np.int64)]).astype(np.uint32)
    return col0, col1


def unary_lut_rows(name: str, inp, lo: int, node=2, input_id=0, mults=(0, 0)):
    """Sin / Exp2 / Log2 rows (`crates/graph/src/op/prim.rs:663-760` and siblings): out = LUT(input),
    lookup multiplicity 1 per row.  Returns (rows, LUT multiplicities indexed by input - lo)."""
    inp = np.asarray(inp, np.int64)
    n = len(inp)
    out = np.rint(_LUT_FN[name](inp / SCALE) * SCALE).astype(np.int64)
    cols = _ids(n, node, input_id) + [to_m31(inp), to_m31(out)]
    cols += [np.full(n, m % P) for m in mults] + [np.ones(n, dtype=np.int64)]
    rows = np.stack([np.asarray(c, dtype=np.int64) % P for c in cols], axis=1).astype(np.uint32)
    return rows, np.bincount(inp - lo)


def lut_lookup_rows(multiplicities, lut_len: int) -> np.ndarray:
    """`SinLookup` & co. trace table: one multiplicity per LUT row (zero on the LUT's padding rows)."""
    m = np.zeros(lut_len, dtype=np.int64)
    m[:len(multiplicities)] = multiplicities
    return m.reshape(-1, 1).astype(np.uint32)


def activation_graph(n: int, seed: int = 42, names=("sin", "exp2", "log2"), ranges=None):
    """y = f(a) for each LUT function in `names`, each on its own fresh input tensor.  Returns
    (tables, luts): the trace tables in `gen_trace` order and the LUT columns the settings carry.
    `ranges`: {name: (lo, hi)} fixed-point input range of each LUT (defaults: a few units wide, 2^14-2^16 rows)."""
    rng = np.random.default_rng(seed)
    ranges = dict({"sin": (-4 * SCALE, 4 * SCALE), "exp2": (-2 * SCALE, 2 * SCALE), "log2": (1, 4 * SCALE)}, **(ranges or {}))
    tables, luts, inputs = [], {}, []
    for t, name in enumerate(names):
        lo, hi = ranges[name]
        a = rng.integers(lo, hi + 1, size=n)
        rows, counts = unary_lut_rows(name, a, lo, node=10 + t, input_id=t, mults=(-1, 0))
        luts[name] = make_lut(name, lo, hi)
        kind, lookup_kind = _LUT_KINDS[name]
        tables += [(kind, rows), (lookup_kind, lut_lookup_rows(counts, len(luts[name][0])))]
        inputs.append(inputs_rows(a, t, 1))
    tables.append((KIND_INPUTS, np.concatenate(inputs)))
    return sorted(tables, key=lambda kt: kt[0]), luts


def less_than_rows(lhs, rhs, node=2, lhs_id=0, rhs_id=1, mults=(-1, -1, 0)):
    """`LuminairLessThan::process_trace` (crates/graph/src/op/prim.rs:1203-1295): out = 1.0 if lhs < rhs,
    diff = rhs - lhs (+ 2^31-1 when borrow), split into four 8-bit limbs that are range-checked.
    Returns (rows, limb multiplicities[256])."""
    lhs, rhs = np.asarray(lhs, np.int64), np.asarray(rhs, np.int64)
    n = len(lhs)
    lt = lhs < rhs
    out = np.where(lt, SCALE, 0)
    borrow = np.where(lt, 0, 1)
    diff = rhs - lhs + borrow * P
    limbs = [(diff >> (8 * k)) & 0xFF for k in range(4)]
    cols = _ids(n, node, lhs_id, rhs_id) + [to_m31(lhs), to_m31(rhs), to_m31(out), diff, borrow] + limbs
    cols += [np.full(n, m % P) for m in mults] + [np.ones(n, dtype=np.int64)]
    rows = np.stack([np.asarray(c, dtype=np.int64) % P for c in cols], axis=1).astype(np.uint32)
    counts = np.zeros(256, dtype=np.int64)
    for l in limbs:
        counts += np.bincount(l, minlength=256)
    return rows, counts


def range_check_lookup_rows(multiplicities) -> np.ndarray:
    """`RangeCheckLookup::add_multiplicities_to_table`: one row per LUT entry (256 for the 8-bit check)."""
    return np.asarray(multiplicities, dtype=np.int64).reshape(-1, 1).astype(np.uint32)


def less_than_graph(n: int, seed: int = 42) -> List[Tuple[int, np.ndarray]]:
    """c = a + b; d = (c < t) with fresh inputs a, b, t (PINNED variant): Add, LessThan, its
    RangeCheckLookup LUT component and the Inputs table, logup sums cancelling."""
    rng = np.random.default_rng(seed)
    a = rng.integers(-2048, 2048, size=n)
    b = rng.integers(-2048, 2048, size=n)
    t = rng.integers(-4096, 4096, size=n)
    c = a + b
    add = add_rows(a, b, node=3, lhs_id=0, rhs_id=1, mults=(-1, -1, 1))
    lt, counts = less_than_rows(c, t, node=4, lhs_id=3, rhs_id=2, mults=(-1, -1, 0))
    inp = np.concatenate([inputs_rows(a, 0, 1), inputs_rows(b, 1, 1), inputs_rows(t, 2, 1)])
    return [(KIND_ADD, add), (KIND_LESS_THAN, lt), (KIND_RANGE_CHECK_LOOKUP, range_check_lookup_rows(counts)),
            (KIND_INPUTS, inp)]


def linear_layer(n_out: int, dim: int, seed: int = 42, with_max: bool = False) -> List[Tuple[int, np.ndarray]]:
    """y = sum_k(x[k] * w[j][k]) + b[j] — the Mul + SumReduce + Add lowering of a linear layer
    (BASELINE config 5's building block), KAT-era multiplicities (initializers consumed with mult 0):
    Mul (node 3) yields each product once, SumReduce (node 4) consumes them and yields y' once,
    Add (node 5) consumes y' [optionally MaxReduce (node 6) consumes the Add outputs]."""
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 2048, size=(n_out, dim))
    w = rng.integers(0, 2048, size=(n_out, dim))
    b = rng.integers(-2048, 2048, size=n_out)
    prod = (x * w) >> 12
    y1 = prod.sum(axis=1)
    y = y1 + b
    tabs = [(KIND_ADD, add_rows(y1, b, node=5, lhs_id=4, rhs_id=9, mults=(-1, 0, 1 if with_max else 0))),
            (KIND_MUL, mul_rows(x.reshape(-1), w.reshape(-1), node=3, lhs_id=7, rhs_id=8, mults=(0, 0, 1))),
            (KIND_SUM_REDUCE, sum_reduce_rows(prod, node=4, input_id=3, input_mult=-1, out_mult=1))]
    if with_max:
        tabs.append((KIND_MAX_REDUCE, max_reduce_rows(y.reshape(1, -1), node=6, input_id=5, input_mult=-1, out_mult=0)))
    return tabs


def config5_linear_layers(n_layers: int = 256, n_out: int = 128, dim: int = 256, seed: int = 42):
    """BASELINE config 5: `n_layers` x (Mul n-row + SumReduce n-row + Add) with n = n_out*dim, laid into the three
    per-kind tables (each layer keeps its own node ids; is_last closes every node's block).  Defaults: 256 layers
    of 128x256 => Mul 2^23 + SumReduce 2^23 + Add 2^15 rows (2^24 total).  KAT-era multiplicities:
    initializers are consumed with multiplicity 0, the Mul -> SumReduce -> Add chain cancels."""
    rng = np.random.default_rng(seed)
    adds, muls, sums = [], [], []
    for l in range(n_layers):
        base = 10 * l
        x = rng.integers(0, 2048, size=(n_out, dim))
        w = rng.integers(0, 2048, size=(n_out, dim))
        b = rng.integers(-2048, 2048, size=n_out)
        prod = (x * w) >> 12
        y1 = prod.sum(axis=1)
        adds.append(add_rows(y1, b, node=base + 5, lhs_id=base + 4, rhs_id=base + 9, mults=(-1, 0, 0)))
        muls.append(mul_rows(x.reshape(-1), w.reshape(-1), node=base + 3, lhs_id=base + 7, rhs_id=base + 8,
                             mults=(0, 0, 1)))
        sums.append(sum_reduce_rows(prod, node=base + 4, input_id=base + 3, input_mult=-1, out_mult=1))
    return [(KIND_ADD, np.concatenate(adds)), (KIND_MUL, np.concatenate(muls)), (KIND_SUM_REDUCE, np.concatenate(sums))]


def config4_black_scholes_shape(batch: int = 1, seed: int = 42):
    """BASELINE config 4: the black-schole-nn *shape* (2 -> 64 -> 64 -> 1 MLP with tanh,
    examples/black-schole-nn/src/main.rs:61-103) with seeded synthetic weights, lowered the way SURVEY.md §8d
    describes: Linear = Mul + SumReduce + Add(bias); tanh(x) = 2*sigmoid(2x) - 1 with
    sigmoid(t) = Recip(1 + Exp2(-t/ln 2)).  Every tensor op is one node; multiplicities follow the KAT-era rule
    (initializers 0, each intermediate yielded once and consumed once), the Exp2 LUT relation is balanced by the
    Exp2Lookup table.  Returns (tables, luts); needs the PINNED variant (Exp2 has no KAT-era claim slot)."""
    rng = np.random.default_rng(seed)
    T = {k: [] for k in (KIND_ADD, KIND_MUL, KIND_RECIP, KIND_SUM_REDUCE, 9)}
    lo, hi = -8 * SCALE, 8 * SCALE
    counts = np.zeros(hi - lo + 1, dtype=np.int64)
    node = [100]

    def new_node():
        node[0] += 1
        return node[0]

    def linear(x, x_id, n_in, n_out, consume):
        """x: (batch, n_in) -> (batch, n_out); returns (values, node id)."""
        w = rng.integers(-1024, 1024, size=(n_out, n_in))
        b = rng.integers(-512, 512, size=n_out)
        xe = np.broadcast_to(x[:, None, :], (len(x), n_out, n_in)).reshape(-1, n_in)
        we = np.broadcast_to(w[None], (len(x), n_out, n_in)).reshape(-1, n_in)
        m_id, s_id, a_id = new_node(), new_node(), new_node()
        prod = (xe * we) >> 12
        # each x element is consumed n_out times by the expanded Mul; its producer yields it n_out times
        T[KIND_MUL].append(mul_rows(xe.reshape(-1), we.reshape(-1), node=m_id, lhs_id=x_id, rhs_id=new_node(),
                                    mults=(-1 if consume else 0, 0, 1)))
        T[KIND_SUM_REDUCE].append(sum_reduce_rows(prod, node=s_id, input_id=m_id, input_mult=-1, out_mult=1))
        y1 = prod.sum(axis=1)
        be = np.tile(b, len(x))
        return y1, be, s_id, a_id, n_out

    def tanh(v, v_id, fanout):
        """elementwise on a flat vector; the result is yielded `fanout` times."""
        n = len(v)
        c = int(round(-2.0 / np.log(2.0) * SCALE))
        ids = [new_node() for _ in range(6)]
        t = (v * c) >> 12
        T[KIND_MUL].append(mul_rows(v, np.full(n, c), node=ids[0], lhs_id=v_id, rhs_id=new_node(), mults=(-1, 0, 1)))
        rows, cnt = unary_lut_rows("exp2", np.clip(t, lo, hi), lo, node=ids[1], input_id=ids[0], mults=(-1, 1))
        assert np.all((t >= lo) & (t <= hi)), "activation outside the exp2 LUT range"
        counts[:len(cnt)] += cnt
        T[9].append(rows)
        e = rows[:, 8].astype(np.int64)
        T[KIND_ADD].append(add_rows(e, np.full(n, SCALE), node=ids[2], lhs_id=ids[1], rhs_id=new_node(), mults=(-1, 0, 1)))
        s = e + SCALE
        T[KIND_RECIP].append(recip_rows(s, node=ids[3], input_id=ids[2], mults=(-1, 1)))
        r = (SCALE * SCALE) // s
        T[KIND_MUL].append(mul_rows(r, np.full(n, 2 * SCALE), node=ids[4], lhs_id=ids[3], rhs_id=new_node(), mults=(-1, 0, 1)))
        u = (r * 2 * SCALE) >> 12
        T[KIND_ADD].append(add_rows(u, np.full(n, -SCALE), node=ids[5], lhs_id=ids[4], rhs_id=new_node(),
                                    mults=(-1, 0, fanout)))
        return u - SCALE, ids[5]

    x = rng.integers(-2048, 2048, size=(batch, 2))
    h, h_id, consume = x, 1, False
    for n_in, n_out, act in ((2, 64, True), (64, 64, True), (64, 1, False)):
        y1, be, s_id, a_id, _ = linear(h, h_id, n_in, n_out, consume)
        fan = {64: 64, 1: 1}[n_out] if act else 0
        # bias add yields its output once to the activation's first Mul (or 0 times for the network output)
        T[KIND_ADD].append(add_rows(y1, be, node=a_id, lhs_id=s_id, rhs_id=new_node(), mults=(-1, 0, 1 if act else 0)))
        y = y1 + be
        if act:
            nxt = {2: 64, 64: 1}[n_in]       # consumers of each activation output = next layer's n_out
            v, h_id = tanh(y, a_id, nxt)
            h = v.reshape(batch, n_out)
            consume = True
    lut = make_lut("exp2", lo, hi)
    tables = [(k, np.concatenate(v)) for k, v in T.items()]
    tables.append((10, lut_lookup_rows(counts, len(lut[0]))))
    return sorted(tables, key=lambda kt: kt[0]), {"exp2": lut}


def config2_add_only(n_rows: int = 1 << 20, seed: int = 42) -> List[Tuple[int, np.ndarray]]:
    """BASELINE config 2a: one Add table, all multiplicities 0 (logup sums are trivially 0)."""
    rng = np.random.default_rng(seed)
    lhs = rng.integers(-2048, 2048, size=n_rows)
    rhs = rng.integers(-2048, 2048, size=n_rows)
    return [(KIND_ADD, add_rows(lhs, rhs))]


def config2_mul_only(n_rows: int = 1 << 20, seed: int = 42) -> List[Tuple[int, np.ndarray]]:
    """The Mul counterpart of config 2a (BASELINE's metric says "Add/Mul trace"): one Mul table, all multiplicities 0,
    non-negative operands (avoids numerair's unverified sign convention for `rem`)."""
    rng = np.random.default_rng(seed)
    lhs = rng.integers(0, 2048, size=n_rows)
    rhs = rng.integers(0, 2048, size=n_rows)
    return [(KIND_MUL, mul_rows(lhs, rhs))]


def config2_graph_faithful(n_rows: int = 1 << 20, seed: int = 42) -> List[Tuple[int, np.ndarray]]:
    """BASELINE config 2b (needs the PINNED variant): Add consumes both inputs with mult -1,
    an Inputs table of 2n rows yields them with multiplicity 1."""
    rng = np.random.default_rng(seed)
    lhs = rng.integers(-2048, 2048, size=n_rows)
    rhs = rng.integers(-2048, 2048, size=n_rows)
    add = add_rows(lhs, rhs, mults=(-1, -1, 0))
    inp = np.concatenate([inputs_rows(lhs, 0, 1), inputs_rows(rhs, 1, 1)])
    return [(KIND_ADD, add), (KIND_INPUTS, inp)]


def chain_graph(n: int, seed: int = 42, with_recip: bool = True) -> List[Tuple[int, np.ndarray]]:
    """c = a*b (node 3); d = c + w (node 4); e = recip(d) (node 5) on n-element tensors, in the
    KAT-era multiplicity rule (graph initializers are consumed with multiplicity 0), so the
    logup sums cancel: Mul yields c once, Add consumes c and yields d once, Recip consumes d."""
    rng = np.random.default_rng(seed)
    a = rng.integers(1, 2048, size=n)
    b = rng.integers(1, 2048, size=n)
    w = rng.integers(8, 2048, size=n)
    c = (a * b) >> 12
    d = c + w
    tabs = [(KIND_ADD, add_rows(c, w, node=4, lhs_id=3, rhs_id=8, mults=(-1, 0, 1 if with_recip else 0))),
            (KIND_MUL, mul_rows(a, b, node=3, lhs_id=6, rhs_id=7, mults=(0, 0, 1)))]
    if with_recip:
        tabs.append((KIND_RECIP, recip_rows(d, node=5, input_id=4, mults=(-1, 0))))
    return tabs


def config3_mixed(log_add: int = 21, log_mul: int = 20, log_recip: int = 20, seed: int = 42):
    """BASELINE config 3: Add 2^21 + Mul 2^20 + Recip 2^20 rows in one pie (independent ops,
    all multiplicities 0)."""
    rng = np.random.default_rng(seed)
    na, nm, nr = 1 << log_add, 1 << log_mul, 1 << log_recip
    return [(KIND_ADD, add_rows(rng.integers(-2048, 2048, size=na), rng.integers(-2048, 2048, size=na))),
            (KIND_MUL, mul_rows(rng.integers(0, 2048, size=nm), rng.integers(0, 2048, size=nm), node=5, lhs_id=3,
                                rhs_id=4)),
            (KIND_RECIP, recip_rows(rng.integers(4, 2048, size=nr), node=7, input_id=6))]


def simple_example() -> List[Tuple[int, np.ndarray]]:
    """`examples/simple/src/main.rs:15-22`: c=a*b; d=c+w; e=c*d on 2x2 tensors — the tables of
    SURVEY.md Appendix A.10 (KAT-era multiplicities)."""
    a, b, w = [1, 2, 3, 4], [10, 20, 30, 40], [-1, -1, -1, -1]
    S = SCALE
    a, b, w = np.array(a) * S, np.array(b) * S, np.array(w) * S
    c = (a * b) >> 12
    d = c + w
    add = add_rows(c, w, node=4, lhs_id=3, rhs_id=8, mults=(-1, 0, 1))
    mul = np.concatenate([mul_rows(a, b, node=3, lhs_id=6, rhs_id=7, mults=(0, 0, 2)),
                          mul_rows(c, d, node=5, lhs_id=3, rhs_id=4, mults=(-1, -1, 0))])
    return [(KIND_ADD, add), (KIND_MUL, mul)]
