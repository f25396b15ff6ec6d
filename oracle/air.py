"""AIR component definitions for the hot-path operators (oracle; test infrastructure only).

Row layouts, padding rows, constraint order and logup relations restate
  Add    `crates/air/src/components/add/{table.rs:20-58,191-216, component.rs:38-116, witness.rs:33-167}`
  Mul    `crates/air/src/components/mul/{table.rs:19-36, component.rs:40-126, witness.rs:18-165}`
  Recip  `crates/air/src/components/recip/{table.rs:20-54, component.rs:38-107}`
  Inputs `crates/air/src/components/inputs/{table.rs:20-44, components.rs:37-85}`
  SumReduce  `crates/air/src/components/sum_reduce/{table.rs:39-56,173-186, component.rs:36-110}`
  MaxReduce  `crates/air/src/components/max_reduce/{table.rs:39-57,178-192, component.rs}`
  Contiguous `crates/air/src/components/contiguous/{table.rs:36-50,162-172, component.rs}`
(the last three use no numerair helper: their constraint forms are fully visible in the reference)
`eval_fixed_{add,mul,recip}` live in numerair@11d1d26 (un-vendored).  KAT evidence (SURVEY.md §2.1,
A.7) pins: eval_fixed_add = 1 constraint `out-(lhs+rhs)`; eval_fixed_mul = 2 constraint slots, the
first `lhs*rhs-(out*scale+rem)`, the second contributing zero whenever rem == 0 — restated here as
a zero slot (**unpinned for rem != 0**).  eval_fixed_recip is **unpinned**; `scale^2-(input*out+rem)`
is used.

Constraint functions are polymorphic: they run on `MV` (numpy M31 vectors, prover side) and on
scalar `QM31` (OODS point, verifier side).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .field import P, U64, QM31, m_add, m_mul, m_sub

FP_SCALE = 1 << 12  # DEFAULT_FP_SCALE = 12, crates/air/src/lib.rs:23-24


class MV:
    """numpy M31 vector with field operators (so constraint code reads like the Rust AIR)."""

    __slots__ = ("a",)

    def __init__(self, a):
        self.a = np.asarray(a, dtype=U64)

    @staticmethod
    def _c(o):
        return o.a if isinstance(o, MV) else U64(int(o) % P)

    def __add__(self, o):
        return MV(m_add(self.a, MV._c(o)))

    __radd__ = __add__

    def __sub__(self, o):
        return MV(m_sub(self.a, MV._c(o)))

    def __rsub__(self, o):
        return MV(m_sub(MV._c(o), self.a))

    def __mul__(self, o):
        return MV(m_mul(self.a, MV._c(o)))

    __rmul__ = __mul__


# kind ids = TraceTable variant order, crates/air/src/pie.rs:31-66
KIND_ADD, KIND_MUL, KIND_RECIP, KIND_SIN, KIND_SIN_LOOKUP, KIND_SUM_REDUCE, KIND_MAX_REDUCE, KIND_SQRT, \
    KIND_REM, KIND_EXP2, KIND_EXP2_LOOKUP, KIND_LOG2, KIND_LOG2_LOOKUP, KIND_LESS_THAN, \
    KIND_RANGE_CHECK_LOOKUP, KIND_INPUTS, KIND_CONTIGUOUS = range(17)
N_KINDS = 17
# KAT-era claim struct had 8 options: add, mul, recip, sin, sin_lookup, sum_reduce, max_reduce, sqrt? —
# only the count (8) and the first two slots are pinned by the KAT bytes.
N_KINDS_KAT = 8


# relation element sets; draw order is node, sin, exp2, log2, range_check (components/mod.rs:227-235,
# lookups/mod.rs:44-51)
ELEMS_NODE, ELEMS_RANGE_CHECK, ELEMS_SIN, ELEMS_EXP2, ELEMS_LOG2 = 0, 1, 2, 3, 4


@dataclass(frozen=True)
class Rel:
    """One `add_to_relation` entry: multiplicity column, first value column, optional second value
    column (tensor id for node relations, LUT output for LUT relations; None for width-1 relations),
    the element set it is combined with, numerator sign, and whether `val`/`id` index the
    component's preprocessed columns instead of its main columns."""
    mult: int
    val: int
    id: Optional[int] = None
    elems: int = ELEMS_NODE
    neg: bool = False
    pre: bool = False


@dataclass
class Component:
    name: str
    kind: int
    n_cols: int
    padding: Tuple[int, ...]
    local: Callable[[Sequence], List]            # cols -> list of constraint values
    relations: Tuple                             # Rel entries (legacy form (mult,(val,id)) is converted)
    pre_cols: Tuple[str, ...] = ()               # ids of the preprocessed (tree 0) columns it reads

    def __post_init__(self):
        self.relations = tuple(r if isinstance(r, Rel) else Rel(r[0], r[1][0], r[1][1]) for r in self.relations)

    @property
    def n_local(self):
        one = [QM31(1)] * self.n_cols
        return len(self.local(one))

    @property
    def n_constraints(self):
        return self.n_local + len(self.relations)


# constraint-form bits (oracle.channel.ProtocolVariant / LMN_PV_* of include/luminair_hip.h)
PV_MUL_ONE_SLOT, PV_RECIP_TWO_SLOTS, PV_RECIP_NEG, PV_SQRT_TWO_SLOTS, PV_SQRT_NEG, PV_REM_TWO_SLOTS, PV_REM_NEG = (
    0x100, 0x200, 0x400, 0x800, 0x1000, 0x2000, 0x4000)


def constraint_layout(comp: "Component", flags: int):
    """(n_protocol, proto_index, neg): how the values `comp.local` + relations produce (the KAT-era shape: eval_fixed_mul
    two slots, eval_fixed_recip / _sqrt / _rem one; "kernel slots") map to the constraints the component contributes
    to the composition polynomial under the constraint-form bits.  Slot 1 is the eval_fixed_* constraint of Mul / Recip
    / Sqrt / Rem, Mul's slot 2 its zero slot.  proto_index[k] = None: the protocol has no such constraint."""
    flags = int(flags)
    drop2 = comp.kind == KIND_MUL and bool(flags & PV_MUL_ONE_SLOT)
    two, neg_bit = {KIND_RECIP: (PV_RECIP_TWO_SLOTS, PV_RECIP_NEG), KIND_SQRT: (PV_SQRT_TWO_SLOTS, PV_SQRT_NEG),
                    KIND_REM: (PV_REM_TWO_SLOTS, PV_REM_NEG)}.get(comp.kind, (0, 0))
    extra, neg1 = bool(flags & two), bool(flags & neg_bit)
    proto_index, neg, p = [], [], 0
    for k in range(comp.n_constraints):
        neg.append(k == 1 and neg1)
        if k == 2 and drop2:
            proto_index.append(None)
            continue
        proto_index.append(p)
        p += 2 if (k == 1 and extra) else 1
    return p, proto_index, neg


def component_coeffs(comp: "Component", flags: int, powers, n_total: int, k0: int):
    """Coefficient (power of the composition randomness, signed; 0 for a slot the protocol lacks) of every kernel
    slot of a component whose first protocol constraint has global index k0.  Returns (coeffs, n_protocol)."""
    n_proto, proto_index, neg = constraint_layout(comp, flags)
    zero = powers[0] - powers[0]
    out = []
    for pi, ng in zip(proto_index, neg):
        c = zero if pi is None else powers[n_total - 1 - (k0 + pi)]
        out.append(-c if ng else c)
    return out, n_proto


def _transition(not_last, pairs, nxt_idx, idx):
    out = [not_last * (n - c) for n, c in pairs]
    out.append(not_last * (nxt_idx - idx - 1))
    return out


def _add_local(c):
    (node, lhs_id, rhs_id, idx, is_last, n_node, n_lhs, n_rhs, n_idx, lhs, rhs, out, _lm, _rm, _om) = c
    cons = [is_last * (is_last - 1), out - (lhs + rhs)]
    not_last = 1 - is_last
    cons += _transition(not_last, [(n_node, node), (n_lhs, lhs_id), (n_rhs, rhs_id)], n_idx, idx)
    return cons


def _mul_local(c):
    (node, lhs_id, rhs_id, idx, is_last, n_node, n_lhs, n_rhs, n_idx, lhs, rhs, out, rem, _lm, _rm, _om) = c
    cons = [is_last * (is_last - 1), lhs * rhs - (out * FP_SCALE + rem), rem * 0]
    not_last = 1 - is_last
    cons += _transition(not_last, [(n_node, node), (n_lhs, lhs_id), (n_rhs, rhs_id)], n_idx, idx)
    return cons


def _recip_local(c):
    (node, in_id, idx, is_last, n_node, n_in, n_idx, inp, out, rem, scale, _im, _om) = c
    cons = [is_last * (is_last - 1), scale * scale - (inp * out + rem)]
    not_last = 1 - is_last
    cons += _transition(not_last, [(n_node, node), (n_in, in_id)], n_idx, idx)
    return cons


def _inputs_local(c):
    (node, idx, is_last, n_node, n_idx, _val, _mult) = c
    not_last = 1 - is_last
    return [is_last * (is_last - 1), not_last * (n_node - node), not_last * (n_idx - idx - 1)]


def _sum_reduce_local(c):
    (node, in_id, idx, is_last, n_node, n_in, n_idx, inp, out, acc, next_acc, is_last_step, _im, _om) = c
    cons = [is_last * (is_last - 1), is_last_step * (is_last_step - 1), next_acc - (acc + inp),
            (out - next_acc) * is_last_step]
    not_last = 1 - is_last
    cons += _transition(not_last, [(n_node, node), (n_in, in_id)], n_idx, idx)
    return cons


def _max_reduce_local(c):
    (node, in_id, idx, is_last, n_node, n_in, n_idx, inp, out, mx, next_mx, is_last_step, is_max, _im, _om) = c
    cons = [is_last * (is_last - 1), is_last_step * (is_last_step - 1), is_max * (is_max - 1),
            is_max * (next_mx - inp), (1 - is_max) * (next_mx - mx), (out - next_mx) * is_last_step]
    not_last = 1 - is_last
    cons += _transition(not_last, [(n_node, node), (n_in, in_id)], n_idx, idx)
    return cons


def _contiguous_local(c):
    (node, in_id, idx, is_last, n_node, n_in, n_idx, _inp, _out, _im, _om) = c
    not_last = 1 - is_last
    return [is_last * (is_last - 1)] + _transition(not_last, [(n_node, node), (n_in, in_id)], n_idx, idx)


def _sqrt_local(c):
    """eval_fixed_sqrt(input, out, rem, scale) is in numerair (un-vendored): **unpinned**; restated as
    the natural fixed-point identity input*scale = out^2 + rem."""
    (node, in_id, idx, is_last, n_node, n_in, n_idx, inp, out, rem, scale, _im, _om) = c
    cons = [is_last * (is_last - 1), inp * scale - (out * out + rem)]
    not_last = 1 - is_last
    cons += _transition(not_last, [(n_node, node), (n_in, in_id)], n_idx, idx)
    return cons


def _rem_local(c):
    """eval_fixed_rem(lhs, rhs, quotient, rem) is in numerair (un-vendored): **unpinned**; restated as
    lhs = rhs*quotient + rem."""
    (node, lhs_id, rhs_id, idx, is_last, n_node, n_lhs, n_rhs, n_idx, lhs, rhs, rem, quo, _lm, _rm, _om) = c
    cons = [is_last * (is_last - 1), lhs - (rhs * quo + rem)]
    not_last = 1 - is_last
    cons += _transition(not_last, [(n_node, node), (n_lhs, lhs_id), (n_rhs, rhs_id)], n_idx, idx)
    return cons


def _pad(n, is_last_col):
    p = [0] * n
    p[is_last_col] = 1
    return tuple(p)


ADD = Component("add", KIND_ADD, 15, _pad(15, 4), _add_local,
                ((12, (9, 1)), (13, (10, 2)), (14, (11, 0))))
MUL = Component("mul", KIND_MUL, 16, _pad(16, 4), _mul_local,
                ((13, (9, 1)), (14, (10, 2)), (15, (11, 0))))
RECIP = Component("recip", KIND_RECIP, 13, _pad(13, 3), _recip_local,
                  ((11, (7, 1)), (12, (8, 0))))
INPUTS = Component("inputs", KIND_INPUTS, 7, _pad(7, 2), _inputs_local,
                   ((6, (5, 0)),))

SUM_REDUCE = Component("sum_reduce", KIND_SUM_REDUCE, 14, _pad(14, 3), _sum_reduce_local,
                       ((12, (7, 1)), (13, (8, 0))))
MAX_REDUCE = Component("max_reduce", KIND_MAX_REDUCE, 15, _pad(15, 3), _max_reduce_local,
                       ((13, (7, 1)), (14, (8, 0))))
CONTIGUOUS = Component("contiguous", KIND_CONTIGUOUS, 11, _pad(11, 3), _contiguous_local,
                       ((9, (7, 1)), (10, (8, 0))))



def _less_than_local(c):
    """`crates/air/src/components/less_than/component.rs:48-185`."""
    (node, lhs_id, rhs_id, idx, is_last, n_node, n_lhs, n_rhs, n_idx, lhs, rhs, out, diff, borrow,
     l0, l1, l2, l3, _lm, _rm, _om, _dm) = c
    cons = [is_last * (is_last - 1), borrow * (borrow - 1), out - (1 - borrow) * FP_SCALE,
            lhs + diff - rhs - borrow * P_MINUS_1_AS_2POW31M1,
            diff - (l3 * (1 << 24) + l2 * (1 << 16) + l1 * (1 << 8) + l0)]
    not_last = 1 - is_last
    cons += _transition(not_last, [(n_node, node), (n_lhs, lhs_id), (n_rhs, rhs_id)], n_idx, idx)
    return cons


# TWO_POW_31_MINUS_1 (crates/air/src/lib.rs:26) as a field constant: 2^31-1 == P == 0 in M31; the
# reference builds it with from_u32_unchecked, i.e. the non-canonical representative of zero.
P_MINUS_1_AS_2POW31M1 = 0

_LT_PAD = [0] * 22
_LT_PAD[4], _LT_PAD[10], _LT_PAD[11], _LT_PAD[12], _LT_PAD[14] = 1, 1, FP_SCALE, 1, 1   # less_than/table.rs:47-72
LESS_THAN = Component("less_than", KIND_LESS_THAN, 22, tuple(_LT_PAD), _less_than_local,
                      (Rel(18, 9, 1), Rel(19, 10, 2), Rel(20, 11, 0),
                       Rel(21, 14, None, ELEMS_RANGE_CHECK), Rel(21, 15, None, ELEMS_RANGE_CHECK),
                       Rel(21, 16, None, ELEMS_RANGE_CHECK), Rel(21, 17, None, ELEMS_RANGE_CHECK)))
# lookups/range_check/component.rs: one multiplicity column + the preprocessed 8-bit enumeration
RANGE_CHECK_LOG = 8
RANGE_CHECK_COL_ID = "range_check_8_column_0"
RANGE_CHECK_LOOKUP = Component("range_check_lookup", KIND_RANGE_CHECK_LOOKUP, 1, (0,), lambda c: [],
                               (Rel(0, 0, None, ELEMS_RANGE_CHECK, neg=True, pre=True),),
                               pre_cols=(RANGE_CHECK_COL_ID,))


def _unary_lut_local(c):
    """Sin / Exp2 / Log2 (`sin/component.rs:50-122`): only the boolean + transition constraints; the
    function value is enforced by the LUT relation (lookup_mult, [input, out])."""
    (node, in_id, idx, is_last, n_node, n_in, n_idx, _inp, _out, _im, _om, _lm) = c
    not_last = 1 - is_last
    return [is_last * (is_last - 1)] + _transition(not_last, [(n_node, node), (n_in, in_id)], n_idx, idx)


def _unary_lut(name, kind, elems):
    return Component(name, kind, 12, _pad(12, 3), _unary_lut_local,
                     (Rel(9, 7, 1), Rel(10, 8, 0), Rel(11, 7, 8, elems)))


def _lut_lookup(name, kind, elems, prefix):
    """`lookups/sin/component.rs:40-59`: multiplicity column + two preprocessed LUT columns,
    relation (-multiplicity, [lut_0, lut_1])."""
    return Component(name, kind, 1, (0,), lambda c: [], (Rel(0, 0, 1, elems, neg=True, pre=True),),
                     pre_cols=(prefix + "_lut_0", prefix + "_lut_1"))


SIN = _unary_lut("sin", KIND_SIN, ELEMS_SIN)
EXP2 = _unary_lut("exp2", KIND_EXP2, ELEMS_EXP2)
LOG2 = _unary_lut("log2", KIND_LOG2, ELEMS_LOG2)
SIN_LOOKUP = _lut_lookup("sin_lookup", KIND_SIN_LOOKUP, ELEMS_SIN, "sin")
EXP2_LOOKUP = _lut_lookup("exp2_lookup", KIND_EXP2_LOOKUP, ELEMS_EXP2, "exp2")
LOG2_LOOKUP = _lut_lookup("log2_lookup", KIND_LOG2_LOOKUP, ELEMS_LOG2, "log2")
# tree-0 column order before the size sort (lookups_to_preprocessed_column, preprocessed.rs:157-179)
PREPROCESSED_ORDER = ("sin_lut_0", "sin_lut_1", "exp2_lut_0", "exp2_lut_1", "log2_lut_0", "log2_lut_1",
                      RANGE_CHECK_COL_ID)

# sqrt/{table.rs:178-190,component.rs}, rem/{table.rs:203-218,component.rs:60-110}: the out relation of Rem carries `rem`
SQRT = Component("sqrt", KIND_SQRT, 13, _pad(13, 3), _sqrt_local, ((11, (7, 1)), (12, (8, 0))))
REM = Component("rem", KIND_REM, 16, _pad(16, 4), _rem_local, ((13, (9, 1)), (14, (10, 2)), (15, (11, 0))))

COMPONENTS = {c.kind: c for c in (ADD, MUL, RECIP, INPUTS, SUM_REDUCE, MAX_REDUCE, CONTIGUOUS, LESS_THAN,
                                  RANGE_CHECK_LOOKUP, SQRT, REM, SIN, EXP2, LOG2, SIN_LOOKUP, EXP2_LOOKUP,
                                  LOG2_LOOKUP)}


def preprocessed_column(col_id: str, luts) -> np.ndarray:
    """Tree-0 column by id.  The range-check column is `RangeCheckPreProcessed::gen_column`
    (crates/air/src/preprocessed.rs:289-296: row r holds r); the sin/exp2/log2 LUT columns
    (`:351-383,434-466,517-549`, generated with f64 math on the reference's host side) are inputs:
    `luts` maps "sin"/"exp2"/"log2" to (col0, col1) arrays."""
    if col_id == RANGE_CHECK_COL_ID:
        return np.arange(1 << RANGE_CHECK_LOG, dtype=U64)
    name, _, idx = col_id.rpartition("_lut_")
    if luts and name in luts:
        return np.asarray(luts[name][int(idx)], dtype=U64)
    raise ValueError("missing preprocessed column " + col_id)


def pad_table(comp: Component, rows: np.ndarray) -> np.ndarray:
    """AoS rows (n, n_cols) -> SoA columns (n_cols, 2^log_size), padded as `write_trace` does
    (`add/witness.rs:43-46`: size = max(next_pow2(n_rows), N_LANES=16))."""
    rows = np.asarray(rows, dtype=U64).reshape(-1, comp.n_cols)
    n = rows.shape[0]
    if n == 0:
        raise ValueError("EmptyTrace")  # TraceError::EmptyTrace, add/witness.rs:39-41
    size = max(1 << (n - 1).bit_length(), 16)
    out = np.empty((size, comp.n_cols), dtype=U64)
    out[:n] = rows
    out[n:] = np.array(comp.padding, dtype=U64)
    return np.ascontiguousarray(out.T)
