"""Device-side `gen_trace` for graphs of the primitives whose `process_trace` runs on the GPU
(Add, Mul, Recip, Sqrt, Rem, LessThan with its range-check table, SumReduce, MaxReduce, Contiguous, the LUT ops
Sin / Exp2 / Log2 with their lookup tables, and the graph inputs,
with expanded ("fake") dimensions as in luminal's ShapeTracker): a small host mirror of
`LuminairGraph::gen_trace` (`crates/graph/src/graph.rs:161-604`) over `lmn_trace_elementwise` /
`lmn_trace_sum_reduce`.  Nodes execute in creation (= topological) order, every tensor stays in HBM
as int32 `Fixed<12>` values, each node appends its rows to its kind's device-resident table, and the
resulting pie feeds `lmn_prove` with `LMN_TABLE_ROWS_ON_DEVICE`.

Multiplicities follow HEAD (`crates/graph/src/op/prim.rs:66-83,1005-1006`): an op consumes each input
with multiplicity -1 and yields its output `num_consumers` times (0 for a final output); a graph input
is yielded `num_consumers` times by its Inputs rows — so the logup sums of a complete graph cancel.
`num_consumers` is expansion-adjusted as in `graph.rs:215-243`: a consumer that reads the tensor through a
view with expanded dimensions counts once per repetition of each element.
The graph front-end itself (luminal's compiler passes, shape tracking, f32 -> fixed conversion rules)
is outside the hot-path scope; this is the execution + table-fill step only."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import backend
from .pie import TraceTableKind

_NCOLS = {0: 15, 1: 16, 2: 13, 3: 12, 5: 14, 6: 15, 7: 13, 8: 16, 9: 12, 11: 12, 13: 22, 15: 7, 16: 11}
_LUT_OF = {3: ("sin", 4), 9: ("exp2", 10), 11: ("log2", 12)}     # op kind -> (LUT name, lookup-table kind)


class _SlabLease(backend.DeviceBuffer):
    """The one device allocation of a gen_trace call.  `free()` hands it back to the context's one-slot cache instead
    of hipFree, so that a stream of small graphs on one context does not pay a hipMalloc / hipFree pair (~0.1 ms) per
    graph; a lease that is still in use is never handed out again (the cache only holds returned slabs)."""

    def free(self):
        if self.ptr and self.owned:
            cached = getattr(self.ctx, "_graph_slab", None)
            if cached is None or cached.nbytes < self.nbytes:
                if cached is not None:
                    cached.free()
                self.ctx._graph_slab = backend.DeviceBuffer(self.ctx, self.ptr, self.nbytes)
            else:
                backend.DeviceBuffer.free(self)
        self.ptr = 0


def _lease_slab(ctx, need: int) -> _SlabLease:
    cached = getattr(ctx, "_graph_slab", None)
    if cached is not None and cached.nbytes >= need:
        ctx._graph_slab = None
        return _SlabLease(ctx, cached.ptr, cached.nbytes)
    b = ctx.alloc(need)
    return _SlabLease(ctx, b.ptr, b.nbytes)


@dataclass
class GraphTensor:
    node_id: int
    shape: Tuple[int, ...]
    consumers: int = 0
    is_output: bool = False
    buf: Optional[backend.DeviceBuffer] = None     # set by gen_trace

    @property
    def size(self) -> int:
        return int(np.prod(self.shape))


@dataclass
class TensorView:
    """A tensor seen through a shape with expanded dimensions (stride 0), slices (offset) or permuted strides: the
    part of luminal's ShapeTracker the ops' index expressions need."""
    base: GraphTensor
    shape: Tuple[int, ...]
    strides: Tuple[int, ...]
    offset: int = 0

    @property
    def size(self) -> int:
        return int(np.prod(self.shape))

    @property
    def expansion(self) -> int:
        return int(np.prod([d for d, st in zip(self.shape, self.strides) if st == 0 and d > 1] or [1]))


def _as_view(t) -> TensorView:
    if isinstance(t, TensorView):
        return t
    strides, acc = [], 1
    for d in reversed(t.shape):
        strides.append(acc)
        acc *= d
    return TensorView(t, t.shape, tuple(reversed(strides)))


@dataclass
class _Node:
    kind: int
    out: GraphTensor
    inputs: List[GraphTensor]
    host: Optional[np.ndarray] = None      # graph inputs
    axis: int = 0                          # SumReduce


class DeviceGraph:
    def __init__(self, ctx: backend.Context):
        self.ctx = ctx
        self.nodes: List[_Node] = []
        self._next_id = 0
        self.luts: Dict[str, tuple] = {}
        self.lut_ranges: Dict[str, list] = {}      # LUTs declared with several ranges (set_lut_ranges)

    def _tensor(self, shape) -> GraphTensor:
        t = GraphTensor(self._next_id, tuple(int(s) for s in shape))
        self._next_id += 1
        return t

    def input(self, values: np.ndarray) -> GraphTensor:
        """A graph input holding Fixed<12> integers (`CopyToStwo`, prim.rs:52-88)."""
        v = np.ascontiguousarray(values, dtype=np.int32)
        t = self._tensor(v.shape)
        self.nodes.append(_Node(int(TraceTableKind.Inputs), t, [], host=v))
        return t

    def constant(self, value: int) -> GraphTensor:
        """`LuminairConstant` (prim.rs:151-200): a one-element tensor, expanded by its consumers."""
        return self.input(np.array([value], dtype=np.int32))

    @staticmethod
    def expand(t, axis: int, size: int) -> TensorView:
        """Insert an expanded dimension of `size` at `axis` (stride 0)."""
        v = _as_view(t)
        return TensorView(v.base, v.shape[:axis] + (int(size),) + v.shape[axis:], v.strides[:axis] + (0,) + v.strides[axis:],
                          v.offset)

    @staticmethod
    def expand_dim(t, axis: int, size: int) -> TensorView:
        """Broadcast an EXISTING dimension of size 1 to `size` (stride 0)."""
        v = _as_view(t)
        if v.shape[axis] != 1:
            raise ValueError("expand_dim broadcasts a dimension of size 1")
        return TensorView(v.base, v.shape[:axis] + (int(size),) + v.shape[axis + 1:],
                          v.strides[:axis] + (0,) + v.strides[axis + 1:], v.offset)

    @staticmethod
    def luminal_expand(t, axis: int, size: int) -> TensorView:
        """`GraphTensor::expand(axis, size)` as the reference's tests use it (crates/graph/src/tests/expansions.rs,
        tests/mod.rs:80-110): a dimension of size 1 at `axis` is broadcast ((2,1).expand(1,3) -> (2,3)); where the
        tensor has no such dimension a new expanded one is inserted ((2,3).expand(2,4) -> (2,3,4)).  Either way the
        element -> buffer index map and the number of repetitions per element are the same."""
        v = _as_view(t)
        if axis < len(v.shape) and v.shape[axis] == 1:
            return DeviceGraph.expand_dim(v, axis, size)
        return DeviceGraph.expand(v, axis, size)

    @staticmethod
    def expand_to(t, shape) -> TensorView:
        """`GraphTensor::expand_to(shape)`: broadcast every dimension of size 1 (leading dimensions are added)."""
        v = _as_view(t)
        shape = tuple(int(d) for d in shape)
        vs, st = (1,) * (len(shape) - len(v.shape)) + v.shape, (0,) * (len(shape) - len(v.shape)) + v.strides
        out = []
        for have, want, stride in zip(vs, shape, st):
            if have == want:
                out.append(stride)
            elif have == 1:
                out.append(0)
            else:
                raise ValueError("expand_to: dimension %d cannot become %d" % (have, want))
        return TensorView(v.base, shape, tuple(out), v.offset)

    @staticmethod
    def slice(t, ranges) -> TensorView:
        """`GraphTensor::slice`: ranges = one (start, stop) per dimension (None = the whole dimension)."""
        v = _as_view(t)
        ranges = list(ranges)
        if len(ranges) > len(v.shape):
            raise ValueError("slice: %d ranges for a %d-dimensional tensor" % (len(ranges), len(v.shape)))
        ranges += [None] * (len(v.shape) - len(ranges))      # missing trailing dimensions are taken whole
        shape, off = [], v.offset
        for d, stride, r in zip(v.shape, v.strides, ranges):
            lo, hi = (0, d) if r is None else r
            if not 0 <= lo < hi <= d:
                raise ValueError("slice out of range")
            shape.append(hi - lo)
            off += lo * stride
        return TensorView(v.base, tuple(shape), v.strides, off)

    @staticmethod
    def permute(t, axes) -> TensorView:
        """`GraphTensor::permute`."""
        v = _as_view(t)
        return TensorView(v.base, tuple(v.shape[a] for a in axes), tuple(v.strides[a] for a in axes), v.offset)

    @staticmethod
    def broadcast_to(t, shape) -> TensorView:
        """View a one-element tensor (a constant) with the given shape."""
        v = _as_view(t)
        if v.base.size != 1:
            raise ValueError("broadcast_to takes a one-element tensor")
        return TensorView(v.base, tuple(int(d) for d in shape), (0,) * len(shape))

    def _consume(self, t) -> TensorView:
        v = _as_view(t)
        v.base.consumers += v.expansion
        return v

    def _binary(self, kind, a, b) -> GraphTensor:
        va, vb = _as_view(a), _as_view(b)
        if va.shape != vb.shape:
            raise ValueError("elementwise operands must have equal (view) shapes: expand / broadcast_to first")
        t = self._tensor(va.shape)
        self.nodes.append(_Node(kind, t, [self._consume(a), self._consume(b)]))
        return t

    def add(self, a, b):
        return self._binary(int(TraceTableKind.Add), a, b)

    def mul(self, a, b):
        return self._binary(int(TraceTableKind.Mul), a, b)

    def _unary(self, kind, a) -> GraphTensor:
        v = self._consume(a)
        t = self._tensor(v.shape)
        self.nodes.append(_Node(kind, t, [v]))
        return t

    def recip(self, a) -> GraphTensor:
        return self._unary(int(TraceTableKind.Recip), a)

    def sqrt(self, a) -> GraphTensor:
        return self._unary(int(TraceTableKind.Sqrt), a)

    def contiguous(self, a) -> GraphTensor:
        """Materialise a view (`LuminairContiguous`, prim.rs:229-301).  A view without expanded dimensions (slice,
        permutation, identity) follows the reference's own row rule - one row per element of the input BUFFER,
        `lmn_trace_contiguous`; a view WITH expanded dimensions consumes the view's element per output row instead
        (the reference's rule cannot balance the logup there)."""
        return self._unary(int(TraceTableKind.Contiguous), a)

    def rem(self, a, b) -> GraphTensor:
        return self._binary(int(TraceTableKind.Rem), a, b)

    def less_than(self, a, b) -> GraphTensor:
        return self._binary(int(TraceTableKind.LessThan), a, b)

    def max_reduce(self, a: GraphTensor, axis: int) -> GraphTensor:
        t = self.sum_reduce(a, axis)
        self.nodes[-1].kind = int(TraceTableKind.MaxReduce)
        return t

    def sum_reduce(self, a: GraphTensor, axis: int) -> GraphTensor:
        if isinstance(a, TensorView):
            raise ValueError("sum_reduce takes a materialised tensor")
        v = self._consume(a)
        shape = a.shape[:axis] + a.shape[axis + 1:]
        t = self._tensor(shape if shape else (1,))
        self.nodes.append(_Node(int(TraceTableKind.SumReduce), t, [v], axis=axis))
        return t

    def set_lut(self, name: str, lo: int, hi: int):
        """Declare the value range of a LUT (what `gen_circuit_settings` derives from a dry run,
        graph.rs:61-159); the columns are generated on the host as the reference does (f64 math)."""
        from . import synthetic
        self.luts[name] = (int(lo), int(hi), synthetic.make_lut(name, int(lo), int(hi)))
        self.lut_ranges.pop(name, None)

    def set_lut_ranges(self, name: str, ranges):
        """A LUT over SEVERAL value ranges - what `gen_circuit_settings` yields when the graph applies the function
        on disjoint input ranges (one padded range per op, coalesced: graph.rs:61-159, 665-691).  `ranges`: ascending,
        disjoint (lo, hi) pairs; columns from `lmn_lut_from_ranges` (`SinPreProcessed::gen_column` and siblings)."""
        rg = [(int(a), int(b)) for a, b in ranges]
        cols = self.ctx.lib.lut_from_ranges(name, rg)
        self.luts[name] = (rg[0][0], rg[-1][1], cols)
        self.lut_ranges[name] = rg

    def _lut_op(self, kind, a) -> GraphTensor:
        v = self._consume(a)
        t = self._tensor(v.shape)
        self.nodes.append(_Node(kind, t, [v]))
        return t

    def sin(self, a):
        return self._lut_op(int(TraceTableKind.Sin), a)

    def exp2(self, a):
        return self._lut_op(int(TraceTableKind.Exp2), a)

    def log2(self, a):
        return self._lut_op(int(TraceTableKind.Log2), a)

    def output(self, t: GraphTensor) -> GraphTensor:
        t.is_output = True
        return t

    def gen_trace(self):
        """Runs every node on the device.  Returns (tables, luts, buffers): tables = [(kind, rows DeviceBuffer,
        n_rows)] in `gen_trace` order (ascending kind) and luts = {name: (col0, col1)} for
        Context.prove_tables(tables, luts); buffers = every device allocation made (free them after proving)."""
        ctx = self.ctx
        K = TraceTableKind
        view_of = lambda v: None if v.strides == _as_view(v.base).strides and v.shape == v.base.shape and not v.offset \
            else backend.LmnView.make(v.shape, v.strides, v.offset)
        reduces = (int(K.SumReduce), int(K.MaxReduce))
        ref_contig = lambda n: n.kind == int(K.Contiguous) and n.inputs[0].expansion == 1

        def rows_of(n):
            if n.kind in reduces:
                return n.inputs[0].size
            if ref_contig(n):
                return max(n.inputs[0].base.size, n.out.size)
            return n.out.size
        total: Dict[int, int] = {}
        for n in self.nodes:
            total[n.kind] = total.get(n.kind, 0) + rows_of(n)
        # one allocation for all tables and tensors (hipMalloc per node would dominate small graphs); the trace
        # calls are stream-ordered, so nothing waits until the tables are proved or a tensor is read back
        al = lambda nbytes: (nbytes + 255) & ~255
        # host data that has to reach the device: graph inputs, LUT output columns, zeroed multiplicity tables - packed
        # into ONE staging array and uploaded with one copy (one synchronisation instead of one per tensor)
        stage_parts, stage_off, cursor_h = [], {}, [0]

        def stage(key, arr):
            a = np.ascontiguousarray(arr).reshape(-1).view(np.uint32)
            stage_off[key] = (cursor_h[0], a.nbytes)
            pad = (al(a.nbytes) - a.nbytes) // 4
            stage_parts.append(a)
            if pad:
                stage_parts.append(np.zeros(pad, dtype=np.uint32))
            cursor_h[0] += al(a.nbytes)

        for n in self.nodes:
            if n.kind in _LUT_OF and _LUT_OF[n.kind][0] not in self.luts:
                raise ValueError("no LUT for %r: call gen_circuit_settings() (or set_lut / set_lut_ranges) before gen_trace"
                                 % _LUT_OF[n.kind][0])
            if n.host is not None:
                stage(("in", n.out.node_id), n.host.astype(np.int32, copy=False))
        luts_out = {}
        for kind in sorted(total):
            if kind in _LUT_OF:
                name, _ = _LUT_OF[kind]
                lo, hi, (c0, c1) = self.luts[name]
                stage(("lut1", name), c1)
                stage(("lutm", name), np.zeros(len(c0), dtype=np.uint32))
                luts_out[name] = (c0, c1)
        if int(K.LessThan) in total:
            stage(("rc",), np.zeros(256, dtype=np.uint32))
        need = sum(al(total[k] * _NCOLS[k] * 4) for k in total) + sum(al(n.out.size * 4) for n in self.nodes) + cursor_h[0]
        slab = _lease_slab(ctx, need)
        cursor = [0]

        def carve(nbytes):
            v = slab.view(cursor[0], nbytes)
            cursor[0] += al(nbytes)
            return v

        staged = carve(cursor_h[0]) if cursor_h[0] else None
        if staged is not None:
            ctx.upload_to(staged, np.concatenate(stage_parts))
        dev_of = lambda key: staged.view(*stage_off[key])
        tables = {k: carve(total[k] * _NCOLS[k] * 4) for k in total}
        offset = {k: 0 for k in total}
        bufs = [slab]
        lut_dev, lut_tables = {}, {}
        for kind in sorted(total):
            if kind in _LUT_OF:
                name, lookup_kind = _LUT_OF[kind]
                lut_dev[name] = (dev_of(("lut1", name)), dev_of(("lutm", name)))   # outputs, multiplicities
                lut_tables[lookup_kind] = (lut_dev[name][1], len(self.luts[name][2][0]))
        rc_mult = None
        if int(K.LessThan) in total:
            rc_mult = dev_of(("rc",))
            lut_tables[int(K.RangeCheckLookup)] = (rc_mult, 256)
        for n in self.nodes:
            t = n.out
            common = dict(num_consumers=t.consumers, is_final_output=t.is_output, rows=tables[n.kind],
                          row_offset=offset[n.kind], out=carve(t.size * 4))
            if n.kind == int(K.Inputs):
                src = dev_of(("in", t.node_id))
                _, t.buf = ctx.trace_elementwise(n.kind, src, None, t.size, node_id=t.node_id, input_ids=(),
                                                 input_mults=(), **common)
            elif n.kind in reduces:
                a = n.inputs[0].base
                front = int(np.prod(a.shape[:n.axis])) if n.axis else 1
                back = int(np.prod(a.shape[n.axis + 1:])) if n.axis + 1 < len(a.shape) else 1
                _, t.buf = ctx.trace_sum_reduce(a.buf, front, a.shape[n.axis], back, node_id=t.node_id,
                                                input_id=a.node_id, maximum=n.kind == int(K.MaxReduce), **common)
            elif n.kind == int(K.LessThan):
                ins = n.inputs
                _, t.buf = ctx.trace_less_than(ins[0].base.buf, ins[1].base.buf, t.size, node_id=t.node_id,
                                               input_ids=tuple(i.base.node_id for i in ins), range_check_mult=rc_mult,
                                               lhs_view=view_of(ins[0]), rhs_view=view_of(ins[1]), **common)
            elif ref_contig(n):
                v = n.inputs[0]
                _, t.buf = ctx.trace_contiguous(v.base.buf, v.base.size, t.size, node_id=t.node_id,
                                                input_id=v.base.node_id, view=view_of(v), **common)
            elif n.kind in _LUT_OF:
                name = _LUT_OF[n.kind][0]
                lo, hi, (c0, _) = self.luts[name]
                v = n.inputs[0]
                _, t.buf = ctx.trace_lut(n.kind, v.base.buf, t.size, node_id=t.node_id, input_id=v.base.node_id,
                                         lut_col1=lut_dev[name][0], lo=lo, lut_len=hi - lo + 1, mult=lut_dev[name][1],
                                         view=view_of(v), ranges=self.lut_ranges.get(name), **common)
            else:
                ins = n.inputs
                _, t.buf = ctx.trace_elementwise(
                    n.kind, ins[0].base.buf, ins[1].base.buf if len(ins) > 1 else None, t.size, node_id=t.node_id,
                    input_ids=tuple(i.base.node_id for i in ins), input_mults=tuple(-1 for _ in ins),
                    lhs_view=view_of(ins[0]), rhs_view=view_of(ins[1]) if len(ins) > 1 else None, **common)
            offset[n.kind] += rows_of(n)
        out = [(k, tables[k], total[k]) for k in tables] + [(k, b, n) for k, (b, n) in lut_tables.items()]
        return sorted(out, key=lambda e: e[0]), luts_out, bufs

    def read(self, t: GraphTensor) -> np.ndarray:
        return self.ctx.download(t.buf, np.int32).reshape(t.shape)

    # ---- gen_circuit_settings (crates/graph/src/graph.rs:61-159): a HOST dry run, as in the reference
    def _dry_run(self) -> Dict[int, np.ndarray]:
        """Every node's values as Fixed<12> integers, computed on the host with the same integer rules as the device
        kernels (`Operator::process` in the reference: plain CPU execution, no trace)."""
        S = 4096
        fn = {int(TraceTableKind.Sin): np.sin, int(TraceTableKind.Exp2): np.exp2, int(TraceTableKind.Log2): np.log2}
        K = TraceTableKind
        vals: Dict[int, np.ndarray] = {}

        def view(v):
            base = vals[v.base.node_id].reshape(-1)
            idx = np.full(v.shape, v.offset, dtype=np.int64)
            for ax, (d, st) in enumerate(zip(v.shape, v.strides)):
                shp = [1] * len(v.shape)
                shp[ax] = d
                idx = idx + (np.arange(d, dtype=np.int64) * st).reshape(shp)
            return base[idx.reshape(-1)]

        for n in self.nodes:
            k = n.kind
            if k == int(K.Inputs):
                out = n.host.astype(np.int64).reshape(-1)
            elif k in (int(K.Add), int(K.Mul), int(K.Rem), int(K.LessThan)):
                a, b = view(n.inputs[0]), view(n.inputs[1])
                out = a + b if k == int(K.Add) else (a * b) >> 12 if k == int(K.Mul) else a % b if k == int(K.Rem) \
                    else np.where(a < b, S, 0)
            elif k == int(K.Recip):
                out = (S * S) // view(n.inputs[0])
            elif k == int(K.Sqrt):
                a = view(n.inputs[0])
                out = np.array([int(np.floor(np.sqrt(float(x) * S))) for x in a], dtype=np.int64)
                out = np.where(out * out > a * S, out - 1, out)
                out = np.where((out + 1) * (out + 1) <= a * S, out + 1, out)
            elif k == int(K.Contiguous):
                out = view(n.inputs[0])
            elif k in (int(K.SumReduce), int(K.MaxReduce)):
                a = n.inputs[0].base
                x = np.moveaxis(vals[a.node_id].reshape(a.shape), n.axis, -1).reshape(-1, a.shape[n.axis])
                out = x.sum(axis=1) if k == int(K.SumReduce) else x.max(axis=1)
            elif k in fn:
                out = np.rint(fn[k](view(n.inputs[0]) / S) * S).astype(np.int64)
            else:
                raise ValueError("dry run: kind %d" % k)
            vals[n.out.node_id] = np.asarray(out, dtype=np.int64).reshape(-1)
        return vals

    def gen_circuit_settings(self):
        """`LuminairGraph::gen_circuit_settings` (crates/graph/src/graph.rs:61-159): dry-run the graph, give every
        Sin / Exp2 / Log2 node the min..max of its source BUFFER padded by 10 % of the span
        (`compute_padded_range_from_srcs` / `buffer_range`, crates/graph/src/utils.rs:44-82), coalesce overlapping or
        adjacent ranges per function (`coalesce_ranges`, graph.rs:665-691), and announce the 8-bit range check when a
        LessThan node exists.  Declares the LUTs on this graph (`set_lut_ranges`) and returns the `CircuitSettings` in
        the reference's own (serialisable) form.  Unpinned: `Fixed::from_f64` of the padded bounds is taken as
        round-to-nearest (numerair is un-vendored); a log2 range whose padding reaches <= 0 is clipped at the smallest
        positive value (the reference would emit LUT rows from log2 of a non-positive number there)."""
        from .pie import CircuitSettings, Lookup, LookupLayout, RangeCheckLookup
        S = 4096.0
        vals = self._dry_run()
        per_fn = {"sin": [], "exp2": [], "log2": []}
        for n in self.nodes:
            if n.kind in _LUT_OF:
                buf = vals[n.inputs[0].base.node_id]
                lo_f, hi_f = float(buf.min()) / S, float(buf.max()) / S
                delta = (hi_f - lo_f) * 0.10
                rnd = lambda x: int(np.floor(abs(x) * S + 0.5)) * (1 if x >= 0 else -1)
                lo, hi = rnd(lo_f - delta), rnd(hi_f + delta)
                name = _LUT_OF[n.kind][0]
                if name == "log2":
                    lo = max(lo, 1)
                per_fn[name].append((lo, hi))
        layouts = {}
        for name, rg in per_fn.items():
            if not rg:
                continue
            rg.sort()
            merged = [list(rg[0])]
            for lo, hi in rg[1:]:
                if lo <= merged[-1][1] + 1:
                    merged[-1][1] = max(merged[-1][1], hi)
                else:
                    merged.append([lo, hi])
            merged = [tuple(r) for r in merged]
            self.set_lut_ranges(name, merged)
            log_size = self.ctx.lib.lut_log_size(merged)
            layouts[name] = Lookup(LookupLayout(merged, log_size), [0] * (1 << log_size))
        rc = RangeCheckLookup([8], 8, [0] * 256) if any(n.kind == int(TraceTableKind.LessThan) for n in self.nodes) else None
        return CircuitSettings(layouts=layouts or None, range_check=rc)

    def fill_multiplicities(self, settings, tables) -> None:
        """What `gen_trace(&mut settings)` leaves in the settings' `AtomicMultiplicityColumn`s: the lookup tables'
        multiplicity columns (downloaded from the device tables `gen_trace` returned)."""
        kinds = {"sin": 4, "exp2": 10, "log2": 12}
        by_kind = {k: (b, n) for k, b, n in tables}
        for name, lk in (settings.layouts or {}).items():
            if kinds[name] in by_kind:
                b, n = by_kind[kinds[name]]
                lk.multiplicities = [int(v) for v in self.ctx.download(b.view(0, n * 4))]
        if settings.range_check is not None and 14 in by_kind:
            b, n = by_kind[14]
            settings.range_check.multiplicities = [int(v) for v in self.ctx.download(b.view(0, n * 4))]
