"""Device-side `gen_trace` for graphs of the primitives whose `process_trace` runs on the GPU
(Add, Mul, Recip, SumReduce and the graph inputs): a small host mirror of
`LuminairGraph::gen_trace` (`crates/graph/src/graph.rs:161-604`) over `lmn_trace_elementwise` /
`lmn_trace_sum_reduce`.  Nodes execute in creation (= topological) order, every tensor stays in HBM
as int32 `Fixed<12>` values, each node appends its rows to its kind's device-resident table, and the
resulting pie feeds `lmn_prove` with `LMN_TABLE_ROWS_ON_DEVICE`.

Multiplicities follow HEAD (`crates/graph/src/op/prim.rs:66-83,1005-1006`): an op consumes each input
with multiplicity -1 and yields its output `num_consumers` times (0 for a final output); a graph input
is yielded `num_consumers` times by its Inputs rows — so the logup sums of a complete graph cancel.
The graph front-end itself (luminal's compiler passes, shape tracking, f32 -> fixed conversion rules)
is outside the hot-path scope; this is the execution + table-fill step only."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import backend
from .pie import TraceTableKind

_NCOLS = {0: 15, 1: 16, 2: 13, 5: 14, 15: 7}


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

    def _binary(self, kind, a: GraphTensor, b: GraphTensor) -> GraphTensor:
        if a.shape != b.shape:
            raise ValueError("elementwise operands must have equal shapes (no broadcasting in this mirror)")
        a.consumers += 1
        b.consumers += 1
        t = self._tensor(a.shape)
        self.nodes.append(_Node(kind, t, [a, b]))
        return t

    def add(self, a, b):
        return self._binary(int(TraceTableKind.Add), a, b)

    def mul(self, a, b):
        return self._binary(int(TraceTableKind.Mul), a, b)

    def recip(self, a: GraphTensor) -> GraphTensor:
        a.consumers += 1
        t = self._tensor(a.shape)
        self.nodes.append(_Node(int(TraceTableKind.Recip), t, [a]))
        return t

    def sum_reduce(self, a: GraphTensor, axis: int) -> GraphTensor:
        a.consumers += 1
        shape = a.shape[:axis] + a.shape[axis + 1:]
        t = self._tensor(shape if shape else (1,))
        self.nodes.append(_Node(int(TraceTableKind.SumReduce), t, [a], axis=axis))
        return t

    def output(self, t: GraphTensor) -> GraphTensor:
        t.is_output = True
        return t

    def gen_trace(self):
        """Runs every node on the device.  Returns (tables, buffers): tables = [(kind, rows DeviceBuffer,
        n_rows)] in `gen_trace` order (ascending kind), ready for Context.prove_tables; buffers = every device
        allocation made (free them after proving)."""
        ctx = self.ctx
        rows_of = lambda n: n.inputs[0].size if n.kind == int(TraceTableKind.SumReduce) else n.out.size
        total: Dict[int, int] = {}
        for n in self.nodes:
            total[n.kind] = total.get(n.kind, 0) + rows_of(n)
        tables = {k: ctx.alloc(total[k] * _NCOLS[k] * 4) for k in total}
        offset = {k: 0 for k in total}
        bufs = list(tables.values())
        for n in self.nodes:
            t = n.out
            common = dict(num_consumers=t.consumers, is_final_output=t.is_output, rows=tables[n.kind],
                          row_offset=offset[n.kind])
            if n.kind == int(TraceTableKind.Inputs):
                src = ctx.upload(n.host.reshape(-1))
                bufs.append(src)
                _, t.buf = ctx.trace_elementwise(n.kind, src, None, t.size, node_id=t.node_id, input_ids=(),
                                                 input_mults=(), **common)
            elif n.kind == int(TraceTableKind.SumReduce):
                a = n.inputs[0]
                front = int(np.prod(a.shape[:n.axis])) if n.axis else 1
                back = int(np.prod(a.shape[n.axis + 1:])) if n.axis + 1 < len(a.shape) else 1
                _, t.buf = ctx.trace_sum_reduce(a.buf, front, a.shape[n.axis], back, node_id=t.node_id,
                                                input_id=a.node_id, **common)
            else:
                ins = n.inputs
                _, t.buf = ctx.trace_elementwise(n.kind, ins[0].buf, ins[1].buf if len(ins) > 1 else None, t.size,
                                                 node_id=t.node_id, input_ids=tuple(i.node_id for i in ins),
                                                 input_mults=tuple(-1 for _ in ins), **common)
            bufs.append(t.buf)
            offset[n.kind] += rows_of(n)
        return [(k, tables[k], total[k]) for k in sorted(tables)], bufs

    def read(self, t: GraphTensor) -> np.ndarray:
        return self.ctx.download(t.buf, np.int32).reshape(t.shape)
