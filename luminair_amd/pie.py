"""Boundary data types, named after the reference's (`crates/air/src/pie.rs:31-66,143-210`,
`crates/air/src/settings.rs`, `crates/prover/src/lib.rs:15-32`, `crates/utils/src/lib.rs:5-34`)."""
from __future__ import annotations

from dataclasses import dataclass, field
from enum import IntEnum
from typing import List, Optional

import numpy as np


class TraceTableKind(IntEnum):
    """`enum TraceTable` variant order (pie.rs:31-66)."""
    Add = 0
    Mul = 1
    Recip = 2
    Sin = 3
    SinLookup = 4
    SumReduce = 5
    MaxReduce = 6
    Sqrt = 7
    Rem = 8
    Exp2 = 9
    Exp2Lookup = 10
    Log2 = 11
    Log2Lookup = 12
    LessThan = 13
    RangeCheckLookup = 14
    Inputs = 15
    Contiguous = 16


# column counts of the components on the hot path (N_TRACE_COLUMNS in each witness.rs)
N_COLUMNS = {TraceTableKind.Add: 15, TraceTableKind.Mul: 16, TraceTableKind.Recip: 13, TraceTableKind.Inputs: 7,
             TraceTableKind.SumReduce: 14, TraceTableKind.MaxReduce: 15, TraceTableKind.Contiguous: 11,
             TraceTableKind.LessThan: 22, TraceTableKind.RangeCheckLookup: 1, TraceTableKind.Sqrt: 13,
             TraceTableKind.Rem: 16, TraceTableKind.Sin: 12, TraceTableKind.Exp2: 12, TraceTableKind.Log2: 12,
             TraceTableKind.SinLookup: 1, TraceTableKind.Exp2Lookup: 1, TraceTableKind.Log2Lookup: 1}


class LuminairError(Exception):
    """`LuminairError` (crates/utils/src/lib.rs:5-34); `.variant` names the Rust variant."""

    def __init__(self, variant: str, message: str = "", code: int = 0):
        super().__init__("%s%s" % (variant, (": " + message) if message else ""))
        self.variant, self.code = variant, code


@dataclass
class TraceTable:
    """One `TraceTable::X { table }`: AoS rows of canonical M31 in `Column::index()` order."""
    kind: TraceTableKind
    rows: np.ndarray  # (n_rows, n_columns) uint32

    @staticmethod
    def from_rows(kind, rows) -> "TraceTable":
        kind = TraceTableKind(kind)
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        if kind not in N_COLUMNS:
            raise LuminairError("InvalidArgument", "component %s is outside the hot-path scope" % kind.name)
        rows = rows.reshape(-1, N_COLUMNS[kind])
        return TraceTable(kind, rows)

    @property
    def n_rows(self) -> int:
        return int(self.rows.shape[0])


@dataclass
class ExecutionResources:
    max_log_size: int = 0


@dataclass
class Metadata:
    execution_resources: ExecutionResources = field(default_factory=ExecutionResources)


@dataclass
class LuminairPie:
    """`LuminairPie { trace_tables, metadata }` (pie.rs:143-150)."""
    trace_tables: List[TraceTable]
    metadata: Metadata = field(default_factory=Metadata)

    @staticmethod
    def from_tables(tables) -> "LuminairPie":
        tts = [t if isinstance(t, TraceTable) else TraceTable.from_rows(t[0], t[1]) for t in tables]
        mx = 0
        for t in tts:
            size = max(16, 1 << max(t.n_rows - 1, 0).bit_length())
            mx = max(mx, size.bit_length() - 1)
        return LuminairPie(tts, Metadata(ExecutionResources(mx)))


@dataclass
class CircuitSettings:
    """`CircuitSettings { lookups }` (crates/air/src/settings.rs).  `lookups` maps "sin" / "exp2" /
    "log2" to that LUT's two preprocessed columns (col0 = inputs, col1 = outputs; uint32 M31 words,
    2^k rows), i.e. what `lookups_to_preprocessed_column` + `gen_column_simd` produce from the
    reference's layouts; the 8-bit range-check LUT is implied by a RangeCheckLookup table."""
    lookups: Optional[dict] = None

    def lut_columns(self) -> dict:
        out = {}
        for name, cols in (self.lookups or {}).items():
            if name == "range_check":
                continue
            if name not in ("sin", "exp2", "log2"):
                raise LuminairError("InvalidArgument", "unknown lookup " + name)
            out[name] = cols
        return out

    def to_bincode(self, kat_era: bool = False) -> bytes:
        """bincode of `CircuitSettings { lookups: Lookups }` for LUT-free graphs: one `None` tag per
        `Lookups` field (HEAD: sin, exp2, log2, range_check; the KAT era had `sin` only, which is what
        `ui/demo/public/settings` holds).  Settings WITH lookups carry numerair `Fixed` ranges and an
        stwo-air-utils multiplicity column whose wire formats are un-vendored: not serialised here."""
        if self.lookups:
            raise LuminairError("SerializationError", "LUT layouts (value ranges) are not carried by this mirror")
        return bytes(1 if kat_era else 4)

    @staticmethod
    def from_bincode(data: bytes) -> "CircuitSettings":
        if len(data) not in (1, 4) or any(data):
            raise LuminairError("SerializationError", "only LUT-free settings can be deserialised")
        return CircuitSettings()


@dataclass
class LuminairProof:
    """`LuminairProof<Blake2sMerkleHasher>` carried as its bincode bytes (`to_bincode`)."""
    bincode: bytes

    def to_bincode(self) -> bytes:
        return self.bincode

    def to_bincode_file(self, path):
        with open(path, "wb") as f:
            f.write(self.bincode)
