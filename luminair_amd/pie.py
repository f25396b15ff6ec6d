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
class LookupLayout:
    """`LookupLayout { ranges: Vec<Range(Fixed, Fixed)>, log_size }` (crates/air/src/preprocessed.rs:34-46)."""
    ranges: List[tuple]
    log_size: int


@dataclass
class Lookup:
    """`SinLookup { layout, multiplicities }` (crates/air/src/components/lookups/sin/mod.rs:20-24; Exp2Lookup and
    Log2Lookup alike).  `multiplicities` = the `AtomicMultiplicityColumn`'s 2^log_size counters."""
    layout: LookupLayout
    multiplicities: List[int]


@dataclass
class RangeCheckLookup:
    """`RangeCheckLookup<1> { layout: RangeCheckLayout { ranges: [u32; 1], log_size }, multiplicities }`
    (crates/air/src/components/lookups/range_check/mod.rs:23-37; graph.rs:142-146 builds it with ranges [8], log 8)."""
    ranges: List[int]
    log_size: int
    multiplicities: List[int]


_LOOKUP_FIELDS = ("sin", "exp2", "log2", "range_check")   # `Lookups` field order, lookups/mod.rs:18-28


@dataclass
class CircuitSettings:
    """`CircuitSettings { lookups: Lookups }` (crates/air/src/settings.rs:14-17).  Two ways to describe a LUT:
    * the reference's own form - `layouts[name] = Lookup(LookupLayout(ranges, log_size), multiplicities)` and
      `range_check = RangeCheckLookup(...)`: this is what (de)serialises (bincode 1.3 and serde-JSON) and from which
      the preprocessed columns are generated behind the boundary (`lmn_lut_from_ranges`);
    * pre-expanded columns - `lookups[name] = (col0, col1)` (uint32 M31 words, 2^k rows) exactly as
      `lookups_to_preprocessed_column` + `gen_column_simd` produce them; not serialisable (the reference has no
      such form), kept for callers that generate LUTs themselves."""
    lookups: Optional[dict] = None
    layouts: Optional[dict] = None
    range_check: Optional[RangeCheckLookup] = None

    def lut_columns(self, library=None) -> dict:
        out = {}
        for name, cols in (self.lookups or {}).items():
            if name == "range_check":
                continue
            if name not in ("sin", "exp2", "log2"):
                raise LuminairError("InvalidArgument", "unknown lookup " + name)
            out[name] = cols
        if self.layouts:
            from . import backend
            lib = library or backend.default_library()
            for name, lk in self.layouts.items():
                if name not in ("sin", "exp2", "log2"):
                    raise LuminairError("InvalidArgument", "unknown lookup " + name)
                if name not in out:
                    out[name] = lib.lut_from_ranges(name, lk.layout.ranges, lk.layout.log_size)
        return out

    # ---- bincode 1.3 (fixed-width little-endian ints, u64 lengths, Option = one tag byte): numerair's `Fixed` is a
    # newtype over i64 and serialises as that i64; `AtomicU32` as u32; `[u32; 1]` (serde_as) without a length
    def to_bincode(self, kat_era: bool = False) -> bytes:
        """bincode of `CircuitSettings`.  HEAD has four `Lookups` fields (sin, exp2, log2, range_check); the KAT era
        had `sin` only, which is what `ui/demo/public/settings` holds (one `None` tag)."""
        import struct
        if self.lookups and not self.layouts:
            raise LuminairError("SerializationError",
                                "pre-expanded LUT columns have no wire form: describe the LUTs as LookupLayout ranges")
        out = bytearray()
        for name in _LOOKUP_FIELDS[:1] if kat_era else _LOOKUP_FIELDS:
            if name == "range_check":
                rc = self.range_check
                if rc is None:
                    out += b"\x00"
                    continue
                out += b"\x01" + b"".join(struct.pack("<I", v) for v in rc.ranges) + struct.pack("<I", rc.log_size)
                out += struct.pack("<Q", len(rc.multiplicities)) + struct.pack("<%dI" % len(rc.multiplicities), *rc.multiplicities)
                continue
            lk = (self.layouts or {}).get(name)
            if lk is None:
                out += b"\x00"
                continue
            out += b"\x01" + struct.pack("<Q", len(lk.layout.ranges))
            for lo, hi in lk.layout.ranges:
                out += struct.pack("<qq", int(lo), int(hi))
            out += struct.pack("<I", lk.layout.log_size)
            out += struct.pack("<Q", len(lk.multiplicities)) + struct.pack("<%dI" % len(lk.multiplicities), *lk.multiplicities)
        return bytes(out)

    @staticmethod
    def from_bincode(data: bytes, kat_era: Optional[bool] = None) -> "CircuitSettings":
        import struct
        if kat_era is None:
            kat_era = len(data) == 1
        pos = [0]

        def take(fmt):
            n = struct.calcsize(fmt)
            if pos[0] + n > len(data):
                raise LuminairError("SerializationError", "truncated settings")
            v = struct.unpack_from(fmt, data, pos[0])
            pos[0] += n
            return v

        def tag():
            t = take("<B")[0]
            if t > 1:
                raise LuminairError("SerializationError", "bad Option tag")
            return t == 1

        def vec_len(elem):
            n = take("<Q")[0]
            if n * elem > len(data) - pos[0]:
                raise LuminairError("SerializationError", "bad length")
            return n
        layouts, rc = {}, None
        for name in _LOOKUP_FIELDS[:1] if kat_era else _LOOKUP_FIELDS:
            if not tag():
                continue
            if name == "range_check":
                bound, log = take("<II")
                m = list(take("<%dI" % vec_len(4)))
                rc = RangeCheckLookup([bound], log, m)
                continue
            ranges = [take("<qq") for _ in range(vec_len(16))]
            log = take("<I")[0]
            m = list(take("<%dI" % vec_len(4)))
            layouts[name] = Lookup(LookupLayout(ranges, log), m)
        if pos[0] != len(data):
            raise LuminairError("SerializationError", "trailing bytes")
        return CircuitSettings(None, layouts or None, rc)

    # ---- serde-JSON (`to_json` = serde_json::to_string_pretty, settings.rs:71-76): tuple structs are arrays,
    # newtypes their inner value
    def to_json(self) -> str:
        import json
        if self.lookups and not self.layouts:
            raise LuminairError("SerializationError",
                                "pre-expanded LUT columns have no wire form: describe the LUTs as LookupLayout ranges")
        lk = {}
        for name in _LOOKUP_FIELDS[:3]:
            v = (self.layouts or {}).get(name)
            lk[name] = None if v is None else {
                "layout": {"ranges": [[int(a), int(b)] for a, b in v.layout.ranges], "log_size": v.layout.log_size},
                "multiplicities": {"data": list(v.multiplicities)}}
        rc = self.range_check
        lk["range_check"] = None if rc is None else {
            "layout": {"ranges": list(rc.ranges), "log_size": rc.log_size}, "multiplicities": {"data": list(rc.multiplicities)}}
        return json.dumps({"lookups": lk}, indent=2)

    @staticmethod
    def from_json(text: str) -> "CircuitSettings":
        import json
        try:
            lk = json.loads(text)["lookups"]
            layouts = {}
            for name in _LOOKUP_FIELDS[:3]:
                v = lk.get(name)
                if v is not None:
                    layouts[name] = Lookup(LookupLayout([(int(a), int(b)) for a, b in v["layout"]["ranges"]],
                                                        int(v["layout"]["log_size"])),
                                           [int(x) for x in v["multiplicities"]["data"]])
            rc = lk.get("range_check")
            rcl = None if rc is None else RangeCheckLookup([int(x) for x in rc["layout"]["ranges"]],
                                                           int(rc["layout"]["log_size"]),
                                                           [int(x) for x in rc["multiplicities"]["data"]])
        except (KeyError, TypeError, ValueError) as e:
            raise LuminairError("SerializationError", "Failed to deserialize settings from JSON: %s" % e)
        return CircuitSettings(None, layouts or None, rcl)


_CLAIM_FIELDS = ("add", "mul", "recip", "sin", "sin_lookup", "sum_reduce", "max_reduce", "sqrt", "rem", "exp2",
                 "exp2_lookup", "log2", "log2_lookup", "less_than", "range_check_lookup", "inputs",
                 "contiguous")   # LuminairClaim / LuminairInteractionClaim field order, crates/air/src/lib.rs:30-48


class _BinReader:
    """bincode 1.3 reader for the proof layout: fixed-width little-endian, u64 lengths."""

    def __init__(self, data: bytes):
        self.b, self.o = data, 0

    def take(self, fmt):
        import struct
        n = struct.calcsize(fmt)
        if self.o + n > len(self.b):
            raise LuminairError("SerializationError", "truncated proof")
        v = struct.unpack_from(fmt, self.b, self.o)
        self.o += n
        return v

    def tag(self):
        t = self.take("<B")[0]
        if t > 1:
            raise LuminairError("SerializationError", "bad Option tag")
        return t == 1

    def length(self, elem):
        n = self.take("<Q")[0]
        if n * elem > len(self.b) - self.o:
            raise LuminairError("SerializationError", "bad length")
        return n

    def q(self):
        a, b, c, d = self.take("<4I")
        return [[a, b], [c, d]]        # QM31(CM31(a, b), CM31(c, d)) as serde writes tuple structs

    def hash(self):
        return list(self.take("<32B"))  # Blake2sHash([u8; 32])

    def decommit(self):
        return {"hash_witness": [self.hash() for _ in range(self.length(32))],
                "column_witness": list(self.take("<%dI" % self.length(4)))}

    def layer(self):
        return {"fri_witness": [self.q() for _ in range(self.length(16))], "decommitment": self.decommit(),
                "commitment": self.hash()}


@dataclass
class LuminairProof:
    """`LuminairProof<Blake2sMerkleHasher>` carried as its bincode bytes (`to_bincode`).

    `to_json` / `from_json` mirror `serde_json` of the same struct (`crates/prover/src/lib.rs:62-106`).  The field names
    of `LuminairProof`, `LuminairClaim`, `Claim` and `InteractionClaim` are the reference's (`prover/src/lib.rs:16-20`,
    `air/src/lib.rs:30-48`, `air/src/components/mod.rs:150-211`); those of stwo's `CommitmentSchemeProof`, `PcsConfig`,
    `FriConfig`, `FriProof`, `FriLayerProof`, `MerkleDecommitment` and `LinePoly` come from the un-vendored crate and
    are restated from its published sources - *unpinned*, like the rest of the HEAD wire format.  The
    field ORDER is the one the reference's known-answer proof confirms for bincode."""
    bincode: bytes

    def to_bincode(self) -> bytes:
        return self.bincode

    def to_bincode_file(self, path):
        with open(path, "wb") as f:
            f.write(self.bincode)

    def to_dict(self, kat_era: Optional[bool] = None) -> dict:
        r = _BinReader(self.bincode)
        if kat_era is None:          # 8 claim slots (KAT era) or 17 (HEAD): try HEAD first, fall back
            for guess in (False, True):
                try:
                    return self.to_dict(guess)
                except LuminairError:
                    continue
            raise LuminairError("SerializationError", "not a LuminairProof")
        names = _CLAIM_FIELDS[:8] if kat_era else _CLAIM_FIELDS
        # `Claim<T> { log_size, _marker: PhantomData<T> }` (components/mod.rs:149-152): serde writes PhantomData as
        # null and `Deserialize` insists on the field, so the JSON form carries `"_marker": null`
        claim = {n: ({"log_size": r.take("<I")[0], "_marker": None} if r.tag() else None) for n in names}
        iclaim = {n: ({"claimed_sum": r.q()} if r.tag() else None) for n in names}
        pow_bits, log_blowup, log_last = r.take("<III")
        n_queries = r.take("<Q")[0]
        commitments = [r.hash() for _ in range(r.length(32))]
        sampled = [[[r.q() for _ in range(r.length(16))] for _ in range(r.length(8))] for _ in range(r.length(8))]
        decommitments = [r.decommit() for _ in range(r.length(16))]
        queried = [list(r.take("<%dI" % r.length(4))) for _ in range(r.length(8))]
        pow_nonce = r.take("<Q")[0]
        first = r.layer()
        inner = [r.layer() for _ in range(r.length(40))]
        coeffs = [r.q() for _ in range(r.length(16))]
        ll_log = r.take("<I")[0]
        if r.o != len(self.bincode) or not (4 <= len(commitments) <= 4):
            raise LuminairError("SerializationError", "trailing bytes or wrong claim layout")
        return {"claim": claim, "interaction_claim": iclaim, "proof": {
            "config": {"pow_bits": pow_bits, "fri_config": {"log_blowup_factor": log_blowup,
                                                            "log_last_layer_degree_bound": log_last, "n_queries": n_queries}},
            "commitments": commitments, "sampled_values": sampled, "decommitments": decommitments,
            "queried_values": queried, "proof_of_work": pow_nonce,
            "fri_proof": {"first_layer": first, "inner_layers": inner,
                          "last_layer_poly": {"coeffs": coeffs, "log_size": ll_log}}}}

    def to_json(self, kat_era: Optional[bool] = None) -> str:
        import json
        return json.dumps(self.to_dict(kat_era), indent=2)

    @staticmethod
    def from_json(text: str) -> "LuminairProof":
        import json
        import struct
        try:
            d = json.loads(text)
            out = bytearray()
            names = list(d["claim"].keys())
            if tuple(names) not in (_CLAIM_FIELDS, _CLAIM_FIELDS[:8]) or list(d["interaction_claim"].keys()) != names:
                raise ValueError("unexpected claim fields")
            for n in names:
                c = d["claim"][n]
                out += b"\x00" if c is None else b"\x01" + struct.pack("<I", c["log_size"])
            q = lambda v: struct.pack("<4I", v[0][0], v[0][1], v[1][0], v[1][1])
            for n in names:
                c = d["interaction_claim"][n]
                out += b"\x00" if c is None else b"\x01" + q(c["claimed_sum"])
            p = d["proof"]
            fc = p["config"]["fri_config"]
            out += struct.pack("<IIIQ", p["config"]["pow_bits"], fc["log_blowup_factor"], fc["log_last_layer_degree_bound"],
                               fc["n_queries"])
            h = lambda v: bytes(v) if len(v) == 32 else (_ for _ in ()).throw(ValueError("hash length"))

            def decommit(m):
                b = struct.pack("<Q", len(m["hash_witness"])) + b"".join(h(x) for x in m["hash_witness"])
                return b + struct.pack("<Q", len(m["column_witness"])) + struct.pack("<%dI" % len(m["column_witness"]), *m["column_witness"])

            def layer(l):
                return (struct.pack("<Q", len(l["fri_witness"])) + b"".join(q(x) for x in l["fri_witness"]) + decommit(l["decommitment"])
                        + h(l["commitment"]))
            out += struct.pack("<Q", len(p["commitments"])) + b"".join(h(x) for x in p["commitments"])
            out += struct.pack("<Q", len(p["sampled_values"]))
            for tree in p["sampled_values"]:
                out += struct.pack("<Q", len(tree))
                for col in tree:
                    out += struct.pack("<Q", len(col)) + b"".join(q(x) for x in col)
            out += struct.pack("<Q", len(p["decommitments"])) + b"".join(decommit(m) for m in p["decommitments"])
            out += struct.pack("<Q", len(p["queried_values"]))
            for t in p["queried_values"]:
                out += struct.pack("<Q", len(t)) + struct.pack("<%dI" % len(t), *t)
            out += struct.pack("<Q", p["proof_of_work"])
            fp = p["fri_proof"]
            out += layer(fp["first_layer"]) + struct.pack("<Q", len(fp["inner_layers"])) + b"".join(layer(l) for l in fp["inner_layers"])
            ll = fp["last_layer_poly"]
            out += struct.pack("<Q", len(ll["coeffs"])) + b"".join(q(x) for x in ll["coeffs"]) + struct.pack("<I", ll["log_size"])
        except (KeyError, TypeError, ValueError, struct.error) as e:
            raise LuminairError("SerializationError", "Failed to deserialize proof from JSON: %s" % e)
        return LuminairProof(bytes(out))
