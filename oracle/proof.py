"""`LuminairProof` container + bincode 1.3 wire format (oracle; test infrastructure only).

Restates `crates/prover/src/lib.rs:15-32` (`LuminairProof{claim, interaction_claim, proof}`,
`to_bincode`), `crates/air/src/lib.rs:30-48` (claim = one `Option<Claim{log_size:u32}>` per
component kind), `crates/air/src/components/mod.rs:202-205` (`InteractionClaim{claimed_sum}`) and
stwo's `StarkProof`/`CommitmentSchemeProof`/`FriProof` serde layout, as pinned by the KAT parse —
SURVEY.md Appendix A.9.  bincode: little-endian, u64 lengths, `Option` = 1 tag byte.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List, Optional

from .field import QM31


@dataclass
class Decommitment:
    hash_witness: List[bytes] = field(default_factory=list)
    column_witness: List[int] = field(default_factory=list)


@dataclass
class FriLayerProof:
    fri_witness: List[QM31]
    decommitment: Decommitment
    commitment: bytes


@dataclass
class StarkProof:
    pow_bits: int
    log_blowup: int
    log_last_layer: int
    n_queries: int
    commitments: List[bytes]
    sampled_values: List[List[List[QM31]]]
    decommitments: List[Decommitment]
    queried_values: List[List[int]]
    proof_of_work: int
    first_layer: FriLayerProof
    inner_layers: List[FriLayerProof]
    last_layer_coeffs: List[QM31]
    last_layer_log_size: int


@dataclass
class LuminairProof:
    claim: List[Optional[int]]                 # log_size per component kind slot
    interaction_claim: List[Optional[QM31]]    # claimed_sum per slot
    proof: StarkProof


class _W:
    def __init__(self):
        self.b = bytearray()

    def u8(self, v):
        self.b += struct.pack("<B", v)

    def u32(self, v):
        self.b += struct.pack("<I", v)

    def u64(self, v):
        self.b += struct.pack("<Q", v)

    def raw(self, b):
        self.b += b

    def q(self, f: QM31):
        self.b += f.to_bytes()


def _w_decommit(w: _W, d: Decommitment):
    w.u64(len(d.hash_witness))
    for h in d.hash_witness:
        w.raw(h)
    w.u64(len(d.column_witness))
    for v in d.column_witness:
        w.u32(v)


def _w_layer(w: _W, l: FriLayerProof):
    w.u64(len(l.fri_witness))
    for f in l.fri_witness:
        w.q(f)
    _w_decommit(w, l.decommitment)
    w.raw(l.commitment)


def to_bincode(p: LuminairProof) -> bytes:
    w = _W()
    for c in p.claim:
        if c is None:
            w.u8(0)
        else:
            w.u8(1)
            w.u32(c)
    for c in p.interaction_claim:
        if c is None:
            w.u8(0)
        else:
            w.u8(1)
            w.q(c)
    s = p.proof
    w.u32(s.pow_bits)
    w.u32(s.log_blowup)
    w.u32(s.log_last_layer)
    w.u64(s.n_queries)
    w.u64(len(s.commitments))
    for c in s.commitments:
        w.raw(c)
    w.u64(len(s.sampled_values))
    for tree in s.sampled_values:
        w.u64(len(tree))
        for col in tree:
            w.u64(len(col))
            for v in col:
                w.q(v)
    w.u64(len(s.decommitments))
    for d in s.decommitments:
        _w_decommit(w, d)
    w.u64(len(s.queried_values))
    for t in s.queried_values:
        w.u64(len(t))
        for v in t:
            w.u32(v)
    w.u64(s.proof_of_work)
    _w_layer(w, s.first_layer)
    w.u64(len(s.inner_layers))
    for l in s.inner_layers:
        _w_layer(w, l)
    w.u64(len(s.last_layer_coeffs))
    for c in s.last_layer_coeffs:
        w.q(c)
    w.u32(s.last_layer_log_size)
    return bytes(w.b)


class _R:
    def __init__(self, b):
        self.b, self.o = b, 0

    def take(self, n):
        if self.o + n > len(self.b):
            raise ValueError("SerializationError: truncated")
        r = self.b[self.o:self.o + n]
        self.o += n
        return r

    def u8(self):
        return self.take(1)[0]

    def u32(self):
        return struct.unpack("<I", self.take(4))[0]

    def u64(self):
        return struct.unpack("<Q", self.take(8))[0]

    def m31(self):
        v = self.u32()
        if v >= (1 << 31) - 1:
            raise ValueError("SerializationError: field word is not a canonical M31")
        return v

    def tag(self):
        t = self.u8()
        if t > 1:
            raise ValueError("SerializationError: bad Option tag")
        return t

    def q(self):
        return QM31(self.m31(), self.m31(), self.m31(), self.m31())


def _r_decommit(r: _R) -> Decommitment:
    hw = [bytes(r.take(32)) for _ in range(r.u64())]
    cw = [r.m31() for _ in range(r.u64())]
    return Decommitment(hw, cw)


def _r_layer(r: _R) -> FriLayerProof:
    fw = [r.q() for _ in range(r.u64())]
    d = _r_decommit(r)
    return FriLayerProof(fw, d, bytes(r.take(32)))


def from_bincode(data: bytes, n_claim_slots: int) -> LuminairProof:
    r = _R(data)
    claim = []
    for _ in range(n_claim_slots):
        claim.append(r.u32() if r.tag() else None)
        if claim[-1] is not None and claim[-1] > 26:
            raise ValueError("SerializationError: claim log_size out of range")
    iclaim = []
    for _ in range(n_claim_slots):
        iclaim.append(r.q() if r.tag() else None)
    pow_bits, log_blowup, log_last = r.u32(), r.u32(), r.u32()
    n_queries = r.u64()
    commitments = [bytes(r.take(32)) for _ in range(r.u64())]
    sampled = []
    for _ in range(r.u64()):
        tree = []
        for _ in range(r.u64()):
            tree.append([r.q() for _ in range(r.u64())])
        sampled.append(tree)
    decommitments = [_r_decommit(r) for _ in range(r.u64())]
    queried = [[r.m31() for _ in range(r.u64())] for _ in range(r.u64())]
    pow_nonce = r.u64()
    first = _r_layer(r)
    inner = [_r_layer(r) for _ in range(r.u64())]
    coeffs = [r.q() for _ in range(r.u64())]
    ll_log = r.u32()
    if r.o != len(data):
        raise ValueError("SerializationError: trailing bytes")
    return LuminairProof(claim, iclaim, StarkProof(pow_bits, log_blowup, log_last, n_queries, commitments,
                                                  sampled, decommitments, queried, pow_nonce, first, inner,
                                                  coeffs, ll_log))
