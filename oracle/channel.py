"""Blake2s Fiat-Shamir channel (oracle; test infrastructure only).

Restates stwo `core/channel/blake2s.rs` (un-vendored) as pinned by the KAT — SURVEY.md
Appendix A.3.  LuminAIR-owned transcript steps that drive it: `crates/air/src/lib.rs:52-104`
(`mix_u64(log_size)` per present component), `crates/air/src/components/mod.rs:173-176,209-211`
(`Claim::mix_into`, `InteractionClaim::mix_into`), `:227-235` (relation draws),
`crates/prover/src/prover.rs:177,186,296`.

Protocol flags (`ProtocolVariant`, the oracle's mirror of the LMN_PV_* bits of include/luminair_hip.h): every
observable difference between the KAT-era protocol and the sources at HEAD is one independent bit, so that a proof
made by any build of the reference can be pinned by search.  KAT (no bit set) = encodings verified against
`ui/demo/public/proof`; PINNED = all transcript bits = what stwo is believed to use at rev 0790eba4
(**parity unpinned**).
"""
from __future__ import annotations

from enum import IntFlag

from .blake2s import blake2s, compress
from .field import P, QM31


class ProtocolVariant(IntFlag):
    KAT = 0                  # KAT-era stwo / LuminAIR: every byte pinned by the reference's known-answer proof
    # transcript bits
    CLAIM17 = 0x1            # 17 Option slots in LuminairClaim (crates/air/src/lib.rs:30-48) instead of 8
    LUT_DRAWS4 = 0x2         # LookupElements::draw: sin, exp2, log2, range_check (lookups/mod.rs:44-51) instead of 1
    MIX_U64_HASHED = 0x4     # mix_u64 = blake2s(digest || lo || hi) instead of the bare compression function
    DRAW_CTR_U32 = 0x8       # draw = blake2s(digest || u32 counter || 0x00) instead of the counter padded to 32 bytes
    POW_PREFIXED = 0x10      # proof of work over a prefixed double hash (verify_pow_nonce), nonce mixed afterwards
    # constraint-form bits (numerair's eval_fixed_* helpers are un-vendored; see oracle/air.py constraint_layout)
    MUL_ONE_SLOT = 0x100
    RECIP_TWO_SLOTS = 0x200
    RECIP_NEG = 0x400
    SQRT_TWO_SLOTS = 0x800
    SQRT_NEG = 0x1000
    REM_TWO_SLOTS = 0x2000
    REM_NEG = 0x4000
    PINNED = 0x1f            # LuminAIR @ reference HEAD as believed: all transcript bits (unverified)


class Blake2sChannel:
    def __init__(self, variant: ProtocolVariant = ProtocolVariant.KAT):
        self.digest = bytes(32)
        self.n_sent = 0
        self.variant = ProtocolVariant(int(variant))

    def clone(self):
        c = Blake2sChannel(self.variant)
        c.digest, c.n_sent = self.digest, self.n_sent
        return c

    def _update(self, d: bytes):
        self.digest = d
        self.n_sent = 0

    def mix_root(self, root: bytes):
        self._update(blake2s(self.digest + root))

    def mix_felts(self, felts):
        self._update(blake2s(self.digest + b"".join(f.to_bytes() for f in felts)))

    def mix_u64(self, v: int):
        lo, hi = v & 0xFFFFFFFF, (v >> 32) & 0xFFFFFFFF
        if not self.variant & ProtocolVariant.MIX_U64_HASHED:
            h = [int.from_bytes(self.digest[4 * i:4 * i + 4], "little") for i in range(8)]
            m = [lo, hi] + [0] * 14
            out = compress(h, m, 0, 0, 0, 0)
            self._update(b"".join(x.to_bytes(4, "little") for x in out))
        else:
            self._update(blake2s(self.digest + lo.to_bytes(4, "little") + hi.to_bytes(4, "little")))

    def draw_random_bytes(self) -> bytes:
        if not self.variant & ProtocolVariant.DRAW_CTR_U32:
            ctr = self.n_sent.to_bytes(8, "little") + bytes(24)
        else:
            ctr = self.n_sent.to_bytes(4, "little") + b"\0"
        self.n_sent += 1
        return blake2s(self.digest + ctr)

    def _draw_base_felts(self):
        while True:
            b = self.draw_random_bytes()
            w = [int.from_bytes(b[4 * i:4 * i + 4], "little") for i in range(8)]
            if all(x < 2 * P for x in w):
                return [x % P for x in w]

    def draw_felt(self) -> QM31:
        f = self._draw_base_felts()
        return QM31(*f[:4])

    def draw_felts(self, n: int):
        out, pool = [], []
        while len(out) < n:
            if len(pool) < 4:
                pool += self._draw_base_felts()
            out.append(QM31(*pool[:4]))
            pool = pool[4:]
        return out

    def trailing_zeros(self) -> int:
        v = int.from_bytes(self.digest[:16], "little")
        if v == 0:
            return 128
        return (v & -v).bit_length() - 1

    # proof of work.  KAT era: the digest after mix_u64(nonce) ends in >= pow_bits zero bits.  POW_PREFIXED (stwo
    # `verify_pow_nonce` at the pinned rev, from memory): blake2s(blake2s(0x12345678 LE || 12 zero bytes || digest ||
    # pow_bits LE) || nonce LE) does; the nonce is mixed afterwards either way.
    def verify_pow_nonce(self, pow_bits: int, nonce: int) -> bool:
        if self.variant & ProtocolVariant.POW_PREFIXED:
            pre = blake2s((0x12345678).to_bytes(4, "little") + bytes(12) + self.digest + pow_bits.to_bytes(4, "little"))
            res = blake2s(pre + nonce.to_bytes(8, "little"))
            v = int.from_bytes(res[:16], "little")
            return (128 if v == 0 else (v & -v).bit_length() - 1) >= pow_bits
        c = self.clone()
        c.mix_u64(nonce)
        return c.trailing_zeros() >= pow_bits

    def grind(self, pow_bits: int) -> int:
        nonce = 0
        while not self.verify_pow_nonce(pow_bits, nonce):
            nonce += 1
        return nonce
