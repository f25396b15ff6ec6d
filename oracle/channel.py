"""Blake2s Fiat-Shamir channel (oracle; test infrastructure only).

Restates stwo `core/channel/blake2s.rs` (un-vendored) as pinned by the KAT — SURVEY.md
Appendix A.3.  LuminAIR-owned transcript steps that drive it: `crates/air/src/lib.rs:52-104`
(`mix_u64(log_size)` per present component), `crates/air/src/components/mod.rs:173-176,209-211`
(`Claim::mix_into`, `InteractionClaim::mix_into`), `:227-235` (relation draws),
`crates/prover/src/prover.rs:177,186,296`.

Variants: KAT = encodings verified against `ui/demo/public/proof`;
PINNED = the encodings stwo is believed to use at rev 0790eba4 (**parity unpinned**).
"""
from __future__ import annotations

from enum import IntEnum

from .blake2s import blake2s, compress
from .field import P, QM31


class ProtocolVariant(IntEnum):
    KAT = 0      # KAT-era stwo / LuminAIR (8-field claim, no Inputs component)
    PINNED = 1   # LuminAIR @ reference HEAD (17-field claim); channel encodings unverified


class Blake2sChannel:
    def __init__(self, variant: ProtocolVariant = ProtocolVariant.KAT):
        self.digest = bytes(32)
        self.n_sent = 0
        self.variant = variant

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
        if self.variant == ProtocolVariant.KAT:
            h = [int.from_bytes(self.digest[4 * i:4 * i + 4], "little") for i in range(8)]
            m = [lo, hi] + [0] * 14
            out = compress(h, m, 0, 0, 0, 0)
            self._update(b"".join(x.to_bytes(4, "little") for x in out))
        else:
            self._update(blake2s(self.digest + lo.to_bytes(4, "little") + hi.to_bytes(4, "little")))

    def draw_random_bytes(self) -> bytes:
        if self.variant == ProtocolVariant.KAT:
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

    # proof of work (KAT era: grind the nonce whose mix yields >= pow_bits trailing zeros)
    def grind(self, pow_bits: int) -> int:
        nonce = 0
        while True:
            c = self.clone()
            c.mix_u64(nonce)
            if c.trailing_zeros() >= pow_bits:
                return nonce
            nonce += 1
