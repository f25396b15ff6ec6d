#!/usr/bin/env python3
"""Parity soak: random component mixes (all 17 kinds, ragged sizes, LUTs) proved on the GPU and by the C oracle, byte for
byte, over many seeds and three size scales.  Usage: soak_random.py [n_seeds] [big] [flags] (default 48 seeds).  With
`flags` every seed also draws its own protocol flags (LMN_PV_*: encodings, proof-of-work form, constraint-form slots and
signs - claim layout and LUT draws stay at HEAD's, which the random pies need).  Test infrastructure (uses oracle/ as the
checker)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import luminair_amd
from level2_checks import random_pie
from oracle.channel import ProtocolVariant
from oracle.proof import to_bincode
from oracle.prover import prove
from oracle.cbackend import CKernels

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
# "big": tables of up to 2^20 rows - random size mixes across the thresholds of the tree storage forms (MerkleCut depth 1 - 3,
# leaf levels fused into the level above, quotient columns joining FRI layers)
scales = (400, 700, 1100, 1500) if "big" in sys.argv[2:] else (1, 40, 150, 400)
B = luminair_amd.backend
rand_flags = "flags" in sys.argv[2:]
p = luminair_amd.Prover(0, protocol_variant=B.VARIANT_PINNED)
ck = CKernels()
FREE_BITS = [B.PV_MIX_U64_HASHED, B.PV_DRAW_CTR_U32, B.PV_POW_PREFIXED, B.PV_MUL_ONE_SLOT, B.PV_RECIP_TWO_SLOTS, B.PV_RECIP_NEG,
             B.PV_SQRT_TWO_SLOTS, B.PV_SQRT_NEG, B.PV_REM_TWO_SLOTS, B.PV_REM_NEG]
bad, t0, rows = [], time.time(), 0
seed0 = int(os.environ.get("SOAK_SEED0", "100"))   # other seeds than the default run's
for seed in range(seed0, seed0 + n):
    scale = scales[seed % 4]
    tabs, luts = random_pie(seed, scale)
    rows += sum(len(r) for _, r in tabs)
    flags = B.VARIANT_PINNED
    if rand_flags:
        import random
        rng = random.Random(seed)
        flags = B.PV_CLAIM17 | B.PV_LUT_DRAWS4 | sum(b for b in FREE_BITS if rng.random() < 0.5)
        p.ctx.close()
        p = luminair_amd.Prover(0, protocol_variant=flags)
    got = p.prove(luminair_amd.LuminairPie.from_tables(tabs), luminair_amd.CircuitSettings(luts)).to_bincode()
    want = to_bincode(prove(tabs, variant=ProtocolVariant(flags), kernels=ck, luts=luts))
    if rand_flags:   # the product verifier under the same flags (random pies do not balance their logup sums: that verdict is expected)
        try:
            luminair_amd.verify(luminair_amd.LuminairProof(got), protocol_variant=flags)
        except luminair_amd.LuminairError as e:
            if e.variant != "InvalidLogUp":
                raise
    if got != want:
        bad.append(seed)
print("random pies: %d seeds, %d trace rows in total, %.0f s, mismatching seeds: %s" % (n, rows, time.time() - t0, bad))
sys.exit(1 if bad else 0)
