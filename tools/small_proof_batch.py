#!/usr/bin/env python3
"""Lock-step batches (lmn_batch_prove) on the reference's own benchmark shape (32x32 Add: 2^10 Add rows + 2^11 Inputs rows,
PINNED variant, crates/graph/benches/ops.rs:92-166) and on BASELINE config 4 (black-scholes MLP shape): byte identity with
lmn_prove, proofs/s and launches per proof for B = 1 .. 64.  Usage: small_proof_batch.py [B ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import luminair_amd
from luminair_amd import backend, synthetic as syn
from luminair_amd.batch import BatchProver

print(json.dumps({"cpus_available": len(os.sched_getaffinity(0)), "cpu_count": os.cpu_count()}), flush=True)
Bs = [int(a) for a in sys.argv[1:]] or [1, 4, 16, 32, 64]
solo = luminair_amd.Prover(0, protocol_variant=backend.VARIANT_PINNED)
tabs4, luts4 = syn.config4_black_scholes_shape()
for name, mk, luts in (("32x32_add", lambda s: syn.config2_graph_faithful(1024, s), None),
                       ("config_4", lambda s: tabs4, luts4)):
    for B in Bs:
        pies = [[(k, r, len(r)) for k, r in mk(100 + i)] for i in range(B)]
        want = [solo.ctx.prove_tables(p, luts) for p in pies[:4]]
        t0 = time.perf_counter()
        for _ in range(5):
            solo.ctx.prove_tables(pies[0], luts)
        solo_ms = 1e3 * (time.perf_counter() - t0) / 5
        bp = BatchProver(0, B, protocol_variant=backend.VARIANT_PINNED)
        got = bp.prove_batch(pies, luts)
        same = got[:4] == want
        c0 = bp.counters()
        for _ in range(3):
            bp.prove_batch(pies, luts)
        reps = max(5, 400 // B)
        t0 = time.perf_counter()
        for _ in range(reps):
            bp.prove_batch(pies, luts)
        dt = time.perf_counter() - t0
        c1 = bp.counters()
        print(json.dumps({"workload": name, "batch": B, "bytes_identical_to_lmn_prove": same,
                          "proofs_per_s": round(B * reps / dt, 1), "ms_per_batch": round(1e3 * dt / reps, 3),
                          "solo_lmn_prove_ms": round(solo_ms, 3),
                          "launches_per_batch": (c1["launches"] - c0["launches"]) // (reps + 3),
                          "host_waits_per_batch": (c1["host_waits"] - c0["host_waits"]) // (reps + 3),
                          "copy_launches_per_batch": (c1["copy_launches"] - c0["copy_launches"]) // (reps + 3),
                          "direct_copies_per_batch": (c1["direct_copies"] - c0["direct_copies"]) / (reps + 3),
                          "arrival_skew_ms_per_batch": round((c1["arrival_skew_ms"] - c0["arrival_skew_ms"]) / (reps + 3), 3),
                          "leader_ms_per_batch": round((c1["leader_ms"] - c0["leader_ms"]) / (reps + 3), 3),
                          "member_host_ms_per_proof": round((c1["member_busy_ms"] - c0["member_busy_ms"]) / (reps + 3) / B, 3)}),
              flush=True)
        bp.close()
