#!/usr/bin/env python3
"""Several lock-step batch groups at once (G x lmn_batch of B slots, one driver thread each) on the reference's own benchmark
shape (32x32 Add, PINNED variant; WORKLOAD=config_4: BASELINE config 4): while one group's members run their host code, another group's launches use the GPU.
Byte identity with lmn_prove is checked for every group's first batch.  Usage: small_proof_groups.py [B] [G ...]"""
import json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import luminair_amd
from luminair_amd import backend, synthetic as syn
from luminair_amd.batch import BatchProver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Gs = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4]
solo = luminair_amd.Prover(0, protocol_variant=backend.VARIANT_PINNED)
WORKLOAD = os.environ.get("WORKLOAD", "32x32_add")       # or config_4 (black-scholes MLP shape with its exp2 LUT)
luts = None
if WORKLOAD == "config_4":
    tabs4, luts = syn.config4_black_scholes_shape()
    pies = [[(k, r, len(r)) for k, r in tabs4] for i in range(B)]
else:
    pies = [[(k, r, len(r)) for k, r in syn.config2_graph_faithful(1024, 100 + i)] for i in range(B)]
want = [solo.ctx.prove_tables(p, luts) for p in pies[:4]]
for G in Gs:
    bps = [BatchProver(0, B, protocol_variant=backend.VARIANT_PINNED) for _ in range(G)]
    same = all(bp.prove_batch(pies, luts)[:4] == want for bp in bps)
    for bp in bps:
        bp.prove_batch(pies, luts)
    reps = max(5, 1200 // (B * G)) * 2
    start = threading.Barrier(G + 1)
    bad = []

    pre = os.environ.get("MARSHAL_ONCE", "0") != "0"     # MARSHAL_ONCE=1: the C argument arrays built once (BatchProver.marshal), as a C / Rust caller has them anyway

    def drive(bp):
        m = bp.marshal(pies, luts) if pre else pies
        start.wait()
        for _ in range(reps):
            out = bp.prove_batch(m, luts)
            if out[0] != want[0]:
                bad.append(1)

    ths = [threading.Thread(target=drive, args=(bp,)) for bp in bps]
    [t.start() for t in ths]
    start.wait()
    t0 = time.perf_counter()
    [t.join() for t in ths]
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": WORKLOAD, "batch": B, "groups": G, "bytes_identical_to_lmn_prove": same and not bad,
                      "marshalled_once": pre, "proofs_per_s": round(B * G * reps / dt, 1), "ms_per_batch_per_group": round(1e3 * dt / reps, 3)}), flush=True)
    for bp in bps:
        bp.close()
