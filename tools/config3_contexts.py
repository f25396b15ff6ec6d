#!/usr/bin/env python3
"""proofs/s of BASELINE config 3 (Add 2^21 + Mul 2^20 + Recip 2^20 rows) for the context counts given as arguments.
Measured on MI355X (gpurun_out/r11b): 4 / 6 / 8 / 12 contexts = 258 / 253 / 260 / 261 proofs/s - flat, unlike the 2^20-row headline."""
import sys, json
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import luminair_amd, bench
from luminair_amd import synthetic as syn, backend as _bk
tabs = syn.config3_mixed(21, 20, 20)
for n in [int(a) for a in sys.argv[1:]]:
    ps = [luminair_amd.Prover(0, protocol_variant=_bk.VARIANT_KAT) for _ in range(n)]
    bs = [[(k, q.ctx.upload(r), len(r)) for k, r in tabs] for q in ps]
    for q, bb in zip(ps, bs):
        q.ctx.prove_tables(bb)
    r = bench.throughput(ps, bs, 48, n)
    print(json.dumps({"workload": "config_3", "contexts": n, "proofs_per_s": round(r["value"], 1)}), flush=True)
    for bb in bs:
        for _, b_, _ in bb:
            b_.free()
    for q in ps:
        q.ctx.close()
