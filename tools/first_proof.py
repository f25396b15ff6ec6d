#!/usr/bin/env python3
"""Time to the first proof: context creation + the first lmn_prove (twiddle tables, arena) for the first and for further
contexts of a process, by table size."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import luminair_amd
from luminair_amd import synthetic as syn
log = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tabs = syn.config2_add_only(1 << log, 5)
out = {"log_rows": log}
for name in ("first_context", "second_context", "third_context"):
    t0 = time.perf_counter()
    p = luminair_amd.Prover(0)
    t1 = time.perf_counter()
    bufs = [(k, p.ctx.upload(r), len(r)) for k, r in tabs]
    t2 = time.perf_counter()
    p.ctx.prove_tables(bufs)
    t3 = time.perf_counter()
    p.ctx.prove_tables(bufs)
    t4 = time.perf_counter()
    out[name] = {"create_ms": round(1e3 * (t1 - t0), 1), "upload_ms": round(1e3 * (t2 - t1), 1),
                 "first_proof_ms": round(1e3 * (t3 - t2), 1), "second_proof_ms": round(1e3 * (t4 - t3), 2)}
    globals()["keep_" + name] = (p, bufs)   # keep the contexts alive: later ones share the device's twiddle set
print(json.dumps(out))
