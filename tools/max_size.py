#!/usr/bin/env python3
"""The largest table lmn_prove accepts with PcsConfig::default(): 2^25 rows (LDE 2^26, composition LDE 2^27, 2^27-leaf
trees).  Proves a 2^LOG-row Add table twice (device-resident rows), checks determinism and runs the product verifier."""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import luminair_amd
from luminair_amd import backend, synthetic as syn

log = int(sys.argv[1]) if len(sys.argv) > 1 else 25
t0 = time.perf_counter()
tabs = syn.config2_add_only(1 << log, 11)
t_gen = time.perf_counter() - t0
p = luminair_amd.Prover(0)
bufs = [(k, p.ctx.upload(r), len(r)) for k, r in tabs]
t0 = time.perf_counter(); a = p.ctx.prove_tables(bufs); t_first = time.perf_counter() - t0
t0 = time.perf_counter(); b = p.ctx.prove_tables(bufs); t_second = time.perf_counter() - t0
for _, bb, _ in bufs:
    bb.free()
p.ctx.close()
t0 = time.perf_counter(); backend.default_library().verify(a, backend.VARIANT_KAT); t_ver = time.perf_counter() - t0
print(json.dumps({"log_rows": log, "proof_bytes": len(a), "deterministic": a == b, "sha256": hashlib.sha256(a).hexdigest(),
                  "first_proof_s": round(t_first, 3), "second_proof_ms": round(1e3 * t_second, 2),
                  "rows_per_s_M": round((1 << log) / t_second / 1e6, 1), "verify_ms": round(1e3 * t_ver, 2),
                  "synthetic_rows_s": round(t_gen, 1), "verified": True}))
