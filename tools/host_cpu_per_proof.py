#!/usr/bin/env python3
"""Host CPU time per proof (user + system, all threads) at 8 proofs in flight, for the wait policy in LMN_SYNC_MODE
(0 spin on hipStreamQuery, 3 blocking-sync event): what a rank costs in CPUs when N ranks share a CPU-limited container."""
import json, os, resource, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import luminair_amd
from luminair_amd import synthetic as syn
import bench

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 384
tabs = syn.config2_add_only(1 << 20, 42)
provers = [luminair_amd.Prover(0) for _ in range(n_ctx)]
bufs = [[(k, p.ctx.upload(r), len(r)) for k, r in tabs] for p in provers]
for p, b in zip(provers, bufs):
    p.ctx.prove_tables(b)
bench.throughput(provers, bufs, 32, n_ctx)
r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter()
r = bench.throughput(provers, bufs, steps, 0)
r1 = resource.getrusage(resource.RUSAGE_SELF); t1 = time.perf_counter()
cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
print(json.dumps({"sync_mode": os.environ.get("LMN_SYNC_MODE", "0"), "contexts": n_ctx, "proofs": steps,
                  "proofs_per_s": round(r["value"], 1), "cpu_ms_per_proof": round(1e3 * cpu / steps, 3),
                  "user_ms_per_proof": round(1e3 * (r1.ru_utime - r0.ru_utime) / steps, 3),
                  "sys_ms_per_proof": round(1e3 * (r1.ru_stime - r0.ru_stime) / steps, 3),
                  "cpus_busy": round(cpu / (t1 - t0), 2),
                  "voluntary_ctx_switches_per_proof": round((r1.ru_nvcsw - r0.ru_nvcsw) / steps, 1)}))
