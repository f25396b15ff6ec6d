#!/usr/bin/env python3
"""proofs/s with 8 proofs in flight (and solo latency) of the headline workload for the LMN_ABLATE mask in the environment
(an experiment build, tools/build_variants.sh ablate "-DLMN_ABLATE"): the change against mask 0 is the MARGINAL cost of
the skipped kernel families under concurrent load.  The proofs of a non-zero mask are garbage and are not checked."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import luminair_amd
from luminair_amd import synthetic as syn
import bench

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 192
tabs = syn.config2_add_only(1 << 20, 42)
provers = [luminair_amd.Prover(0) for _ in range(n_ctx)]
bufs = [[(k, p.ctx.upload(r), len(r)) for k, r in tabs] for p in provers]
for p, b in zip(provers, bufs):
    p.ctx.prove_tables(b)
r = bench.throughput(provers, bufs, steps, n_ctx)
print(json.dumps({"mask": int(os.environ.get("LMN_ABLATE", "0")), "contexts": n_ctx, "proofs": steps,
                  "proofs_per_s": round(r["value"], 1), "ms_per_proof": round(1e3 / r["value"], 4),
                  "solo_latency_ms": round(bench.solo_latency(provers[0].ctx, bufs[0]), 3)}), flush=True)
