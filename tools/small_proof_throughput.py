#!/usr/bin/env python3
"""Throughput of the reference's own benchmark shape (32x32 Add: 2^10 Add rows + 2^11 Inputs rows, PINNED variant,
docs/snippets/benchmark-component.mdx:172) with N contexts in flight: small proofs are latency-bound (0.86 ms alone), so
concurrency is the lever.  Usage: small_proof_throughput.py [N ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import luminair_amd
from luminair_amd import backend, synthetic as syn
import bench
tabs = syn.config2_graph_faithful(1024, 42)
for n in [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16, 32]:
    ps = [luminair_amd.Prover(0, protocol_variant=backend.VARIANT_PINNED) for _ in range(n)]
    for dev in (False, True):
        bs = [[(k, (q.ctx.upload(r) if dev else r), len(r)) for k, r in tabs] for q in ps]
        res = bench.throughput(ps, bs, 200 * n, 4 * n)
        print(json.dumps({"contexts": n, "rows": "device" if dev else "host", "proofs_per_s": round(res["value"], 1),
                          "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}), flush=True)
    for q in ps:
        q.ctx.close()
