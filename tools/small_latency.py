#!/usr/bin/env python3
"""Solo latency (median of N) of the small shapes: the reference's 32x32 Add benchmark shape (host rows) and BASELINE config 4."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import luminair_amd
from luminair_amd import backend, synthetic as syn

def med(ctx, tabs, luts, n):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); ctx.prove_tables(tabs, luts); ts.append(1e3 * (time.perf_counter() - t0))
    return round(sorted(ts)[n // 2], 4)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 101
p = luminair_amd.Prover(0, protocol_variant=backend.VARIANT_PINNED)
a = [(k, r, len(r)) for k, r in syn.config2_graph_faithful(1024, 42)]
t4, l4 = syn.config4_black_scholes_shape()
b = [(k, p.ctx.upload(r), len(r)) for k, r in t4]
for tabs, luts in ((a, None), (b, l4)):
    for _ in range(5):
        p.ctx.prove_tables(tabs, luts)
print(json.dumps({"tag": os.environ.get("TAG", ""), "shape_32x32_ms": med(p.ctx, a, None, n), "config4_ms": med(p.ctx, b, l4, n)}))
