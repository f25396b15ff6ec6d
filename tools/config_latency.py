#!/usr/bin/env python3
"""Solo prove latency + stage breakdown for the BASELINE configs (device-resident rows)."""
import json, sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import luminair_amd
from luminair_amd import backend, synthetic as syn

def run(name, tabs, luts=None, variant=backend.VARIANT_KAT, reps=5):
    p = luminair_amd.Prover(0, protocol_variant=variant)
    bufs = [(k, p.ctx.upload(r), len(r)) for k, r in tabs]
    p.ctx.prove_tables(bufs, luts)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); p.ctx.prove_tables(bufs, luts); ts.append(1e3 * (time.perf_counter() - t0))
    p.ctx.set_profiling(True); p.ctx.prove_tables(bufs, luts); p.ctx.set_profiling(False)
    tm = p.timings()
    rows = sum(len(r) for _, r in tabs)
    print(json.dumps({"config": name, "rows": rows, "latency_ms": round(sorted(ts)[len(ts) // 2], 3),
                      "rows_per_s": round(rows / (1e-3 * sorted(ts)[len(ts) // 2])),
                      "stage_ms": {k: round(v, 3) for k, v in tm.items() if k.endswith("_ms")}}))
    for _, b, _ in bufs:
        b.free()

run("2a add 2^20", syn.config2_add_only(1 << 20, 42))
run("2b add 2^20 + inputs 2^21", syn.config2_graph_faithful(1 << 20, 42), variant=backend.VARIANT_PINNED)
run("3 add 2^21 + mul 2^20 + recip 2^20", syn.config3_mixed())
t, l = syn.config4_black_scholes_shape()
run("4 black-scholes shape", t, l, variant=backend.VARIANT_PINNED)
run("5 256 linear layers 2^24 rows", syn.config5_linear_layers(), reps=3)

# BASELINE config 4 end to end on the device: DeviceGraph gen_trace + prove + verify
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "tests"))
from level2_checks import device_mlp
lib = backend.default_library()
cfg = lib.default_config(); cfg.protocol_variant = backend.VARIANT_PINNED
ctx = backend.Context(0, cfg, lib)
res = []
for _ in range(4):
    t0 = time.perf_counter()
    g, out, ref = device_mlp(ctx)
    tables, luts, bufs = g.gen_trace()
    t1 = time.perf_counter()
    proof = ctx.prove_tables(tables, luts)
    t2 = time.perf_counter()
    lib.verify(proof, backend.VARIANT_PINNED)
    t3 = time.perf_counter()
    for b in bufs:
        b.free()
    res.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
res.sort(key=lambda r: r[0] + r[1])
print(json.dumps({"config": "4 on device: gen_trace / prove / verify ms", "ms": [round(v, 3) for v in res[1]]}))
