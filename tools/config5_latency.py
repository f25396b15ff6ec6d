#!/usr/bin/env python3
"""Solo latency + transform time of BASELINE config 5 (2^24 rows) and of a single 2^23-row Add table (3-pass sizes)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import luminair_amd
from luminair_amd import synthetic as syn
for name, tabs in (("config5", syn.config5_linear_layers()), ("add 2^23", syn.config2_add_only(1 << 23, 3)), ("add 2^22", syn.config2_add_only(1 << 22, 3))):
    p = luminair_amd.Prover(0)
    bufs = [(k, p.ctx.upload(r), len(r)) for k, r in tabs]
    p.ctx.prove_tables(bufs)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); p.ctx.prove_tables(bufs); ts.append(1e3 * (time.perf_counter() - t0))
    p.ctx.set_profiling(True); p.ctx.prove_tables(bufs); p.ctx.set_profiling(False)
    tm = p.timings()
    print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("LMN_")}, "workload": name,
                      "latency_ms": round(sorted(ts)[1], 2), "fft_ms": round(tm["fft_ms"], 2), "fft_launches": tm["fft_launches"]}))
    for _, b, _ in bufs:
        b.free()
    p.ctx.close()
