#!/usr/bin/env python3
"""Solo latency and transform time of the 2^20-row Add proof under the transform launch knobs (LMN_FFT_CPB, LMN_FFT_THREADS)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import luminair_amd
from luminair_amd import synthetic as syn
tabs = syn.config2_add_only(1 << 20, 42)
p = luminair_amd.Prover(0)
bufs = [(k, p.ctx.upload(r), len(r)) for k, r in tabs]
p.ctx.prove_tables(bufs)
ts = []
for _ in range(15):
    t0 = time.perf_counter(); p.ctx.prove_tables(bufs); ts.append(1e3 * (time.perf_counter() - t0))
fs = []
p.ctx.set_profiling(True)
for _ in range(5):
    p.ctx.prove_tables(bufs); fs.append(p.timings()["fft_ms"])
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("LMN_")}, "latency_ms": round(sorted(ts)[7], 3),
                  "fft_ms": round(sorted(fs)[2], 4)}))
