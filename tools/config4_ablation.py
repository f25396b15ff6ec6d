#!/usr/bin/env python3
"""Solo latency of BASELINE config 4 (many small components) - used with the ablation switches
LMN_LOGUP_SCAN_V1 / LMN_NO_FFT_FUSION / LMN_NO_FOLD_FUSION to see what each round-3 kernel costs at small sizes."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import luminair_amd
from luminair_amd import backend, synthetic as syn
t, l = syn.config4_black_scholes_shape()
p = luminair_amd.Prover(0, protocol_variant=backend.VARIANT_PINNED)
bufs = [(k, p.ctx.upload(r), len(r)) for k, r in t]
p.ctx.prove_tables(bufs, l)
ts = []
for _ in range(15):
    t0 = time.perf_counter(); p.ctx.prove_tables(bufs, l); ts.append(1e3 * (time.perf_counter() - t0))
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("LMN_")}, "latency_ms": round(sorted(ts)[7], 3),
                  "sizes": [(k, len(r)) for k, r in t]}))
