#!/bin/bash
# round 4: 32-word vs 16-word runs in the strided passes of the three-pass transform sizes (config 5, config 3)
set -u
OUT=gpurun_out/r5k
mkdir -p $OUT
python - <<'PY' 2>&1 | tail -6
import luminair_amd, time
p = luminair_amd.Prover(0)
for log in (17, 22, 23, 24, 25):
    t = time.time(); p.ctx.fft_selftest(log, 2); print("selftest", log, "ok", round(time.time() - t, 2), flush=True)
PY
for cb in 5 4 5 4; do
LMN_FFT_CB3=$cb timeout 600 python - <<'PY'
import json, sys, time, os
sys.path.insert(0, ".")
import luminair_amd
from luminair_amd import backend, synthetic as syn
for name, tabs in (("config3", syn.config3_mixed()), ("config5", syn.config5_linear_layers())):
    p = luminair_amd.Prover(0)
    bufs = [(k, p.ctx.upload(r), len(r)) for k, r in tabs]
    p.ctx.prove_tables(bufs)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); p.ctx.prove_tables(bufs); ts.append(1e3 * (time.perf_counter() - t0))
    p.ctx.set_profiling(True); p.ctx.prove_tables(bufs); p.ctx.set_profiling(False)
    tm = p.timings()
    print("cb3", os.environ["LMN_FFT_CB3"], name, round(sorted(ts)[2], 2), "ms  fft", round(tm["fft_ms"], 2), "merkle", round(tm["merkle_ms"], 2), flush=True)
    for _, b, _ in bufs: b.free()
    p.ctx.close()
PY
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3 or config5 or 2_24 or full" > $OUT/parity.log 2>&1; tail -1 $OUT/parity.log
