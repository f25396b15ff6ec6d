#!/bin/bash
# round 4: fixed-shape passes for the three-pass sizes (5-layer 32-word pass, zero-extended top pass): selftests, full GPU
# suite, config latencies
set -u
OUT=gpurun_out/r5j
mkdir -p $OUT
python - <<'PY' 2>&1 | tail -14
import luminair_amd, time
p = luminair_amd.Prover(0)
for log in (13, 16, 17, 18, 20, 21, 22, 23, 24, 25):
    t = time.time(); p.ctx.fft_selftest(log, 2); print("selftest", log, "ok", round(time.time() - t, 2), flush=True)
PY
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/gpu_suite.log 2>&1; grep -E "passed|failed|rror" $OUT/gpu_suite.log | tail -3
timeout 900 python tools/config_latency.py > $OUT/config_latency.jsonl 2> $OUT/config_latency.err; echo "config rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r5j/config_latency.jsonl"):
    d = json.loads(l)
    if "stage_ms" in d: print(d["config"], d["latency_ms"], "fft", d["stage_ms"]["fft_ms"], "merkle", d["stage_ms"]["merkle_ms"])
PY
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('bench', round(d['value'],1), 'solo', round(d['prove_latency_ms'],3), d['roofline_other'][0]['traffic_source'])"
