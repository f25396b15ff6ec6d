#!/bin/bash
# round 4 (third session): full GPU suite at HEAD, Merkle occupancy experiments (waves_per_eu 6, idle climb waves end early),
# steady-state host-rows throughput
set -u
OUT=gpurun_out/r6a
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
cp luminair_amd/csrc/libluminair_hip.so /tmp/product.so
for v in base mw6 cexit mw6cexit base mw6 cexit mw6cexit; do
  cp tools/bin/variants/$v.so luminair_amd/csrc/libluminair_hip.so
  if [ ! -f $OUT/parity_$v.log ]; then timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kat or full_size_2_20 or op_level or ragged or random" > $OUT/parity_$v.log 2>&1; tail -1 $OUT/parity_$v.log; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), {k:v for k,v in d["stage_ms"].items() if k in ("fft_ms","merkle_fused_ms","merkle_ms","fri_ms")})
PY
done
cp /tmp/product.so luminair_amd/csrc/libluminair_hip.so
timeout 900 python tools/host_rows_steady.py --steps 48,192,384 --contexts 8,12 > $OUT/host_rows_steady.jsonl 2> $OUT/host_rows_steady.err; cat $OUT/host_rows_steady.jsonl
