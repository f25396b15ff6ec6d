#!/bin/bash
# round 4: Blake2s bookkeeping in k_merkle_fused (literal first half round, ping-pong message buffers, hoisted zero words):
# parity subset + throughput, against the previous build (tools/bin/variants/prev.so) on the same box
set -u
OUT=gpurun_out/r5g
mkdir -p $OUT
cp luminair_amd/csrc/libluminair_hip.so /tmp/new.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kat or full or op_level or sizes or ragged or config or random" > $OUT/parity.log 2>&1; tail -1 $OUT/parity.log
for v in new prev new prev; do
  if [ $v = prev ]; then cp tools/bin/variants/prev.so luminair_amd/csrc/libluminair_hip.so; else cp /tmp/new.so luminair_amd/csrc/libluminair_hip.so; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), {k:v for k,v in d["stage_ms"].items() if k in ("fft_ms","merkle_fused_ms","merkle_ms","fri_ms")}, "alu frac", round(d["roofline"]["alu_ceiling"]["frac"],3))
PY
done
cp /tmp/new.so luminair_amd/csrc/libluminair_hip.so
