#!/bin/bash
# round 4: leaf level under a level with columns hashed by that level's launch (MerkleFold::below): parity, then throughput
# against the previous storage forms on one box (LMN_MERKLE_BELOW_MIN_LOG=99: separate leaf launch; LMN_MERKLE_FULL=1: whole trees)
set -u
OUT=gpurun_out/r6g
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/parity.log 2>&1; grep -E "passed|failed|rror" $OUT/parity.log | tail -3
for v in below cut full below cut full; do
  unset LMN_MERKLE_FULL LMN_MERKLE_BELOW_MIN_LOG
  if [ $v = full ]; then export LMN_MERKLE_FULL=1; fi
  if [ $v = cut ]; then export LMN_MERKLE_BELOW_MIN_LOG=99; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), {k:v for k,v in d["stage_ms"].items() if k in ("fft_ms","merkle_fused_ms","merkle_ms","fri_ms","decommit_ms")}, "launches", d["roofline"]["launches_per_proof"])
PY
done
unset LMN_MERKLE_FULL LMN_MERKLE_BELOW_MIN_LOG
LMN_HOST_PROFILE=1 python tools/ablate_throughput.py 1 1 2>&1 | grep "decommit:" | tail -1
