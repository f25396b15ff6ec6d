#!/bin/bash
# round 4: QM31 issue-phase experiment (tools/bin/variants/qphase.so = -DLMN_QM31_PHASES) against the product on one box:
# parity subset, then throughput / solo latency / stage times, alternating
set -u
OUT=gpurun_out/r5f
mkdir -p $OUT
cp luminair_amd/csrc/libluminair_hip.so /tmp/product.so
for v in product qphase product qphase; do
  if [ $v = qphase ]; then cp tools/bin/variants/qphase.so luminair_amd/csrc/libluminair_hip.so; else cp /tmp/product.so luminair_amd/csrc/libluminair_hip.so; fi
  if [ ! -f $OUT/parity_$v.log ]; then timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kat or full_size_2_20 or op_level or ragged or random" > $OUT/parity_$v.log 2>&1; tail -1 $OUT/parity_$v.log; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), {k:v for k,v in d["stage_ms"].items() if k in ("quotients_ms","oods_ms","composition_ms","logup_ms","fft_ms","merkle_fused_ms","fri_ms")})
PY
done
cp /tmp/product.so luminair_amd/csrc/libluminair_hip.so
