#!/bin/bash
# round 4: a quotient column that joins a FRI layer is folded inside that layer's leaf hashing too (MerkleFold::src2):
# parity, then throughput / latency against the two separate fold launches (LMN_NO_JOIN_FUSION=1) on one box
set -u
OUT=gpurun_out/r6i
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/parity.log 2>&1; grep -E "passed|failed|rror" $OUT/parity.log | tail -3
for v in join sep join sep join sep; do
  unset LMN_NO_JOIN_FUSION
  if [ $v = sep ]; then export LMN_NO_JOIN_FUSION=1; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), {k:v for k,v in d["stage_ms"].items() if k in ("merkle_fused_ms","merkle_ms","fri_ms")})
PY
done
unset LMN_NO_JOIN_FUSION
# k_quotients at 8 waves per SIMD (64 VGPRs; <2> with 2 rows per lane) against the product
cp luminair_amd/csrc/libluminair_hip.so /tmp/product.so
for v in base qocc qocc4 base qocc qocc4; do
  cp tools/bin/variants/$v.so luminair_amd/csrc/libluminair_hip.so
  if [ ! -f $OUT/parity_$v.log ]; then timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kat or full_size_2_20 or op_level or ragged or random" > $OUT/parity_$v.log 2>&1; tail -1 $OUT/parity_$v.log; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), {k:v for k,v in d["stage_ms"].items() if k in ("quotients_ms","oods_ms","composition_ms")})
PY
done
cp /tmp/product.so luminair_amd/csrc/libluminair_hip.so
