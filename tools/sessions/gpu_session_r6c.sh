#!/bin/bash
# round 4: remaining ablation masks (transforms, OODS evaluation), k_quotients<1> with 8 rows per lane, proofs in flight 8 / 12 / 16
set -u
OUT=gpurun_out/r6c
mkdir -p $OUT
cp luminair_amd/csrc/libluminair_hip.so /tmp/product.so
cp tools/bin/variants/ablate.so luminair_amd/csrc/libluminair_hip.so
for m in 0 2 16 0 2 16; do
  LMN_ABLATE=$m timeout 300 python tools/ablate_throughput.py 8 192 2>> $OUT/ablate.err | tee -a $OUT/ablate.jsonl
done
for v in base q8 base q8; do
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
for n in 8 12 16 8 12 16; do
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight $n --steps 24 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight', $n, 'proofs/s %.1f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'short %.1f' % d['short_region']['value'])" | tee -a $OUT/inflight.txt
done
for n in 8 12 16; do
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight $n --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver-cmd inflight', $n, 'proofs/s %.1f' % d['value'], 'ms/step %.3f' % d['ms_per_step'])" | tee -a $OUT/inflight.txt
done
