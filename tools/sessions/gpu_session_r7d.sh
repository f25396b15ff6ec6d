#!/bin/bash
# round 4: k_fft_fx at 7 / 8 waves per SIMD (72 / 64 VGPRs with 2-8 spilled registers) against the product (77 / 78 VGPRs, 6 waves)
set -u
OUT=gpurun_out/r7d
mkdir -p $OUT
cp luminair_amd/csrc/libluminair_hip.so /tmp/product.so
for v in base fftw8 fftw7 base fftw8 fftw7; do
  cp tools/bin/variants/$v.so luminair_amd/csrc/libluminair_hip.so
  if [ ! -f $OUT/parity_$v.log ]; then timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kat or full_size_2_20 or op_level" > $OUT/parity_$v.log 2>&1; tail -1 $OUT/parity_$v.log; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), {k:v for k,v in d["stage_ms"].items() if k in ("fft_ms",)})
PY
done
cp /tmp/product.so luminair_amd/csrc/libluminair_hip.so
