#!/bin/bash
for v in 0 1 0 1; do
  LMN_FRI_OVERLAP=$v python tools/fft_knobs.py 2>/dev/null
done
LMN_FRI_OVERLAP=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kat or full_size_2_20 or config3 or equals_oracle" 2>&1 | grep -E "passed|failed"
for v in 0 1; do
  LMN_FRI_OVERLAP=$v python bench.py --gpus 1 --no-cpu-baseline --no-extras --no-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap=$v k192', round(d['value'],1), 'lat', round(d['prove_latency_ms'],3))"
done
