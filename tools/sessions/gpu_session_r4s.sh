#!/bin/bash
# transforms: all twiddles of a register stage loaded before the stage's data
set -u
OUT=gpurun_out/r4s
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kat or full or fft or op_level or sizes or ragged or selftest" > $OUT/pytest.log 2>&1; grep -E "passed|failed|rror" $OUT/pytest.log | tail -3
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$i.json 2> $OUT/bench_$i.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_$i.json").read().strip().splitlines()[-1])
print(round(d["value"],1), "solo", round(d["prove_latency_ms"],3), "fft launch us", round(d["roofline_other"][0]["avg_launch_ms"]*1e3,1), "merkle", round(d["roofline"]["avg_launch_ms"]*1e3,1))
PY
done
