#!/bin/bash
# a kernel change: parity subset, then throughput / solo latency / stage times
set -u
OUT=gpurun_out/r4u
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${KEXPR:-kat or full or op_level or sizes or ragged or config or random}" > $OUT/pytest.log 2>&1; grep -E "passed|failed|rror" $OUT/pytest.log | tail -3
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$i.json 2> $OUT/bench_$i.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_$i.json").read().strip().splitlines()[-1])
print(round(d["value"],1), "solo", round(d["prove_latency_ms"],3), {k:v for k,v in d["stage_ms"].items() if k in ("quotients_ms","oods_ms","composition_ms","logup_ms","fft_ms","merkle_fused_ms","transpose_ms")})
PY
done
