#!/bin/bash
# round 6, session e: k_fft_rows_fx with the XCD-aware (tile, column group) order, 8 columns per workgroup (135 KB LDS, one
# workgroup per CU) and 4 (68 KB, two per CU), against the transpose launch + plain first pass (LMN_NO_ROWS_FUSION=1),
# alternating on one box; solo durations from the library's own stage timers.
set -u
OUT=gpurun_out/r10e
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kat or config2 or full or ragged or sizes" > $OUT/parity8.log 2>&1; grep -n "passed\|failed" $OUT/parity8.log | tail -1
LMN_ROWS_FX_COLS=4 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kat or config2 or full or ragged or sizes" > $OUT/parity4.log 2>&1; grep -n "passed\|failed" $OUT/parity4.log | tail -1
for rep in 1 2 3 4; do
for v in fused8 fused4 plain; do
  unset LMN_NO_ROWS_FUSION LMN_ROWS_FX_COLS
  [ $v = plain ] && export LMN_NO_ROWS_FUSION=1
  [ $v = fused4 ] && export LMN_ROWS_FX_COLS=4
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_${v}_$rep.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_${v}_$rep.json").read().strip().splitlines()[-1])
st=d["stage_ms"]
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "transpose", st.get("transpose_ms"), "main_commit", st.get("main_commit_ms"), "logup", st.get("logup_ms"), "fft", st.get("fft_ms"))
PY
done
done
