#!/bin/bash
# round 6, session i: batch library after the per-member table region (TSan finding in the emulated batch build):
# batch + parity GPU tests; k_transpose_pad at wave priority 3 (tools/bin/variants/tprio3.so, -DLMN_TRANSPOSE_PRIO=3)
# against the unchanged tree, alternating.
set -u
OUT=gpurun_out/r10i
mkdir -p $OUT
LIB=luminair_amd/csrc/libluminair_hip.so
cp $LIB /tmp/new.so
timeout 1500 python -m pytest tests/test_batch.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/parity.log 2>&1; grep -n "passed\|failed" $OUT/parity.log | tail -2
for rep in 1 2 3 4; do
for v in new tprio3; do
  [ $v = new ] && cp /tmp/new.so $LIB || cp tools/bin/variants/$v.so $LIB
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_${v}_$rep.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_${v}_$rep.json").read().strip().splitlines()[-1])
st=d["stage_ms"]
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "p95", round(d["prove_latency_p95_ms"],3), "transpose", st.get("transpose_ms"))
PY
done
done
cp /tmp/new.so $LIB
