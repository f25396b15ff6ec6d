#!/bin/bash
set -u
OUT=gpurun_out/r3f
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
for i in 1 2 3 4 5; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-anchor > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.err; echo "rc=$?"
done
for n in 1 2 3 4 5 6 8; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --inflight $n --no-cpu-baseline --no-extras --no-anchor > $OUT/inflight_$n.json 2> /dev/null
done
timeout 400 python bench.py --gpus 1 --no-cpu-baseline --no-extras --no-anchor > $OUT/default_192.json 2> /dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3f/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), round(d["prove_latency_ms"],3), "errors", d["errors"])
    except Exception as e:
        print(f, "ERR", e)
PY
