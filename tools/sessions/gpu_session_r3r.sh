#!/bin/bash
set -u
OUT=gpurun_out/r3r
mkdir -p $OUT
( time timeout 1800 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2; grep real $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2 3 4 5; do
  /usr/bin/time -f "wall %e s" timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.err; echo "driver rc=$? $(tail -1 $OUT/driver_cmd_$i.err)"
done
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "default rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3r/driver_cmd_*.json"))+["gpurun_out/r3r/bench_default.json"]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"],1), "lat", round(d["prove_latency_ms"],3), "errors", d["errors"], {k:round(d[k]["value"],1) for k in ("host_rows","config_2b","mul_only","config_3") if k in d and "value" in d[k]})
    except Exception as e:
        print(f, "ERR", e)
PY
