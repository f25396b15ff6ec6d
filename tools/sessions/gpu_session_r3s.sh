#!/bin/bash
set -u
OUT=gpurun_out/r3s
mkdir -p $OUT
for i in 1 2 3 4 5 6; do
  s=$(date +%s.%N)
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.err; rc=$?
  e=$(date +%s.%N); echo "driver rc=$rc wall $(python -c "print(round($e-$s,1))") s"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3s/driver_cmd_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"],1), "lat", round(d["prove_latency_ms"],3), "errors", d["errors"], {k:round(d[k]["value"],1) for k in ("host_rows","config_2b","mul_only","config_3") if k in d and "value" in d[k]})
    except Exception as e:
        print(f, "ERR", e)
PY
