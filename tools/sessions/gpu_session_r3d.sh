#!/bin/bash
set -u
OUT=gpurun_out/r3d
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "host_rows or serialised or four_contexts" > $OUT/pytest_new.log 2>&1; tail -3 $OUT/pytest_new.log
for i in 1 2 3; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.err; echo "rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3d/driver_cmd_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), round(d["prove_latency_ms"],3), "errors", d["errors"])
        for k in ("host_rows","config_2b","mul_only","config_3"):
            print("   ", k, {a:(round(b,2) if isinstance(b,float) else b) for a,b in d[k].items() if a in ("value","prove_latency_ms","error","proofs_in_flight_per_gpu")})
        print("    cpu", d["cpu_baseline"].get("value"), d["cpu_baseline"].get("cores"), {t:round(v["value"],2) for t,v in d["cpu_baseline"].get("by_threads",{}).items()})
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 0; echo "gpus2 rc=$?"
