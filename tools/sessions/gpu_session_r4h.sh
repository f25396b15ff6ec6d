#!/bin/bash
# FFT butterflies issued in phases: parity + bench
set -u
OUT=gpurun_out/r4h
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kat or full or fft or op_level or sizes or ragged" 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/driver_$i.json; python - <<PY
import json; d=json.load(open("$OUT/driver_$i.json")); print("driver cmd", round(d["value"],1), d.get("errors"), {k:(round(v["value"],1) if isinstance(v,dict) and "value" in v else None) for k,v in d.items() if k in ("host_rows","config_2b","mul_only","config_3")})
PY
done
timeout 300 python bench.py 2>/dev/null | tail -1 > $OUT/default.json; python - <<PY
import json; d=json.load(open("$OUT/default.json")); print("default", round(d["value"],1), d["ms_per_step"])
PY
