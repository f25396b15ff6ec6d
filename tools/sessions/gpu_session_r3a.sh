#!/bin/bash
# r3 session A: (1) host_rows repro on the UNFIXED library, (2) the driver's exact bench command 5x
set -u
OUT=gpurun_out/r3a
mkdir -p $OUT
timeout 900 python tools/repro_host_rows.py --rounds 30 --per-round 48 > $OUT/repro_before.jsonl 2> $OUT/repro_before.err
echo "repro rc=$?" >> $OUT/repro_before.jsonl
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3a/driver_cmd_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), round(d["prove_latency_ms"],3), d.get("host_rows",{}).get("error") or d.get("host_rows",{}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
cat $OUT/repro_before.jsonl
