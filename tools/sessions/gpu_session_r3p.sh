#!/bin/bash
set -u
OUT=gpurun_out/r3p
mkdir -p $OUT
B="python bench.py --gpus 1 --no-cpu-baseline --no-extras --no-anchor"
for v in off on; do
  if [ $v = on ]; then export LMN_STREAM_PRIO_CYCLE=1; else unset LMN_STREAM_PRIO_CYCLE; fi
  for rep in 1 2 3 4 5 6; do
    timeout 300 $B --steps 20 --warmup 5 > $OUT/${v}_k20_$rep.json 2>/dev/null
  done
  timeout 300 $B --steps 192 --warmup 16 > $OUT/${v}_k192_1.json 2>/dev/null
  timeout 300 $B --steps 192 --warmup 16 --inflight 8 > $OUT/${v}_k192n8_1.json 2>/dev/null
done
python - <<'PY'
import json,glob,collections
acc=collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/r3p/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        acc.setdefault(f.split("/")[-1].rsplit("_",1)[0],[]).append((round(d["value"],1), round(d["prove_latency_ms"],2)))
    except Exception as e:
        acc.setdefault(f,[]).append("ERR")
for k,v in acc.items(): print(k,v)
PY
