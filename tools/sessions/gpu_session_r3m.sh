#!/bin/bash
set -u
OUT=gpurun_out/r3m
mkdir -p $OUT
B="python bench.py --gpus 1 --no-cpu-baseline --no-extras --no-anchor"
for n in 4 6 7 8 10 12 16; do
  for rep in 1 2 3; do
    timeout 300 $B --steps 20 --warmup 5 --inflight $n > $OUT/n${n}_k20_w5_$rep.json 2>/dev/null
  done
  timeout 300 $B --steps 192 --warmup 16 --inflight $n > $OUT/n${n}_k192_w16_1.json 2>/dev/null
done
python - <<'PY'
import json,glob,collections
acc=collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/r3m/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        acc.setdefault(f.split("/")[-1].rsplit("_",1)[0],[]).append(round(d["value"],1))
    except Exception as e:
        acc.setdefault(f,[]).append("ERR")
for k,v in acc.items(): print(k,v)
PY
