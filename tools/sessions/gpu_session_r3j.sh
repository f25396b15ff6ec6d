#!/bin/bash
set -u
OUT=gpurun_out/r3k
mkdir -p $OUT
B="python bench.py --gpus 1 --no-cpu-baseline --no-extras --no-anchor"
for q in 0; do
  for n in 4 5 8 10; do
    for rep in 1 2 3; do
      if [ $q = 0 ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
      timeout 300 $B --steps 20 --warmup 5 --inflight $n > $OUT/q${q}_n${n}_s20_$rep.json 2>/dev/null
    done
    timeout 300 $B --steps 192 --warmup 16 --inflight $n > $OUT/q${q}_n${n}_s192_1.json 2>/dev/null; timeout 300 $B --steps 192 --warmup 16 --inflight $((n*2)) > $OUT/q${q}_n$((n*2))_s192_1.json 2>/dev/null
  done
done
python - <<'PY'
import json,glob,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/r3k/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        key=f.split("/")[-1].rsplit("_",1)[0]
        acc[key].append(round(d["value"],1))
    except Exception as e:
        acc[f].append("ERR")
for k,v in acc.items(): print(k, v)
PY
