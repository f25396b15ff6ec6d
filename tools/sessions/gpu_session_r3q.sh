#!/bin/bash
set -u
OUT=gpurun_out/r3q
mkdir -p $OUT
python - <<'PY'
import torch
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,"priority_range") else None)
PY
B="python bench.py --gpus 1 --no-cpu-baseline --no-extras --no-anchor"
for n in 3 4 5 6 8 10; do
  for rep in 1 2 3 4; do
    timeout 300 $B --steps 20 --warmup 5 --inflight $n > $OUT/n${n}_k20_$rep.json 2>/dev/null
  done
done
python - <<'PY'
import json,glob,collections
acc=collections.OrderedDict()
for f in sorted(glob.glob("gpurun_out/r3q/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        acc.setdefault(f.split("/")[-1].rsplit("_",1)[0],[]).append(round(d["value"],1))
    except Exception as e:
        acc.setdefault(f,[]).append("ERR")
for k,v in acc.items(): print(k,v)
PY
