#!/bin/bash
set -u
OUT=gpurun_out/r3l
mkdir -p $OUT
B="python bench.py --gpus 1 --no-cpu-baseline --no-extras --no-anchor"
for n in 8; do
  for k in 16 20 24 32 40 64; do
    for w in 5 8 16; do
      timeout 300 $B --steps $k --warmup $w --inflight $n > $OUT/n${n}_k${k}_w${w}.json 2>/dev/null
    done
  done
done
for n in 6 7 9 12; do
  for k in 20 24; do
    timeout 300 $B --steps $k --warmup 5 --inflight $n > $OUT/n${n}_k${k}_w5.json 2>/dev/null
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3l/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"],1), "ms total", round(d["ms_per_step"]*d["steps"],1))
    except Exception as e:
        print(f, "ERR")
PY
