#!/bin/bash
# shader clock and board power while the bench runs (8 proofs in flight, then one)
set -u
OUT=gpurun_out/r4m
mkdir -p $OUT
sample() { for i in $(seq 1 $1); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.5; done; }
sample 3 > $OUT/idle.txt
python bench.py --no-cpu-baseline --no-extras --no-anchor --steps 6000 --warmup 24 > $OUT/bench8.json 2>/dev/null &
BP=$!
sleep 6; sample 12 > $OUT/load8.txt; wait $BP
python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight 1 --steps 3000 --warmup 4 > $OUT/bench1.json 2>/dev/null &
BP=$!
sleep 6; sample 8 > $OUT/load1.txt; wait $BP
echo idle; cat $OUT/idle.txt | cut -c1-300; echo load8; cat $OUT/load8.txt | cut -c1-300; echo load1; cat $OUT/load1.txt | cut -c1-300
python - <<PY
import json
for f in ("bench8","bench1"):
    d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1]); print(f, round(d["value"],1))
PY
