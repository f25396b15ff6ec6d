#!/bin/bash
# round 5, session a: baseline of the tree as round 4 left it on today's box: whole GPU suite, driver command x3, default
# bench, host-side marks of solo proofs (LMN_HOST_PROFILE=1) next to a one-proof kernel timeline
set -u
OUT=gpurun_out/r8a
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
for i in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.err; done
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
for f in ["driver_cmd_1","driver_cmd_2","driver_cmd_3","bench_default"]:
    try:
        d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1])
        print(f, round(d["value"],1), "solo", round(d["prove_latency_ms"],3), "frac", round(d["roofline"]["frac"],3), "cpu", d.get("cpu_baseline",{}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
LMN_HOST_PROFILE=1 timeout 300 python tools/host_marks.py 2> $OUT/host_marks.txt > /dev/null
grep -c "\[host\]" $OUT/host_marks.txt
lscpu | head -25 > $OUT/lscpu.txt; nproc >> $OUT/lscpu.txt
