#!/bin/bash
# round 5, session f: Context::prove split into phase functions (no behaviour change intended): whole GPU suite, smoke,
# driver command x2
set -u
OUT=gpurun_out/r8f
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; grep -n "passed\|failed" $OUT/gpu_tests.log | tail -3
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.err; python -c "
import json; d=json.loads(open('$OUT/driver_cmd_$i.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['prove_latency_ms'],3), d['errors'])"; done
