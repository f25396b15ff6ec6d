#!/bin/bash
# round 5, final session: validation of the final tree: whole GPU suite, smoke, driver command x3, default
# bench with the CPU baseline, small proofs (solo and in batches), host marks, the round's profile set
# (tools/profile_round.sh r5)
set -u
OUT=gpurun_out/r8zz
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; grep -n "passed\|failed" $OUT/gpu_tests.log | tail -3
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
for i in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.err; done
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
for f in ["driver_cmd_1","driver_cmd_2","driver_cmd_3","bench_default"]:
    try:
        d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1])
        r=d["roofline"]; c=d.get("cpu_baseline",{})
        print(f, round(d["value"],1), "solo", round(d["prove_latency_ms"],3), "frac", round(r["frac"],3), "by counter", r.get("frac_by_counter_traffic"), "alu", round(r["alu_ceiling"]["frac"],3), "arch", round(r["alu_ceiling"]["architectural"]["frac"],3), "cpu", c.get("value"), "scalar", c.get("port_scalar",{}).get("value"), "anchor", c.get("reference_shape_32x32_add_ms"), d["errors"])
    except Exception as e: print(f, "ERR", e)
PY
TAG=final timeout 120 python tools/small_latency.py 101 2>> $OUT/err.log | tail -2
timeout 600 python tools/small_proof_batch.py > $OUT/small_proof_batch.jsonl 2> $OUT/small_proof_batch.err; grep 32x32 $OUT/small_proof_batch.jsonl | tail -2 | cut -c1-200
LMN_HOST_PROFILE=1 timeout 300 python tools/host_marks.py 2> $OUT/host_marks.txt > /dev/null; tail -17 $OUT/host_marks.txt
timeout 1500 bash tools/profile_round.sh r5 > $OUT/profile_round.log 2>&1; tail -30 $OUT/profile_round.log
