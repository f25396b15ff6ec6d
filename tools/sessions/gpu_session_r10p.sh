#!/bin/bash
# round 6, session p: validation of the FINAL tree with bench.py's new default of 24 proofs in flight: whole GPU suite, smoke,
# the driver's command x3 at the default and at --inflight 8, default bench with the CPU baseline, hardware-queue count at
# 24 / 48 in flight (is "multiples of 12" three priority levels x four queues?).
set -u
OUT=gpurun_out/r10p
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; grep -n "passed\|failed" $OUT/gpu_tests.log | tail -3
timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
for i in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.err
  timeout 600 python bench.py --steps 20 --warmup 5 --inflight 8 --no-cpu-baseline --no-extras --no-anchor > $OUT/driver_cmd_inflight8_$i.json 2> $OUT/driver_cmd8_$i.err
done
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
for f in ["driver_cmd_1","driver_cmd_inflight8_1","driver_cmd_2","driver_cmd_inflight8_2","driver_cmd_3","driver_cmd_inflight8_3","bench_default"]:
    try:
        d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1])
        r=d["roofline"]; c=d.get("cpu_baseline") or {}
        print(f, round(d["value"],1), "ms_per_step", round(d["ms_per_step"],2), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "frac", round(r["frac"],3), "alu", round(r["alu_ceiling"]["frac"],3), "host_rows", d.get("host_rows_proofs_per_s"), "pinned", d.get("host_rows_pinned_proofs_per_s"), "cpu", c.get("value"), c.get("cores"), "host_cpu", d.get("host_cpu_ms_per_proof"), d["errors"])
    except Exception as e: print(f, "ERR", e)
PY
for rep in 1 2; do
for q in 4 8 16; do
for inf in 24 48; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight $inf > $OUT/bench_q${q}_$inf.json 2> $OUT/bench.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_q${q}_$inf.json").read().strip().splitlines()[-1])
print("GPU_MAX_HW_QUEUES $q inflight $inf", round(d["value"],1))
PY
done
done
done
