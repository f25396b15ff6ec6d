#!/bin/bash
# round 6, session h: validation and measurement set of the tree with k_quot_prepare: whole GPU suite, smoke, the driver's
# command x3, default bench with the CPU baseline, small proofs (solo and in batches), host marks, the round's profile set
# (tools/profile_round.sh r6), BASELINE config latencies, marginal costs per kernel family on this tree (ablation build).
set -u
OUT=gpurun_out/r10h
mkdir -p $OUT
LIB=luminair_amd/csrc/libluminair_hip.so
cp $LIB /tmp/new.so
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
        print(f, round(d["value"],1), "solo", round(d["prove_latency_ms"],3), "frac", round(r["frac"],3), "by counter", r.get("frac_by_counter_traffic"), "alu", round(r["alu_ceiling"]["frac"],3), "arch", round(r["alu_ceiling"]["architectural"]["frac"],3), "host_rows", d.get("host_rows_proofs_per_s"), d.get("host_rows_prove_latency_ms"), "cpu", c.get("value"), c.get("cores"), c.get("memory_policy"), "by_threads", {k:v.get("value") for k,v in c.get("by_threads",{}).items()}, d["errors"])
    except Exception as e: print(f, "ERR", e)
PY
TAG=final timeout 120 python tools/small_latency.py 101 2>> $OUT/err.log | tail -2
timeout 600 python tools/small_proof_batch.py > $OUT/small_proof_batch.jsonl 2> $OUT/small_proof_batch.err; grep 32x32 $OUT/small_proof_batch.jsonl | tail -2 | cut -c1-200
LMN_HOST_PROFILE=1 timeout 300 python tools/host_marks.py 2> $OUT/host_marks.txt > /dev/null; tail -16 $OUT/host_marks.txt
timeout 600 python tools/config_latency.py > $OUT/config_latency.jsonl 2>> $OUT/err.log; cut -c1-120 $OUT/config_latency.jsonl
timeout 1500 bash tools/profile_round.sh r6 > $OUT/profile_round.log 2>&1; tail -30 $OUT/profile_round.log
cp tools/bin/variants/ablate.so $LIB
for rep in 1 2 3; do
for m in 0 1 2 4 8 16 32 63; do
  LMN_ABLATE=$m timeout 300 python tools/ablate_throughput.py 8 192 2>> $OUT/ablate.err | tee -a $OUT/ablate.jsonl
done
done
cp /tmp/new.so $LIB
