#!/bin/bash
# round 5, session g: device-resident Fiat-Shamir of the commitment phases (k_chan_root_elems / _claims_root_alpha /
# _root_oods; no host wait before the sampled values): parity (batches and sharded ranks included), then solo latency /
# throughput / small-proof latency against the host transcript (LMN_HOST_FS=1), alternating on one box
set -u
OUT=gpurun_out/r8g
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; grep -n "passed\|failed" $OUT/gpu_tests.log | tail -3
for v in dev host dev host dev host; do
  if [ $v = host ]; then export LMN_HOST_FS=1; else unset LMN_HOST_FS; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "p95", round(d["prove_latency_p95_ms"],3), "cpu/proof", d["host_cpu_ms_per_proof"])
PY
  TAG=$v timeout 120 python tools/small_latency.py 101 2>> $OUT/err.log | tail -2
done
unset LMN_HOST_FS
LMN_HOST_PROFILE=1 timeout 300 python tools/host_marks.py 2> $OUT/host_marks_device_fs.txt > /dev/null; tail -16 $OUT/host_marks_device_fs.txt
timeout 600 python tools/small_proof_batch.py > $OUT/small_proof_batch.jsonl 2> $OUT/small_proof_batch.err; grep 32x32 $OUT/small_proof_batch.jsonl | tail -2 | cut -c1-200
