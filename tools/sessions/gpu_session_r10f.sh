#!/bin/bash
# round 6, session f: the transcript step and the tables in front of the FRI quotient kernels on the device (k_quot_prepare):
# the host's first wait of a proof is the one for the FRI results (2 waits per proof instead of 3).  Whole GPU suite, then
# solo latency / throughput against LMN_HOST_QUOT=1 (the wait for the sampled values, as in round 5) alternating on one box;
# kernel timeline + host marks of one solo proof.
set -u
OUT=gpurun_out/r10f
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; grep -n "passed\|failed" $OUT/gpu_tests.log | tail -3
for rep in 1 2 3 4; do
for v in dev host; do
  unset LMN_HOST_QUOT
  [ $v = host ] && export LMN_HOST_QUOT=1
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_${v}_$rep.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_${v}_$rep.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "p95", round(d["prove_latency_p95_ms"],3), "host_cpu_ms", d["host_cpu_ms_per_proof"])
PY
done
done
unset LMN_HOST_QUOT
TAG=dev timeout 120 python tools/small_latency.py 101 2>> $OUT/err.log | tail -1
LMN_HOST_QUOT=1 TAG=host timeout 120 python tools/small_latency.py 101 2>> $OUT/err.log | tail -1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --output-format csv --kernel-trace -d $OUT/kt -o kt -- python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight 1 --steps 8 --warmup 2 > /dev/null 2> $OUT/kt.log
python tools/timeline.py $(find $OUT/kt -name '*kernel_trace.csv' | head -1) v > $OUT/kernel_timeline_one_proof.txt
rm -rf $OUT/kt
LMN_HOST_PROFILE=1 python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight 1 --steps 6 --warmup 2 > /dev/null 2> $OUT/host_marks.txt
grep -n "gap \+[1-9][0-9]\.\|end-to-end\|_gaps\|k_quot_prepare" $OUT/kernel_timeline_one_proof.txt | tail -20
tail -16 $OUT/host_marks.txt
