#!/bin/bash
# round 5, session x: the fold in front of the FRI tail inside the tail, the tail writes the FRI result block to page-locked memory, the evaluation job table copied by the OODS step (two transfers and one launch less),
# against the previous commit (tools/bin/variants/prev_x.so).  Small proofs too.
# Parity subset, then solo latency / throughput of the new library against the previous
# commit's (tools/bin/variants/prev_x.so), alternating; kernel timeline and host marks of the new one.
set -u
OUT=gpurun_out/r8x
mkdir -p $OUT
LIB=luminair_amd/csrc/libluminair_hip.so
cp $LIB /tmp/new.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_batch.py -m gpu -x -q -k "kat or full or config or random or ragged or sizes or blowups or batch or switches or less_than or lut or flags" > $OUT/parity.log 2>&1; grep -n "passed\|failed" $OUT/parity.log | tail -2
for v in new prev new prev new prev; do
  [ $v = new ] && cp /tmp/new.so $LIB
  [ $v = prev ] && cp tools/bin/variants/prev_x.so $LIB
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "p95", round(d["prove_latency_p95_ms"],3))
PY
done
cp /tmp/new.so $LIB
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --output-format csv --kernel-trace -d $OUT/kt -o kt -- python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight 1 --steps 8 --warmup 2 > /dev/null 2> $OUT/kt.log
python tools/timeline.py $(find $OUT/kt -name '*kernel_trace.csv' | head -1) v > $OUT/kernel_timeline_one_proof.txt
rm -rf $OUT/kt
LMN_HOST_PROFILE=1 python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight 1 --steps 6 --warmup 2 > /dev/null 2> $OUT/host_marks.txt
grep -n "gap \+[1-9][0-9]\.\|end-to-end\|_gaps\|copyBuffer" $OUT/kernel_timeline_one_proof.txt | tail -30
tail -22 $OUT/host_marks.txt
grep -n "k_eval_at_point\|k_logup\|k_scan_blocksums\|end-to-end" $OUT/kernel_timeline_one_proof.txt | head -12
TAG=new timeout 120 python tools/small_latency.py 101 2>> $OUT/err.log | tail -1
cp tools/bin/variants/prev_x.so $LIB; TAG=prev timeout 120 python tools/small_latency.py 101 2>> $OUT/err.log | tail -1; cp /tmp/new.so $LIB
