#!/bin/bash
set -u
OUT=gpurun_out/r3g
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size or equals_oracle or config or kat" > $OUT/pytest_scan.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_scan.log
LMN_HOST_PROFILE=1 timeout 300 python tools/host_marks.py > $OUT/host_marks.txt 2>&1; tail -40 $OUT/host_marks.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
BENCH="python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight 1"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof -o ks -- $BENCH --steps 32 --warmup 4 > $OUT/bench_under_rocprof_inflight1.json 2> $OUT/prof.log
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_inflight1.csv \;
KT=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
python tools/timeline.py $KT v > $OUT/kernel_timeline_one_proof.txt
rm -rf $OUT/prof
tail -30 $OUT/kernel_timeline_one_proof.txt
for v in 0 1; do
  if [ $v = 1 ]; then export LMN_LOGUP_SCAN_V1=1; fi
  timeout 300 python bench.py --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_scanv$v.json 2>/dev/null
done
unset LMN_LOGUP_SCAN_V1
python - <<'PY'
import json
for v in (0,1):
    d=json.loads(open("gpurun_out/r3g/bench_scanv%d.json"%v).read().strip().splitlines()[-1])
    print("scan v1=%d"%v, round(d["value"],1), round(d["prove_latency_ms"],3), d["stage_ms"]["logup_ms"])
PY
