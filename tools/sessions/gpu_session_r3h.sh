#!/bin/bash
set -u
OUT=gpurun_out/r3h
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_size or equals_oracle or config or kat or soak" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for v in fused nofuse; do
  if [ $v = nofuse ]; then export LMN_NO_FFT_FUSION=1; fi
  for i in 1 2 3; do
    timeout 300 python bench.py --gpus 1 --steps 64 --warmup 8 --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_${v}_$i.json 2>/dev/null
  done
done
unset LMN_NO_FFT_FUSION
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3h/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    r=[x for x in [d["roofline"]]+d["roofline_other"] if x["kernel"]=="k_fft_staged"][0]
    print(f, round(d["value"],1), round(d["prove_latency_ms"],3), "fft_ms", d["stage_ms"]["fft_ms"], "launches", r["launches_per_proof"], "alu frac", round(r["alu_ceiling"]["frac"],3))
PY
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
BENCH="python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight 1"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof -o ks -- $BENCH --steps 32 --warmup 4 > $OUT/bench_under_rocprof_inflight1.json 2> $OUT/prof.log
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_inflight1.csv \;
KT=$(find $OUT/prof -name '*kernel_trace.csv' | head -1)
python tools/timeline.py $KT v > $OUT/kernel_timeline_one_proof.txt
rm -rf $OUT/prof
tail -26 $OUT/kernel_timeline_one_proof.txt
