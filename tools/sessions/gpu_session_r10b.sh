#!/bin/bash
# round 6, session b: uniform-base buffer loads in the QM31 / Merkle leaf kernels (ld_ub / ld_col), quotient entry table
# through the scalar cache, Blake2s zero-word half rounds as one asm statement per quarter-round step.
# Parity, then throughput / solo latency alternating: new | prev_r5 (round 5's library) | quot_lds (new tree with the
# quotient table staged in LDS as before); solo kernel durations (rocprofv3 --kernel-trace --stats, one proof in flight)
# of new and prev_r5; marginal cost per kernel family under 8 proofs in flight (ablation build of the new tree).
set -u
OUT=gpurun_out/r10b
mkdir -p $OUT
LIB=luminair_amd/csrc/libluminair_hip.so
cp $LIB /tmp/new.so
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_batch.py -m gpu -x -q > $OUT/parity.log 2>&1; grep -n "passed\|failed" $OUT/parity.log | tail -2
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
for rep in 1 2 3; do
for v in new prev_r5 quot_lds; do
  [ $v = new ] && cp /tmp/new.so $LIB || cp tools/bin/variants/$v.so $LIB
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_${v}_$rep.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_${v}_$rep.json").read().strip().splitlines()[-1])
st=d["stage_ms"]
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "p95", round(d["prove_latency_p95_ms"],3), "merkle_fused", st.get("merkle_fused_ms"), "quot", st.get("quotients_ms"), "comp", st.get("composition_ms"), "oods", st.get("oods_ms"), "logup", st.get("logup_ms"), "fft", st.get("fft_ms"))
PY
done
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in new prev_r5; do
  [ $v = new ] && cp /tmp/new.so $LIB || cp tools/bin/variants/$v.so $LIB
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof_$v -o ks -- python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight 1 --steps 32 --warmup 4 > /dev/null 2> $OUT/prof_$v.log
  find $OUT/prof_$v -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_inflight1_$v.csv \;
  rm -rf $OUT/prof_$v
done
cp tools/bin/variants/ablate.so $LIB
for rep in 1 2 3; do
for m in 0 1 2 4 8 16 32 63; do
  LMN_ABLATE=$m timeout 300 python tools/ablate_throughput.py 8 192 2>> $OUT/ablate.err | tee -a $OUT/ablate.jsonl
done
done
cp /tmp/new.so $LIB
