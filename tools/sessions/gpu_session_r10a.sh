#!/bin/bash
# round 6, session a: Blake2s leaf compressions with compile-time-zero message words (b2_compress_fresh_nz: v_add_u32 instead
# of v_add3_u32 for the zero words of 4- / 8- / 12- / 15-column leaves and FRI layers): parity on the GPU, then throughput /
# solo latency against round 5's library (tools/bin/variants/prev_r5.so) alternating on one box, solo Merkle commits, and
# the marginal cost of every kernel family under 8 proofs in flight on the new tree (ablation build).
set -u
OUT=gpurun_out/r10a
mkdir -p $OUT
LIB=luminair_amd/csrc/libluminair_hip.so
cp $LIB /tmp/new.so
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/parity.log 2>&1; grep -n "passed\|failed" $OUT/parity.log | tail -2
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
for v in new prev new prev new prev; do
  [ $v = new ] && cp /tmp/new.so $LIB
  [ $v = prev ] && cp tools/bin/variants/prev_r5.so $LIB
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "p95", round(d["prove_latency_p95_ms"],3), "merkle_avg_launch_ms", d["roofline"].get("avg_launch_ms"), "alu", round(d["roofline"]["alu_ceiling"]["frac"],3))
PY
  timeout 200 python tools/merkle_solo.py 2>&1 | sed "s/^/$v /"
done
cp tools/bin/variants/ablate.so $LIB
for rep in 1 2 3; do
for m in 0 1 2 4 8 16 32 63; do
  LMN_ABLATE=$m timeout 300 python tools/ablate_throughput.py 8 192 2>> $OUT/ablate.err | tee -a $OUT/ablate.jsonl
done
done
cp /tmp/new.so $LIB
