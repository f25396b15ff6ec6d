#!/bin/bash
# round 6, session d: the AoS -> SoA transpose inside the first pass of the interpolation (k_fft_rows_fx; no k_transpose_pad
# launch, no column-major evaluations in HBM, logup fractions read the table's rows): parity, then throughput / solo latency
# against the same library with LMN_NO_ROWS_FUSION=1 (transpose launch + plain first pass), alternating on one box; solo
# kernel durations of both.
set -u
OUT=gpurun_out/r10d
mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_batch.py -m gpu -x -q > $OUT/parity.log 2>&1; grep -n "passed\|failed" $OUT/parity.log | tail -2
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
for rep in 1 2 3 4; do
for v in fused plain; do
  if [ $v = plain ]; then export LMN_NO_ROWS_FUSION=1; else unset LMN_NO_ROWS_FUSION; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_${v}_$rep.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_${v}_$rep.json").read().strip().splitlines()[-1])
st=d["stage_ms"]
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "p95", round(d["prove_latency_p95_ms"],3), "transpose", st.get("transpose_ms"), "main_commit", st.get("main_commit_ms"), "logup", st.get("logup_ms"), "fft", st.get("fft_ms"))
PY
done
done
unset LMN_NO_ROWS_FUSION
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in fused plain; do
  if [ $v = plain ]; then export LMN_NO_ROWS_FUSION=1; else unset LMN_NO_ROWS_FUSION; fi
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof_$v -o ks -- python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight 1 --steps 32 --warmup 4 > /dev/null 2> $OUT/prof_$v.log
  find $OUT/prof_$v -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_inflight1_$v.csv \;
  rm -rf $OUT/prof_$v
done
unset LMN_NO_ROWS_FUSION
python - <<PY
import csv,re
for v in ("fused","plain"):
    rows=list(csv.DictReader(open("$OUT/kernel_stats_inflight1_%s.csv"%v)))
    np_=[int(r["Calls"]) for r in rows if "k_fft_interp_extend_fx<8>" in r["Name"]][0]//2
    print(v, "proofs", np_)
    for r in rows:
        if any(k in r["Name"] for k in ("k_transpose_pad","k_fft_rows_fx","k_fft_fx<true","k_logup_fracs","k_fft_interp_extend_fx<8>")):
            print("   %-60s %8.1f us/proof"%(re.sub(r"\(.*","",r["Name"])[:60], float(r["TotalDurationNs"])/np_/1e3))
PY
