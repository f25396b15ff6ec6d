#!/bin/bash
# round 6, session u: launch / wait knobs tuned at 8 proofs in flight, re-measured at 24 (one variable at a time, alternating).
set -u
OUT=gpurun_out/r10u
mkdir -p $OUT
run() {   # label, env assignments...
  local label=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_${label}_$REP.json 2> $OUT/bench.err || tail -3 $OUT/bench.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_${label}_$REP.json").read().strip().splitlines()[-1])
print("$label", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "host_cpu", d["host_cpu_ms_per_proof"])
PY
}
for REP in 1 2 3; do
  run default LMN_NOP=1
  run fft_cpb1 LMN_FFT_CPB=1
  run fft_cpb2 LMN_FFT_CPB=2
  run fft_cpb3 LMN_FFT_CPB=3
  run merkle_sub2 LMN_MERKLE_SUB=2
  run spin_poll LMN_SPIN_US=-1
  run spin_1200 LMN_SPIN_US=1200
  run spin_0 LMN_SPIN_US=0
  run sync_mode2 LMN_SYNC_MODE=2
  run dev_kernarg HIP_FORCE_DEV_KERNARG=1
done
