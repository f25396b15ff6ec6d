#!/bin/bash
# round 6, session d of the last day: the wait policy with the direct "GPU is shared" signal (proofs in flight in this process
# > 1 -> sleep after 100 us) against the policy before it (tools/bin/variants/noshared.so, -DLMN_NO_SHARED_SIGNAL), alternating:
# throughput, host CPU per proof, solo latency at 24 and 8 proofs in flight; then the GPU suite on the new library.
set -u
OUT=gpurun_out/r11d
mkdir -p $OUT
LIB=luminair_amd/csrc/libluminair_hip.so
cp $LIB /tmp/new.so
for rep in 1 2 3 4; do
for v in new noshared; do
  [ $v = new ] && cp /tmp/new.so $LIB || cp tools/bin/variants/$v.so $LIB
  for inf in 24 8; do
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight $inf > $OUT/bench_${v}_${inf}_$rep.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_${v}_${inf}_$rep.json").read().strip().splitlines()[-1])
print("$v inflight $inf", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "p95", round(d["prove_latency_p95_ms"],3), "host_cpu", d["host_cpu_ms_per_proof"])
PY
  done
done
done
cp /tmp/new.so $LIB
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; grep -n "passed\|failed" $OUT/gpu_tests.log | tail -2
timeout 300 python tools/host_cpu_per_proof.py 2>/dev/null | tail -3
