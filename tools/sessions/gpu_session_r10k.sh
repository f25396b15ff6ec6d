#!/bin/bash
# round 6, session k: do the HBM-bound launches of other proofs find room next to the issue-bound Merkle launches?
# k_merkle_fused: 87 VGPRs, 24 KB LDS, 256 lanes -> 5 workgroups per CU by registers, leaving 72 VGPRs per SIMD - less than a
# transform wave needs (77 - 78).  LMN_MERKLE_LDS_PAD caps the Merkle workgroups per CU through unused dynamic LDS
# (8704 -> 4, 16640 -> 3, 30720 -> 2), LMN_FFT_LDS_PAD the transforms'.  Throughput / solo latency, alternating.
set -u
OUT=gpurun_out/r10k
mkdir -p $OUT
run() {
  LMN_MERKLE_LDS_PAD=$1 LMN_FFT_LDS_PAD=$2 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$1_$2_$3.json 2> $OUT/bench.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$1_$2_$3.json").read().strip().splitlines()[-1])
st=d["stage_ms"]
print("merkle_pad $1 fft_pad $2", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "merkle_fused", st.get("merkle_fused_ms"), "fft", st.get("fft_ms"))
PY
}
for rep in 1 2 3; do
  for m in 0 8704 16640 30720; do run $m 0 $rep; done
  run 0 24576 $rep
  run 8704 24576 $rep
  run 16640 24576 $rep
done
