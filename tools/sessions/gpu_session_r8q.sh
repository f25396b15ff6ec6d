#!/bin/bash
# round 5, session q: small-proof batches (tools/small_proof_batch.py, B = 1 / 16 / 64) with this tree's batch library
# (ChanStep fetched through a pointer) against the libraries of two earlier commits of the round, alternating
set -u
OUT=gpurun_out/r8q
mkdir -p $OUT
LIB=luminair_amd/csrc/libluminair_hip_batch.so
cp $LIB /tmp/new_batch.so
timeout 900 python -m pytest tests/test_batch.py tests/test_gpu_parity.py -m gpu -x -q -k "batch or kat or switches or config" > $OUT/tests.log 2>&1; grep -n "passed\|failed" $OUT/tests.log | tail -2
for v in new 902cc16 4a194a5 new 902cc16 4a194a5; do
  [ $v = new ] && cp /tmp/new_batch.so $LIB || cp tools/bin/variants/batch_$v.so $LIB
  timeout 600 python tools/small_proof_batch.py 1 16 64 > $OUT/batch_$v.jsonl 2> $OUT/batch_$v.err
  python - <<PY
import json
for l in open("$OUT/batch_$v.jsonl"):
    d=json.loads(l)
    if "workload" in d: print("$v", d["workload"], d["batch"], d["proofs_per_s"], d["ms_per_batch"], "solo", d["solo_lmn_prove_ms"], d["launches_per_batch"], d.get("copy_launches_per_batch"))
PY
done
cp /tmp/new_batch.so $LIB
