#!/bin/bash
# round 6, session y: several lock-step batch groups at once (BatchPool): GPU tests of the batch library, groups x slots on the
# reference's benchmark shape and on BASELINE config 4, the default bench line with the new small_proofs.concurrent_groups.
set -u
OUT=gpurun_out/r10y
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_batch.py -m gpu -x -q > $OUT/batch_tests.log 2>&1; grep -n "passed\|failed" $OUT/batch_tests.log | tail -2
for rep in 1 2; do
  timeout 300 python tools/small_proof_groups.py 64 1 2 3 4 2>> $OUT/err.log | tee -a $OUT/groups.jsonl
  timeout 300 python tools/small_proof_groups.py 32 2 3 4 6 2>> $OUT/err.log | tee -a $OUT/groups.jsonl
  WORKLOAD=config_4 timeout 600 python tools/small_proof_groups.py 16 1 2 3 4 2>> $OUT/err.log | tee -a $OUT/groups.jsonl
done
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print(round(d["value"],1), d["errors"], json.dumps(d.get("small_proofs"))[:900])
PY
tail -3 $OUT/err.log
