#!/bin/bash
# round 5, session c: batch-library fixes (twiddle set-up outside lock-step, per-proof mark for the non-canonical-word
# verdict, ordered direct-copy fallback, small solo groups, guarded fiber stacks): batch + parity tests, small-proof throughput
set -u
OUT=gpurun_out/r8c
mkdir -p $OUT
timeout 1800 python -m pytest tests/test_batch.py tests/test_gpu_parity.py tests/test_sharded_prove.py -m gpu -x -q > $OUT/gpu_tests.log 2>&1; grep -n "passed\|failed\|Error" $OUT/gpu_tests.log | tail -5
timeout 600 python tools/small_proof_batch.py > $OUT/small_proof_batch.jsonl 2> $OUT/small_proof_batch.err; tail -4 $OUT/small_proof_batch.jsonl
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/driver_cmd.json 2> $OUT/driver_cmd.err; python -c "
import json; d=json.loads(open('$OUT/driver_cmd.json').read().strip().splitlines()[-1]); print(d['value'], d['prove_latency_ms'])"
