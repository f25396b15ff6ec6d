#!/bin/bash
# round 4: lock-step batch build (libluminair_hip_batch.so) - byte identity with lmn_prove and throughput on small proofs
set -u
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
OUT=gpurun_out/r5b
mkdir -p $OUT
for mode in dev; do
LMN_BATCH_ARGS=$mode timeout 900 python tools/small_proof_batch.py ${BS:-16 32 48 64} > $OUT/small_proof_batch_$mode.jsonl 2> $OUT/small_proof_batch_$mode.err
echo "args=$mode"; cat $OUT/small_proof_batch_$mode.jsonl; tail -5 $OUT/small_proof_batch_$mode.err
done
