#!/bin/bash
# round 6, session v: marginal cost per kernel family under 24 proofs in flight (the new default of bench.py), ablation build of
# the final tree, three alternations of masks 0 / 1 / 2 / 4 / 8 / 16 / 32 / 64 / 63.
set -u
OUT=gpurun_out/r10v
mkdir -p $OUT
LIB=luminair_amd/csrc/libluminair_hip.so
cp $LIB /tmp/new.so
cp tools/bin/variants/ablate.so $LIB
for rep in 1 2 3; do
for m in 0 1 2 4 8 16 32 64 63; do
  LMN_ABLATE=$m timeout 300 python tools/ablate_throughput.py 24 576 2>> $OUT/ablate.err | tee -a $OUT/ablate.jsonl
done
done
cp /tmp/new.so $LIB
