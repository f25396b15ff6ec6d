#!/bin/bash
# round 4: (1) host -> device bandwidth by kind of host memory and NUMA node (tools/microbench_h2d.hip);
# (2) marginal cost of each kernel family under 8 proofs in flight (experiment build -DLMN_ABLATE, tools/ablate_throughput.py)
set -u
OUT=gpurun_out/r6b
mkdir -p $OUT
timeout 300 tools/bin/mb_h2d > $OUT/h2d.txt 2> $OUT/h2d.err; cat $OUT/h2d.txt
cp luminair_amd/csrc/libluminair_hip.so /tmp/product.so
cp tools/bin/variants/ablate.so luminair_amd/csrc/libluminair_hip.so
for m in 0 1 2 4 8 16 32 3 60 63 0 1 2; do
  LMN_ABLATE=$m timeout 300 python tools/ablate_throughput.py 8 192 2>> $OUT/ablate.err | tee -a $OUT/ablate.jsonl
done
cp /tmp/product.so luminair_amd/csrc/libluminair_hip.so
