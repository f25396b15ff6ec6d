#!/bin/bash
# round 4: traffic experiment - level `sub` (1/8 of a big tree's nodes) not written either (mask 256, garbage proofs)
set -u
OUT=gpurun_out/r6p
mkdir -p $OUT
cp luminair_amd/csrc/libluminair_hip.so /tmp/product.so
cp tools/bin/variants/ablate.so luminair_amd/csrc/libluminair_hip.so
for m in 0 256 0 256 0 256; do
  LMN_ABLATE=$m timeout 300 python tools/ablate_throughput.py 8 192 2>> $OUT/ablate.err | tee -a $OUT/ablate.jsonl
done
cp /tmp/product.so luminair_amd/csrc/libluminair_hip.so
