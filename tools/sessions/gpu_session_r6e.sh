#!/bin/bash
# round 4: traffic experiment - how much throughput the 1.0 GB of bottom-level Merkle hash writes per proof cost under load
# (experiment build -DLMN_ABLATE, mask 128: levels 0..2 of every fused subtree are not written; proofs are garbage)
set -u
OUT=gpurun_out/r6e
mkdir -p $OUT
cp luminair_amd/csrc/libluminair_hip.so /tmp/product.so
cp tools/bin/variants/ablate.so luminair_amd/csrc/libluminair_hip.so
for m in 0 128 0 128 0 128; do
  LMN_ABLATE=$m timeout 300 python tools/ablate_throughput.py 8 192 2>> $OUT/ablate.err | tee -a $OUT/ablate.jsonl
done
cp /tmp/product.so luminair_amd/csrc/libluminair_hip.so
