#!/bin/bash
# round 4: small-proof solo latency, this round's first library (commit 24f20ed) against the final one, alternating on one box
set -u
OUT=gpurun_out/r6l
mkdir -p $OUT
cp luminair_amd/csrc/libluminair_hip.so /tmp/product.so
for v in new old new old new old; do
  if [ $v = old ]; then cp tools/bin/variants/old_24f20ed.so luminair_amd/csrc/libluminair_hip.so; else cp /tmp/product.so luminair_amd/csrc/libluminair_hip.so; fi
  TAG=$v timeout 300 python tools/small_latency.py 101 2>> $OUT/err.log | tee -a $OUT/small_latency.jsonl
done
cp /tmp/product.so luminair_amd/csrc/libluminair_hip.so
