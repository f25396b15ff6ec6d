#!/bin/bash
# round 4: host CPU per proof with minimal polling (LMN_SPIN_US=0: sleep between polls from the start), 1 and 8 contexts
set -u
OUT=gpurun_out/r7b
mkdir -p $OUT
for n in 1 8; do for sp in 0 1200; do LMN_SPIN_US=$sp timeout 120 python tools/host_cpu_per_proof.py $n 256 2>> $OUT/err.log | sed "s/^{/{\"spin_us\": $sp, /" | tee -a $OUT/host_cpu.jsonl; done; done
