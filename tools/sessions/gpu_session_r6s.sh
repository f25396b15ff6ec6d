#!/bin/bash
# round 4: host CPU time per proof by wait policy (what N ranks need from a CPU-limited container)
set -u
OUT=gpurun_out/r6s
mkdir -p $OUT
for m in 0 3 1 2; do LMN_SYNC_MODE=$m timeout 300 python tools/host_cpu_per_proof.py 8 384 2>> $OUT/err.log | tee -a $OUT/host_cpu.jsonl; done
for m in 0 3; do LMN_SYNC_MODE=$m taskset -c 0-1 timeout 300 python tools/host_cpu_per_proof.py 8 384 2>> $OUT/err.log | sed 's/^{/{"taskset": "2 cpus", /' | tee -a $OUT/host_cpu.jsonl; done
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
