#!/bin/bash
# round 4: a wait policy that really sleeps (LMN_SYNC_MODE=4: hipDeviceScheduleBlockingSync + hipStreamSynchronize): host CPU per
# proof and throughput with all CPUs, with 2 CPUs (what one of 8 ranks gets in a 16-CPU container), with 1 CPU
set -u
OUT=gpurun_out/r6t
mkdir -p $OUT
for m in 0 4; do LMN_SYNC_MODE=$m timeout 300 python tools/host_cpu_per_proof.py 8 384 2>> $OUT/err.log | tee -a $OUT/host_cpu.jsonl; done
for c in 0-1 0; do for m in 0 4; do LMN_SYNC_MODE=$m taskset -c $c timeout 300 python tools/host_cpu_per_proof.py 8 384 2>> $OUT/err.log | sed "s/^{/{\"taskset\": \"$c\", /" | tee -a $OUT/host_cpu.jsonl; done; done
for m in 0 4; do LMN_SYNC_MODE=$m timeout 300 python bench.py --no-cpu-baseline --no-extras --no-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mode $m', round(d['value'],1), 'short', round(d['short_region']['value'],1), 'solo', round(d['prove_latency_ms'],3))"; done
tail -3 $OUT/err.log
