#!/bin/bash
# round 4: polling wait that backs off to 50 us sleeps after LMN_SPIN_US (default 1200 us; -1 = never, the previous behaviour):
# host CPU per proof and throughput with all CPUs / 2 CPUs / 1 CPU, solo latency, the bench line
set -u
OUT=gpurun_out/r6v
mkdir -p $OUT
for sp in -1 1200 -1 1200; do LMN_SPIN_US=$sp timeout 120 python tools/host_cpu_per_proof.py 8 384 2>> $OUT/err.log | sed "s/^{/{\"spin_us\": $sp, /" | tee -a $OUT/host_cpu.jsonl; done
for c in 0-1 0; do for sp in -1 1200; do LMN_SPIN_US=$sp taskset -c $c timeout 120 python tools/host_cpu_per_proof.py 8 384 2>> $OUT/err.log | sed "s/^{/{\"taskset\": \"$c\", \"spin_us\": $sp, /" | tee -a $OUT/host_cpu.jsonl; done; done
for sp in -1 1200 -1 1200; do LMN_SPIN_US=$sp timeout 200 python bench.py --no-cpu-baseline --no-extras --no-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('spin_us $sp', round(d['value'],1), 'short', round(d['short_region']['value'],1), 'solo', round(d['prove_latency_ms'],3), 'p95', round(d['prove_latency_p95_ms'],3))"; done
tail -3 $OUT/err.log
