#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-extras --no-anchor --steps 96 --warmup 8 --inflight $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 inflight', $1, 'proofs/s %.1f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'solo %.3f' % d['prove_latency_ms'])"; }
for n in 4 6 8; do GPU_MAX_HW_QUEUES=8 run $n hwq8; done
for n in 4 6 8; do GPU_MAX_HW_QUEUES=16 run $n hwq16; done
for n in 2 4; do GPU_MAX_HW_QUEUES=2 run $n hwq2; done
for n in 4 6; do HIP_FORCE_DEV_KERNARG=1 run $n devkernarg; done
for n in 4 6; do LMN_SYNC_MODE=2 run $n syncmode2; done
