#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-extras --no-anchor --steps 128 --warmup 8 --inflight $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 inflight', $1, 'proofs/s %.1f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'solo %.3f' % d['prove_latency_ms'])"; }
for sub in 3 2 1 0; do for n in 4 6; do LMN_MERKLE_SUB=$sub run $n sub$sub; done; done
