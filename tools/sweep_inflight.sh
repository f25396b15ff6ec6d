#!/bin/bash
# proofs/s of the headline workload vs proofs in flight (one line per setting)
for n in 1 2 3 4 5 6 8; do
  python bench.py --no-cpu-baseline --no-extras --no-anchor --steps 96 --warmup 8 --inflight $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight', $n, 'proofs/s %.1f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'solo %.3f' % d['prove_latency_ms'])"
done
