#!/bin/bash
# round 4: host-wait policy under CPU scarcity (what 8 ranks x 8 contexts meet inside a 16-CPU container): the same bench
# with all CPUs and pinned to 2 CPUs, LMN_SYNC_MODE 0 (spin on hipStreamQuery), 1 (hipStreamSynchronize), 2 (spin, then block)
set -u
OUT=gpurun_out/r5i
mkdir -p $OUT
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for cpus in all 0-1; do
for mode in ${MODES:-0 3}; do
  if [ $cpus = all ]; then pre=""; else pre="taskset -c $cpus"; fi
  LMN_SYNC_MODE=$mode $pre timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor --steps 12 --warmup 2 > $OUT/bench_${cpus}_$mode.json 2> $OUT/bench_${cpus}_$mode.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_${cpus}_$mode.json").read().strip().splitlines()[-1])
print("cpus $cpus sync_mode $mode:", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3))
PY
done
done
