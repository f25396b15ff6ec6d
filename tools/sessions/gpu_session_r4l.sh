#!/bin/bash
# priority policy variants (tools/build_variants.sh): throughput with 8 proofs in flight and solo latency
set -u
OUT=gpurun_out/r4l
mkdir -p $OUT
for V in ${VARIANTS:-base nofftphase nob2phase nophases}; do
  cp tools/bin/variants/$V.so luminair_amd/csrc/libluminair_hip.so
  for i in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline --no-extras --no-anchor --steps 192 --warmup 24 2>/dev/null | tail -1 > $OUT/${V}_$i.json
  done
  timeout 300 python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight 1 --steps 32 --warmup 4 2>/dev/null | tail -1 > $OUT/${V}_solo.json
  python - <<PY
import json
a=[json.load(open("$OUT/${V}_%d.json"%i))["value"] for i in (1,2)]; e=json.load(open("$OUT/${V}_solo.json"))
print("%-16s 8 ctx %.1f %.1f   solo %.1f proofs/s (%.3f ms)" % ("$V", a[0], a[1], e["value"], e["ms_per_step"]))
PY
done
