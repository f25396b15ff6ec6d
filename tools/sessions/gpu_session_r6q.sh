#!/bin/bash
# round 4: FRI tail without global round trips inside the layer loop (layer values, digest in LDS; early twiddle fetch): parity,
# solo kernel durations, throughput / latency against the previous library (tools/bin/variants/prev.so) on one box
set -u
OUT=gpurun_out/r6q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_batch.py -m gpu -x -q > $OUT/parity.log 2>&1; tail -1 $OUT/parity.log
cp luminair_amd/csrc/libluminair_hip.so /tmp/new.so
for v in new prev new prev new prev; do
  if [ $v = prev ]; then cp tools/bin/variants/prev.so luminair_amd/csrc/libluminair_hip.so; else cp /tmp/new.so luminair_amd/csrc/libluminair_hip.so; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), {k:v for k,v in d["stage_ms"].items() if k in ("fri_ms","merkle_ms","merkle_fused_ms")})
PY
done
cp /tmp/new.so luminair_amd/csrc/libluminair_hip.so
BENCH="python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight 1"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof -o ks -- $BENCH --steps 16 --warmup 2 > $OUT/bench_rocprof.json 2> $OUT/prof.log
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/r6q/kernel_stats.csv")):
    n=r["Name"].split("(")[0].replace("void lmn::","").replace("lmn::","")
    if "merkle" in n or "fri_tail" in n: print("%-28s calls %5s avg %8.1f us" % (n[:28], r["Calls"], float(r["AverageNs"])/1e3))
PY
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace -d $OUT/pmc_lds -o lds -- $BENCH --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_lds.log
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r6q/pmc_lds/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void lmn::", "")[:44]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_LDS": n[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_LDS_BANK_CONFLICT"])[:8]:
    print("%-46s %8d %14.0f %14.0f %8.3f" % (k, n[k], v["SQ_INSTS_LDS"], v["SQ_LDS_BANK_CONFLICT"], v["SQ_LDS_BANK_CONFLICT"] / max(1.0, v["SQ_LDS_IDX_ACTIVE"])))
PY
rm -rf $OUT/prof $OUT/pmc_lds
