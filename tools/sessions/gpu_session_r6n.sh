#!/bin/bash
# round 4: transpose tile with an odd row stride (LDS bank conflicts); LDS bank-conflict counters of every kernel
set -u
OUT=gpurun_out/r6n
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kat or full_size_2_20 or ragged or proof_equals_oracle or random" > $OUT/parity.log 2>&1; tail -1 $OUT/parity.log
BENCH="python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight 1"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof -o ks -- $BENCH --steps 16 --warmup 2 > $OUT/bench_rocprof.json 2> $OUT/prof.log
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
head -40 $OUT/kernel_stats.csv | cut -d, -f1-6
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace -d $OUT/pmc_lds -o lds -- $BENCH --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_lds.log
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r6n/pmc_lds/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void lmn::", "")[:44]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_LDS": n[k] += 1
print("%-46s %8s %14s %14s %8s" % ("kernel", "launches", "LDS insts", "conflict cyc", "conf/idx"))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_LDS_BANK_CONFLICT"]):
    print("%-46s %8d %14.0f %14.0f %8.3f" % (k, n[k], v["SQ_INSTS_LDS"], v["SQ_LDS_BANK_CONFLICT"], v["SQ_LDS_BANK_CONFLICT"] / max(1.0, v["SQ_LDS_IDX_ACTIVE"])))
PY
rm -rf $OUT/prof $OUT/pmc_lds
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product', round(d['value'],1), 'solo', round(d['prove_latency_ms'],3), d['stage_ms']['transpose_ms'])"; done
