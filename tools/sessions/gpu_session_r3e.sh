#!/bin/bash
set -u
OUT=gpurun_out/r3e
mkdir -p $OUT
timeout 300 tools/bin/mb_reconcile > $OUT/reconcile.txt 2> $OUT/reconcile.err; echo "rc=$?"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 300 rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_a -o pmc -- tools/bin/mb_reconcile pmc > $OUT/reconcile_pmc_a.txt 2> $OUT/pmc_a.log; echo "pmc_a rc=$?"
timeout 300 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d $OUT/pmc_b -o pmc -- tools/bin/mb_reconcile pmc > $OUT/reconcile_pmc_b.txt 2> $OUT/pmc_b.log; echo "pmc_b rc=$?"
for d in pmc_a pmc_b; do f=$(find $OUT/$d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && cp $f $OUT/${d}_counters.csv; rm -rf $OUT/$d; done
cat $OUT/reconcile.txt; tail -3 $OUT/pmc_a.log $OUT/pmc_b.log
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r3e/pmc_*_counters.csv")):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f)
    for k, v in acc.items():
        print("  %-22s" % k, {c: [round(x) for x in xs] for c, xs in v.items()})
PY
