#!/bin/bash
# counters for the co-issue streams: SQ_INSTS_VALU / (GRBM_GUI_ACTIVE / 8) per launch
set -u
OUT=gpurun_out/r4w
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 300 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace -d $OUT/pmc -o pmc -- tools/bin/mb_reconcile pmc2 > $OUT/pmc2.txt 2> $OUT/pmc2.log; echo "rc=$?"
F=$(find $OUT/pmc -name '*counter_collection.csv' | head -1)
python - <<PY
import csv, collections
rows = collections.defaultdict(dict)
for r in csv.DictReader(open("$F")):
    rows[(r["Kernel_Name"].split("(")[0], r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for (k, d), v in sorted(rows.items(), key=lambda kv: int(kv[0][1])):
    if "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"] > 0:
        print("%-46s dispatch %3s  SQ_INSTS_VALU %12.0f  GRBM_GUI_ACTIVE/8 %10.0f  instr/clk/SIMD %.3f  waves %d" % (k[:46], d, v["SQ_INSTS_VALU"], v["GRBM_GUI_ACTIVE"]/8, v["SQ_INSTS_VALU"]/1024/(v["GRBM_GUI_ACTIVE"]/8), v.get("SQ_WAVES",0)))
PY
cut -c1-75 $OUT/pmc2.txt
rm -rf $OUT/pmc
