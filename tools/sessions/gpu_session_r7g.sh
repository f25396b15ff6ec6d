#!/bin/bash
# round 4: kernel durations of one BASELINE config 5 proof (2^24 rows; three-pass transforms)
set -u
OUT=gpurun_out/r7g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof -o ks -- python tools/config5_latency.py > $OUT/config5.json 2> $OUT/prof.log
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r7g/kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    n=r["Name"].split("(")[0].replace("void lmn::","").replace("lmn::","")
    print("%-44s calls %5s avg %9.1f us  total %8.2f ms  %5.1f%%" % (n[:44], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, 100*float(r["TotalDurationNs"])/tot))
PY
cat $OUT/config5.json | tail -2 | cut -c1-400
rm -rf $OUT/prof
