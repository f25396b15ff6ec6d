#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats (one proof in flight, so kernel durations are solo
# durations), three separate PMC passes (FETCH_SIZE, WRITE_SIZE, VALU activity), and the summaries that go to
# profiles/.  Usage: tools/profile_round.sh r2      (outputs under gpurun_out/<tag>_*)
set -u
TAG=${1:-r2}
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
BENCH="python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight 1"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/${TAG}_prof -o ks -- $BENCH --steps 32 --warmup 4 > $OUT/${TAG}_bench_under_rocprof_inflight1.json 2> $OUT/${TAG}_prof.log
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --output-format csv --pmc $C --kernel-trace -d $OUT/${TAG}_pmc_$C -o pmc -- $BENCH --steps 3 --warmup 1 > /dev/null 2> $OUT/${TAG}_pmc_$C.log
done
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/${TAG}_pmc_sq -o sq -- $BENCH --steps 4 --warmup 1 > /dev/null 2> $OUT/${TAG}_pmc_sq.log
find $OUT/${TAG}_prof -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_kernel_stats_inflight1.csv \;
KT=$(find $OUT/${TAG}_prof -name '*kernel_trace.csv' | head -1)
python tools/timeline.py $KT v > $OUT/${TAG}_kernel_timeline_one_proof.txt
F=$(find $OUT/${TAG}_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)
W=$(find $OUT/${TAG}_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python tools/pmc_summary.py $F $W $OUT/${TAG}_pmc_summary.json > $OUT/${TAG}_pmc_summary.txt
S=$(find $OUT/${TAG}_pmc_sq -name '*counter_collection.csv' | head -1)
# proofs in the SQ pass: warmup 1 + steps 4 + one per context init + 21 latency + 1 profiled
NP=$(python - <<PY
import csv
n = sum(1 for r in csv.DictReader(open("$S")) if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "k_transpose_pad" in r["Kernel_Name"])
print(n)
PY
)
python tools/valu_summary.py $S $NP > $OUT/${TAG}_valu_summary.txt
# multi-stream run: GPU busy fraction and kernels in flight
rocprofv3 --output-format csv --kernel-trace -d $OUT/${TAG}_conc -o conc -- python bench.py --no-cpu-baseline --no-extras --no-anchor --steps 8 --warmup 1 > $OUT/${TAG}_bench_under_rocprof_inflight4.json 2> $OUT/${TAG}_conc.log
python tools/overlap.py $(find $OUT/${TAG}_conc -name '*kernel_trace.csv' | head -1) 16 64 > $OUT/${TAG}_overlap.txt   # skip = 8 contexts + 8 warm-up proofs
rm -rf $OUT/${TAG}_prof $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_sq $OUT/${TAG}_conc
tail -25 $OUT/${TAG}_valu_summary.txt; cat $OUT/${TAG}_overlap.txt; tail -12 $OUT/${TAG}_kernel_timeline_one_proof.txt
