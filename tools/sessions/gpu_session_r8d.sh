#!/bin/bash
# round 5, session d: (1) the round's profile set (tools/profile_round.sh r5: kernel stats, FETCH/WRITE PMC, VALU issue,
# overlap); (2) stall attribution per kernel: SQ wave-cycle buckets in separate --pmc passes (kernel-trace only), solo and
# with 8 contexts (to see whether rocprofv3 lets kernels of different streams overlap while it collects counters)
set -u
OUT=gpurun_out/r8d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash tools/profile_round.sh r5 > $OUT/profile_round.log 2>&1; tail -5 $OUT/profile_round.log
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counters_available.txt; wc -l $OUT/sq_counters_available.txt
BENCH="python bench.py --no-cpu-baseline --no-extras --no-anchor"
pass() {  # tag inflight counters...
  local tag=$1 infl=$2; shift 2
  timeout 600 rocprofv3 --output-format csv --pmc "$@" --kernel-trace -d $OUT/pmc_$tag -o p -- $BENCH --inflight $infl --steps 3 --warmup 1 > /dev/null 2> $OUT/pmc_$tag.log
  find $OUT/pmc_$tag -name '*counter_collection.csv' | head -1
}
A=$(pass a1 1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY)
B=$(pass b1 1 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS)
C=$(pass c1 1 SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU)
NP=$(python - <<PY
import csv
print(sum(1 for r in csv.DictReader(open("$A")) if r["Counter_Name"] == "SQ_WAVE_CYCLES" and "k_transpose_pad" in r["Kernel_Name"]))
PY
)
python tools/stall_summary.py $NP $A $B $C > $OUT/r5_stall_attribution_solo.txt; cat $OUT/r5_stall_attribution_solo.txt
A8=$(pass a8 8 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY)
B8=$(pass b8 8 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS)
NP8=$(python - <<PY
import csv
print(sum(1 for r in csv.DictReader(open("$A8")) if r["Counter_Name"] == "SQ_WAVE_CYCLES" and "k_transpose_pad" in r["Kernel_Name"]))
PY
)
python tools/stall_summary.py $NP8 $A8 $B8 > $OUT/r5_stall_attribution_8ctx.txt; cat $OUT/r5_stall_attribution_8ctx.txt
# do kernels of different contexts overlap while counters are collected?
KT=$(find $OUT/pmc_a8 -name '*kernel_trace.csv' | head -1)
python tools/overlap.py $KT 16 64 > $OUT/r5_overlap_under_pmc.txt 2>&1; cat $OUT/r5_overlap_under_pmc.txt
rm -rf $OUT/pmc_a1 $OUT/pmc_b1 $OUT/pmc_c1 $OUT/pmc_a8 $OUT/pmc_b8
