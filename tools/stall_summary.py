#!/usr/bin/env python3
"""Where the wave-cycles of every kernel go: rocprofv3 --pmc passes holding SQ_WAVE_CYCLES, SQ_WAIT_ANY (waves parked on
s_waitcnt / barriers: memory or LDS latency), SQ_WAIT_INST_ANY (waves with an instruction ready that could not issue:
port / pipe pressure), SQ_ACTIVE_INST_ANY, and in further passes SQ_ACTIVE_INST_VALU, SQ_INSTS_VALU, SQ_WAIT_INST_LDS,
SQ_ACTIVE_INST_LDS, SQ_INSTS_SALU, SQ_ACTIVE_INST_SCA, SQ_INSTS_VMEM_RD / _WR, SQ_INSTS_LDS.  WAIT_ANY + WAIT_INST_ANY +
ACTIVE_INST_ANY ~ WAVE_CYCLES (MI355X_MICROARCH.md); all in quad-cycles summed over the chip.
Usage: stall_summary.py <n_proofs> <counter_collection.csv> [more csv ...]"""
import csv
import sys
from collections import defaultdict

nproofs = float(sys.argv[1])
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lmn::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1


def per_launch_norm(k, name, ref="SQ_WAVE_CYCLES"):
    """counter total rescaled to the number of launches the reference counter saw (passes see the same launches)"""
    if not cnt[k][name]:
        return None
    return acc[k][name] * (cnt[k][ref] / cnt[k][name] if cnt[k][ref] else 1.0)


rows = []
for k in acc:
    wc = acc[k]["SQ_WAVE_CYCLES"]
    if not wc:
        continue
    g = lambda n: per_launch_norm(k, n)
    wait, stall, active = g("SQ_WAIT_ANY"), g("SQ_WAIT_INST_ANY"), g("SQ_ACTIVE_INST_ANY")
    valu_act, valu_n = g("SQ_ACTIVE_INST_VALU"), g("SQ_INSTS_VALU")
    lds_stall, lds_act = g("SQ_WAIT_INST_LDS"), g("SQ_ACTIVE_INST_LDS")
    rows.append((wc, k, cnt[k]["SQ_WAVE_CYCLES"], wait, stall, active, valu_act, valu_n, lds_stall, lds_act))
tot = sum(r[0] for r in rows)
fmt = lambda x, d: ("%6.1f%%" % (100.0 * x / d)) if (x is not None and d) else "     - "
print("%-30s %5s %9s | %7s %7s %7s | %7s %7s %7s | %s" % ("kernel", "n", "wave-cyc", "parked", "stalled", "active", "VALU", "LDSstal", "LDSact", "reading"))
for wc, k, n, wait, stall, active, va, vn, ls, la in sorted(rows, reverse=True)[:20]:
    verdict = []
    if wait is not None and wait / wc > 0.5:
        verdict.append("latency-bound (waves parked on waitcnt/barrier)")
    if stall is not None and stall / wc > 0.25:
        verdict.append("issue-stalled" + (" on LDS" if ls and ls / max(stall, 1) > 0.5 else " (port / dependency)"))
    if va is not None and va / wc > 0.35:
        verdict.append("VALU-busy")
    print("%-30s %5d %8.1f%% | %s %s %s | %s %s %s | %s" % (k[:30], n, 100 * wc / tot, fmt(wait, wc), fmt(stall, wc), fmt(active, wc),
                                                         fmt(va, wc), fmt(ls, wc), fmt(la, wc), "; ".join(verdict) or "mixed"))
print("wave-cycles per proof (quad-cycles, chip): %.3g" % (tot / nproofs))
