#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of a multi-stream run: GPU busy fraction (union of kernel intervals) and the
time-weighted number of kernels in flight, over the window that holds the middle 80 % of the trace."""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo, hi = t0 + (t1 - t0) // 10, t1 - (t1 - t0) // 10
ev = []
for s, e, _ in rows:
    s, e = max(s, lo), min(e, hi)
    if e > s:
        ev += [(s, 1), (e, -1)]
ev.sort()
busy = conc = 0
depth, last = 0, lo
for t, d in ev:
    if depth > 0:
        busy += t - last
    conc += depth * (t - last)
    depth += d
    last = t
print("window %.1f ms  busy %.1f %%  mean kernels in flight %.2f" % ((hi - lo) / 1e6, 100.0 * busy / (hi - lo), conc / (hi - lo)))
