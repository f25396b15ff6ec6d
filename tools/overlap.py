#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of a multi-context bench run: GPU busy fraction (union of kernel intervals) and
the time-weighted number of kernels in flight over the TIMED REGION, located by counting proofs: every proof starts with
one k_transpose_pad launch per table, so the region spans from the launch that opens proof `skip` to the end of the
last kernel before proof `skip + count` starts (for bench.py: skip = contexts + warmup, count = steps).
Usage: overlap.py <kernel_trace.csv> <skip> <count>"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
skip, count = int(sys.argv[2]), int(sys.argv[3])
starts = [s for s, _, k in rows if "k_transpose_pad" in k]
lo = starts[skip]
hi = starts[skip + count] if skip + count < len(starts) else max(e for _, e, _ in rows)
# the proofs of the region end after the next region's first transposes start only if there is a next region; clip
ev = []
for s, e, _ in rows:
    s, e = max(s, lo), min(e, hi)
    if e > s:
        ev += [(s, 1), (e, -1)]
ev.sort()
busy = conc = 0
depth, last = 0, lo
hist = {}
for t, d in ev:
    if depth > 0:
        busy += t - last
    conc += depth * (t - last)
    hist[min(depth, 8)] = hist.get(min(depth, 8), 0) + (t - last)
    depth += d
    last = t
span = hi - lo
print("timed region: proofs %d..%d, %.2f ms (%.3f ms per proof)  GPU busy %.1f %%  mean kernels in flight %.2f" % (
    skip, skip + count, span / 1e6, span / 1e6 / count, 100.0 * busy / span, conc / span))
print("time share by number of kernels in flight (8 = 8 or more): " + "  ".join("%d: %.1f%%" % (k, 100.0 * v / span) for k, v in sorted(hist.items())))

# per-kernel mean duration inside the region (compare with the solo durations of the one-proof-in-flight run: the ratio
# says which kernels pay for sharing the chip)
per = {}
for s_, e_, k in rows:
    if s_ >= lo and e_ <= hi:
        a = per.setdefault(k.replace("void ", "").replace("lmn::", ""), [0, 0])
        a[0] += 1
        a[1] += e_ - s_
print("kernel                               launches/proof   mean us   sum us/proof")
for k, (n, tot) in sorted(per.items(), key=lambda kv: -kv[1][1])[:18]:
    print("%-36s %8.1f %12.1f %12.1f" % (k[:36], n / count, tot / n / 1e3, tot / count / 1e3))
