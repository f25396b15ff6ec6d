#!/usr/bin/env python3
"""Average every PMC counter per kernel from a rocprofv3 --pmc counter_collection.csv.
Usage: pmc_generic.py <counter_collection.csv> [kernel-substring ...]"""
import csv, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lmn::", "")
    a = acc[k][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
want = sys.argv[2:]
for k in sorted(acc):
    if want and not any(w in k for w in want):
        continue
    print(k)
    for c, (n, s) in sorted(acc[k].items()):
        print("   %-28s n=%4d avg %16.1f" % (c, n, s / n))
