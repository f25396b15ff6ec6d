#!/usr/bin/env python3
"""Per-kernel VALU-busy share from a rocprofv3 --pmc pass holding SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES and
GRBM_GUI_ACTIVE (quad-cycle SQ units summed over the chip; GRBM summed over 8 XCDs).
Usage: valu_summary.py <counter_collection.csv> <n_proofs>"""
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lmn::", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        n[k] += 1
nproofs = float(sys.argv[2])
tv = tg = 0
rows = []
for k, v in acc.items():
    valu = v["SQ_ACTIVE_INST_VALU"] * 4 / 1024; gui = v["GRBM_GUI_ACTIVE"] / 8
    rows.append((valu, k, n[k], gui, valu / gui if gui else 0, v["SQ_WAVE_CYCLES"] * 4 / 1024 / gui if gui else 0))
    tv += valu; tg += gui
print("%-28s %6s %14s %14s %6s %10s" % ("kernel", "n", "VALU cyc/proof", "GPU cyc/proof", "util", "waves/SIMD"))
for valu, k, c, g, u, occ in sorted(rows, reverse=True)[:18]:
    print("%-28s %6d %14.0f %14.0f %6.2f %10.2f" % (k[:28], c, valu / nproofs, g / nproofs, u, occ))
print("total VALU-busy SIMD cycles per proof: %.0f   GPU-active cycles per proof: %.0f" % (tv / nproofs, tg / nproofs))
