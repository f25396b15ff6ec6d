#!/usr/bin/env python3
"""Per-kernel VALU issue from a rocprofv3 --pmc pass holding SQ_ACTIVE_INST_VALU (wave-instructions issued, summed over
the chip), SQ_WAVE_CYCLES (quad-cycle units) and GRBM_GUI_ACTIVE (summed over 8 XCDs).  "1-port cyc" = instructions x 4
cycles / 1024 SIMDs: the time the launch would need if every instruction took a whole issue slot (the pre-co-issue model,
profiles/ceilings/valu_issue_reconciled.txt); instr/clk/SIMD above 0.25 means both issue ports were used
(profiles/ceilings/valu_coissue_two_ports.txt).
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
print("%-28s %6s %16s %14s %14s %10s" % ("kernel", "n", "1-port cyc/proof", "GPU cyc/proof", "instr/clk/SIMD", "waves/SIMD"))
for valu, k, c, g, u, occ in sorted(rows, reverse=True)[:18]:
    print("%-28s %6d %16.0f %14.0f %14.3f %10.2f" % (k[:28], c, valu / nproofs, g / nproofs, u / 4, occ))
print("total: %.0f wave-instructions per SIMD per proof (x 4 = %.0f one-port cycles)   GPU-active cycles per proof, solo: %.0f" % (tv / nproofs / 4, tv / nproofs, tg / nproofs))
