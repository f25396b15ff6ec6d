#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into per-kernel HBM
traffic per launch.  Units: the counters are in KiB (MI355X_MICROARCH.md §HBM); on gfx950 FETCH_SIZE
reports half the bytes of a wide coalesced read, so the corrected read traffic is 2 x FETCH_SIZE.
Usage: pmc_summary.py <fetch_csv> <write_csv> <out_json>"""
import csv, json, sys
from collections import defaultdict


def per_kernel(path, name):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lmn::", "")
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    return acc


f = per_kernel(sys.argv[1], "FETCH_SIZE")
w = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    nf, sf = f.get(k, [0, 0.0])
    nw, sw = w.get(k, [0, 0.0])
    out[k] = {
        "launches": nf or nw,
        "fetch_kib_per_launch_raw": sf / nf if nf else None,
        "write_kib_per_launch_raw": sw / nw if nw else None,
        "hbm_bytes_per_launch_corrected": (2.0 * (sf / nf if nf else 0.0) + (sw / nw if nw else 0.0)) * 1024.0,
    }
json.dump({"note": "corrected = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md (gfx950 FETCH_SIZE "
                   "undercounts wide coalesced reads by 2x; WRITE_SIZE uncalibrated)", "kernels": out},
          open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print("%-28s n=%4d fetch %10.1f KiB write %10.1f KiB -> %8.1f MB corrected" % (
        k[:28], v["launches"], v["fetch_kib_per_launch_raw"] or 0, v["write_kib_per_launch_raw"] or 0,
        v["hbm_bytes_per_launch_corrected"] / 1e6))
