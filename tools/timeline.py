#!/usr/bin/env python3
"""Print the kernel timeline of the last proof in a rocprofv3 --kernel-trace CSV (dev helper)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'transpose' in r['Kernel_Name']]
# third-from-last proof: the last one in a bench.py run is the solo proof with HIP-event profiling on
idx, end = (starts[-3], starts[-2]) if len(starts) >= 3 else (starts[-1], len(rows))
rows = rows[:end]
t0 = int(rows[idx]['Start_Timestamp'])
prev_end = t0
tot = {}
for r in rows[idx:]:
    name = r['Kernel_Name'].split('(')[0].replace('lmn::', '').replace('void ', '')
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if len(sys.argv) > 2:
        print("%8.1f gap %6.1f dur %7.1f  %-28s grid %s" % ((st - t0) / 1e3, (st - prev_end) / 1e3, (en - st) / 1e3, name[:28], r['Grid_Size_X']))
    tot[name] = tot.get(name, 0) + (en - st) / 1e3
    tot['_gaps'] = tot.get('_gaps', 0) + max(0, (st - prev_end) / 1e3)
    prev_end = en
print("end-to-end %.1f us" % ((prev_end - t0) / 1e3))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("%10.1f us  %s" % (v, k))
