#!/usr/bin/env python3
"""Per kernel of a rocprofv3 --kernel-trace csv: average duration AND the average idle gap to the next kernel's start, over
the last `n` dispatches (one graph replay of the FMT chain) - kernel durations alone do not add up to the wall clock.
    python tools/trace_gaps.py <dir> [n_last] [name filter]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:64], r.get("Grid_Size_X", r.get("Grid_Size", ""))))
rows.sort()
rows = rows[-n_last:]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for i, (s, e, n, g) in enumerate(rows):
    if flt and flt not in n:
        continue
    a = agg[(n, g)]
    a[0] += 1
    a[1] += (e - s) / 1e3
    if i + 1 < len(rows):
        a[2] += (rows[i + 1][0] - e) / 1e3
span = (rows[-1][1] - rows[0][0]) / 1e6
busy = sum(e - s for s, e, _, _ in rows) / 1e6
print("span %.2f ms, sum of durations %.2f ms, %d dispatches" % (span, busy, len(rows)))
print("kernel,grid,calls,avg_us,avg_gap_after_us,total_ms(dur+gap)")
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print("%s,%s,%d,%.2f,%.2f,%.2f" % (k[0], k[1], v[0], v[1] / v[0], v[2] / v[0], (v[1] + v[2]) / 1e3))
