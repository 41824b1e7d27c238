#!/usr/bin/env python3
"""Per (kernel, grid) average duration from a rocprofv3 --kernel-trace csv: tells which decoder level / which GEMM
shape the time of a kernel family goes to.   python tools/summarize_trace.py <dir> [name filter]"""
import collections
import csv
import glob
import sys

d, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if flt and flt not in n:
            continue
        key = (n.split("(")[0][:60], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("LDS_Block_Size", ""))
        a = agg[key]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print("kernel,grid_x,grid_y,lds,calls,avg_us,total_ms,share")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%s,%s,%s,%s,%d,%.1f,%.2f,%.1f%%" % (k + (v[0], v[1] / v[0], v[1] / 1e3, 100 * v[1] / tot)))
