#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection csv by kernel (and grid size)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"].split("(")[0][:44], r.get("Grid_Size", r.get("Grid_Size_X", "")))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(key, r["Counter_Name"])] += 1
names = sorted({c for v in agg.values() for c in v})
print("kernel,grid,dispatches," + ",".join(names))
for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    n = max(cnt[(key, c)] for c in v)
    print("%s,%s,%d," % (key[0], key[1], n) + ",".join("%.4g" % (v.get(c, 0.0) / n) for c in names))
