#!/usr/bin/env python3
"""Per (kernel, grid) average of each counter in a rocprofv3 --pmc counter_collection csv.
   python tools/summarize_pmc.py <dir> > profiles/rNN_pmc_<what>.csv"""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
names = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"].split("(")[0][:70], r.get("Grid_Size", ""))
        c = r["Counter_Name"]
        if c not in names:
            names.append(c)
        agg[key][c] += float(r["Counter_Value"])
        cnt[key][c] += 1
print("kernel,grid,dispatches," + ",".join(names))
for key in sorted(agg, key=lambda k: -sum(agg[k].values())):
    n = max(cnt[key].values())
    print("%s,%s,%d,%s" % (key[0], key[1], n, ",".join("%.4g" % (agg[key][c] / max(1, cnt[key][c])) for c in names)))
