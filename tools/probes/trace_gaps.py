#!/usr/bin/env python3
"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace csv (one queue): is the chain GPU-bound (gaps ~ the dispatch
boundary) or submission-bound (gaps grow)?   python tools/probes/trace_gaps.py <dir>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
rows.sort()
# the longest run of back-to-back fmt_ kernels
best, cur = [], []
for r in rows:
    if r[2].startswith("void fmt_") or r[2].startswith("fmt_"):
        cur.append(r)
    else:
        if len(cur) > len(best):
            best = cur
        cur = []
if len(cur) > len(best):
    best = cur
n = len(best)
span = best[-1][1] - best[0][0]
dur = sum(e - s for s, e, _ in best)
gaps = sorted(best[i + 1][0] - best[i][1] for i in range(n - 1))
print("chain of %d kernels: span %.1f us, sum of durations %.1f us (%.1f %%), gaps: median %.2f us, p10 %.2f, p90 %.2f, max %.1f, negative %d" % (
    n, span / 1e3, dur / 1e3, 100.0 * dur / span, gaps[n // 2] / 1e3, gaps[n // 10] / 1e3, gaps[9 * n // 10] / 1e3, gaps[-1] / 1e3,
    sum(1 for g in gaps if g < 0)))
durs = sorted(e - s for s, e, _ in best)
print("durations: median %.2f us, p10 %.2f, p90 %.2f" % (durs[n // 2] / 1e3, durs[n // 10] / 1e3, durs[9 * n // 10] / 1e3))
