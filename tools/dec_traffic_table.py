#!/usr/bin/env python3
"""Decoder HBM traffic by launch from the PMC summaries (profiles/<round>_pmc_{fetch,write}_size_dec.csv; tools/profile_hotpath.py
--what dec decodes 16 frames twice), beside the bytes each launch must move (input once + output once, 16-bit NHWC).
python tools/dec_traffic_table.py [round]  ->  one line per (kernel, grid): MB per frame."""
import re
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
FR = 16  # frames per launch in profile_hotpath.py


def load(fn):
    d = {}
    for line in open(fn).read().splitlines()[1:]:
        m = re.match(r'^(.*),(\d+),(\d+),([0-9.e+\-]+)$', line)
        if m:
            d[(m.group(1), int(m.group(2)))] = (int(m.group(3)), float(m.group(4)))
    return d


f = load("profiles/%s_pmc_fetch_size_dec.csv" % rnd)
w = load("profiles/%s_pmc_write_size_dec.csv" % rnd)
C = {64: 256, 128: 128, 256: 64, 512: 32, 32: 512, 16: 512, 8: 512}  # channels by resolution (channel_multiplier 1)


def act(R):
    return R * R * C[R] * 2 / 1e6  # MB per frame


# (kernel substring, grid) -> (label, algorithmic read MB/frame, algorithmic write MB/frame); grids of the 512-px decoder at 16 frames
rows = []
for (k, grid), (n, fs) in sorted(f.items(), key=lambda kv: -kv[1][1]):
    if "dec_" not in k:
        continue
    n2, ws = w.get((k, grid), (n, 0.0))
    rows.append((k.replace("void ", "")[:40], grid, n, 2 * fs * 1024 / n / FR / 1e6, ws * 1024 / n2 / FR / 1e6))
tot_f = sum(r[3] * r[2] for r in rows) / 2  # two decodes
tot_w = sum(r[4] * r[2] for r in rows) / 2
print("%-40s %9s %3s %12s %12s" % ("kernel", "grid", "n", "fetch MB/fr", "write MB/fr"))
for r in rows:
    print("%-40s %9d %3d %12.2f %12.2f" % r)
print("sum over one decode: fetch %.1f + write %.1f = %.1f MB per frame (feature repack of the clip included: once per clip)" % (tot_f, tot_w, tot_f + tot_w))
print("activations, MB per frame: " + ", ".join("%d px %.1f" % (R, act(R)) for R in (64, 128, 256, 512)))
