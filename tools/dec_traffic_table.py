#!/usr/bin/env python3
"""Decoder HBM traffic by launch from the PMC summaries (profiles/<round>_pmc_{fetch,write}_size_dec.csv; tools/profile_hotpath.py
--what dec decodes 16 frames twice), beside the bytes each launch must move (input once + output once, 16-bit NHWC).
python tools/dec_traffic_table.py [round]  ->  one line per (kernel, grid): MB per frame."""
import re
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
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
# Calibration (profiles/r05_counter_calibration.json, tools/probes/counter_calib.hip): WRITE_SIZE is EXACT for every store shape of
# the decoder (8 B per lane, 32-byte runs per pixel included); FETCH_SIZE tallies 64 B per request - exact for the 64-byte pieces of
# a halo at >= 128-byte pixel stride (Cin >= 64), HALF for full 128-byte lines (16-B-per-lane streams, the Cin = 32 halo at 512 px).
# Per row: the counter as it reads (x1), doubled (x2), and the calibrated pick for the row's dominant read shape.
def fetch_factor(kernel, grid):
    if "dec_conv16_kernel<FP16, 2" in kernel:  # the 32-channel tiles: 512 px, Cin = 32 -> full lines
        return 2.0
    if "dec_conv" in kernel or "dec_zconv" in kernel or "dec_zblur" in kernel:
        return 2.0 if ("dec_zblur" in kernel and grid <= 300000) else 1.0  # halo pieces of 64 B; (the 16 -> 32 px blur reads whole rows)
    return 2.0  # flow / blur / repack / style kernels stream 16 B per lane


rows = []
decodes = 2  # tools/profile_hotpath.py --what dec decodes twice
for (k, grid), (n, fs) in sorted(f.items(), key=lambda kv: -kv[1][1] * kv[1][0]):
    if "dec_" not in k:
        continue
    n2, ws = w.get((k, grid), (n, 0.0))
    per = n / decodes  # dispatches of this (kernel, grid) per decode; the csv holds the AVERAGE per dispatch
    x1 = fs * 1024 / FR / 1e6
    rows.append((k.replace("void ", "")[:40], grid, n, x1 * per, 2 * x1 * per, fetch_factor(k, grid) * x1 * per, ws * 1024 / FR / 1e6 * (n2 / decodes)))
print("%-40s %9s %3s %12s %12s %12s %12s" % ("kernel", "grid", "n", "fetch x1", "fetch x2", "fetch calib", "write (exact)"))
for r in rows:
    print("%-40s %9d %3d %12.2f %12.2f %12.2f %12.2f" % r)
print("sum over one decode, MB per frame: fetch x1 %.1f / x2 %.1f / calibrated %.1f, write %.1f (feature repack of the clip included: once per clip)"
      % tuple(sum(r[i] for r in rows) for i in (3, 4, 5, 6)))
print("activations, MB per frame: " + ", ".join("%d px %.1f" % (R, act(R)) for R in (64, 128, 256, 512)))
