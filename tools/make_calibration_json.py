#!/usr/bin/env python3
"""profiles/rNN_counter_calibration.json from tools/probes/counter_calib.hip run under `rocprofv3 --pmc FETCH_SIZE` and
`--pmc WRITE_SIZE` (separate passes): counter bytes / bytes the kernel moved, per access shape.
    python tools/make_calibration_json.py <calib stdout> <fetch dir> <write dir> <out.json>"""
import csv
import glob
import json
import re
import sys

log, fetch_dir, write_dir, dst = sys.argv[1:5]
known = {}
for l in open(log):
    m = re.match(r"CALIB (\S+) (\d+) (\d+)(?: requested (\d+))?", l)
    if m:
        known[m.group(1)] = {"bytes_read": int(m.group(2)), "bytes_written": int(m.group(3)), "bytes_requested": int(m.group(4) or 0)}


def counters(d, name):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            k = re.sub(r"^void |\(.*$", "", r["Kernel_Name"])
            out[k] = out.get(k, 0.0) + float(r["Counter_Value"])
    return out


F, W = counters(fetch_dir, "FETCH_SIZE"), counters(write_dir, "WRITE_SIZE")
res = {"method": "tools/probes/counter_calib.hip: each kernel moves a known byte count once over 2 GiB; factor = counter x 1024 / bytes "
                 "(rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; counters in KB)", "shapes": {}}
for k, v in known.items():
    e = dict(v)
    if v["bytes_read"]:
        e["fetch_size_kb"] = F.get(k)
        e["fetch_factor"] = round(F[k] * 1024.0 / v["bytes_read"], 4) if k in F else None
    if v["bytes_written"]:
        e["write_size_kb"] = W.get(k)
        e["write_factor"] = round(W[k] * 1024.0 / v["bytes_written"], 4) if k in W else None
    res["shapes"][k] = e
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res, indent=1))
