#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected
SEPARATELY, MI355X_MICROARCH.md 'rocprofv3 PMC slots') of tools/profile_hotpath.py.
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE is in KB and reads exactly half of a wide
coalesced stream on gfx950 (MI355X_MICROARCH.md 'HBM'); WRITE_SIZE is exact for 16-B-per-lane stores."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash  # noqa: E402  (identity of the kernel sources the counters were taken on)


def per_class(d):
    out = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            cls = ("fmt_adaln_gemm" if n.startswith(("void fmt_gemm_wide", "void fmt_gemm_dma")) else "fmt_gemm" if n.startswith("void fmt_gemm_kernel") else
                   "dec_conv" if n.startswith(("void dec_conv", "void dec_zconv")) else "dec_flow" if n.startswith("void dec_flow") else
                   "dec_zblur" if n.startswith("void dec_zblur") else "dec_other" if n.startswith("void dec_") else None)
            if cls:
                out[cls][0] += float(r["Counter_Value"])
                out[cls][1] += 1
    return out


fetch_dirs, write_dirs, dst = sys.argv[1].split(","), sys.argv[2].split(","), sys.argv[3]
F, W = collections.defaultdict(lambda: [0.0, 0]), collections.defaultdict(lambda: [0.0, 0])
for d in fetch_dirs:
    for k, v in per_class(d).items():
        F[k][0] += v[0]
        F[k][1] += v[1]
for d in write_dirs:
    for k, v in per_class(d).items():
        W[k][0] += v[0]
        W[k][1] += v[1]
res = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/profile_hotpath.py; "
                 "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE reads 1/2)"}
for k in F:
    n = F[k][1]
    fetch_kb, write_kb = F[k][0] / n, W[k][0] / max(W[k][1], 1)
    res[k] = {"launches_profiled": n, "fetch_size_kb_per_launch": round(fetch_kb, 1), "write_size_kb_per_launch": round(write_kb, 1),
              "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024)}
res["source_hash"] = source_hash()
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res))
