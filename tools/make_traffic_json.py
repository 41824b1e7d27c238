#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected
SEPARATELY, MI355X_MICROARCH.md 'rocprofv3 PMC slots') of tools/profile_hotpath.py.
HBM bytes = (f * FETCH_SIZE + WRITE_SIZE) * 1024 with the factors CALIBRATED on this repository's access shapes
(profiles/r05_counter_calibration.json, tools/probes/counter_calib.hip): WRITE_SIZE is exact for every store shape;
FETCH_SIZE tallies 64 B per request - f = 2 for full 128-byte lines (16-B-per-lane streams: the FMT's weight and LDS-DMA
streams, the decoder's flow / blur kernels, the 32-channel halo at 512 px), f = 1 for the 64-byte halo pieces of the
decoder's convs at >= 64 input channels."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash  # noqa: E402  (identity of the kernel sources the counters were taken on)


def per_class(d):
    out = collections.defaultdict(lambda: [0.0, 0, 0.0])  # counter sum, launches, calibrated sum
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            cls = ("fmt_adaln_gemm" if n.startswith(("void fmt_gemm_wide", "void fmt_gemm_dma", "void fmt_gemm_big4")) else "fmt_gemm" if n.startswith("void fmt_gemm_kernel") else
                   "fmt_gemm_rb" if n.startswith("void fmt_gemm_rbs") else
                   "dec_conv" if n.startswith(("void dec_conv", "void dec_zconv")) else "dec_flow" if n.startswith("void dec_flow") else
                   "dec_zblur" if n.startswith("void dec_zblur") else "dec_other" if n.startswith("void dec_") else None)
            if cls and d.rstrip("/").endswith("_fmtb") and cls != "fmt_gemm_rb":
                cls = None  # the stacked-clip pass only speaks for the row-blocked tiles (its other launches run 2 880 rows)
            if cls:
                # calibrated FETCH_SIZE factor of the launch's dominant read shape (only applied to FETCH_SIZE rows)
                f = 1.0 if (cls in ("dec_conv", "dec_zblur") and not n.startswith("void dec_conv16_kernel<FP16, 2")) else 2.0
                v = float(r["Counter_Value"])
                out[cls][0] += v
                out[cls][1] += 1
                out[cls][2] += v * (f if r["Counter_Name"] == "FETCH_SIZE" else 1.0)
    return out


fetch_dirs, write_dirs, dst = sys.argv[1].split(","), sys.argv[2].split(","), sys.argv[3]
F, W = collections.defaultdict(lambda: [0.0, 0, 0.0]), collections.defaultdict(lambda: [0.0, 0, 0.0])
for d in fetch_dirs:
    for k, v in per_class(d).items():
        for i in range(3):
            F[k][i] += v[i]
for d in write_dirs:
    for k, v in per_class(d).items():
        for i in range(3):
            W[k][i] += v[i]
res = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/profile_hotpath.py; "
                 "bytes = (f*FETCH_SIZE + WRITE_SIZE)*1024, f calibrated per access shape (profiles/r05_counter_calibration.json): "
                 "2 for full-line streams, 1 for the decoder convs' 64-byte halo pieces; WRITE_SIZE exact"}
for k in F:
    n = F[k][1]
    fetch_kb, write_kb, fetch_cal_kb = F[k][0] / n, W[k][0] / max(W[k][1], 1), F[k][2] / n
    res[k] = {"launches_profiled": n, "fetch_size_kb_per_launch": round(fetch_kb, 1), "write_size_kb_per_launch": round(write_kb, 1),
              "fetch_factor": round(fetch_cal_kb / fetch_kb, 3) if fetch_kb else None,
              "hbm_bytes_per_launch": int((fetch_cal_kb + write_kb) * 1024)}
res["source_hash"] = source_hash()
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res))
