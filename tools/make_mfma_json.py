#!/usr/bin/env python3
"""MFMA utilisation per kernel class from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE + wave-state
counters) over tools/profile_hotpath.py:  util = sum(MFMA busy cycles over the SIMDs) / (kernel cycles x 1024 SIMDs), kernel
cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs, MI355X_MICROARCH.md 'DVFS give-back').  Also the wave-state split
WAIT_ANY / WAIT_INST_ANY / ACTIVE_INST_ANY as fractions of SQ_WAVE_CYCLES.
    python tools/make_mfma_json.py <dir fmt>,<dir dec> out.json"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash  # noqa: E402  (identity of the kernel sources the counters were taken on)

N_SIMD = 256 * 4
CLASSES = (("fmt_adaln_gemm", ("void fmt_gemm_wide", "void fmt_gemm_dma", "void fmt_gemm_big4")), ("fmt_gemm", ("void fmt_gemm_kernel",)), ("fmt_gemm_rb", ("void fmt_gemm_rbs",)), ("dec_conv", ("void dec_conv", "void dec_zconv")), ("fmt_small", ("void fmt_lnmod", "void fmt_attn")),
           ("dec_flow", ("void dec_flow",)), ("dec_blur", ("void dec_blur",)))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for d in sys.argv[1].split(","):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            for cls, pre in CLASSES:
                # the stacked-clip pass (..._fmtb) is there for the row-blocked tiles: its step-chain / adaLN launches (2 880 rows)
                # must not be averaged into the one-clip classes
                if d.rstrip("/").endswith("_fmtb") and cls != "fmt_gemm_rb":
                    continue
                if n.startswith(pre):
                    agg[cls][r["Counter_Name"]] += float(r["Counter_Value"])
                    disp[cls].add((f, r["Dispatch_Id"]))
out = {"method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS "
                 "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE over tools/profile_hotpath.py; mfma_util = MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs)"}
for cls, c in agg.items():
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    out[cls] = {"dispatches": len(disp[cls]),
                "mfma_util": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * N_SIMD), 4) if cyc else None,
                "wave_wait_any": round(c.get("SQ_WAIT_ANY", 0.0) / wc, 3), "wave_wait_inst": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3),
                "wave_active": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3), "wave_active_lds": round(c.get("SQ_ACTIVE_INST_LDS", 0.0) / wc, 3),
                "wave_active_valu": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 3)}
out["source_hash"] = source_hash()
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
