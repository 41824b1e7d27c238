#!/usr/bin/env python3
"""Rewrite the figures DESIGN.md section 6 / README.md / BASELINE.md quote from profiles/r05_bench.json (the bench line of the
profile round) after a new profile round: python tools/fill_design_numbers.py   (prints what it changed; idempotent)."""
import json
import re

d = json.load(open("profiles/r05_bench.json"))
T = d["config"].get("frames_per_clip", 250)
ms = lambda k: "%.1f" % d[k]
fps = lambda k: "%.0f" % d[k]
st = d["stage_ms"]
enc = st["appearance_encoder"] + st["audio_encoder"]
host = d["ms_per_step"] - enc - st["fmt_sample"] - st["decode_and_d2h"]
b = [d["value_batch%d" % n] for n in (4, 8, 16)]
bms = [d["ms_per_step_batch%d" % n] / n for n in (4, 8, 16)]
r1, r3, rb = d["roofline"], d["roofline_tertiary"], d["roofline_batch"]
ksum = d["kernel_class_ms"]
subs = {
    "DESIGN.md": [
        (r"encoders [\d.]+ \+ FMT [\d.]+ \+ decode and hand-over [\d.]+ \+ host ~[\d.]+ \| \*\*[\d.]+\*\* \| \*\*\d+\*\* \|",
         "encoders %.1f + FMT %.1f + decode and hand-over %.1f + host ~%.1f | **%s** | **%s** |" % (enc, st["fmt_sample"], st["decode_and_d2h"], host, ms("ms_per_step"), fps("value"))),
        (r"\(`value_s2e`\) \| [\d.]+ \| \d+ \|", "(`value_s2e`) | %s | %s |" % (ms("ms_per_step_s2e"), fps("value_s2e"))),
        (r"\(`value_nfe50`\) \| [\d.]+ \| \d+ \|", "(`value_nfe50`) | %s | %s |" % (ms("ms_per_step_nfe50"), fps("value_nfe50"))),
        (r"\(`value_from_host_inputs`\) \| [\d.]+ \| \d+ \|", "(`value_from_host_inputs`) | %s | %s |" % (ms("ms_per_step_from_host_inputs"), fps("value_from_host_inputs"))),
        (r"\(`value_bf16`\) \| [\d.]+ \| \d+ \|", "(`value_bf16`) | %s | %s |" % (ms("ms_per_step_bf16"), fps("value_bf16"))),
        (r"B = 4 / 8 / 16 \| [\d.]+ / [\d.]+ / [\d.]+ \| \*\*\d+ / \d+ / \d+\*\* \|", "B = 4 / 8 / 16 | %.1f / %.1f / %.1f | **%.0f / %.0f / %.0f** |" % (*bms, *b)),
        (r"now \d+ at 4, \d+ at 16", "now %.0f at 4, %.0f at 16" % (b[0], b[2])),
        (r"`value_batch4` \*\*\d+\*\* \(was 3035\), `value_batch8` \d+, `value_batch16` \d+", "`value_batch4` **%.0f** (was 3035), `value_batch8` %.0f, `value_batch16` %.0f" % tuple(b)),
        (r"\(single clip, [\d.]+ of \d+ ms\): 8 507 launches per clip, [\d.]+ us average, 6.13 MB algorithmic per launch -> \*\*\d+ GB/s =\n  [\d.]+ of the 8 TB/s HBM peak\*\* \([\d.]+ of",
         "(single clip, %.1f of %.0f ms): 8 507 launches per clip, %.2f us average, 6.13 MB algorithmic per launch -> **%.0f GB/s =\n  %.3f of the 8 TB/s HBM peak** (%.3f of" % (ksum["fmt_gemm"], d["ms_per_step"], r1["avg_launch_us"], r1["achieved"], r1["frac"], r1["frac_measured"])),
        (r"8 000 launches per 250 evaluations, [\d.]+ us average =\n  \*\*\d+ TFLOP/s = [\d.]+", "8 000 launches per 250 evaluations, %.1f us average =\n  **%.0f TFLOP/s = %.2f" % (rb["avg_launch_us"], rb["achieved"], rb["frac"])),
    ],
    "README.md": [
        (r"\*\*\d+ frames/s end to end\*\* \([\d.]+ ms per clip: encoders [\d.]+ \+ FMT [\d.]+ \+ decode and hand-over [\d.]+;",
         "**%s frames/s end to end** (%s ms per clip: encoders %.1f + FMT %.1f + decode and hand-over %.1f;" % (fps("value"), ms("ms_per_step"), enc, st["fmt_sample"], st["decode_and_d2h"])),
        (r"\(the reference's default widget\) \d+; from host inputs through `run_inference` \d+; bf16 FMT operands \d+;",
         "(the reference's default widget) %s; from host inputs through `run_inference` %s; bf16 FMT operands %s;" % (fps("value_s2e"), fps("value_from_host_inputs"), fps("value_bf16"))),
        (r"\*\*\d+ / \d+ / \d+ frames/s end to end at B = 4 / 8 / 16\*\*", "**%.0f / %.0f / %.0f frames/s end to end at B = 4 / 8 / 16**" % tuple(b)),
    ],
    "BASELINE.md": [
        (r"\*\*\d+ frames/s end to end\*\* \([\d.]+ ms per clip: encoders [\d.]+, FMT sampling [\d.]+, decode \+ hand-over to host [\d.]+\); `value_bf16` \d+; with the speech-emotion model in the step \d+; literal nfe = 50 grid \d+; from host inputs \d+",
         "**%s frames/s end to end** (%s ms per clip: encoders %.1f, FMT sampling %.1f, decode + hand-over to host %.1f); `value_bf16` %s; with the speech-emotion model in the step %s; literal nfe = 50 grid %s; from host inputs %s"
         % (fps("value"), ms("ms_per_step"), enc, st["fmt_sample"], st["decode_and_d2h"], fps("value_bf16"), fps("value_s2e"), fps("value_nfe50"), fps("value_from_host_inputs"))),
        (r"\*\*\d+ / \d+ / \d+ frames/s end to end at B = 4 / 8 / 16\*\*", "**%.0f / %.0f / %.0f frames/s end to end at B = 4 / 8 / 16**" % tuple(b)),
        (r"step-chain GEMM class \d+ GB/s = [\d.]+", "step-chain GEMM class %.0f GB/s = %.3f" % (r1["achieved"], r1["frac"])),
        (r"stacked-clip GEMMs \(16 clips\) \d+ TFLOP/s = [\d.]+", "stacked-clip GEMMs (16 clips) %.0f TFLOP/s = %.2f" % (rb["achieved"], rb["frac"])),
        (r"adaLN GEMM \(`fmt_gemm_big4_kernel`\) [\d ]+ TFLOP/s = [\d.]+", "adaLN GEMM (`fmt_gemm_big4_kernel`) %s TFLOP/s = %.2f" % ("{:,.0f}".format(r3["achieved"]).replace(",", " "), r3["frac"])),
    ],
}
for f, pairs in subs.items():
    s = open(f).read()
    for pat, rep in pairs:
        s2, n = re.subn(pat, lambda m: rep, s)
        print("%-12s %d x  %s" % (f, n, rep[:90].replace("\n", " ")))
        s = s2
    open(f, "w").write(s)
print("adaLN: %.0f us = %.0f TFLOP/s = %.3f (measured-peak %.3f of %.2f PFLOP/s); 16 clips %.1f ms = %.3f" % (
    r3["avg_launch_us"], r3["achieved"], r3["frac"], r3["frac_measured"], r3["peak_measured"] / 1e3,
    rb["adaln_gemm"]["avg_launch_us"] / 1e3, rb["adaln_gemm"]["frac"]))
