#!/usr/bin/env python3
"""Fill the @PLACEHOLDERS@ of DESIGN.md section 6 / README.md from profiles/r04_bench.json (the bench line of the profile round)."""
import json
import sys

d = json.load(open("profiles/r04_bench.json"))
T = d["config"]["frames_per_clip"]


def pair(fps):
    return "%.1f" % (T / fps * 1e3), "%.0f" % fps


rep = {}
for tag, key in (("", "value"), ("_S2E", "value_s2e"), ("_NFE", "value_nfe50"), ("_HOST", "value_from_host_inputs"), ("_BF", "value_bf16")):
    ms, fps = pair(d[key])
    rep["@MS%s@" % tag], rep["@FPS%s@" % tag] = ms, fps
rep["@MS_HOT@"], rep["@FPS_HOT@"] = "%.1f" % d["hot_path"]["ms_per_clip"], "%.0f" % d["hot_path"]["frames_per_s"]
rep["@HOSTMS@"] = "%.2f ms" % d["host_inputs_ms"]
rep["@MS_B4@"], rep["@FPS_B4@"] = "%.1f per 4 clips" % d["ms_per_step_batch4"], "%.0f" % d["value_batch4"]
for f in sys.argv[1:]:
    s = open(f).read()
    for k, v in rep.items():
        s = s.replace(k, v)
    open(f, "w").write(s)
print(rep)
