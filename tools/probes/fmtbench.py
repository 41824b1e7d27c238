"""FMT-only timing probe: one 10 s clip (5 windows x 50 Euler evaluations, 3-way CFG) through
float_fmt_sample.  Tuning switches come from the environment (FLOAT_FMT_*), so A/B runs are separate
processes:  FLOAT_FMT_FC2_SPLIT=0 python tools/probes/fmtbench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from tests.util import load_pkg  # noqa: E402

pkg = load_pkg()
cfg = pkg.config.FmtConfig()
sd = pkg.weights.synth_fmt_state(cfg, seed=1)
dt = os.environ.get("FMT_DTYPE", "bf16")
fmt = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", dt, use_graph=int(os.environ.get("FMT_GRAPH", "2")))
T = 250
cond = pkg.pipeline.synth_conditions(cfg, T, seed=0, device="cuda:0")
noise = pkg.fmt.draw_noise(5, 1, cfg, 15).cuda()
for _ in range(2):
    r_d = fmt.sample(cond["r_s"], cond["wa"], cond["we"], noise, 51, 2.0, 1.0, 1.0)
torch.cuda.synchronize()
n = int(os.environ.get("FMT_REPS", "3"))
each = []
for _ in range(n):
    t0 = time.perf_counter()
    r_d = fmt.sample(cond["r_s"], cond["wa"], cond["we"], noise, 51, 2.0, 1.0, 1.0)
    torch.cuda.synchronize()
    each.append((time.perf_counter() - t0) * 1e3)
ms = sum(each) / n
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("FLOAT_FMT"))
print("fmt sample 250 frames: %.2f ms (%.1f us/eval) min %.2f  mean %.6f absmean %.6f  [%s]"
      % (ms, ms * 1e3 / 250, min(each), float(r_d.mean()), float(r_d.abs().mean()), tag))
if os.environ.get("FMT_SAVE"):
    torch.save(r_d.cpu(), os.environ["FMT_SAVE"])
if os.environ.get("FMT_CMP") and os.path.exists(os.environ["FMT_CMP"]):
    ref = torch.load(os.environ["FMT_CMP"])
    print("   rel-L2 vs %s: %.3e" % (os.environ["FMT_CMP"], float((r_d.cpu() - ref).norm() / ref.norm())))
