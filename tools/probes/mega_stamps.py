"""Per-stage timeline of the persistent evaluation kernel (library built with EXTRA=-DMEGA_STAMPS): for workgroup
FLOAT_FMT_MEGA_STAMP_WG (default 0), per stage: body time, store drain, barrier wait - from s_memrealtime stamps of the LAST launch."""
import ctypes as C
import os
import sys

import torch

os.environ.setdefault("FLOAT_FMT_MEGA", "1")  # the persistent kernel is opt-in
sys.path.insert(0, ".")
from tests.util import load_pkg  # noqa: E402

pkg = load_pkg()
N = pkg.native
cfg = pkg.config.FmtConfig()
fmt = pkg.fmt.FlowMatchingTransformerHIP(pkg.weights.synth_fmt_state(cfg, seed=1), cfg, "cuda:0", "fp16", use_graph=2)
cond = pkg.pipeline.synth_conditions(cfg, 50, seed=0, device="cuda:0")
noise = pkg.fmt.draw_noise(1, 1, cfg, 15).cuda()
for _ in range(3):
    fmt.sample(cond["r_s"], cond["wa"], cond["we"], noise, 51, 2.0, 1.0, 1.0)
torch.cuda.synchronize()
out = torch.zeros(64 * 3, device="cuda:0")
N.check(N.lib().float_fmt_debug(fmt._h, 2, None, N.dev_ptr(out), N.stream_ptr("cuda:0")))
t = out.cpu().reshape(64, 3)
names = ["xembed"] + ["ln1", "qkv", "attn", "proj", "ln2", "fc1", "fc2"] * cfg.fmt_depth + ["lnF", "head"]
agg = {}
for s, nm in enumerate(names):
    if s + 1 >= len(names):
        nxt = None
    else:
        nxt = float(t[s + 1, 0])
    body, drain = float(t[s, 1] - t[s, 0]), float(t[s, 2] - t[s, 1])
    bar = (nxt - float(t[s, 2])) if nxt is not None else 0.0
    a = agg.setdefault(nm, [0, 0.0, 0.0, 0.0])
    a[0] += 1
    a[1] += body
    a[2] += drain
    a[3] += bar
print("workgroup %s; total %.1f us" % (os.environ.get("FLOAT_FMT_MEGA_STAMP_WG", "0"), float(t[len(names) - 1, 1])))
print("stage   n   body  drain  barrier  (us, mean)")
for nm, (n, b, d, w) in agg.items():
    print("%-7s %2d %6.2f %6.2f %6.2f" % (nm, n, b / n, d / n, w / n))
