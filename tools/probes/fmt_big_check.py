"""B stacked clips through float_fmt_sample_batch against the same clips one by one (rel-L2 per clip), bitwise run to run, and the
time per window - the check used for tile experiments on the stacked-clip step chain (FLOAT_FMT_RB_QKV / _PROJ / _FC1 / _FC2 =
"shape[,ksplit]" select the tiles).  B=13 T=50 python3 tools/probes/fmt_big_check.py"""
import os, sys, time
import torch
sys.path.insert(0, ".")
from tests.util import load_pkg
pkg = load_pkg()
cfg = pkg.config.FmtConfig()
sd = pkg.weights.synth_fmt_state(cfg, seed=1)
B = int(os.environ.get("B", "13"))
T = int(os.environ.get("T", "50"))
fmt = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", os.environ.get("FMT_DTYPE", "fp16"), max_batch=B)
one = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", os.environ.get("FMT_DTYPE", "fp16"), max_batch=1)
cs = [pkg.pipeline.synth_conditions(cfg, T, seed=q, device="cuda:0") for q in range(B)]
cat = lambda k: torch.cat([c[k] for c in cs])
nw = (T + 49) // 50
noise = pkg.fmt.draw_noise(nw, B, cfg, 15).cuda()
r = fmt.sample(cat("r_s"), cat("wa"), cat("we"), noise, 11, 2.0, 1.0, 1.0)
r2 = fmt.sample(cat("r_s"), cat("wa"), cat("we"), noise, 11, 2.0, 1.0, 1.0)
worst = 0.0
for q in (0, 1, B // 2, B - 1):
    c = cs[q]
    rq = one.sample(c["r_s"], c["wa"], c["we"], noise[:, q:q + 1].contiguous(), 11, 2.0, 1.0, 1.0)
    e = float((r[q] - rq[0]).norm() / rq[0].norm())
    worst = max(worst, e)
print("B=%d (%d rows): stacked vs one by one rel-L2 max %.2e; bitwise run to run: %s; finite: %s; saturation %d" % (
    B, B * 180, worst, torch.equal(r, r2), bool(torch.isfinite(r).all()), fmt.saturation()))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): fmt.sample(cat("r_s"), cat("wa"), cat("we"), noise, 11, 2.0, 1.0, 1.0)
torch.cuda.synchronize(); print("%.2f ms per window of 10 evaluations" % ((time.perf_counter() - t0) / 3 / nw * 1e3))
