"""Throughput of float_fmt_sample_batch: B clips of 10 s (5 windows x 50 Euler evaluations, 3-way CFG) per launch chain."""
import os, sys, time
import torch
sys.path.insert(0, ".")
from tests.util import load_pkg
pkg = load_pkg()
cfg = pkg.config.FmtConfig()
sd = pkg.weights.synth_fmt_state(cfg, seed=1)
dt = os.environ.get("FMT_DTYPE", "fp16")
base = None
for B in [int(b) for b in os.environ.get("BATCHES", "1,2,4,8").split(",")]:
    fmt = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", dt, max_batch=B)
    T = 250
    cs = [pkg.pipeline.synth_conditions(cfg, T, seed=q, device="cuda:0") for q in range(B)]
    cat = lambda k: torch.cat([c[k] for c in cs])
    r_s, wa, we = cat("r_s"), cat("wa"), cat("we")
    noise = pkg.fmt.draw_noise(5, B, cfg, 15).cuda()
    for _ in range(2):
        r_d = fmt.sample(r_s, wa, we, noise, 51, 2.0, 1.0, 1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = int(os.environ.get("REPS", "3"))
    for _ in range(n):
        r_d = fmt.sample(r_s, wa, we, noise, 51, 2.0, 1.0, 1.0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    base = base or ms
    print("B=%d: %.2f ms per batch, %.2f ms per clip, throughput x%.2f vs B=1 (%.0f latent frames/s)" % (B, ms, ms / B, base * B / ms, B * T / ms * 1e3))
    fmt.close()
    del fmt
    torch.cuda.empty_cache()
