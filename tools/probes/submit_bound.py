"""Is the FMT chain bound by the GPU or by the host's submission of its ~15 000 graph nodes?  Put a long kernel in front, so the
host can run ahead of the GPU, and compare the host time inside the sampling call and the GPU time of the chain."""
import os, sys, time
import torch
sys.path.insert(0, ".")
from tests.util import load_pkg
pkg = load_pkg()
cfg = pkg.config.FmtConfig()
sd = pkg.weights.synth_fmt_state(cfg, seed=1)
fmt = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", "fp16", use_graph=int(os.environ.get("FMT_GRAPH", "2")))
cond = pkg.pipeline.synth_conditions(cfg, 250, seed=0, device="cuda:0")
noise = pkg.fmt.draw_noise(5, 1, cfg, 15).cuda()
run = lambda: fmt.sample(cond["r_s"], cond["wa"], cond["we"], noise, 51, 2.0, 1.0, 1.0)
for _ in range(2):
    run()
torch.cuda.synchronize()
big = torch.randn(8192, 8192, device="cuda:0")
def blocker(n):
    x = big
    for _ in range(n):
        x = x @ big  # ~1.1 TFLOP fp32 each: tens of ms
    return x
for n_block in (0, 3):
    for _ in range(2):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        blocker(n_block)
        e1.record()
        c0 = time.perf_counter()
        run()
        host = (time.perf_counter() - c0) * 1e3
        e2.record()
        torch.cuda.synchronize()
        print("blocker %d: blocker GPU %.1f ms, host inside sample %.1f ms, chain GPU (after the blocker) %.1f ms" % (
            n_block, e0.elapsed_time(e1), host, e1.elapsed_time(e2)), flush=True)
