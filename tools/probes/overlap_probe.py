"""Stage overlap with a decoder whose launches are kept small (FLOAT_DEC_TPW / FLOAT_DEC_FLOW_WGS): sequential vs
FMT(window k+1) beside decode(window k) on two streams."""
import os, sys, time
import torch
sys.path.insert(0, ".")
from tests.util import load_pkg
pkg = load_pkg()
cfg = pkg.config.FmtConfig()
dev = torch.device("cuda:0")
T, size = 250, 512
hp = pkg.pipeline.FloatHotPath(pkg.weights.synth_fmt_state(cfg, seed=1), pkg.weights.synth_decoder_state(size, seed=1), cfg, dev, size,
                               "fp16", "fp16", int(os.environ.get("DEC_MAXF", "32")))
hp.dec.set_feats(pkg.weights.synth_feats(size, seed=1))
cond = pkg.pipeline.synth_conditions(cfg, T, seed=0, device=dev)
noise = pkg.fmt.draw_noise(5, 1, cfg, seed=15).to(dev)


def timed(fn, n=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


r_d = hp.sample(cond["r_s"], cond["wa"], cond["we"], 51, 2.0, 1.0, 1.0, noise=noise)
args = (cond["r_s"], cond["wa"], cond["we"], cond["s_r"], None, 51)
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("FLOAT_DEC"))
print("fmt %.1f | decode %.1f | sequential %.1f | overlapped %.1f ms   [%s]" % (
    timed(lambda: hp.sample(cond["r_s"], cond["wa"], cond["we"], 51, 2.0, 1.0, 1.0, noise=noise)),
    timed(lambda: hp.decode(cond["s_r"], None, r_d)),
    timed(lambda: hp.generate(*args, noise=noise)),
    timed(lambda: hp.generate(*args, noise=noise, overlap=True)), tag), flush=True)
