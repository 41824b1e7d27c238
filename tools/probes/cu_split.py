"""FMT chain and decoder on CU-masked streams (hipExtStreamCreateWithCUMask): each stage alone on its share of the
CUs, then both at once (FMT of window k+1 beside the decode of window k).  SPLITS = CU counts given to the FMT."""
import os, sys, time
import torch
sys.path.insert(0, ".")
from tests.util import load_pkg
pkg = load_pkg()
N = pkg.native
cfg = pkg.config.FmtConfig()
dev = torch.device("cuda:0")
T, size = 250, 512
fmt_sd = pkg.weights.synth_fmt_state(cfg, seed=1)
dec_sd = pkg.weights.synth_decoder_state(size, seed=1)
hp = pkg.pipeline.FloatHotPath(fmt_sd, dec_sd, cfg, dev, size, "fp16", "fp16", 32, use_graph=int(os.environ.get("GRAPH", "2")))
hp.dec.set_feats(pkg.weights.synth_feats(size, seed=1))
cond = pkg.pipeline.synth_conditions(cfg, T, seed=0, device=dev)
noise = pkg.fmt.draw_noise(5, 1, cfg, seed=15).to(dev)
n_cu = torch.cuda.get_device_properties(dev).multi_processor_count


def timed(fn, n=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


def sample():
    return hp.sample(cond["r_s"], cond["wa"], cond["we"], 51, 2.0, 1.0, 1.0, noise=noise)


r_d = sample()
print("whole chip: fmt %.1f ms, decode %.1f ms, sequential %.1f ms" % (
    timed(sample), timed(lambda: hp.decode(cond["s_r"], None, r_d)),
    timed(lambda: hp.generate(cond["r_s"], cond["wa"], cond["we"], cond["s_r"], None, 51, noise=noise))), flush=True)
ref = hp.generate(cond["r_s"], cond["wa"], cond["we"], cond["s_r"], None, 51, noise=noise)
for split in [int(v) for v in os.environ.get("SPLITS", "64,128,192").split(",")]:
    with torch.cuda.device(dev):
        s_f = N.cu_range_stream(0, split, dev)
        s_d = N.cu_range_stream(split, n_cu, dev)
    cur = torch.cuda.current_stream(dev)

    def on(stream, fn):
        def run():
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                fn()
            cur.wait_stream(stream)
        return run
    t_f = timed(on(s_f, sample))
    t_d = timed(on(s_d, lambda: hp.decode(cond["s_r"], None, r_d)))
    hp.cu_split = split
    for a in ("_s_fmt", "_s_dec"):
        if hasattr(hp, a):
            delattr(hp, a)
    both = lambda: hp.generate(cond["r_s"], cond["wa"], cond["we"], cond["s_r"], None, 51, noise=noise, overlap=True)
    t_b = timed(both)
    same = bool(torch.equal(both(), ref))
    print("fmt on %3d CUs: %.1f ms | decode on %3d CUs: %.1f ms | overlapped pipeline %.1f ms (bitwise %s)" % (
        split, t_f, n_cu - split, t_d, t_b, same), flush=True)
