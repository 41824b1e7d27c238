import os, sys, time, torch
sys.path.insert(0, ".")
from tests.util import load_pkg
pkg = load_pkg()
cfg = pkg.config.FmtConfig()
sd = pkg.weights.synth_fmt_state(cfg, seed=1)
fmt = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", "fp16")
cond = pkg.pipeline.synth_conditions(cfg, 250, seed=0, device="cuda:0")
noise = pkg.fmt.draw_noise(5, 1, cfg, 15).cuda()
run = lambda: fmt.sample(cond["r_s"], cond["wa"], cond["we"], noise, 51, 2.0, 1.0, 1.0)
def timed(tag):
    for _ in range(2): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): run()
    torch.cuda.synchronize(); print("%-28s %.2f ms per clip" % (tag, (time.perf_counter() - t0) * 250), flush=True)
timed("null stream")
for pr in (0, -1):
    s = torch.cuda.Stream(priority=pr)
    with torch.cuda.stream(s):
        timed("created stream, priority %d" % pr)
timed("null stream again")
