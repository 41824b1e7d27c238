"""N independent FMT chains on N HIP streams (one handle each): do latency-bound chains overlap on one GPU?"""
import os, sys, time
import torch
sys.path.insert(0, ".")
from tests.util import load_pkg
pkg = load_pkg()
cfg = pkg.config.FmtConfig()
sd = pkg.weights.synth_fmt_state(cfg, seed=1)
T = 250
for N in [int(v) for v in os.environ.get("STREAMS", "1,2,3,4").split(",")]:
    fmts = [pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", "fp16") for _ in range(N)]
    streams = [torch.cuda.Stream() for _ in range(N)]
    conds = [pkg.pipeline.synth_conditions(cfg, T, seed=q, device="cuda:0") for q in range(N)]
    noise = pkg.fmt.draw_noise(5, 1, cfg, 15).cuda()
    def run():
        outs = []
        for q in range(N):
            with torch.cuda.stream(streams[q]):
                outs.append(fmts[q].sample(conds[q]["r_s"], conds[q]["wa"], conds[q]["we"], noise, 51, 2.0, 1.0, 1.0))
        return outs
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        outs = run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    ref = fmts[0].sample(conds[N - 1]["r_s"], conds[N - 1]["wa"], conds[N - 1]["we"], noise, 51, 2.0, 1.0, 1.0)
    torch.cuda.synchronize()
    print("%d chains on %d streams: %.2f ms per round, %.2f ms per clip (x%.2f vs 85.0); last chain == single-stream result: %s" % (
        N, N, ms, ms / N, 85.0 * N / ms, bool(torch.equal(ref, outs[-1]))))
    for f in fmts:
        f.close()
    del fmts
    torch.cuda.empty_cache()
