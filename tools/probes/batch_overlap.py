"""Do a stacked-clip FMT chain (stream B) and the decodes of the previous half-batch (stream A) overlap?  Times, for B clips:
the chain alone, B decodes alone, both one after the other on one stream, both at once on two streams."""
import os, sys, time
import torch
sys.path.insert(0, ".")
from tests.util import load_pkg
pkg = load_pkg()
cfg = pkg.config.FmtConfig()
B = int(os.environ.get("B", "8"))
dev = "cuda:0"
fmt = pkg.fmt.FlowMatchingTransformerHIP(pkg.weights.synth_fmt_state(cfg, seed=1), cfg, dev, "fp16", max_batch=B)
dec = pkg.decoder.SynthesisHIP(pkg.weights.synth_decoder_state(512, seed=1), 512, 512, dev, "fp16", max_frames=32)
dec.set_feats(pkg.weights.synth_feats(512, seed=1))
T = 250
cs = [pkg.pipeline.synth_conditions(cfg, T, seed=q, device=dev) for q in range(B)]
cat = lambda k: torch.cat([c[k] for c in cs])
r_s, wa, we = cat("r_s"), cat("wa"), cat("we")
noise = pkg.fmt.draw_noise(5, B, cfg, 15).cuda()
g = torch.Generator().manual_seed(0)
s_r, r_dd = torch.randn(1, 512, generator=g).cuda(), (torch.randn(1, T, 512, generator=g) * 0.5).cuda()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def chain():
    return fmt.sample(r_s, wa, we, noise, 51, 2.0, 1.0, 1.0)

def decodes():
    for _ in range(B):
        out = dec.decode_latent_into_processed_images(s_r, r_dd)
    return out

def timed(f, n=2):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / n

def both_seq():
    chain(); decodes()

def both_par():
    with torch.cuda.stream(sa):
        decodes()
    with torch.cuda.stream(sb):
        chain()

tc, td, ts, tp = timed(chain), timed(decodes), timed(both_seq), timed(both_par)
print("B=%d: chain %.1f ms, decodes %.1f ms, one stream %.1f ms, two streams %.1f ms (max %.1f, sum %.1f)" % (B, tc, td, ts, tp, max(tc, td), tc + td))
