import sys, time, torch
sys.path.insert(0, '.')
from tests.util import load_pkg
pkg = load_pkg()
sd = pkg.weights.synth_decoder_state(512, seed=1)
dec = pkg.decoder.SynthesisHIP(sd, 512, 512, "cuda:0", "fp16", max_frames=int(__import__("os").environ.get("DEC_MAXF", "32")))
dec.set_feats(pkg.weights.synth_feats(512, seed=1))
g = torch.Generator().manual_seed(0)
s_r, r_d = torch.randn(1, 512, generator=g), torch.randn(1, 250, 512, generator=g) * 0.5
host = torch.empty(250, 512, 512, 3).pin_memory()
st = None
for _ in range(2): st = dec.decode_into_host(s_r, r_d, host, st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): st = dec.decode_into_host(s_r, r_d, host, st)
torch.cuda.synchronize(); print("decode + hand-over, 250 frames: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
