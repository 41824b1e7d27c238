import sys, time, torch
sys.path.insert(0, '.')
from tests.util import load_pkg
pkg = load_pkg()
sd = pkg.weights.synth_decoder_state(512, seed=1)
dec = pkg.decoder.SynthesisHIP(sd, 512, 512, "cuda:0", "fp16", max_frames=int(__import__("os").environ.get("DEC_MAXF", "16")))
dec.set_feats(pkg.weights.synth_feats(512, seed=1))
g = torch.Generator().manual_seed(0)
s_r, r_d = torch.randn(1, 512, generator=g), torch.randn(1, 250, 512, generator=g) * 0.5
for _ in range(2): out = dec.decode_latent_into_processed_images(s_r, r_d)
torch.cuda.synchronize(); t0 = time.perf_counter()
REPS = int(__import__("os").environ.get("REPS", "3"))
for _ in range(REPS): out = dec.decode_latent_into_processed_images(s_r, r_d)
torch.cuda.synchronize(); print("decode 250 frames: %.2f ms" % ((time.perf_counter() - t0) / REPS * 1e3), float(out.mean()),
                                "sha", __import__("hashlib").sha1(out[::7].contiguous().cpu().numpy().tobytes()).hexdigest()[:16], "sat", dec.saturation())
