"""A/B of the fused transposed-conv + blur path against the separate kernels (run twice with FLOAT_DEC_ZBLUR_MIN set / unset)."""
import os, sys, torch
sys.path.insert(0, '.')
from tests.util import load_pkg
pkg = load_pkg()
sd = pkg.weights.synth_decoder_state(512, seed=1)
dec = pkg.decoder.SynthesisHIP(sd, 512, 512, "cuda:0", "fp16", max_frames=16)
dec.set_feats(pkg.weights.synth_feats(512, seed=1))
g = torch.Generator().manual_seed(0)
s_r, r_d = torch.randn(1, 512, generator=g), torch.randn(1, 3, 512, generator=g) * 0.5
out = dec.decode_latent_into_processed_images(s_r, r_d).cpu()
path = "/tmp/zb_ref.pt"
if os.environ.get("SAVE"):
    torch.save(out, path)
else:
    ref = torch.load(path)
    d = (out - ref).abs()
    print("max", float(d.max()), "mean", float(d.mean()))
    dm = d[0].amax(-1)
    rows = dm.amax(1); cols = dm.amax(0)
    print("rows with large err:", [int(i) for i in torch.nonzero(rows > 0.3 * d.max()).flatten()[:40]])
    print("cols with large err:", [int(i) for i in torch.nonzero(cols > 0.3 * d.max()).flatten()[:40]])
    print("mean err by row block of 28:", [round(float(d[0, i:i+28].mean()), 5) for i in range(0, 140, 28)])
    print("err at rows 0..6:", [round(float(d[0, i].mean()), 5) for i in range(7)])
    print("err at rows 26..30:", [round(float(d[0, i].mean()), 5) for i in range(26, 31)])
if not os.environ.get("SAVE"):
    big = (d > 1e-3)
    print("count > 1e-3:", int(big.sum()), "of", d.numel(), " > 1e-4:", int((d > 1e-4).sum()), " psnr vs ref: %.1f dB" % float(-10 * torch.log10(((out - ref) ** 2).mean())))
    idx = torch.nonzero(big)[:12]
    print("where:", idx.tolist())
