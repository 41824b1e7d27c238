import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from tests.util import load_pkg
pkg = load_pkg()
esd = pkg.weights.synth_encoder_state(512, seed=1)
dsd = pkg.weights.synth_decoder_state(512, seed=1)
enc = pkg.encoder.EncoderHIP(esd, 512, 512, 20, "cuda:0", "fp16", direction_weight=dsd["direction.weight"])
img = (torch.from_numpy(np.random.RandomState(0).rand(1, 3, 512, 512).astype("float32")) * 2 - 1).cuda()
for _ in range(3): out = enc.encode_image_into_latent(img, want_feats=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): out = enc.encode_image_into_latent(img, want_feats=False)
torch.cuda.synchronize(); print("encoder: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3), float(out[0].mean()))
