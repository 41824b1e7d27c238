"""Time of the once-per-clip host-side stage (PyTorch-ROCm): appearance encoder + Direction +
wav2vec2-base audio encoder (random weights, full-size config) for a 10 s clip."""
import sys, time, torch
sys.path.insert(0, '.')
from tests.util import load_pkg
pkg = load_pkg()
hm = pkg.host_models
dev = "cuda:0"
enc_sd = {k: v.to(dev) for k, v in pkg.weights.synth_encoder_state(512, seed=1).items()}
dec_sd = pkg.weights.synth_decoder_state(512, seed=1)
q = hm.direction_basis(dec_sd, dev)
aud = hm.AudioEncoderHost().to(dev)   # default Wav2Vec2Config() = wav2vec2-base
img = torch.rand(1, 3, 512, 512, device=dev) * 2 - 1
a = torch.randn(1, 160000, device=dev)
def run():
    s_r, feats, lam = hm.encode_appearance(enc_sd, img)
    r_s = hm.direction(q, lam)
    wa = aud.inference(a, seq_len=250)
    return s_r, wa
with torch.no_grad():
    for _ in range(2): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): run()
    torch.cuda.synchronize()
print("conditioning (encoder + direction + wav2vec2-base, 10 s): %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
