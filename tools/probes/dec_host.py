"""Decode 250 frames into pinned host memory (float_dec_frames_host), timing only; FLOAT_DEC_COPY / FLOAT_DEC_RIDE_* from the env."""
import os, sys, time
import torch
sys.path.insert(0, ".")
from tests.util import load_pkg
pkg = load_pkg()
size, T = 512, int(os.environ.get("FRAMES", "250"))
dec = pkg.decoder.SynthesisHIP(pkg.weights.synth_decoder_state(size, seed=1), size, 512, "cuda:0", "fp16", max_frames=32)
dec.set_feats(pkg.weights.synth_feats(size, seed=1))
g = torch.Generator().manual_seed(0)
s_r, r_d = torch.randn(1, 512, generator=g).cuda(), (torch.randn(1, T, 512, generator=g) * 0.5).cuda()
host = torch.empty(T, size, size, 3, dtype=torch.float32, pin_memory=True)
staging = torch.empty(T, size, size, 3, dtype=torch.float32, device="cuda:0")
mode = os.environ.get("MODE", "host")
def run():
    if mode == "host":
        dec.decode_into_host(s_r, r_d, host, staging)
    else:
        dec.decode_latent_into_processed_images(s_r, r_d)
for _ in range(2):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = int(os.environ.get("REPS", "5"))
for _ in range(n):
    run()
torch.cuda.synchronize()
print("%s: %.2f ms per %d frames" % (mode, (time.perf_counter() - t0) * 1e3 / n, T))
if os.environ.get("STAMPS"):
    import ctypes as C
    L = C.CDLL(os.path.join("comfyui-float_optimized_amd", "csrc", "libfloat_hip_stamps.so"))
