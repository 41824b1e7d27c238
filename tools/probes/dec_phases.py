"""Diagnostic (library built with -DDEC_PHASES, e.g. into build_ab/phases.so and swapped in by tools/ab_lib.sh): where does a
workgroup of dec_zblur_kernel / dec_conv16_kernel spend its wall time?  Sums of wave 0's 10-ns stamps per phase, per level."""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from tests.util import load_pkg
pkg = load_pkg()
L = C.CDLL(pkg.native.LIB_PATH)
if not hasattr(L, "float_dec_debug_phases"):
    sys.exit("library without -DDEC_PHASES")
size, T = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 32
dec = pkg.decoder.SynthesisHIP(pkg.weights.synth_decoder_state(size, seed=1), size, 512, "cuda:0", "fp16", max_frames=32)
dec.set_feats(pkg.weights.synth_feats(size, seed=1))
g = torch.Generator().manual_seed(0)
s_r, r_d = torch.randn(1, 512, generator=g), torch.randn(1, T, 512, generator=g) * 0.5
out = (C.c_ulonglong * 16)()
dec.decode_latent_into_processed_images(s_r, r_d)
L.float_dec_debug_phases(out, 1)
dec.decode_latent_into_processed_images(s_r, r_d)
L.float_dec_debug_phases(out, 1)
v = [x / 100.0 for x in out]  # us (counts stay counts * 0.01)
nz, nc = out[7], out[15]
print("zblur : %d tiles; per tile us: prologue+first issue %.2f | wait+commit %.2f | mfma %.2f | z write %.2f | filter+store %.2f | total %.2f"
      % (nz, v[0] / nz, v[1] / nz, v[2] / nz, v[3] / nz, v[4] / nz, sum(v[:5]) / nz))
print("conv16: %d tiles; per tile us: prologue %.2f (per wg) | wait+commit %.2f | issue+mfma %.2f | epilogue %.2f | total %.2f"
      % (nc, v[8] / nc, v[9] / nc, v[10] / nc, v[11] / nc, sum(v[8:12]) / nc))
