"""Diagnostic (library built with -DDEC_STAMPS into csrc/libfloat_hip.so): when do the copy workgroups of the last 512-px
flow launch run relative to its compute workgroups?"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, ".")
from tests.util import load_pkg
pkg = load_pkg()
size, T = 512, 64
dec = pkg.decoder.SynthesisHIP(pkg.weights.synth_decoder_state(size, seed=1), size, 512, "cuda:0", "fp16", max_frames=32)
dec.set_feats(pkg.weights.synth_feats(size, seed=1))
g = torch.Generator().manual_seed(0)
s_r, r_d = torch.randn(1, 512, generator=g).cuda(), (torch.randn(1, T, 512, generator=g) * 0.5).cuda()
host = torch.empty(T, size, size, 3, dtype=torch.float32, pin_memory=True)
staging = torch.empty(T, size, size, 3, dtype=torch.float32, device="cuda:0")
L = C.CDLL(pkg.native.LIB_PATH)
for rep in range(3):
    dec.decode_into_host(s_r, r_d, host, staging)
    out = (C.c_ulonglong * 4)()
    assert L.float_dec_debug_stamps(out) == 0
    c0, c1, k0, k1 = [v / 100.0 for v in out]  # us
    print("copy wgs: start %+.1f us, end %+.1f us | compute wgs: first start 0, last end %+.1f us" % (c0 - k0, c1 - k0, k1 - k0))
