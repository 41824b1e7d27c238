"""Diagnostic (library built with -DDEC_PHASES swapped in): per-phase wall time of a workgroup of dec_zblur_kernel /
dec_conv16_kernel, one StyledConv at a time through float_dec_debug_styled_conv (32 frames)."""
import ctypes as C, sys, time
import torch
sys.path.insert(0, ".")
from tests.util import load_pkg, seeded_normal as rnd
pkg = load_pkg()
D = pkg.decoder
L = C.CDLL(pkg.native.LIB_PATH)
has = hasattr(L, "float_dec_debug_phases")
out = (C.c_ulonglong * 16)()
F = 32
style = rnd(1, F, 512)
for (cin, cout, R, up) in [(64, 32, 256, 1), (32, 32, 512, 0), (128, 64, 128, 1), (64, 64, 256, 0), (256, 128, 64, 1), (128, 128, 128, 0),
                           (512, 256, 32, 1), (256, 256, 64, 0)]:
    sd = {"conv.weight": rnd(2, 1, cout, cin, 3, 3), "conv.modulation.weight": rnd(3, cin, 512), "conv.modulation.bias": 1 + rnd(4, cin, std=0.1),
          "activate.bias": rnd(5, 1, cout, 1, 1, std=0.1)}
    x = rnd(6, 1, cin, R, R).expand(F, cin, R, R)
    D.debug_styled_conv(sd, x, style, upsample=bool(up))
    if has:
        L.float_dec_debug_phases(out, 1)
    D.debug_styled_conv(sd, x, style, upsample=bool(up))
    if not has:
        continue
    L.float_dec_debug_phases(out, 1)
    v = [t / 100.0 for t in out]
    nz, nc = out[7], out[15]
    tag = "%3d->%3d %3d px %s" % (cin, cout, R, "up  " if up else "conv")
    if nz:
        print(tag, "zblur : %6d tiles; us per tile: prologue %.2f | wait+commit %.2f | mfma %.2f | z write %.2f | filter+store %.2f | total %.2f"
              % (nz, v[0] / nz, v[1] / nz, v[2] / nz, v[3] / nz, v[4] / nz, sum(v[:5]) / nz))
    if nc:
        print(tag, "conv16: %6d tiles; us per tile: prologue/wg %.2f | wait+commit %.2f | issue+mfma %.2f | epilogue %.2f | total %.2f"
              % (nc, v[8] / nc, v[9] / nc, v[10] / nc, v[11] / nc, sum(v[8:12]) / nc))
