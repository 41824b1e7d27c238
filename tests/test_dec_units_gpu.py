"""One decoder op at a time through the production kernels (float_dec_debug_styled_conv / float_dec_debug_flow_level) against
the reference modules' own outputs (tests/golden/dec_units_hip.npz, made by tools/make_goldens.py::gen_dec_units_hip from
StyledConv / ToFlow / ToRGB of /root/reference styledecoder.py:302-425).  One case per kernel route.
Tolerances: fp32 verification mode max |d| <= 2e-5 (outputs are O(1)); fp16 rel-L2 <= 2e-3."""
import pytest
import torch

from tests.util import golden, load_pkg, max_abs, rel_l2, seeded_normal as rnd

pkg = load_pkg()
D = pkg.decoder
pytestmark = pytest.mark.gpu
LIM = {"fp32": dict(max=2e-5, rel=2e-6), "fp16": dict(max=3e-2, rel=2e-3)}


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_styled_conv_routes(dtype):
    g = golden("dec_units_hip")
    seed = g["seed"]
    style = rnd(seed + 1, 2, 512)
    names = ["plain8", "plain32", "up4", "up8", "up32"]
    for i, (name, (cin, cout, R, up)) in enumerate(zip(names, g["sc_cases"].tolist())):
        k = seed + 100 * (i + 1)
        sd = {"conv.weight": rnd(k + 2, 1, cout, cin, 3, 3), "conv.modulation.weight": rnd(k + 3, cin, 512),
              "conv.modulation.bias": 1 + rnd(k + 4, cin, std=0.1), "activate.bias": rnd(k + 5, 1, cout, 1, 1, std=0.1)}
        x = rnd(k + 6, 2, cin, R, R)
        out, sat = D.debug_styled_conv(sd, x, style, upsample=bool(up), dtype=dtype)
        want = g["sc_%s_out" % name]
        m, r = max_abs(out.cpu(), want), rel_l2(out.cpu(), want)
        print("%s styled_conv %-8s max|d| %.2e rel %.2e" % (dtype, name, m, r))
        assert out.shape == want.shape and sat == 0
        assert m <= LIM[dtype]["max"] and r <= LIM[dtype]["rel"], name


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_flow_level(dtype):
    g = golden("dec_units_hip")
    seed = g["seed"]
    style = rnd(seed + 1, 2, 512)
    for j, (name, (C, R, prev)) in enumerate(zip(["c32", "c128"], g["fl_cases"].tolist())):
        k = seed + 1000 * (j + 1)
        sd = {"to_flow.bias": rnd(k + 6, 1, 3, 1, 1, std=0.1), "to_flow.conv.weight": rnd(k + 7, 1, 3, C, 1, 1, std=0.3),
              "to_flow.conv.modulation.weight": rnd(k + 8, C, 512), "to_flow.conv.modulation.bias": 1 + rnd(k + 9, C, std=0.1),
              "to_rgb.bias": rnd(k + 13, 1, 3, 1, 1, std=0.1), "to_rgb.conv.0.weight": rnd(k + 14, 3, C, 1, 1),
              "to_rgb.conv.1.bias": rnd(k + 15, 1, 3, 1, 1, std=0.1)}
        x, feat = rnd(k + 10, 2, C, R, R), rnd(k + 11, 1, C, R, R)
        pflow = rnd(k + 12, 2, 3, R // 2, R // 2, std=0.5) if prev else None
        prgb = rnd(k + 16, 2, 3, R // 2, R // 2) if prev else None
        of, ob, org = D.debug_flow_level(sd, x, feat, style, pflow, prgb, dtype=dtype)
        for what, got in (("out", of), ("blend", ob), ("rgb", org)):
            want = g["fl_%s_%s" % (name, what)]
            m, r = max_abs(got.cpu(), want), rel_l2(got.cpu(), want)
            print("%s flow level %-5s %-5s max|d| %.2e rel %.2e" % (dtype, name, what, m, r))
            # the warp is a bilinear gather of white noise positioned by the flow: fp32 rounding of the position moves it
            lim_m = LIM[dtype]["max"] * (5 if what != "out" else 1)
            assert m <= lim_m and r <= LIM[dtype]["rel"] * (5 if what != "out" else 1), (name, what)


def test_unit_op_argument_errors():
    sd = {"conv.weight": torch.zeros(1, 16, 32, 3, 3), "conv.modulation.weight": torch.zeros(32, 512),
          "conv.modulation.bias": torch.zeros(32), "activate.bias": torch.zeros(1, 16, 1, 1)}
    with pytest.raises(ValueError, match="multiples of 32"):
        D.debug_styled_conv(sd, torch.zeros(1, 32, 8, 8), torch.zeros(1, 512))
    with pytest.raises(ValueError):
        D.debug_styled_conv(sd, torch.zeros(1, 32, 8, 8), torch.zeros(1, 512), dtype="bf16")


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_up_conv_blur_kernels(dtype):
    """StyledConv(upsample=True, blur_kernel=k) of the reference for a symmetric and an asymmetric 4-tap kernel
    (tests/golden/dec_blur.npz) through the three up-conv routes (per-class conv + dec_blur_kernel, dec_zconv4_kernel +
    dec_blur_kernel, dec_zblur_kernel).  The asymmetric kernel pins the orientation (upfirdn2d convolves: styledecoder.py:28-29).
    Same limits as the [1,3,3,1] routes; the kernel arrives once as the loader's widget value and once as the checkpoint's
    `conv.blur.kernel` buffer (which wins over a contradicting widget, like the reference's strict load)."""
    g = golden("dec_blur")
    seed = g["seed"]
    style = rnd(seed + 1, 2, 512)
    for ki, bk in enumerate(g["kernels"].tolist()):
        for i, (name, (cin, cout, R, F)) in enumerate(zip(["up4", "up8", "up32"], g["sc_cases"].tolist())):
            k = seed + 100 * (i + 1)
            sd = {"conv.weight": rnd(k + 2, 1, cout, cin, 3, 3), "conv.modulation.weight": rnd(k + 3, cin, 512),
                  "conv.modulation.bias": 1 + rnd(k + 4, cin, std=0.1), "activate.bias": rnd(k + 5, 1, cout, 1, 1, std=0.1)}
            x = rnd(k + 6, F, cin, R, R)
            want = g["sc_k%d_%s_out" % (ki, name)]
            out, sat = D.debug_styled_conv(sd, x, style[:F], upsample=True, dtype=dtype, blur_kernel=bk)
            m, r = max_abs(out.cpu(), want), rel_l2(out.cpu(), want)
            print("%s up-conv %-5s kernel %s max|d| %.2e rel %.2e" % (dtype, name, bk, m, r))
            assert sat == 0 and m <= LIM[dtype]["max"] and r <= LIM[dtype]["rel"], (name, bk)
            k1 = torch.tensor(bk, dtype=torch.float32)
            sd["conv.blur.kernel"] = k1[:, None] * k1[None, :] / k1.sum() ** 2 * 4  # make_kernel(k) * 4 (styledecoder.py:39-44,118)
            out2, _ = D.debug_styled_conv(sd, x, style[:F], upsample=True, dtype=dtype, blur_kernel=[1, 3, 3, 1])
            assert torch.equal(out2, out), "the checkpoint's buffer decides"


def test_blur_kernel_refusals():
    sd = {"conv.weight": torch.zeros(1, 32, 32, 3, 3), "conv.modulation.weight": torch.zeros(32, 512),
          "conv.modulation.bias": torch.zeros(32), "activate.bias": torch.zeros(1, 32, 1, 1)}
    x, st = torch.zeros(1, 32, 8, 8), torch.zeros(1, 512)
    with pytest.raises(ValueError, match="4-tap"):
        D.debug_styled_conv(sd, x, st, upsample=True, blur_kernel=[1, 1])
    with pytest.raises(ValueError, match="sums to zero"):
        D.debug_styled_conv(sd, x, st, upsample=True, blur_kernel=[1, -1, 1, -1])
    bad = dict(sd)
    bad["conv.blur.kernel"] = torch.eye(4)  # not an outer product k (x) k
    with pytest.raises(ValueError, match="outer product"):
        D.debug_styled_conv(bad, x, st, upsample=True)
    bad["conv.blur.kernel"] = torch.ones(6, 6) / 9
    with pytest.raises(ValueError, match="4 x 4"):
        D.debug_styled_conv(bad, x, st, upsample=True)
