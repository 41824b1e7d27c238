"""The tuned launch variants against the plain ones: the tuning switches are read once per process (static), so each variant runs
in its own child process and the results are compared here.
  * FMT: weight touch (FLOAT_FMT_TOUCH), the token-blocked head GEMM (FLOAT_FMT_NO_TOKBLK) and the hoisting of the adaLN
    projection out of the Euler step (FLOAT_FMT_HOIST, FLOAT_FMT_ZGROUP) only change WHERE and WHEN work is done - every output element is produced by the same arithmetic in the same order, so r_d must be bitwise identical.
  * decoder: the fused transposed-conv + blur kernel (FLOAT_DEC_ZBLUR_MIN) filters the same fp16-rounded z values as the separate
    kernels; its horizontal filter sums are formed by the matrix pipe, the separate kernel's by packed fp32 FMAs, so frames agree
    to ~80 dB, not bitwise.  The 32- and 64-channel tiles of the 3x3 conv (FLOAT_DEC_CONV_BN) accumulate every output in the
    same order: bitwise.
  * decoder grid order (FLOAT_DEC_CB_ORDER): bitwise."""
import os
import subprocess
import sys

import pytest
import torch

from .util import ROOT

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys, torch
sys.path.insert(0, %(root)r)
from tests.util import load_pkg
pkg = load_pkg()
what, out = sys.argv[1], sys.argv[2]
if what == "fmt":
    cfg = pkg.config.FmtConfig()
    sd = pkg.weights.synth_fmt_state(cfg, seed=3)
    fmt = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", "bf16")
    cond = pkg.pipeline.synth_conditions(cfg, 75, seed=2, device="cuda:0")
    noise = pkg.fmt.draw_noise(2, 1, cfg, 15).cuda()
    r3 = fmt.sample(cond["r_s"], cond["wa"], cond["we"], noise, 6, 2.0, 1.0, 1.0).cpu()
    r4 = fmt.sample(cond["r_s"], cond["wa"], cond["we"], noise, 6, 2.0, 1.5, 1.0, include_r_cfg=True).cpu()
    torch.save({"r3": r3, "r4": r4}, out)
elif what == "fmtbig":
    # 10 evaluations per window: 1 800 (3-way CFG) / 2 400 (4-way) dense rows - the persistent adaLN projection kernel's range
    cfg = pkg.config.FmtConfig()
    sd = pkg.weights.synth_fmt_state(cfg, seed=3)
    dt = os.environ.get("CHILD_DTYPE", "fp16")
    fmt = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", dt, max_batch=2)
    cs = [pkg.pipeline.synth_conditions(cfg, 75, seed=q, device="cuda:0") for q in range(2)]
    cat = lambda k: torch.cat([c[k] for c in cs])
    n1, n2 = pkg.fmt.draw_noise(2, 1, cfg, 15).cuda(), pkg.fmt.draw_noise(2, 2, cfg, 15).cuda()
    c = cs[0]
    r3 = fmt.sample(c["r_s"], c["wa"], c["we"], n1, 11, 2.0, 1.0, 1.0).cpu()
    r4 = fmt.sample(c["r_s"], c["wa"], c["we"], n1, 11, 2.0, 1.5, 1.0, include_r_cfg=True).cpu()
    rb = fmt.sample(cat("r_s"), cat("wa"), cat("we"), n2, 11, 2.0, 1.0, 1.0).cpu()
    torch.save({"r3": r3, "r4": r4, "rb": rb}, out)
elif what == "fmtbatch":
    cfg = pkg.config.FmtConfig()
    sd = pkg.weights.synth_fmt_state(cfg, seed=3)
    fmt = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", "fp16", max_batch=2)
    cs = [pkg.pipeline.synth_conditions(cfg, 75, seed=q, device="cuda:0") for q in range(2)]
    cat = lambda k: torch.cat([c[k] for c in cs])
    noise = pkg.fmt.draw_noise(2, 2, cfg, 15).cuda()
    torch.save({"r": fmt.sample(cat("r_s"), cat("wa"), cat("we"), noise, 6, 2.0, 1.0, 1.0).cpu()}, out)
else:
    sd = pkg.weights.synth_decoder_state(512, seed=1)
    dec = pkg.decoder.SynthesisHIP(sd, 512, 512, "cuda:0", "fp16", max_frames=4)
    dec.set_feats(pkg.weights.synth_feats(512, seed=1))
    g = torch.Generator().manual_seed(0)
    s_r, r_d = torch.randn(1, 512, generator=g), torch.randn(1, 5, 512, generator=g) * 0.5
    torch.save({"frames": dec.decode_latent_into_processed_images(s_r, r_d).cpu()}, out)
'''


def run_child(tmp_path, what, tag, env):
    script = tmp_path / "child.py"
    script.write_text(CHILD % {"root": ROOT})
    out = tmp_path / ("%s_%s.pt" % (what, tag))
    e = dict(os.environ)
    e.update(env)
    subprocess.run([sys.executable, str(script), what, str(out)], check=True, env=e, cwd=ROOT, timeout=600)
    return torch.load(out)


def test_fmt_touch_and_token_blocked_head_are_bitwise_neutral(tmp_path):
    base = run_child(tmp_path, "fmt", "plain", {"FLOAT_FMT_TOUCH": "0", "FLOAT_FMT_NO_TOKBLK": "1"})
    tuned = run_child(tmp_path, "fmt", "tuned", {})
    every = run_child(tmp_path, "fmt", "every", {"FLOAT_FMT_TOUCH": "255"})
    # the adaLN projection per evaluation (one launch per step) instead of once per window (one batched launch): the same
    # kernel on the same operands; and other XCD groupings of the batched launch
    per_step = run_child(tmp_path, "fmt", "perstep", {"FLOAT_FMT_HOIST": "0"})
    grouped = run_child(tmp_path, "fmt", "zgroup", {"FLOAT_FMT_ZGROUP": "3"})
    # the register-staged 192 x 128 tile instead of the LDS-DMA 192 x 320 one: same k order, one accumulator per output
    regstage = run_child(tmp_path, "fmt", "wide2", {"FLOAT_FMT_WIDE_VARIANT": "2"})
    for k in ("r3", "r4"):
        assert torch.isfinite(base[k]).all()
        assert torch.equal(base[k], tuned[k]), k
        assert torch.equal(base[k], every[k]), k
        assert torch.equal(base[k], per_step[k]), k
        assert torch.equal(base[k], grouped[k]), k
        assert torch.equal(base[k], regstage[k]), k


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_persistent_adaln_projection_is_bitwise_neutral(tmp_path, dtype):
    """fmt_gemm_big4_kernel (one persistent launch over the DENSE rows of a window's evaluations, from 1 536 rows on) against
    fmt_gemm_dma_kernel on one padded row block per evaluation (FLOAT_FMT_BIG=0) and against one launch per evaluation
    (FLOAT_FMT_HOIST=0): every modulation is the same MFMA sequence over k plus one bias add, so r_d is bitwise identical - for
    one clip with 3- and 4-way CFG (1 800 / 2 400 rows, the last row block ragged) and for two stacked clips (3 600 rows)."""
    env = {"CHILD_DTYPE": dtype}
    big = run_child(tmp_path, "fmtbig", "big_" + dtype, env)
    dma = run_child(tmp_path, "fmtbig", "dma_" + dtype, dict(env, FLOAT_FMT_BIG="0"))
    per_step = run_child(tmp_path, "fmtbig", "step_" + dtype, dict(env, FLOAT_FMT_HOIST="0"))
    for k in ("r3", "r4", "rb"):
        assert torch.isfinite(big[k]).all()
        assert torch.equal(big[k], dma[k]), k
        assert torch.equal(big[k], per_step[k]), k


def test_fused_upsample_matches_separate_kernels(tmp_path):
    sep = run_child(tmp_path, "dec", "separate", {"FLOAT_DEC_ZBLUR_MIN": "9999"})["frames"]
    fused = run_child(tmp_path, "dec", "fused", {})["frames"]
    lower = run_child(tmp_path, "dec", "fused32", {"FLOAT_DEC_ZBLUR_MIN": "32"})["frames"]
    assert sep.shape == (5, 512, 512, 3)
    for got in (fused, lower):
        # the fp32 filter sums are contracted differently, which flips a 16-bit rounding here and there; later layers and the
        # warp amplify a flip into an isolated pixel difference (measured: 82 dB, max 9e-3, mean 3e-5; parity with the reference
        # is 58 dB).  A tiling or border mistake would show up as tens of dB less.
        d = (got - sep).abs()
        psnr = float(-10 * torch.log10(((got - sep) ** 2).mean()))
        assert psnr > 72.0 and float(d.mean()) < 2e-4 and float(d.max()) < 0.05, (psnr, float(d.mean()), float(d.max()))


def test_conv_tile_width_is_bitwise_neutral(tmp_path):
    """FLOAT_DEC_CONV_BN=32: the 3x3 convs on 32-channel tiles (the default takes 64 where the layer has them).  An output's K
    chunks and taps are accumulated in the same order either way."""
    wide = run_child(tmp_path, "dec", "bn64", {})["frames"]
    narrow = run_child(tmp_path, "dec", "bn32", {"FLOAT_DEC_CONV_BN": "32"})["frames"]
    assert torch.equal(wide, narrow)


def test_channel_block_order_is_bitwise_neutral(tmp_path):
    """dec_group_cb (the output-channel blocks of a tile as consecutive slots of one XCD) against the round-1 order (grid.y):
    which workgroup computes a tile does not change a single operation."""
    new = run_child(tmp_path, "dec", "cbnew", {})["frames"]
    old = run_child(tmp_path, "dec", "cbold", {"FLOAT_DEC_CB_ORDER": "0"})["frames"]
    assert torch.equal(new, old)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_device_packed_weights_equal_host_packed(tmp_path, dtype):
    """Round 6: every Linear's fragment-major 16-bit image is written by fmt_pack_w_kernel from the fp32 rows on the device
    (to_target 3.4 s -> 0.54 s) instead of a host loop (FLOAT_PACK_HOST=1).  Same rounding (nearest even, fp16 saturating), same
    layout: the chain's output is bitwise identical."""
    env = {"CHILD_DTYPE": dtype}
    dev = run_child(tmp_path, "fmtbig", "packdev_" + dtype, env)
    host = run_child(tmp_path, "fmtbig", "packhost_" + dtype, dict(env, FLOAT_PACK_HOST="1"))
    for k in ("r3", "r4", "rb"):
        assert torch.equal(dev[k], host[k]), k


def test_toflow_in_the_conv_epilogue_matches_the_flow_kernel(tmp_path):
    """Round 6, last level: ToFlow's 1x1 conv as one more MFMA in conv2's epilogue (weights hi + lo * 2^-11, the same 16-bit V
    values as operands, V itself never stored) + dec_flowlast_kernel, against dec_flow_kernel on the stored V
    (FLOAT_DEC_FLOW_EPI=0).  Both are fp32 sums of the same exact products in different orders: the frames agree to the rounding
    of a flow value (one in ~1e5 pixels moves a bilinear tap; parity with the reference is 56 dB)."""
    epi = run_child(tmp_path, "dec", "flowepi", {})["frames"]
    old = run_child(tmp_path, "dec", "flowker", {"FLOAT_DEC_FLOW_EPI": "0"})["frames"]
    d = (epi - old).abs()
    psnr = float(-10 * torch.log10(((epi - old) ** 2).mean() + 1e-20))
    print("ToFlow in the epilogue vs flow kernel: %.1f dB, max %.2e, mean %.2e" % (psnr, float(d.max()), float(d.mean())))
    assert psnr > 80.0 and float(d.mean()) < 5e-5
