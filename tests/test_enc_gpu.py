"""GPU parity of the appearance-encoder operator (float_enc_*) against the
goldens made from the reference's Encoder and against the live CPU oracle, through the C ABI.
16-bit activations / conv weights with fp32 accumulation; s_r, Encoder.fc and Direction are fp32.
Stated tolerances (rel-L2 per tensor): fp16 2e-3, bf16 1.5e-2; measured 7e-4 / 5e-3 (seven ResBlocks of 16-bit activations)."""
import math

import numpy as np
import pytest
import torch

from oracle import float_oracle as O
from tests.util import golden, load_pkg, rel_l2

pkg = load_pkg()
W = pkg.weights
pytestmark = pytest.mark.gpu

TOL = {"fp16": 2e-3, "bf16": 1.5e-2, "fp32": 2e-5}  # fp32 = the verification mode (SURVEY 8d asks 1e-4; measured <= 2e-6)


def _image(seed, size):
    return torch.from_numpy(np.random.RandomState(seed).rand(1, 3, size, size).astype(np.float32)) * 2 - 1


@pytest.mark.parametrize("dtype", ["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("size", [64, 512])
def test_encoder_golden(size, dtype):
    g = golden("enc_%d" % size)
    esd = W.synth_encoder_state(size, seed=g["seed"])
    dsd = W.synth_decoder_state(size, seed=g["seed"])
    enc = pkg.encoder.EncoderHIP(esd, size, 512, 20, "cuda:0", dtype, direction_weight=dsd["direction.weight"])
    s_r, lam, feats, r_s = enc.encode_image_into_latent(_image(g["seed"], size))
    errs = dict(s_r=rel_l2(s_r.cpu(), g["s_r"]), lam=rel_l2(lam.cpu(), g["lam"]), r_s=rel_l2(r_s.cpu(), g["r_s"]))
    assert len(feats) == int(math.log2(size)) - 2
    for i, f in enumerate(feats):
        st = int(g["feat%d_stride" % i])
        assert tuple(f.shape[1:]) == enc.feat_shapes()[i]
        errs["feat%d" % i] = rel_l2(f.cpu()[:, :, ::st, ::st], g["feat%d" % i])
    print(size, dtype, " ".join("%s %.2e" % kv for kv in errs.items()))
    assert max(errs.values()) < TOL[dtype], errs
    assert enc.saturation() == 0  # no 16-bit store had to be clamped (always 0 for bf16 / fp32)


def _stress_errs(tag, dtype):
    g = golden("enc_stress_64_" + tag)
    size = 64
    esd = W.scale_encoder_convs(W.synth_encoder_state(size, seed=g["seed"]), g["gain"])
    dsd = W.synth_decoder_state(size, seed=g["seed"])
    enc = pkg.encoder.EncoderHIP(esd, size, 512, 20, "cuda:0", dtype, direction_weight=dsd["direction.weight"])
    s_r, lam, feats, r_s = enc.encode_image_into_latent(_image(g["seed"], size))
    errs = dict(s_r=rel_l2(s_r.cpu(), g["s_r"]), lam=rel_l2(lam.cpu(), g["lam"]))
    for i, f in enumerate(feats):
        st = int(g["feat%d_stride" % i])
        errs["feat%d" % i] = rel_l2(f.cpu()[:, :, ::st, ::st], g["feat%d" % i])
    return errs, enc.saturation(), g


def test_encoder_range_stress():
    """The reference Encoder with every conv weight x gain (encoder.py:146-247; tools/make_goldens.py GOLDENS_ONLY=encx).
    gain 2: activations up to ~300, inside fp16's range -> counter 0 and the fp16 result within its tolerance.
    gain 6: the 8 x 8 skip map reaches 1.9e5 > 65504 -> the fp16 operator MUST report it (float_enc_saturation > 0) and its
    result is indeed wrong; the fp32 mode reproduces the reference with counter 0.  This is what makes `saturation() == 0`
    in the other tests (and the warning of the product path) mean something."""
    e2, sat2, _ = _stress_errs("g2", "fp16")
    print("gain 2 fp16:", " ".join("%s %.2e" % kv for kv in e2.items()), "saturation", sat2)
    assert sat2 == 0 and max(e2.values()) < TOL["fp16"], (sat2, e2)
    e6, sat6, g6 = _stress_errs("g6", "fp16")
    print("gain 6 fp16:", " ".join("%s %.2e" % kv for kv in e6.items()), "saturation", sat6, "ref |feat0|max %.3g" % g6["feat0_absmax"])
    assert sat6 > 0, "an out-of-range checkpoint went unnoticed"
    # (the fp32 copies of the skip maps are written before the 16-bit conversion and stay right; what the 16-bit chain feeds
    # forward - s_r, lambda, and the 16-bit maps the decoder receives - is clamped)
    assert e6["s_r"] > 10 * TOL["fp16"], "gain 6 was expected to break the fp16 result"
    e32, sat32, _ = _stress_errs("g6", "fp32")
    print("gain 6 fp32:", " ".join("%s %.2e" % kv for kv in e32.items()), "saturation", sat32)
    assert sat32 == 0 and max(e32.values()) < TOL["fp32"], (sat32, e32)


def test_encoder_live_oracle_and_determinism():
    size = 128
    esd = W.synth_encoder_state(size, seed=5)
    dsd = W.synth_decoder_state(size, seed=5)
    img = _image(77, size)
    enc = pkg.encoder.EncoderHIP(esd, size, 512, 20, "cuda:0", "fp16", direction_weight=dsd["direction.weight"])
    s_r, lam, feats, r_s = enc.encode_image_into_latent(img)
    o_s, o_f, o_l = O.encode_appearance(esd, img)
    assert rel_l2(s_r.cpu(), o_s) < TOL["fp16"] and rel_l2(lam.cpu(), o_l) < TOL["fp16"]
    assert rel_l2(r_s.cpu(), O.direction(dsd, o_l)) < TOL["fp16"]
    for a, b in zip(feats, o_f):
        assert rel_l2(a.cpu(), b) < TOL["fp16"]
    # Direction alone (fp32 Householder Q vs torch.linalg.qr): feed the oracle's lambda through r_s = lam @ Q^T
    s2, l2, f2, r2 = enc.encode_image_into_latent(img)
    assert torch.equal(s_r, s2) and torch.equal(r_s, r2) and all(torch.equal(a, b) for a, b in zip(feats, f2))
    # reference-shaped call: Encoder.forward(img, None) -> (h_source, None, feats)
    h, none, f3 = enc(img, None)
    assert none is None and torch.equal(h, s_r) and len(f3) == len(feats)


def test_encoder_hands_feats_to_decoder():
    """float_enc_feats16 -> float_dec_set_feats16 must give the frames of the fp32 NCHW hand-over bit for bit."""
    size = 64
    esd = W.synth_encoder_state(size, seed=9)
    dsd = W.synth_decoder_state(size, seed=9)
    enc = pkg.encoder.EncoderHIP(esd, size, 512, 20, "cuda:0", "fp16")
    dec = pkg.decoder.SynthesisHIP(dsd, size, 512, "cuda:0", "fp16", max_frames=4)
    s_r, lam, feats, r_s = enc.encode_image_into_latent(_image(3, size))
    assert r_s is None  # no direction.weight given
    g = torch.Generator().manual_seed(0)
    r_d = torch.randn(1, 3, 512, generator=g) * 0.3
    a = dec.decode_latent_into_processed_images(s_r, r_d, [f * 0.05 for f in feats]).clone()  # different feats first
    b = dec.decode_latent_into_processed_images(s_r, r_d, feats).clone()
    enc.hand_feats_to(dec)
    c = dec.decode_latent_into_processed_images(s_r, r_d)
    assert not torch.equal(a, b)
    assert torch.equal(b, c)


def test_encoder_errors():
    esd = W.synth_encoder_state(64, seed=1)
    with pytest.raises(KeyError):
        pkg.encoder.EncoderHIP({k: v for k, v in esd.items() if "convs.2.conv2" not in k}, 64, 512, 20, "cuda:0")
    with pytest.raises(ValueError):
        pkg.encoder.EncoderHIP(esd, 48, 512, 20, "cuda:0")
    enc = pkg.encoder.EncoderHIP(esd, 64, 512, 20, "cuda:0")
    with pytest.raises(ValueError):
        enc.encode_image_into_latent(torch.zeros(1, 3, 32, 32))
    # the Blur buffers come from the checkpoint in the reference (strict load): any 4 x 4 values are applied
    # (test_encoder_blur_buffers_from_checkpoint); another size is refused - the layer's padding belongs to 4 taps
    odd = dict(esd)
    key = [k for k in esd if k.endswith("conv2.0.kernel")][0]
    odd[key] = torch.ones(3, 3) / 9
    with pytest.raises(ValueError, match="4 x 4"):
        pkg.encoder.EncoderHIP(odd, 64, 512, 20, "cuda:0")


@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
def test_encoder_blur_buffers_from_checkpoint(dtype):
    """Blur buffers other than make_kernel([1,3,3,1]) (tests/golden/fir_buffers.npz: the reference Encoder after a strict load of
    weights.fir_buffer_states - `conv2.0.kernel` = make_kernel([1,2,4,1]), asymmetric: pins the flip of upfirdn2d; `skip.0.kernel`
    = a non-separable positive kernel).  enc_blur_kernel takes every layer's 16 taps from the checkpoint; a state without the
    buffers encodes bitwise like one holding the default buffers."""
    g = golden("fir_buffers")
    size, seed = int(g["size"]), int(g["seed"])
    esd, _ = W.fir_buffer_states(size, seed)
    enc = pkg.encoder.EncoderHIP(esd, size, 512, 20, "cuda:0", dtype)
    s_r, lam, feats, _ = enc.encode_image_into_latent(_image(seed, size))
    errs = dict(s_r=rel_l2(s_r.cpu(), g["enc_s_r"]), lam=rel_l2(lam.cpu(), g["enc_lam"]))
    for i, f in enumerate(feats):
        st = int(g["enc_feat%d_stride" % i])
        errs["feat%d" % i] = rel_l2(f.cpu()[:, :, ::st, ::st], g["enc_feat%d" % i])
    print(dtype, " ".join("%s %.2e" % kv for kv in errs.items()))
    assert max(errs.values()) < TOL[dtype], errs
    esd0 = W.synth_encoder_state(size, seed=seed)
    a = pkg.encoder.EncoderHIP(esd0, size, 512, 20, "cuda:0", dtype).encode_image_into_latent(_image(seed, size))
    assert rel_l2(a[2][0].cpu(), g["enc_feat0"]) > 0.05  # the default kernels: a different encoder (the 8-px map, behind every Blur)
    b = pkg.encoder.EncoderHIP({k: v for k, v in esd0.items() if not k.endswith(".kernel")}, size, 512, 20, "cuda:0", dtype).encode_image_into_latent(_image(seed, size))
    assert torch.equal(a[0], b[0]) and all(torch.equal(x, y) for x, y in zip(a[2], b[2]))


def test_encoder_fp32_feeds_decoder_fp32():
    """The fp32 verification modes chain: encoder (fp32 NHWC skip features) -> float_dec_set_feats16 with dtype fp32 -> decoder.
    Image -> frames against the oracle (which is bit-identical to the reference's Encoder and Synthesis on the goldens) at 1e-4,
    and a 16-bit encoder is refused by an fp32 decoder."""
    size = 64
    esd = W.synth_encoder_state(size, seed=9)
    dsd = W.synth_decoder_state(size, seed=9)
    enc = pkg.encoder.EncoderHIP(esd, size, 512, 20, "cuda:0", "fp32")
    dec = pkg.decoder.SynthesisHIP(dsd, size, 512, "cuda:0", "fp32", max_frames=4)
    img = _image(3, size)
    s_r, lam, feats, _ = enc.encode_image_into_latent(img)
    enc.hand_feats_to(dec)
    g = torch.Generator().manual_seed(0)
    r_d = torch.randn(1, 3, 512, generator=g) * 0.3
    frames = dec.decode_latent_into_processed_images(s_r, r_d).cpu()
    o_s, o_f, _ = O.encode_appearance(esd, img)
    want = O.decode_frames(dsd, o_s, r_d, o_f)
    m = float((frames - want).abs().max())
    print("image -> frames, fp32 chain: max|d| %.2e" % m)
    assert m <= 1e-4
    enc16 = pkg.encoder.EncoderHIP(esd, size, 512, 20, "cuda:0", "fp16")
    enc16.encode_image_into_latent(img)
    with pytest.raises(ValueError, match="dtype"):
        enc16.hand_feats_to(dec)


def test_exported_skip_maps_feed_the_decoder_like_the_direct_hand_over():
    """float_enc_export_feats16 (ABI 6): the copies a batch of portraits keeps per item decode to the same frames, bit for bit, as
    the direct hand-over of the last forward (float_enc_feats16 -> float_dec_set_feats16) - also after ANOTHER image went through
    the encoder in between, which is what the copies are for (nodes.py:189-209: B items, one decode each)."""
    size = 64
    enc = pkg.encoder.EncoderHIP(W.synth_encoder_state(size, seed=3), size, 512, 20, "cuda:0", "fp16")
    dec = pkg.decoder.SynthesisHIP(W.synth_decoder_state(size, seed=3), size, 512, "cuda:0", "fp16", max_frames=4)
    g = torch.Generator().manual_seed(5)
    img_a, img_b = torch.rand(1, 3, size, size, generator=g) * 2 - 1, torch.rand(1, 3, size, size, generator=g) * 2 - 1
    r_d = torch.randn(1, 3, 512, generator=g) * 0.5
    s_a, _, _, _ = enc.encode_image_into_latent(img_a, want_feats=False)
    enc.hand_feats_to(dec)
    direct = dec.decode_latent_into_processed_images(s_a, r_d[0]).clone()
    kept = enc.export_feats16()
    enc.encode_image_into_latent(img_b, want_feats=False)  # the encoder's own buffers now hold image b
    enc.hand_feats_to(dec)
    other = dec.decode_latent_into_processed_images(s_a, r_d[0]).clone()
    dec.set_feats16(kept, enc.dtype)
    again = dec.decode_latent_into_processed_images(s_a, r_d[0])
    assert torch.equal(again, direct) and not torch.equal(other, direct)
    with pytest.raises(ValueError):
        pkg.native.check(pkg.native.lib().float_enc_export_feats16(enc._h, None, 3, None))
