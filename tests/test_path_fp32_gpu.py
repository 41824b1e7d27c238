"""The whole path - image + waveform -> frames - with EVERY operator in its fp32 verification mode (appearance encoder, wav2vec2
audio encoder, speech-emotion model, FMT sampler, decoder: the production launch chains and kernels with 4-byte operands), through
the product object (InferenceAgent.infer_device), against the CPU oracle chained the same way.  The oracle is pinned to the
reference per operator (tests/test_oracle_golden.py: FMT <= 2.2e-6, decoder / encoder bit-identical, audio 4e-6), so this is the
HIP logic of the complete path at reference precision: r_d rel-L2 <= 2e-5, frames max-abs <= 1e-4 at 64 px."""
import importlib
import math

import numpy as np
import pytest
import torch

from oracle import float_oracle as O
from tests.util import load_pkg, max_abs, rel_l2

pkg = load_pkg()
W, C = pkg.weights, pkg.config
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("emo", ["happy", None])
def test_image_and_audio_to_frames_fp32(emo):
    gen = importlib.import_module(pkg.__name__ + ".src.nodes.generate")
    opt = importlib.import_module(pkg.__name__ + ".src.nodes.options.base_options").BaseOptions()
    opt.input_size, opt.nfe = 64, 6
    cfg = C.FmtConfig.from_options(opt)
    acfg, ecfg = C.small_audio_config(), C.small_emotion_config()
    acfg.dim_w = opt.dim_w
    parts = dict(enc=W.synth_encoder_state(64, seed=31), dec=W.synth_decoder_state(64, seed=31), fmt=W.synth_fmt_state(cfg, seed=31),
                 audio_encoder=(W.synth_audio_state(acfg, seed=31), acfg), emotion_encoder=(W.synth_audio_state(ecfg, seed=32), ecfg))
    agent = gen.InferenceAgent(opt, parts, "cuda:0", max_frames=8, fmt_dtype="fp32", dec_dtype="fp32", aud_dtype="fp32")
    img = torch.from_numpy(np.random.RandomState(5).rand(1, 3, 64, 64).astype(np.float32)) * 2 - 1
    wav = W.synth_waveform(1.4, seed=9)  # 35 frames: one window, replicate-padded
    frames = agent.infer_device(img.cuda(), wav.cuda(), 2.0, 1.0, 1.0, emo=emo, seed=7)
    assert frames.shape == (35, 64, 64, 3) and frames.is_pinned()
    # the oracle, chained like FLOAT.inference (FLOAT.py:172-253, 113-169)
    s_r, feats, lam = O.encode_appearance(parts["enc"], img)
    r_s = O.direction(parts["dec"], lam)
    T = math.ceil(wav.shape[-1] * opt.fps / opt.sampling_rate)
    wa = O.audio_encoder_inference(parts["audio_encoder"][0], acfg, wav, T)
    if emo is None:
        we = O.audio2emotion_predict(parts["emotion_encoder"][0], ecfg, wav).reshape(1, 1, -1)
    else:
        we = pkg.host_models.emotion_one_hot(emo, "cpu")
    noise = pkg.fmt.draw_noise(1, 1, cfg, 7)
    r_d = O.sample_rd(parts["fmt"], cfg, r_s, wa, we, noise, opt.nfe, 2.0, 1.0, 1.0)
    want = O.decode_frames(parts["dec"], s_r, r_d, feats)
    m = max_abs(frames, want)
    print("whole path in fp32 (emo=%s): frames max|d| %.2e" % (emo, m))
    assert m <= 1e-4
    assert agent.G.dec.saturation() == 0


def test_product_path_reports_fp16_overflow(monkeypatch):
    """An encoder checkpoint that leaves fp16's range (conv weights x 6: the 8 x 8 skip map reaches 1.9e5) through the PRODUCT
    call: InferenceAgent.infer_device warns (RuntimeWarning, default) or raises (FLOAT_AMD_RANGE=raise) instead of handing
    back wrong / black regions silently; the same checkpoint with the fp32 decoder + encoder passes without a message."""
    gen = importlib.import_module(pkg.__name__ + ".src.nodes.generate")
    opt = importlib.import_module(pkg.__name__ + ".src.nodes.options.base_options").BaseOptions()
    opt.input_size, opt.nfe = 64, 3
    cfg = C.FmtConfig.from_options(opt)
    acfg = C.small_audio_config()
    acfg.dim_w = opt.dim_w
    parts = dict(enc=W.scale_encoder_convs(W.synth_encoder_state(64, seed=1010), 6.0), dec=W.synth_decoder_state(64, seed=1010),
                 fmt=W.synth_fmt_state(cfg, seed=31), audio_encoder=(W.synth_audio_state(acfg, seed=31), acfg))
    img = torch.from_numpy(np.random.RandomState(1010).rand(1, 3, 64, 64).astype(np.float32)) * 2 - 1
    wav = W.synth_waveform(0.5, seed=9)
    agent = gen.InferenceAgent(opt, parts, "cuda:0", max_frames=8)
    monkeypatch.setenv("FLOAT_AMD_RANGE", "warn")
    with pytest.warns(RuntimeWarning, match="fp16 range exceeded in .*encoder"):
        agent.infer_device(img.cuda(), wav.cuda(), 2.0, 1.0, 1.0, emo="happy", seed=7)
    monkeypatch.setenv("FLOAT_AMD_RANGE", "raise")
    with pytest.raises(pkg.pipeline.Fp16RangeError):
        agent.infer_device(img.cuda(), wav.cuda(), 2.0, 1.0, 1.0, emo="happy", seed=7)
    # FLOAT_AMD_RANGE=auto: warn, rebuild the operators that overflowed in a type with the range, run the clip again
    monkeypatch.setenv("FLOAT_AMD_RANGE", "auto")
    auto = gen.InferenceAgent(opt, parts, "cuda:0", max_frames=8)
    with pytest.warns(RuntimeWarning, match="fp16 range exceeded"):
        frames_auto = auto.infer_device(img.cuda(), wav.cuda(), 2.0, 1.0, 1.0, emo="happy", seed=7)
    assert auto.G.dec.dtype == "fp32" and auto.enc.dtype == "fp32" and auto.G.fmt.dtype == "bf16" and torch.isfinite(frames_auto).all()
    monkeypatch.setenv("FLOAT_AMD_RANGE", "raise")
    # the remedy the message names: fp32 decoder + encoder, and an FMT operand type with fp32's exponent range (the identity
    # latent r_s ~ 1e7 of this checkpoint leaves fp16's range inside the FMT's condition rows too)
    agent32 = gen.InferenceAgent(opt, parts, "cuda:0", max_frames=8, dec_dtype="fp32", fmt_dtype="bf16")
    frames = agent32.infer_device(img.cuda(), wav.cuda(), 2.0, 1.0, 1.0, emo="happy", seed=7)  # raise mode: no error
    assert torch.isfinite(frames).all() and agent32.range_counts() == {"audio": 0}
    assert torch.equal(frames, frames_auto)  # the automatic rebuild ended in the same operators
