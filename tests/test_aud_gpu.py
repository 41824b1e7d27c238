"""GPU parity of the audio conditioning operator (float_aud_*) against goldens made
from the reference's AudioEncoder (transformers' Wav2Vec2Model underneath) and against the live CPU oracle,
through the C ABI.  16-bit activations / weights, fp32 accumulation, statistics and residual stream.
Stated tolerance on wa (rel-L2): fp16 3e-3, bf16 2.5e-2 (measured 9e-4 / 8e-3 on the base config)."""
import pytest
import torch

from oracle import float_oracle as O
from tests.util import golden, load_pkg, rel_l2

pkg = load_pkg()
W, C = pkg.weights, pkg.config
pytestmark = pytest.mark.gpu

TOL = {"fp16": 3e-3, "bf16": 2.5e-2, "fp32": 2e-5}  # fp32 = the verification mode (the same kernels, 4-byte operands)


@pytest.mark.parametrize("dtype", ["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("tag", ["small", "base"])
def test_audio_golden(tag, dtype):
    g = golden("aud_" + tag)
    cfg = C.small_audio_config() if tag == "small" else C.AudioConfig()
    sd = W.synth_audio_state(cfg, seed=g["seed"])
    enc = pkg.audio.AudioEncoderHIP(sd, cfg, "cuda:0", dtype)
    wa = enc.inference(W.synth_waveform(g["seconds"], seed=g["seed"] + 1), int(g["T"])).cpu()
    e = rel_l2(wa, g["wa"])
    print(tag, dtype, "wa rel-L2 %.3e max|d| %.3e" % (e, float((wa - g["wa"]).abs().max())))
    assert wa.shape == g["wa"].shape and e < TOL[dtype]
    assert enc.saturation() == 0  # no 16-bit activation store was clamped (float_aud_saturation; always 0 for bf16 / fp32)


def test_audio_live_oracle_lengths_and_determinism():
    """Ragged lengths: replicate padding (FLOAT.py:371-373), a clip longer than the first (workspace regrowth),
    only_last_features, batch of two, bitwise run-to-run determinism."""
    cfg = C.small_audio_config()
    sd = W.synth_audio_state(cfg, seed=21)
    enc = pkg.audio.AudioEncoderHIP(sd, cfg, "cuda:0", "fp16")
    for seconds, T in ((0.5, 13), (3.1, 78), (12.0, 300), (1.0, 25)):  # 300 frames: more than one row block per GEMM
        a = W.synth_waveform(seconds, seed=int(seconds * 10))
        got = enc.inference(a, T)
        want = O.audio_encoder_inference(sd, cfg, a, T)
        assert rel_l2(got.cpu(), want) < TOL["fp16"], (seconds, T)
        assert torch.equal(got, enc.inference(a, T))
    a2 = torch.cat([W.synth_waveform(1.0, seed=4), W.synth_waveform(1.0, seed=5)])
    got = enc.inference(a2, 25).cpu()
    assert rel_l2(got, O.audio_encoder_inference(sd, cfg, a2, 25)) < TOL["fp16"]
    cfg2 = C.small_audio_config()
    cfg2.only_last_features = True
    sd2 = W.synth_audio_state(cfg2, seed=22)
    enc2 = pkg.audio.AudioEncoderHIP(sd2, cfg2, "cuda:0", "fp16")
    a = W.synth_waveform(1.0, seed=6)
    assert rel_l2(enc2.inference(a, 25).cpu(), O.audio_encoder_inference(sd2, cfg2, a, 25)) < TOL["fp16"]


def test_audio_weight_norm_key_variants_and_errors():
    cfg = C.small_audio_config()
    sd = W.synth_audio_state(cfg, seed=23)
    a = W.synth_waveform(1.0, seed=7)
    base = pkg.audio.AudioEncoderHIP(sd, cfg, "cuda:0", "fp16").inference(a, 25)
    p = "wav2vec2.encoder.pos_conv_embed.conv."
    legacy = {k: v for k, v in sd.items() if "parametrizations" not in k}
    legacy[p + "weight_g"] = sd[p + "parametrizations.weight.original0"]
    legacy[p + "weight_v"] = sd[p + "parametrizations.weight.original1"]
    assert torch.equal(base, pkg.audio.AudioEncoderHIP(legacy, cfg, "cuda:0", "fp16").inference(a, 25))
    plain = {k: v for k, v in sd.items() if "parametrizations" not in k}
    plain[p + "weight"] = O.pos_conv_weight(sd)
    assert rel_l2(pkg.audio.AudioEncoderHIP(plain, cfg, "cuda:0", "fp16").inference(a, 25).cpu(), base.cpu()) < 1e-3
    with pytest.raises(KeyError):
        pkg.audio.AudioEncoderHIP({k: v for k, v in sd.items() if "layers.1.feed_forward" not in k}, cfg, "cuda:0")
    bad = C.small_audio_config()
    bad.hidden_size, bad.num_attention_heads = 320, 5  # LayerNorm width not a multiple of 256
    with pytest.raises(ValueError):
        pkg.audio.AudioEncoderHIP(sd, bad, "cuda:0")


@pytest.mark.parametrize("dtype", ["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("tag", ["small", "xlsr"])
def test_speech_emotion_golden(tag, dtype):
    """float_aud_classify (wav2vec2-large variant + classification head) vs the golden assembled from transformers'
    Wav2Vec2Model and the reference's Wav2Vec2ClassificationHead.  Scores are probabilities: max |d| <= 1e-3 fp16, 1e-2 bf16 (measured 2.8e-4 / 3.0e-3 on the xlsr shape)."""
    g = golden("emo_" + tag)
    cfg = C.small_emotion_config() if tag == "small" else C.emotion_audio_config()
    sd = W.synth_audio_state(cfg, seed=g["seed"])
    ser = pkg.audio.Audio2EmotionHIP(sd, cfg, "cuda:0", dtype)
    a = W.synth_waveform(g["seconds"], seed=g["seed"] + 1)
    scores = ser.predict_emotion(a).cpu()
    d = float((scores - g["scores"]).abs().max())
    print(tag, dtype, "scores max|d| %.3e" % d, [round(float(v), 4) for v in scores[0]])
    assert scores.shape == g["scores"].shape and d < {"fp16": 1e-3, "bf16": 1e-2, "fp32": 2e-6}[dtype]
    assert abs(float(scores.sum()) - 1.0) < 1e-5 and int(scores.argmax()) == int(g["scores"].argmax())
    assert ser.saturation() == 0
    assert torch.equal(ser.predict_emotion(a).cpu(), scores)
    with pytest.raises(TypeError):
        ser.inference(a, 25)


def test_speech_emotion_live_oracle_ragged():
    cfg = C.small_emotion_config()
    sd = W.synth_audio_state(cfg, seed=41)
    ser = pkg.audio.Audio2EmotionHIP(sd, cfg, "cuda:0", "fp16")
    for seconds in (0.3, 2.7, 1.0):
        a = W.synth_waveform(seconds, seed=int(seconds * 10))
        assert float((ser.predict_emotion(a).cpu() - O.audio2emotion_predict(sd, cfg, a)).abs().max()) < 2e-3
    # prev_a is concatenated in front (FLOAT.py:397-398)
    a, p = W.synth_waveform(1.0, seed=2), W.synth_waveform(0.4, seed=3)
    assert torch.equal(ser.predict_emotion(a, p), ser.predict_emotion(torch.cat([p, a], dim=1)))


def test_run_time_calls_do_not_allocate():
    """float_aud_inference refuses a clip beyond the reserved workspace (float_aud_reserve is the only allocating call);
    the Python mirror reserves before it calls."""
    import ctypes as C
    cfg = pkg.config.small_audio_config()
    sd = W.synth_audio_state(cfg, seed=3)
    enc = pkg.audio.AudioEncoderHIP(sd, cfg, "cuda:0", "fp16")
    a = W.synth_waveform(1.0, seed=2).cuda()
    out = torch.empty(25, cfg.dim_w, device="cuda:0")
    L, N = pkg.native.lib(), pkg.native
    rc = L.float_aud_inference(enc._h, N.dev_ptr(a[0]), a.shape[1], 25, N.dev_ptr(out), N.stream_ptr("cuda:0"))
    assert rc == 1 and b"float_aud_reserve" in L.float_last_error()
    N.check(L.float_aud_reserve(enc._h, a.shape[1], 25, N.stream_ptr("cuda:0")))
    N.check(L.float_aud_inference(enc._h, N.dev_ptr(a[0]), a.shape[1], 25, N.dev_ptr(out), N.stream_ptr("cuda:0")))
    wa = enc.inference(a, 25)[0]
    assert torch.equal(wa, out)
    longer = enc.inference(W.synth_waveform(2.0, seed=2).cuda(), 50)  # the mirror grows the reservation itself
    assert longer.shape == (1, 50, cfg.dim_w) and torch.isfinite(longer).all()


def test_long_audio_beyond_the_round1_frame_limit():
    """More than 3900 transformer frames per call (round 1's limit: one score row per wave in 64 KiB of LDS): 84 s through the
    50 Hz speech-emotion model (4199 frames) and 170 s of audio conditioning at 25 fps (4250 frames), small configs, against
    the oracle."""
    cfg = C.small_emotion_config()
    sd = W.synth_audio_state(cfg, seed=43)
    ser = pkg.audio.Audio2EmotionHIP(sd, cfg, "cuda:0", "fp16")
    a = W.synth_waveform(84.0, seed=5)
    got, ref = ser.predict_emotion(a).cpu(), O.audio2emotion_predict(sd, cfg, a)
    assert float((got - ref).abs().max()) < 2e-3
    acfg = C.small_audio_config()
    asd = W.synth_audio_state(acfg, seed=44)
    enc = pkg.audio.AudioEncoderHIP(asd, acfg, "cuda:0", "fp16")
    a = W.synth_waveform(170.0, seed=6)
    wa = enc.inference(a, 4250).cpu()
    assert rel_l2(wa, O.audio_encoder_inference(asd, acfg, a, 4250)) < 3e-3


def test_audio_length_is_not_capped_by_lds():
    """the reference has no audio length limit (FLOAT.py:190-198); rounds 1-2 stopped at 3900 / 10 000 frames
    because the attention kernel kept a whole score row in LDS.  The keys are tiled now (online softmax): 12 000 frames =
    480 s at 25 fps = six key tiles, against the oracle; and a clip whose length is not a multiple of the tile."""
    acfg = C.small_audio_config()
    asd = W.synth_audio_state(acfg, seed=45)
    enc = pkg.audio.AudioEncoderHIP(asd, acfg, "cuda:0", "fp16")
    a = W.synth_waveform(480.0, seed=8)
    wa = enc.inference(a, 12000).cpu()
    ref = O.audio_encoder_inference(asd, acfg, a, 12000)
    e = rel_l2(wa, ref)
    print("12 000 frames: rel-L2 %.2e" % e)
    assert wa.shape == (1, 12000, acfg.dim_w) and e < 3e-3 and enc.saturation() == 0
    a = W.synth_waveform(100.0, seed=9)
    wa = enc.inference(a, 2500).cpu()  # one tile + 452 keys
    assert rel_l2(wa, O.audio_encoder_inference(asd, acfg, a, 2500)) < 3e-3
    with pytest.raises(ValueError, match="200000"):
        enc.inference(W.synth_waveform(8.0, seed=7), 200001)
