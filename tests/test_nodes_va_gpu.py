"""The very-advanced node graph end to end on the GPU: split checkpoint files -> VA loaders (architecture inferred from the
tensors) -> Apply FLOAT Encoder -> Get Identity Reference -> audio nodes -> Sample Motion Sequence RD -> Apply FLOAT Synthesis,
checked against the same operators driven directly and against the CPU oracle."""
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import float_oracle as O
from tests.util import load_pkg, rel_l2

pkg = load_pkg()
W, C = pkg.weights, pkg.config
pytestmark = pytest.mark.gpu
SIZE = 64


@pytest.fixture(scope="module")
def models(tmp_path_factory):
    from safetensors.torch import save_file
    root = tmp_path_factory.mktemp("models")
    os.environ["FLOAT_MODELS_DIR"] = str(root)
    fcfg, acfg = C.FmtConfig(), C.small_audio_config()
    acfg.dim_w = 512
    sds = dict(enc=W.synth_encoder_state(SIZE, seed=31), dec=W.synth_decoder_state(SIZE, seed=31), fmt=W.synth_fmt_state(fcfg, seed=31),
               aud=W.synth_audio_state(acfg, seed=31))
    cont = lambda d: {k: v.contiguous() for k, v in d.items()}  # noqa: E731
    os.makedirs(root / "float" / "motion_autoencoder")
    save_file(cont(sds["enc"]), str(root / "float" / "motion_autoencoder" / "encoder.safetensors"))
    # decoder and fmt only exist inside the unified file: the loaders must extract them
    uni = {"motion_autoencoder.dec." + k: v for k, v in sds["dec"].items()}
    uni.update({"fmt." + k: v for k, v in sds["fmt"].items()})
    save_file(cont(uni), str(root / "float" / "FLOAT.safetensors"))
    wdir = root / "audio" / "wav2vec2-base-960h"
    os.makedirs(wdir)
    save_file(cont({k: v for k, v in sds["aud"].items() if k.startswith("wav2vec2.")}), str(wdir / "model.safetensors"))
    acfg.to_hf().save_pretrained(str(wdir))
    os.makedirs(root / "float" / "audio_projections")
    save_file(cont({k[len("audio_projection."):]: v for k, v in sds["aud"].items() if k.startswith("audio_projection.")}),
              str(root / "float" / "audio_projections" / "projection.safetensors"))
    ecfg = C.small_emotion_config()
    sds["emo"] = W.synth_audio_state(ecfg, seed=31)
    edir = root / "audio" / "wav2vec-english-speech-emotion-recognition"
    os.makedirs(edir)
    save_file(cont(sds["emo"]), str(edir / "model.safetensors"))
    hf = ecfg.to_hf()
    hf.id2label = {i: n for i, n in enumerate(["angry", "disgust", "fear", "happy", "neutral", "sad", "surprise"])}
    hf.label2id = {n: i for i, n in hf.id2label.items()}
    hf.save_pretrained(str(edir))
    return dict(root=root, sds=sds, fcfg=fcfg, acfg=acfg, ecfg=ecfg)


def test_va_graph(models):
    n = pkg.NODE_CLASS_MAPPINGS
    dev = "cuda:0"
    size, dim_w, dim_m, enc = n["LoadFloatEncoderModel"]().load_encoder_infer_arch("encoder.safetensors", dev, False)
    dec, dsize, style_dim, motion_dim = n["LoadFloatSynthesisModel"]().load_synthesis_infer_arch("decoder.safetensors", dev, 1, "[1, 3, 3, 1]", False)
    fmt, fps, fopts, chunk = n["LoadFMTModel"]().load_fmt_model("fmt.safetensors", dev, False, 7, 8, 2, 10, 25.0, 2.0)
    assert (size, dim_w, dim_m) == (SIZE, 512, 20) and (dsize, style_dim, motion_dim) == (SIZE, 512, 20)
    assert fps == 25.0 and chunk == 60 and fopts["dim_h"] == 1024 and fopts["fmt_depth"] == 8 and fopts["dim_a"] == 512
    assert abs(fopts["mlp_ratio"] - 4.0) < 1e-9 and os.path.exists(models["root"] / "float" / "fmt" / "fmt.safetensors")
    sr, pipe = n["LoadWav2VecModel"]().load_float_wav2vec_model("wav2vec2-base-960h", dev)
    proj, din, dim_a = n["LoadAudioProjectionLayer"]().load_projection_layer("projection.safetensors", dev)
    assert sr == 16000 and (din, dim_a) == (512, 512)  # 2 layers x 256

    img = torch.from_numpy(np.random.RandomState(5).rand(2, SIZE, SIZE, 3).astype(np.float32))
    pipe_app, lam, _ = n["ApplyFloatEncoder"]().apply_encoder(img, enc)
    assert pipe_app["h_source"].shape == (2, 512) and lam.shape == (2, 20) and len(pipe_app["feats"]) == 4
    o_s, o_f, o_l = O.encode_appearance(models["sds"]["enc"], img.permute(0, 3, 1, 2) * 2 - 1)
    assert rel_l2(pipe_app["h_source"], o_s) < 2e-3 and rel_l2(lam, o_l) < 2e-3
    _, r_s = n["FloatGetIdentityReferenceVA"]().get_identity_reference_batch(lam, dec)
    assert rel_l2(r_s, O.direction(models["sds"]["dec"], lam)) < 1e-5

    wav = torch.cat([W.synth_waveform(1.0, seed=8), W.synth_waveform(1.0, seed=9)])[:, None]
    feats, T, a_norm, _, _, fps2 = n["FloatAudioPreprocessAndFeatureExtract"]().extract_features_with_custom_model(
        {"waveform": wav, "sample_rate": 16000}, pipe, 25.0, False)
    assert T == 25 and feats.shape == (2, 25, 512) and fps2 == 25.0
    (wa,) = n["FloatApplyAudioProjection"]().apply_projection(feats, proj)
    assert rel_l2(wa, O.audio_encoder_inference(models["sds"]["aud"], models["acfg"], a_norm, 25)) < 3e-3

    we = torch.softmax(torch.randn(2, 1, 7, generator=torch.Generator().manual_seed(1)), -1)
    S = n["FloatSampleMotionSequenceRD_VA"]()
    args = dict(r_s_latent=r_s, wa_latent=wa, we_latent=we, audio_num_frames=T, float_fmt_model=fmt, a_cfg_scale=2.0, r_cfg_scale=1.0,
                e_cfg_scale=1.0, include_r_cfg=False, nfe=4, torchdiffeq_ode_method="euler", ode_atol=1e-5, ode_rtol=1e-5,
                audio_dropout_prob=0.1, ref_dropout_prob=0.1, emotion_dropout_prob=0.1, fix_noise_seed=True, seed=15)
    r_d, _ = S.sample_rd_sequence_va(**args)
    assert r_d.shape == (2, 25, 512) and torch.equal(r_d, S.sample_rd_sequence_va(**args)[0])
    noise = pkg.fmt.draw_noise(1, 2, models["fcfg"], 15, device="cuda:0")
    assert torch.equal(r_d, fmt.sample(r_s, wa, we, noise, 4, 2.0, 1.0, 1.0).cpu())
    want = O.sample_rd(models["sds"]["fmt"], models["fcfg"], r_s[:1], wa[:1], we[:1], noise[:, :1].cpu(), 4, 2.0, 1.0, 1.0)
    assert rel_l2(r_d[:1], want) < 5e-3

    r_d = r_d * 0.3
    frames, _ = n["ApplyFloatSynthesis"]().apply_synthesis(pipe_app, dec, r_d[:, :3])
    assert frames.shape == (6, SIZE, SIZE, 3) and frames.min() >= 0 and frames.max() <= 1
    direct = dec.decode_latent_into_processed_images(pipe_app["h_source"][1:2], r_d[1, :3], [f[1:2] for f in pipe_app["feats"]]).cpu()
    assert torch.equal(frames[3:], direct)
    empty, _ = n["ApplyFloatSynthesis"]().apply_synthesis(pipe_app, dec, r_d[:, :0])
    assert empty.shape == (0, SIZE, SIZE, 3)


def test_va_errors(models):
    n = pkg.NODE_CLASS_MAPPINGS
    _, _, _, enc = n["LoadFloatEncoderModel"]().load_encoder_infer_arch("encoder.safetensors", "cuda:0", False)
    with pytest.raises(ValueError):
        n["ApplyFloatEncoder"]().apply_encoder(torch.zeros(1, 32, 32, 3), enc)
    with pytest.raises(ValueError):
        n["LoadFloatSynthesisModel"]().load_synthesis_infer_arch("decoder.safetensors", "cuda:0", 2, "[1, 3, 3, 1]", False)
    with pytest.raises(ValueError, match="4-tap"):  # the Blur's padding follows the tap count; the reference's strict load fails too
        n["LoadFloatSynthesisModel"]().load_synthesis_infer_arch("decoder.safetensors", "cuda:0", 1, "[1, 2, 1]", False)
    # another 4-tap widget value loads; the checkpoint's [1,3,3,1] buffers decide what the decoder computes (strict load)
    dec, _, _, _ = n["LoadFloatSynthesisModel"]().load_synthesis_infer_arch("decoder.safetensors", "cuda:0", 1, "[1, 2, 2, 1]", False)
    assert dec.blur_kernel_setting == [1, 2, 2, 1]
    with pytest.raises(ValueError):
        n["LoadFMTModel"]().load_fmt_model("fmt.safetensors", "cuda:0", False, 600, 8, 2, 10, 25.0, 2.0)  # dim_a <= 0
    with pytest.raises(FileNotFoundError):
        n["LoadAudioProjectionLayer"]().load_projection_layer("nope.safetensors", "cuda:0")


def test_va_emotion_nodes(models):
    n = pkg.NODE_CLASS_MAPPINGS
    pipe, dim_e = n["LoadEmotionRecognitionModel"]().load_emotion_model("wav2vec-english-speech-emotion-recognition", "cuda:0")
    assert dim_e == 7 and pipe[2]["label2id"]["happy"] == 3
    a = torch.cat([W.synth_waveform(1.0, seed=12), W.synth_waveform(1.0, seed=13)])
    E = n["FloatExtractEmotionWithCustomModel"]()
    we, _ = E.extract_emotion_from_features(a, pipe, "none")
    assert we.shape == (2, 1, 7)
    assert float((we[:, 0] - O.audio2emotion_predict(models["sds"]["emo"], models["ecfg"], a)).abs().max()) < 2e-3
    we, _ = E.extract_emotion_from_features(a, pipe, "happy")
    assert torch.equal(we, torch.nn.functional.one_hot(torch.tensor(3), 7).float()[None, None].repeat(2, 1, 1))
    with pytest.raises(ValueError):
        E.extract_emotion_from_features(a[0], pipe, "none")
    # dynamic: 4.5 s in 2 s chunks -> 3 score vectors, nearest-upsampled to ceil(4.5 * 25) = 113 frames
    wav = torch.from_numpy(np.random.RandomState(3).standard_normal((1, 1, 72000)).astype(np.float32)) * 0.1
    we_dyn, _, seq = n["FloatExtractEmotionWithCustomModelDyn"]().extract_dynamic_emotion({"waveform": wav, "sample_rate": 16000}, pipe, 25.0, 2.0)
    assert seq.shape == (1, 3, 7) and we_dyn.shape == (1, 113, 7)
    assert torch.equal(we_dyn[0, 0], seq[0, 0]) and torch.equal(we_dyn[0, 112], seq[0, 2])
    chunk0 = pkg.host_models.preprocess_audio(wav[0, :, :32000], 16000, 16000)
    assert float((seq[0, 0] - O.audio2emotion_predict(models["sds"]["emo"], models["ecfg"], chunk0)[0]).abs().max()) < 2e-3
    with pytest.raises(ValueError):
        n["FloatExtractEmotionWithCustomModelDyn"]().extract_dynamic_emotion({"waveform": wav, "sample_rate": 8000}, pipe, 25.0, 2.0)


def test_va_offload_after_every_node(models, monkeypatch):
    """FLOAT_AMD_OFFLOAD=always: every VA node releases its operator's handle after the call - the reference's
    `with model_to_target(logger, model)` around every node body (nodes_vadv.py:107,192,275,347,437,520,697,807) - and the next
    call rebuilds it from the host weights the loader object keeps.  Same graph, resident vs released: bitwise the same tensors;
    after a released run no handle is left and the device memory the operators held is free again."""
    n = pkg.NODE_CLASS_MAPPINGS
    dev = "cuda:0"
    _, _, _, enc = n["LoadFloatEncoderModel"]().load_encoder_infer_arch("encoder.safetensors", dev, False)
    dec, _, _, _ = n["LoadFloatSynthesisModel"]().load_synthesis_infer_arch("decoder.safetensors", dev, 1, "[1, 3, 3, 1]", False)
    fmt, _, _, _ = n["LoadFMTModel"]().load_fmt_model("fmt.safetensors", dev, False, 7, 8, 2, 10, 25.0, 2.0)
    _, pipe = n["LoadWav2VecModel"]().load_float_wav2vec_model("wav2vec2-base-960h", dev)
    proj, _, _ = n["LoadAudioProjectionLayer"]().load_projection_layer("projection.safetensors", dev)
    emo, _ = n["LoadEmotionRecognitionModel"]().load_emotion_model("wav2vec-english-speech-emotion-recognition", dev)
    img = torch.from_numpy(np.random.RandomState(6).rand(1, SIZE, SIZE, 3).astype(np.float32))
    wav = W.synth_waveform(1.0, seed=8)[:, None]

    def graph():
        pipe_app, lam, _ = n["ApplyFloatEncoder"]().apply_encoder(img, enc)
        _, r_s = n["FloatGetIdentityReferenceVA"]().get_identity_reference_batch(lam, dec)
        feats, T, a_norm, _, _, _ = n["FloatAudioPreprocessAndFeatureExtract"]().extract_features_with_custom_model(
            {"waveform": wav, "sample_rate": 16000}, pipe, 25.0, False)
        (wa,) = n["FloatApplyAudioProjection"]().apply_projection(feats, proj)
        we, _ = n["FloatExtractEmotionWithCustomModel"]().extract_emotion_from_features(a_norm, emo, "none")
        r_d, _ = n["FloatSampleMotionSequenceRD_VA"]().sample_rd_sequence_va(
            r_s_latent=r_s, wa_latent=wa, we_latent=we, audio_num_frames=T, float_fmt_model=fmt, a_cfg_scale=2.0, r_cfg_scale=1.0,
            e_cfg_scale=1.0, include_r_cfg=False, nfe=3, torchdiffeq_ode_method="midpoint", ode_atol=1e-5, ode_rtol=1e-5,
            audio_dropout_prob=0.1, ref_dropout_prob=0.1, emotion_dropout_prob=0.1, fix_noise_seed=True, seed=15)
        frames, _ = n["ApplyFloatSynthesis"]().apply_synthesis(pipe_app, dec, r_d[:, :3] * 0.3)
        return [pipe_app["h_source"], r_s, wa, we, r_d, frames.clone()]

    ops = lambda: [enc, dec, fmt, emo[0]] + list(proj.get("_encoders", {}).values())  # noqa: E731
    resident = graph()
    assert all(op.resident for op in ops())
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    held = torch.cuda.mem_get_info()[0]
    monkeypatch.setenv("FLOAT_AMD_OFFLOAD", "always")
    released = graph()
    assert not any(op.resident for op in ops())
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    freed = torch.cuda.mem_get_info()[0] - held
    print("VA operators released: %.2f GB of device memory came back" % (freed / 2**30))
    assert freed > 0.5 * 2**30  # the small FMT / decoder / encoder / two wav2vec2 models of this fixture
    for a, b in zip(resident, released):
        assert torch.equal(a, b)
    again = graph()  # rebuilt from the released state
    for a, b in zip(resident, again):
        assert torch.equal(a, b)
