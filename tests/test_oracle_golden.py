"""Pin the CPU oracle: oracle/float_oracle.py must reproduce what the reference produced
(tests/golden/*.npz, made by tools/make_goldens.py from the imported reference) on the same
seeded weights and inputs.  fp32 vs fp32 with different summation orders: tolerance 1e-4 rel."""
import math

import numpy as np
import pytest
import torch

from oracle import float_oracle as O
from tests.util import golden, load_pkg, max_abs, rel_l2

pkg = load_pkg()
W, C = pkg.weights, pkg.config

TOL_REL = 1e-4  # fp32 restatement vs fp32 reference


@pytest.mark.parametrize("tag", ["small", "full"])
@pytest.mark.parametrize("case", ["nocfg", "cfg3", "cfg4", "cfg3dyn"])
def test_fmt_eval(tag, case):
    g = golden("fmt_eval_" + tag)
    cfg = C.small_fmt_config() if tag == "small" else C.FmtConfig()
    sd = W.synth_fmt_state(cfg, g["seed"])
    a, r, e, rc = [float(v) for v in g[case + "_scales"]]
    out = O.fmt_forward_cfv(sd, cfg, g["t"], g[case + "_x"], g[case + "_wa"], g[case + "_wr"], g[case + "_we"],
                            g[case + "_prev_x"], g[case + "_prev_wa"], g.get(case + "_prev_we"),
                            a, r, e, bool(rc))
    assert out.shape == g[case + "_out"].shape
    assert rel_l2(out, g[case + "_out"]) < TOL_REL


@pytest.mark.parametrize("tag", ["small_static", "small_dynamic", "full_static"])
def test_fmt_sample(tag):
    g = golden("fmt_sample_" + tag)
    cfg = C.FmtConfig() if tag.startswith("full") else C.small_fmt_config()
    sd = W.synth_fmt_state(cfg, g["seed"])
    r_d = O.sample_rd(sd, cfg, g["r_s"], g["wa"], g["we"], g["noise"], g["nfe"], g["a"], 1.0, g["e"])
    assert r_d.shape == (1, g["T"], cfg.dim_w)
    assert rel_l2(r_d, g["r_d"]) < TOL_REL


@pytest.mark.parametrize("tag,windows", [("config2", 1), ("config5", 2)])
def test_fmt_sample_full_length_prefix(tag, windows):
    """The full-length reference runs (250 / 750 frames at 50 evaluations per window) are 25 / 75 s of oracle time; the AR
    chain is causal, so the oracle is held to their first window(s) - config5's second window covers the dynamic-emotion
    prev_we hand-off at the headline grid - and the HIP path to the whole clips (tests/test_configs_gpu.py)."""
    from tests.util import sample_inputs
    g = golden("fmt_sample_" + tag)
    cfg = C.FmtConfig()
    inp = sample_inputs(cfg, g["seed"], g["T"], bool(g["dynamic"]), g["noise_seed"])
    n = windows * cfg.num_frames_for_clip
    we = inp["we"][:, :n] if g["dynamic"] else inp["we"]
    sd = W.synth_fmt_state(cfg, g["seed"])
    r_d = O.sample_rd(sd, cfg, inp["r_s"], inp["wa"][:, :n], we, inp["noise"][:windows], g["nfe"], g["a"], 1.0, g["e"])
    assert rel_l2(r_d, g["r_d"][:, :n]) < TOL_REL


def test_dynamic_we_requires_prev_we():
    cfg = C.small_fmt_config()
    sd = W.synth_fmt_state(cfg, 1)
    z = torch.zeros
    with pytest.raises(ValueError):
        O.fmt_forward(sd, cfg, torch.tensor([0.1]), z(1, 50, 128), z(1, 50, 128), z(1, 128), z(1, 50, 7),
                      z(1, 10, 128), z(1, 10, 128), None)


def test_dec_units():
    g = golden("dec_units")
    st = g["style"]
    for name, up in (("plain", False), ("up", True)):
        sd = {"c.weight": g["mc_%s_w" % name], "c.modulation.weight": g["mc_%s_mw" % name],
              "c.modulation.bias": g["mc_%s_mb" % name]}
        out = O.modulated_conv(g["mc_%s_x" % name], st, sd, "c", True, up)
        assert rel_l2(out, g["mc_%s_out" % name]) < TOL_REL
    sd = {"f.bias": g["tf_bias"], "f.conv.weight": g["tf_w"], "f.conv.modulation.weight": g["tf_mw"],
          "f.conv.modulation.bias": g["tf_mb"]}
    fw, bl, o3, grid = O.to_flow(g["tf_x"], st, g["tf_feat"], sd, "f", g["tf_prev"])
    assert max_abs(fw, g["tf_warp"]) < 1e-5 and max_abs(bl, g["tf_blend"]) < 1e-5
    assert max_abs(o3, g["tf_out"]) < 1e-5 and max_abs(grid, g["tf_grid"]) < 1e-6
    sdr = {"r.bias": g["tr_bias"], "r.conv.0.weight": g["tr_w"], "r.conv.1.bias": g["tr_b1"]}
    assert max_abs(O.to_rgb(g["tf_warp"], sdr, "r", g["tr_prev"]), g["tr_out"]) < 1e-5
    assert max_abs(O.direction({"direction.weight": g["dir_w"]}, g["dir_lam"]), g["dir_out"]) < 1e-5


def test_dec_64():
    g = golden("dec_64")
    sd = W.synth_decoder_state(64, seed=g["seed"])
    feats = W.synth_feats(64, seed=g["seed"])
    frames = O.decode_frames(sd, g["s_r"], g["r_d"], feats)
    assert frames.shape == g["frames"].shape == (3, 64, 64, 3)
    assert max_abs(frames, g["frames"]) < 1e-4
    raw, flow, _ = O.synthesis(sd, g["s_r"] + g["r_d"][:, 0], feats, return_all=True)
    assert max_abs(raw, g["raw0"]) < 2e-4 and max_abs(flow, g["flow0"]) < 1e-5


def test_dec_512_lattice():
    g = golden("dec_512")
    sd = W.synth_decoder_state(512, seed=g["seed"])
    feats = W.synth_feats(512, seed=g["seed"])
    frames = O.decode_frames(sd, g["s_r"], g["r_d"][:, :1], feats)
    assert max_abs(frames[:, ::7, ::5], g["lattice"][:1]) < 1e-4
    assert max_abs(frames[:, 250:258], g["band"][:1]) < 1e-4
    assert abs(float(frames.mean()) - float(g["mean"][0])) < 1e-5


def test_euler_grid_and_chunking():
    # "S steps" = S evaluations = nfe - 1 (FLOAT.py:188,247)
    ts = O.euler_grid(11)
    assert len(ts) == 11 and float(ts[0]) == 0.0 and float(ts[-1]) == 1.0
    assert int(math.ceil(125 / 50)) == 3
    a = torch.arange(6.0).reshape(1, 3, 2)
    p = O.pad_replicate(a, 5)
    assert p.shape == (1, 5, 2) and torch.equal(p[0, 3], a[0, 2]) and torch.equal(p[0, 4], a[0, 2])


def test_e2e_config1():
    """BASELINE configs[0] (1 s audio, 25 frames, nfe = 10, fp32): the reference's sampler + decode loop chained."""
    g = golden("e2e_config1")
    cfg = C.FmtConfig()
    fsd = W.synth_fmt_state(cfg, g["seed"])
    dsd = W.synth_decoder_state(512, seed=g["seed"])
    feats = W.synth_feats(512, seed=g["seed"])
    r_d = O.sample_rd(fsd, cfg, g["r_s"], g["wa"], g["we"], g["noise"], 10, 2.0, 1.0, 1.0)
    assert rel_l2(r_d, g["r_d"]) < TOL_REL
    pick = [int(i) for i in g["pick"]]
    frames = O.decode_frames(dsd, g["s_r"], r_d[:, pick[:1]], feats)
    assert max_abs(frames[:, ::7, ::5], g["lattice"][:1]) < 2e-4


def _enc_image(seed, size):
    import numpy as np
    return torch.from_numpy(np.random.RandomState(seed).rand(1, 3, size, size).astype(np.float32)) * 2 - 1


@pytest.mark.parametrize("size", [64, 512])
def test_encoder(size):
    """Encoder.forward(img, None) + Encoder.fc + Direction (encoder.py:203-281)."""
    g = golden("enc_%d" % size)
    esd = W.synth_encoder_state(size, seed=g["seed"])
    dsd = W.synth_decoder_state(size, seed=g["seed"])
    s_r, feats, lam = O.encode_appearance(esd, _enc_image(g["seed"], size))
    assert rel_l2(s_r, g["s_r"]) < TOL_REL and rel_l2(lam, g["lam"]) < TOL_REL
    assert rel_l2(O.direction(dsd, lam), g["r_s"]) < TOL_REL
    assert len(feats) == int(math.log2(size)) - 2
    for i, f in enumerate(feats):
        st = int(g["feat%d_stride" % i])
        assert f.shape[-1] == 8 << i
        assert rel_l2(f[:, :, ::st, ::st], g["feat%d" % i]) < TOL_REL
        assert rel_l2(f.mean(dim=(2, 3)), g["feat%d_mean" % i]) < TOL_REL


@pytest.mark.parametrize("tag", ["small", "base"])
def test_audio_encoder(tag):
    """AudioEncoder.inference (FLOAT.py:370-375) on transformers' wav2vec2 (wav2vec2.py:33-98)."""
    g = golden("aud_" + tag)
    cfg = C.small_audio_config() if tag == "small" else C.AudioConfig()
    sd = W.synth_audio_state(cfg, seed=g["seed"])
    a = W.synth_waveform(g["seconds"], seed=g["seed"] + 1)
    wa = O.audio_encoder_inference(sd, cfg, a, int(g["T"]))
    assert wa.shape == g["wa"].shape and rel_l2(wa, g["wa"]) < TOL_REL
    # the interpolated feature sequence on its own (wav2vec2.py:100-121 feature_extract)
    need = int(g["T"]) * 640
    ap = a if a.shape[1] % need == 0 else torch.nn.functional.pad(a[:, None], (0, need - a.shape[1]), mode="replicate")[:, 0]
    f = O.linear_interpolation(O.wav2vec_features(sd, cfg, ap).transpose(1, 2), int(g["T"]))
    assert rel_l2(f[:, :, ::8], g["feat_interp"]) < TOL_REL


@pytest.mark.parametrize("tag", ["small", "xlsr"])
def test_speech_emotion(tag):
    """Audio2Emotion.predict_emotion (FLOAT.py:396-401; wav2vec2_ser.py:23-96): layer-norm feature extractor with conv
    bias, pre-LayerNorm encoder, mean pooling, classification head, softmax."""
    g = golden("emo_" + tag)
    if tag == "xlsr" and not __import__("os").environ.get("FLOAT_SLOW_TESTS"):
        pytest.skip("316 M parameters: ~40 s of CPU; set FLOAT_SLOW_TESTS=1 (the GPU suite covers this shape)")
    cfg = C.small_emotion_config() if tag == "small" else C.emotion_audio_config()
    sd = W.synth_audio_state(cfg, seed=g["seed"])
    scores = O.audio2emotion_predict(sd, cfg, W.synth_waveform(g["seconds"], seed=g["seed"] + 1))
    assert scores.shape == g["scores"].shape and max_abs(scores, g["scores"]) < 1e-5
    assert abs(float(scores.sum()) - 1.0) < 1e-5


def test_dec_units_kernel_shapes():
    """The unit-op fixture the HIP kernels are held to (dec_units_hip.npz: StyledConv / ToFlow / ToRGB of the reference at
    kernel shapes) must also be what the oracle computes from the regenerated inputs - the fixture carries outputs only."""
    from tests.util import seeded_normal as rnd
    g = golden("dec_units_hip")
    seed = g["seed"]
    style = rnd(seed + 1, 2, 512)
    names = ["plain8", "plain32", "up4", "up8", "up32"]
    for i, (name, (cin, cout, R, up)) in enumerate(zip(names, g["sc_cases"].tolist())):
        k = seed + 100 * (i + 1)
        sd = {"c.conv.weight": rnd(k + 2, 1, cout, cin, 3, 3), "c.conv.modulation.weight": rnd(k + 3, cin, 512),
              "c.conv.modulation.bias": 1 + rnd(k + 4, cin, std=0.1), "c.activate.bias": rnd(k + 5, 1, cout, 1, 1, std=0.1)}
        out = O.styled_conv(rnd(k + 6, 2, cin, R, R), style, sd, "c", bool(up))
        assert rel_l2(out, g["sc_%s_out" % name]) < 2e-6, name
    for j, (name, (Cc, R, prev)) in enumerate(zip(["c32", "c128"], g["fl_cases"].tolist())):
        k = seed + 1000 * (j + 1)
        fsd = {"f.bias": rnd(k + 6, 1, 3, 1, 1, std=0.1), "f.conv.weight": rnd(k + 7, 1, 3, Cc, 1, 1, std=0.3),
               "f.conv.modulation.weight": rnd(k + 8, Cc, 512), "f.conv.modulation.bias": 1 + rnd(k + 9, Cc, std=0.1)}
        rsd = {"r.bias": rnd(k + 13, 1, 3, 1, 1, std=0.1), "r.conv.0.weight": rnd(k + 14, 3, Cc, 1, 1),
               "r.conv.1.bias": rnd(k + 15, 1, 3, 1, 1, std=0.1)}
        x, feat = rnd(k + 10, 2, Cc, R, R), rnd(k + 11, 1, Cc, R, R)
        pflow = rnd(k + 12, 2, 3, R // 2, R // 2, std=0.5) if prev else None
        prgb = rnd(k + 16, 2, 3, R // 2, R // 2) if prev else None
        fw, bl, o3, _ = O.to_flow(x, style, feat.repeat(2, 1, 1, 1), fsd, "f", pflow)
        rgb = O.to_rgb(fw, rsd, "r", prgb)
        assert max_abs(o3, g["fl_%s_out" % name]) < 1e-5 and max_abs(bl, g["fl_%s_blend" % name]) < 1e-5
        assert max_abs(rgb, g["fl_%s_rgb" % name]) < 1e-5


@pytest.mark.parametrize("kind", ["warp", "range"])
def test_dec_stress_64(kind):
    """Stress fixtures of the reference Synthesis (full-range warp on white-noise features; styles of +-300 on activations of
    1e2..1e4) against the oracle: bit-identical op order, so the tolerance is the fp32 one."""
    g = golden("dec_stress_%s_64" % kind)
    sd, feats = W.stress_decoder(64, seed=g["seed"], kind=kind)
    raw = torch.cat([O.synthesis(sd, g["s_r"] + g["r_d"][:, t], feats) for t in range(g["r_d"].shape[1])])
    assert rel_l2(raw, g["raw"]) < TOL_REL
    assert g["ref_vs_f64_rel"] < (1e-3 if kind == "warp" else 1e-5)  # the recorded sensitivity the GPU limits scale with


def test_dec_channel_multiplier_2():
    g = golden("dec_cm2_128")
    sd = W.synth_decoder_state(128, seed=g["seed"], channel_multiplier=2)
    feats = W.synth_feats(128, seed=g["seed"], channel_multiplier=2)
    raw = torch.cat([O.synthesis(sd, g["s_r"] + g["r_d"][:, t], feats) for t in range(2)])
    assert rel_l2(raw, g["raw"]) < TOL_REL


def test_dec_blur_kernels():
    """Other 4-tap blur kernels than [1,3,3,1] (tests/golden/dec_blur.npz, reference StyledConv(blur_kernel=...) and a whole
    Synthesis whose checkpoint holds the kernel's buffers): the oracle's StyledConv with the kernel as constructor argument,
    and its Synthesis reading the checkpoint's `conv.blur.kernel` like the strictly loaded reference does."""
    from tests.util import seeded_normal as rnd
    g = golden("dec_blur")
    seed = g["seed"]
    style = rnd(seed + 1, 2, 512)
    for ki, bk in enumerate(g["kernels"].tolist()):
        for i, (name, (cin, cout, R, F)) in enumerate(zip(["up4", "up8", "up32"], g["sc_cases"].tolist())):
            k = seed + 100 * (i + 1)
            sd = {"c.conv.weight": rnd(k + 2, 1, cout, cin, 3, 3), "c.conv.modulation.weight": rnd(k + 3, cin, 512),
                  "c.conv.modulation.bias": 1 + rnd(k + 4, cin, std=0.1), "c.activate.bias": rnd(k + 5, 1, cout, 1, 1, std=0.1)}
            x = rnd(k + 6, F, cin, R, R)
            want = g["sc_k%d_%s_out" % (ki, name)]
            assert rel_l2(O.styled_conv(x, style[:F], sd, "c", True, blur_kernel=bk), want) < TOL_REL
            assert rel_l2(O.styled_conv(x, style[:F], sd, "c", True), want) > 0.1  # the default kernel is a different answer
    bk = g["kernels"].tolist()[1]
    sd = W.synth_decoder_state(128, seed=seed, blur_kernel=bk)
    feats = W.synth_feats(128, seed=seed)
    raw = torch.cat([O.synthesis(sd, g["s_r"] + g["r_d"][:, t], feats) for t in range(2)])
    assert rel_l2(raw, g["raw"]) < TOL_REL
    sd0 = W.synth_decoder_state(128, seed=seed)  # default buffers in the checkpoint: the constructor argument alone changes nothing
    raw0 = torch.cat([O.synthesis(sd0, g["s_r"] + g["r_d"][:, t], feats, blur_kernel=bk) for t in range(2)])
    assert rel_l2(raw0, g["raw"]) > 0.05


def test_fir_buffers_from_the_checkpoint():
    """FIR buffers other than make_kernel([1,3,3,1]) outside the up-sampling StyledConvs (tests/golden/fir_buffers.npz: the
    reference's Encoder / Synthesis after a strict load of weights.fir_buffer_states): the oracle's encoder Blur reads
    `conv2.0.kernel` / `skip.0.kernel` (asymmetric and non-separable kernels: the flip of upfirdn2d is pinned), its ToRGB / ToFlow
    Upsample reads `upsample.kernel`; with the default kernels the answers differ by 24 % / 41 %."""
    g = golden("fir_buffers")
    size, seed = int(g["size"]), int(g["seed"])
    esd, dsd = W.fir_buffer_states(size, seed)
    img = torch.from_numpy(np.random.RandomState(seed).rand(1, 3, size, size).astype(np.float32)) * 2 - 1
    s_r, feats, lam = O.encode_appearance(esd, img)
    assert rel_l2(s_r, g["enc_s_r"]) < TOL_REL and rel_l2(lam, g["enc_lam"]) < TOL_REL
    for i, f in enumerate(feats):
        st = int(g["enc_feat%d_stride" % i])
        assert rel_l2(f[:, :, ::st, ::st], g["enc_feat%d" % i]) < TOL_REL
        assert rel_l2(f.mean(dim=(2, 3)), g["enc_feat%d_mean" % i]) < TOL_REL
    d_feats = O.encode_appearance(W.synth_encoder_state(size, seed=seed), img)[1]
    assert rel_l2(d_feats[0], g["enc_feat0"]) > 0.05  # make_kernel([1,3,3,1]) is a different encoder
    dfeats = W.synth_feats(size, seed=seed)
    raw = torch.cat([O.synthesis(dsd, g["dec_s_r"] + g["dec_r_d"][:, t], dfeats) for t in range(2)])
    assert rel_l2(raw, g["dec_raw"]) < TOL_REL
    dflt = torch.cat([O.synthesis(W.synth_decoder_state(size, seed=seed), g["dec_s_r"] + g["dec_r_d"][:, t], dfeats) for t in range(2)])
    assert rel_l2(dflt, g["dec_raw"]) > 0.1
