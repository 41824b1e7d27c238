"""Seeded synthetic weights in the reference's checkpoint layout.

No checkpoint can be downloaded where this is built or benchmarked, so parity tests,
smoke() and bench.py use weights from this deterministic synthesiser.  Keys and shapes
are exactly those of the reference state dicts (SURVEY.md appendix B;
reference FMT.py:218-236, styledecoder.py:447-495), so the same dict loads into the
reference modules (goldens) and into the HIP handles (product), and a real
`fmt.safetensors` / `decoder.safetensors` drops in unchanged.

The reference zero-initialises the adaLN and output layers of the FMT
(FMT.py:260-269), which makes an untrained FMT output identically 0; every tensor is
therefore re-randomised here with a scale that keeps activations O(1).
"""
import math
import zlib

import numpy as np
import torch


def _rng(seed, name):
    return np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 32))


def _randn(seed, name, shape, std=1.0):
    a = _rng(seed, name).standard_normal(size=shape).astype(np.float32) * np.float32(std)
    return torch.from_numpy(a)


def sinusoid_table(n_pos, d_hid):
    """pos_embed[p, j] = sin/cos(p / 10000^(2*(j//2)/d)), even j sin, odd j cos
    (reference FMT.py:22-40).  The angle is evaluated in float64 then cast, like the reference's
    python-float list -> torch.Tensor path, and sin / cos are torch's fp32 ones: bit-identical to the
    reference's table (tests/golden/fmt_tables.npz)."""
    p = np.arange(n_pos, dtype=np.float64)[:, None]
    j = np.arange(d_hid, dtype=np.float64)[None, :]
    ang = torch.from_numpy((p / np.power(10000.0, 2.0 * np.floor(j / 2.0) / d_hid)).astype(np.float32))
    tab = ang.clone()
    tab[:, 0::2] = torch.sin(ang[:, 0::2])
    tab[:, 1::2] = torch.cos(ang[:, 1::2])
    return tab


def band_mask(n, window):
    """True = blocked, |i-j| > window (reference FMT.py:15-19)."""
    i = torch.arange(n)
    return (i[:, None] - i[None, :]).abs() > window


def synth_fmt_state(cfg, seed=0):
    """cfg: FmtConfig.  Returns {key: fp32 tensor} with the `fmt.` prefix stripped."""
    D, W = cfg.dim_h, cfg.dim_w
    n_tok = cfg.num_prev_frames + cfg.num_frames_for_clip
    cdim = cfg.dim_w + cfg.dim_a + cfg.dim_e
    H = int(cfg.dim_h * cfg.mlp_ratio)
    sd = {}

    def lin(name, n_out, n_in, gain=1.0, bstd=0.02):
        sd[name + ".weight"] = _randn(seed, name + ".weight", (n_out, n_in), gain / math.sqrt(n_in))
        sd[name + ".bias"] = _randn(seed, name + ".bias", (n_out,), bstd)

    sd["pos_embed"] = sinusoid_table(n_tok, D)[None]
    sd["alignment_mask"] = band_mask(n_tok, cfg.attention_window)
    lin("x_embedder.proj", D, W)
    lin("t_embedder.mlp.0", D, 256)
    lin("t_embedder.mlp.2", D, D)
    lin("c_embedder", D, cdim)
    for b in range(cfg.fmt_depth):
        p = "blocks.%d." % b
        lin(p + "attn.qkv", 3 * D, D)
        lin(p + "attn.proj", D, D)
        lin(p + "mlp.fc1", H, D)
        lin(p + "mlp.fc2", D, H)
        lin(p + "adaLN_modulation.1", 6 * D, D, gain=0.5, bstd=0.1)
    lin("decoder.adaLN_modulation.1", 2 * D, D, gain=0.5, bstd=0.1)
    lin("decoder.linear", W, D)
    return sd


DEC_CHANNELS = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256, 128: 128, 256: 64, 512: 32, 1024: 16}


def synth_decoder_state(size=512, style_dim=512, motion_dim=20, seed=0, flow_gain=0.1, rgb_gain=0.4, channel_multiplier=1,
                        blur_kernel=(1, 3, 3, 1)):
    """Synthesis (motion-AE decoder) weights, prefix `motion_autoencoder.dec.` stripped.
    StyleGAN2-style layers carry their 1/sqrt(fan_in) equalised-lr scale in the forward
    pass, so N(0,1) weights are the natural scale.  `flow_gain` shrinks the ToFlow conv so
    the synthetic warp stays a moderate displacement (a real checkpoint's flows are smooth;
    unit-variance random flows make the fp32 reference itself differ from fp64 by 5e-2 per
    pixel); `rgb_gain` keeps most pixels inside the clamp range so errors are not hidden.
    `blur_kernel`: what Synthesis(blur_kernel=...) registers as the up-sampling StyledConvs' `conv.blur.kernel` buffers
    (styledecoder.py:209-213,486-488); ToRGB / ToFlow keep [1,3,3,1] (:489-491)."""
    sd = {}
    log_size = int(math.log2(size))
    DEC_CHANNELS = {r: (c if r <= 32 else c * channel_multiplier) for r, c in globals()["DEC_CHANNELS"].items()}  # styledecoder.py:457-467
    blur = torch.tensor([1.0, 3.0, 3.0, 1.0])
    k2 = blur[None, :] * blur[:, None]
    k2 = k2 / k2.sum()
    bk = torch.tensor([float(t) for t in blur_kernel])
    kc = bk[None, :] * bk[:, None]
    kc = kc / kc.sum()

    def styled(prefix, cin, cout, up):
        sd[prefix + ".conv.weight"] = _randn(seed, prefix + ".conv.weight", (1, cout, cin, 3, 3))
        if up:
            sd[prefix + ".conv.blur.kernel"] = (kc * 4.0).clone()
        sd[prefix + ".conv.modulation.weight"] = _randn(seed, prefix + ".conv.modulation.weight", (cin, style_dim))
        sd[prefix + ".conv.modulation.bias"] = 1.0 + _randn(seed, prefix + ".conv.modulation.bias", (cin,), 0.1)
        sd[prefix + ".noise.weight"] = torch.zeros(1)
        sd[prefix + ".activate.bias"] = _randn(seed, prefix + ".activate.bias", (1, cout, 1, 1), 0.1)

    def to_rgb(prefix, cin, up):
        sd[prefix + ".bias"] = _randn(seed, prefix + ".bias", (1, 3, 1, 1), 0.1)
        if up:
            sd[prefix + ".upsample.kernel"] = (k2 * 4.0).clone()
        sd[prefix + ".conv.0.weight"] = _randn(seed, prefix + ".conv.0.weight", (3, cin, 1, 1), rgb_gain)
        sd[prefix + ".conv.1.bias"] = _randn(seed, prefix + ".conv.1.bias", (1, 3, 1, 1), 0.1)

    def to_flow(prefix, cin):
        sd[prefix + ".bias"] = _randn(seed, prefix + ".bias", (1, 3, 1, 1), 0.1)
        sd[prefix + ".upsample.kernel"] = (k2 * 4.0).clone()
        sd[prefix + ".conv.weight"] = _randn(seed, prefix + ".conv.weight", (1, 3, cin, 1, 1), flow_gain)
        sd[prefix + ".conv.modulation.weight"] = _randn(seed, prefix + ".conv.modulation.weight", (cin, style_dim))
        sd[prefix + ".conv.modulation.bias"] = 1.0 + _randn(seed, prefix + ".conv.modulation.bias", (cin,), 0.1)

    sd["direction.weight"] = _randn(seed, "direction.weight", (512, motion_dim))
    sd["input.input"] = _randn(seed, "input.input", (1, DEC_CHANNELS[4], 4, 4))
    styled("conv1", DEC_CHANNELS[4], DEC_CHANNELS[4], False)
    to_rgb("to_rgb1", DEC_CHANNELS[4], False)
    cin = DEC_CHANNELS[4]
    for li, i in enumerate(range(3, log_size + 1)):
        cout = DEC_CHANNELS[2 ** i]
        styled("convs.%d" % (2 * li), cin, cout, True)
        styled("convs.%d" % (2 * li + 1), cout, cout, False)
        to_rgb("to_rgbs.%d" % li, cout, True)
        to_flow("to_flows.%d" % li, cout)
        cin = cout
    return sd


def synth_feats(size=512, seed=0, smooth=8, hi=0.02, channel_multiplier=1):
    """Appearance skip features in the reference order (encoder.py:220-231): spatial
    8,16,...,size with the decoder's channel map.  Smooth random fields (low-res noise,
    bilinearly enlarged) plus a little per-pixel noise, O(1) amplitude."""
    feats = []
    r = 8
    while r <= size:
        c = DEC_CHANNELS[r] * (channel_multiplier if r > 32 else 1)
        lo = max(2, r // smooth)
        base = _randn(seed, "feat%d.lo" % r, (1, c, lo, lo))
        f = torch.nn.functional.interpolate(base, size=(r, r), mode="bilinear", align_corners=False)
        f = f + _randn(seed, "feat%d.hi" % r, (1, c, r, r), hi)
        feats.append(f.contiguous())
        r *= 2
    return feats


def stress_decoder(size=512, seed=0, kind="warp", style_gain=100.0, act_gain=300.0):
    """Hard cases for the 16-bit decoder (tests/test_dec_stress_gpu.py, tools/make_goldens.py::gen_dec_stress), same layout
    as synth_decoder_state / synth_feats; returns (state dict, feats).
    kind "warp":  unit-gain ToFlow convs and white-noise skip features - a displacement field of the full tanh range
                  sampling a feature map with no smoothness at all (the tame goldens use flow_gain 0.1 on smooth features).
                  With random weights this map is chaotic: the reference's own fp32 output differs from an fp64 evaluation
                  by 2.6e-3 at 64 px and by 2.4 (rel-L2 0.25) at 512 px, so it is only a test case at 64 px;
    kind "warp_smooth": unit-gain ToFlow convs on the smooth features of the tame goldens - the 512-px form (reference fp32
                  vs fp64: max 0.21, rel-L2 2.8e-3; the fixtures record that sensitivity and the tests scale their limits by it).
    kind "warp_half": the same at flow gain 0.5 - between the tame goldens (0.1) and the chaotic unit gain: the second point of
                  the 16-bit tolerance table (how fast the fp16 frame error grows with the amplitude of the warp).
    kind "range": modulation weights x style_gain (styles reach +-3 style_gain = 300: StyleGAN2's fp16 overflow case) and
                  ConstantInput / skip features x act_gain (activations 1e2..1e4); the ToFlow convs (whose own modulation is
                  left alone) are scaled back by 1 / act_gain so that the warp stays a warp instead of saturating tanh."""
    if kind in ("warp", "warp_smooth", "warp_half"):
        sd = synth_decoder_state(size, seed=seed, flow_gain=0.5 if kind == "warp_half" else 1.0)
        feats = synth_feats(size, seed=seed, smooth=1, hi=0.5) if kind == "warp" else synth_feats(size, seed=seed)
        return sd, feats
    if kind != "range":
        raise ValueError(kind)
    sd = synth_decoder_state(size, seed=seed)
    feats = [f * act_gain for f in synth_feats(size, seed=seed)]
    for k in list(sd):
        if k.endswith(".conv.modulation.weight") and not k.startswith("to_flows."):
            sd[k] = sd[k] * style_gain
        if k.startswith("to_flows.") and k.endswith(".conv.weight"):
            sd[k] = sd[k] / (act_gain)
    sd["input.input"] = sd["input.input"] * act_gain
    return sd, feats


ENC_CHANNELS = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256, 128: 128, 256: 64, 512: 32, 1024: 16}


def synth_encoder_state(size=512, dim=512, dim_motion=20, seed=0):
    """Appearance/motion encoder weights in the reference layout (`motion_autoencoder.enc.` prefix
    stripped; encoder.py:203-247).  Equalised-lr layers: N(0,1) weights."""
    sd = {}
    blur = torch.tensor([1.0, 3.0, 3.0, 1.0])
    k2 = blur[None, :] * blur[:, None]
    k2 = k2 / k2.sum()
    p = "net_app.convs."
    c = ENC_CHANNELS[size]
    sd[p + "0.0.weight"] = _randn(seed, p + "0.0.weight", (c, 3, 1, 1))
    sd[p + "0.1.bias"] = _randn(seed, p + "0.1.bias", (1, c, 1, 1), 0.1)
    i = 1
    r = size
    while r > 4:
        co = ENC_CHANNELS[r // 2]
        q = p + "%d." % i
        sd[q + "conv1.0.weight"] = _randn(seed, q + "conv1.0.weight", (c, c, 3, 3))
        sd[q + "conv1.1.bias"] = _randn(seed, q + "conv1.1.bias", (1, c, 1, 1), 0.1)
        sd[q + "conv2.0.kernel"] = k2.clone()
        sd[q + "conv2.1.weight"] = _randn(seed, q + "conv2.1.weight", (co, c, 3, 3))
        sd[q + "conv2.2.bias"] = _randn(seed, q + "conv2.2.bias", (1, co, 1, 1), 0.1)
        sd[q + "skip.0.kernel"] = k2.clone()
        sd[q + "skip.1.weight"] = _randn(seed, q + "skip.1.weight", (co, c, 1, 1))
        c = co
        r //= 2
        i += 1
    sd[p + "%d.weight" % i] = _randn(seed, p + "%d.weight" % i, (dim, c, 4, 4))
    for j in range(4):
        sd["fc.%d.weight" % j] = _randn(seed, "fc.%d.weight" % j, (dim, dim))
        sd["fc.%d.bias" % j] = _randn(seed, "fc.%d.bias" % j, (dim,), 0.1)
    sd["fc.4.weight"] = _randn(seed, "fc.4.weight", (dim_motion, dim))
    sd["fc.4.bias"] = _randn(seed, "fc.4.bias", (dim_motion,), 0.1)
    return sd


def fir_buffer_states(size=64, seed=0):
    """(encoder state, decoder state) of synth_*_state(size, seed) whose FIR BUFFERS outside the up-sampling StyledConvs are not
    make_kernel([1,3,3,1]) - what a checkpoint may hold and the reference's strict load takes over (tests/golden/fir_buffers.npz):
    encoder `conv2.0.kernel` = make_kernel([1,2,4,1]) (asymmetric: pins upfirdn2d's flip), `skip.0.kernel` = a seeded non-separable
    positive 4 x 4 kernel of sum 1; decoder `to_rgbs.N.upsample.kernel` = make_kernel([1,2,4,1]) * 4, `to_flows.N.upsample.kernel` =
    4 * outer([1,3,3,1], [1,2,4,1]) / 64 (rank 1 with different factors per axis)."""
    def mk(a, b, gain):
        k = torch.tensor(a, dtype=torch.float32)[:, None] * torch.tensor(b, dtype=torch.float32)[None, :]
        return k / k.sum() * gain
    esd = synth_encoder_state(size, seed=seed)
    nsk = 0
    for k in sorted(esd):
        if k.endswith("conv2.0.kernel"):
            esd[k] = mk([1, 2, 4, 1], [1, 2, 4, 1], 1.0)
        elif k.endswith("skip.0.kernel"):
            r = torch.from_numpy(np.random.RandomState(seed + 50 + nsk).rand(4, 4).astype(np.float32)) + 0.1
            esd[k] = r / r.sum()
            nsk += 1
    dsd = synth_decoder_state(size, seed=seed)
    for k in sorted(dsd):
        if k.startswith("to_rgbs.") and k.endswith("upsample.kernel"):
            dsd[k] = mk([1, 2, 4, 1], [1, 2, 4, 1], 4.0)
        elif k.startswith("to_flows.") and k.endswith("upsample.kernel"):
            dsd[k] = mk([1, 3, 3, 1], [1, 2, 4, 1], 4.0)
    return esd, dsd


def scale_encoder_convs(sd, gain):
    """Every conv weight (4-D `*.weight`) of an encoder state times `gain`: the range-stress fixtures (tools/make_goldens.py,
    tests/test_enc_gpu.py) - activations grow by `gain` per conv layer."""
    if gain == 1.0:
        return sd
    return {k: (v * gain if k.endswith(".weight") and v.dim() == 4 else v) for k, v in sd.items()}


def synth_audio_state(cfg, seed=0):
    """AudioEncoder weights in the reference layout (`audio_encoder.` prefix stripped): transformers' Wav2Vec2Model
    keys under `wav2vec2.` (FLOAT.py:318,326) and `audio_projection.{0,1}` (FLOAT.py:338-342).  cfg: AudioConfig.
    Scales keep activations O(1) through the GELU conv stack and the post-LayerNorm encoder."""
    sd = {}
    p = "wav2vec2."
    cin = 1
    for i, (co, k) in enumerate(zip(cfg.conv_dim, cfg.conv_kernel)):
        q = p + "feature_extractor.conv_layers.%d." % i
        sd[q + "conv.weight"] = _randn(seed, q + "conv.weight", (co, cin, k), math.sqrt(2.0 / (cin * k)))
        if cfg.conv_bias:
            sd[q + "conv.bias"] = _randn(seed, q + "conv.bias", (co,), 0.05)
        if i == 0 or cfg.feat_extract_norm == "layer":
            sd[q + "layer_norm.weight"] = 1.0 + _randn(seed, q + "layer_norm.weight", (co,), 0.1)
            sd[q + "layer_norm.bias"] = _randn(seed, q + "layer_norm.bias", (co,), 0.1)
        cin = co
    D, C = cfg.hidden_size, cfg.conv_dim[-1]

    def lin(name, n, k, std=None):
        sd[name + ".weight"] = _randn(seed, name + ".weight", (n, k), std if std is not None else 1.0 / math.sqrt(k))
        sd[name + ".bias"] = _randn(seed, name + ".bias", (n,), 0.05)

    def ln(name, n):
        sd[name + ".weight"] = 1.0 + _randn(seed, name + ".weight", (n,), 0.1)
        sd[name + ".bias"] = _randn(seed, name + ".bias", (n,), 0.1)

    sd[p + "masked_spec_embed"] = _randn(seed, p + "masked_spec_embed", (D,))  # unused at inference (mask_time_indices=None)
    ln(p + "feature_projection.layer_norm", C)
    lin(p + "feature_projection.projection", D, C)
    q = p + "encoder.pos_conv_embed.conv."
    cpg, K = D // cfg.num_conv_pos_embedding_groups, cfg.num_conv_pos_embeddings
    v = _randn(seed, q + "v", (D, cpg, K))
    sd[q + "parametrizations.weight.original1"] = v
    sd[q + "parametrizations.weight.original0"] = (v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt() * (1.0 / math.sqrt(cpg * K))
                                                   * (1.0 + _randn(seed, q + "g", (1, 1, K), 0.1)))
    sd[q + "bias"] = _randn(seed, q + "bias", (D,), 0.05)
    ln(p + "encoder.layer_norm", D)
    for l in range(cfg.num_hidden_layers):
        q = p + "encoder.layers.%d." % l
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin(q + "attention." + nm, D, D)
        ln(q + "layer_norm", D)
        lin(q + "feed_forward.intermediate_dense", cfg.intermediate_size, D)
        lin(q + "feed_forward.output_dense", D, cfg.intermediate_size)
        ln(q + "final_layer_norm", D)
    if cfg.num_labels:  # Wav2Vec2ClassificationHead (wav2vec2_ser.py:23-38)
        lin("classifier.dense", D, D)
        lin("classifier.out_proj", cfg.num_labels, D, 2.0 / math.sqrt(D))
        return sd
    din = D if cfg.only_last_features else D * cfg.num_hidden_layers
    lin("audio_projection.0", cfg.dim_w, din)
    ln("audio_projection.1", cfg.dim_w)
    return sd


def synth_waveform(seconds, seed=1, sr=16000):
    """SURVEY.md 8d synthetic audio: 0.1 N(0,1) + 0.3 sin(2 pi 220 t), then zero-mean / unit-variance like
    Wav2Vec2FeatureExtractor(do_normalize=True) (generate.py:69-73).  Returns (1, N) fp32."""
    n = int(round(seconds * sr))
    t = np.arange(n, dtype=np.float64) / sr
    w = 0.1 * np.random.RandomState(seed).standard_normal(n) + 0.3 * np.sin(2 * np.pi * 220.0 * t)
    w = (w - w.mean()) / np.sqrt(w.var() + 1e-7)
    return torch.from_numpy(w.astype(np.float32))[None]
