"""CPU oracle for the FLOAT hot path (FMT Euler sampling loop + Synthesis decoder) and the rows it was widened into
(appearance encoder, wav2vec2 audio conditioning, speech-emotion recogniser: SURVEY.md section 8f).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg as the checker / baseline.  The product package never imports it and
fails loudly when its HIP library is missing.

It is an independent restatement (torch-CPU tensor algebra, fp32 by default, fp64 on
request) of the reference's algorithm; every function cites the reference lines it
follows.  Pinning: tests/test_oracle_golden.py checks it against fixtures under
tests/golden/ that tools/make_goldens.py produced by running the reference itself
(imported from /root/reference in the build container) on the same seeded weights and
inputs; when /root/reference is present the same test also runs the reference live.

Third-party arithmetic that is not in /root/reference:
  * torchdiffeq (unpinned, requirements.txt:3) - only `method='euler'` on a fixed grid is
    restated: y_{i+1} = y_i + (t_{i+1}-t_i) f(t_i, y_i).  Parity with torchdiffeq itself
    is unpinned (the package is absent); the stand-in used to make the goldens implements
    the same published rule.
  * timm>=1.0.9 Mlp = Linear -> GELU(tanh) -> Linear (FMT.py:160-162); unpinned likewise.
  * transformers>=4 (requirements.txt:8; 5.15.0 where the goldens were made) - the wav2vec2 modules under the reference's
    Wav2VecModel / Wav2Vec2ForSpeechClassification.  Restated from the package's module definitions and PINNED through
    the reference's own AudioEncoder, which instantiates that package (aud_*.npz, 4e-6), and, for the speech-emotion
    model, through transformers' Wav2Vec2Model + the reference's Wav2Vec2ClassificationHead (emo_*.npz, 1e-7).
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- FMT


def timestep_embedding(t, dim=256, max_period=10000.0):
    """[cos(t f_k), sin(t f_k)], f_k = exp(-ln(max_period) k / half)  (FMT.py:107-126)."""
    half = dim // 2
    k = torch.arange(half, dtype=torch.float32)
    freqs = torch.exp(-math.log(max_period) * k / half).to(t.dtype)
    args = t.reshape(-1, 1) * freqs[None, :]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


_EMULATE = None  # torch.bfloat16 / torch.float16: round GEMM operands like the 16-bit HIP path does


def set_emulation(dt):
    """Diagnostic only: make every Linear round its input and weight to `dt` first (fp32
    accumulate), which is where the HIP kernels round.  Separates precision effects from logic
    errors when a parity test fails.  None restores exact arithmetic."""
    global _EMULATE
    _EMULATE = dt


def _q(t):
    return t if _EMULATE is None else t.to(_EMULATE).to(t.dtype)


def _linear(x, sd, name):
    return _q(x) @ _q(sd[name + ".weight"].to(x.dtype)).T + sd[name + ".bias"].to(x.dtype)


def _silu(x):
    return x * torch.sigmoid(x)


def _gelu_tanh(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def _layernorm(x, eps=1e-6):
    """No affine, biased variance (FMT.py:157,159,185)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps)


def alignment_mask(n, window):
    """enc_dec_mask(n, n, 1, expansion=window) (FMT.py:15-19, registered at 234-236): True = blocked.  Row i leaves
    columns max(0, i - window) .. i + window open, i.e. blocks |i - j| > window.  Pinned by fmt_tables.npz."""
    i = torch.arange(n)
    return (i[:, None] - i[None, :]).abs() > window


def sinusoid_table(n_position, d_hid):
    """get_sinusoid_encoding_table (FMT.py:22-40, copied into pos_embed at 249-250): angle(p, j) = p / 10000^(2 (j//2) / d)
    evaluated in python floats (float64), cast to fp32 by torch.Tensor(...), then sin on even / cos on odd columns in fp32.
    Pinned bit for bit by fmt_tables.npz."""
    p = torch.arange(n_position, dtype=torch.float64)[:, None]
    j = torch.arange(d_hid, dtype=torch.float64)[None, :]
    ang = (p / torch.pow(torch.tensor(10000.0, dtype=torch.float64), 2.0 * torch.floor(j / 2.0) / d_hid)).float()
    tab = ang.clone()
    tab[:, 0::2] = torch.sin(ang[:, 0::2])
    tab[:, 1::2] = torch.cos(ang[:, 1::2])
    return tab


def band_attention(q, k, v, window):
    """softmax(q k^T / sqrt(d) + mask) v with mask = -inf where |i-j| > window
    (FMT.py:15-19, 75-80).  q,k,v: (B, H, N, d)."""
    n = q.shape[-2]
    blocked = alignment_mask(n, window)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    s = s.masked_fill(blocked, float("-inf"))
    return torch.softmax(s, dim=-1) @ v


def fmt_forward(sd, cfg, t, x, wa, wr, we, prev_x, prev_wa, prev_we=None, dtype=torch.float32):
    """FlowMatchingTransformer.forward with train=False (FMT.py:277-340).
    t: (1,) or scalar; x, wa: (B,L,dim); wr: (B,dim_w); we: (B,1|L,dim_e);
    prev_*: (B,L',.).  Returns (B, L'+L, dim_w)."""
    cv = lambda a: a.to(dtype)  # noqa: E731
    t = cv(torch.as_tensor(t)).reshape(-1)
    x, wa, wr, we, prev_x, prev_wa = cv(x), cv(wa), cv(wr), cv(we), cv(prev_x), cv(prev_wa)
    dynamic_we = we.shape[1] > 1
    if dynamic_we and prev_we is None:
        raise ValueError("`we` is dynamic (T>1), but prev_we was not provided with prev_x/prev_wa.")
    # time embedding MLP (FMT.py:128-131,294)
    te = _linear(timestep_embedding(t), sd, "t_embedder.mlp.0")
    te = _linear(_silu(te), sd, "t_embedder.mlp.2")[:, None, :]
    # sequence assembly (FMT.py:312-326)
    x = torch.cat([prev_x, x], dim=1)
    wa = torch.cat([prev_wa, wa], dim=1)
    n = wa.shape[1]
    if dynamic_we:
        we = torch.cat([cv(prev_we), we], dim=1)
        if we.shape[1] != n:
            raise ValueError("Dynamic emotion latent `we` time dimension (%d) does not match "
                             "audio latent `wa` time dimension (%d)." % (we.shape[1], n))
    else:
        we = we.expand(-1, n, -1)
    # pos_embed: the checkpoint's when it carries one (the unified file does), else regenerated like the VA loader does
    # (nodes_vadv_loader.py:822-840 -> FMT.py:249-250)
    pos = sd["pos_embed"] if "pos_embed" in sd else sinusoid_table(n, sd["x_embedder.proj.weight"].shape[0])[None]
    h = _linear(x, sd, "x_embedder.proj") + cv(pos)
    c = torch.cat([wr[:, None, :].expand(-1, n, -1), wa, we], dim=-1)
    c = te + _linear(c, sd, "c_embedder")
    sc = _silu(c)
    B, N, D = h.shape
    H = cfg.num_heads
    for b in range(cfg.fmt_depth):
        p = "blocks.%d." % b
        mod = _linear(sc, sd, p + "adaLN_modulation.1")
        sh1, s1, g1, sh2, s2, g2 = mod.chunk(6, dim=-1)  # FMT.py:173
        a_in = _layernorm(h) * (1 + s1) + sh1
        qkv = _linear(a_in, sd, p + "attn.qkv").reshape(B, N, 3, H, D // H).permute(2, 0, 3, 1, 4)
        qkv = _q(qkv)
        o = band_attention(qkv[0], qkv[1], qkv[2], cfg.attention_window)
        o = o.transpose(1, 2).reshape(B, N, D)
        h = h + g1 * _linear(o, sd, p + "attn.proj")
        m_in = _layernorm(h) * (1 + s2) + sh2
        m = _linear(_gelu_tanh(_linear(m_in, sd, p + "mlp.fc1")), sd, p + "mlp.fc2")
        h = h + g2 * m
    sh, s = _linear(sc, sd, "decoder.adaLN_modulation.1").chunk(2, dim=-1)  # FMT.py:196
    return _linear(_layernorm(h) * (1 + s) + sh, sd, "decoder.linear")


def fmt_forward_cfv(sd, cfg, t, x, wa, wr, we, prev_x, prev_wa, prev_we=None,
                    a_cfg_scale=1.0, r_cfg_scale=1.0, e_cfg_scale=1.0, include_r_cfg=False,
                    dtype=torch.float32):
    """Classifier-free vector field (FMT.py:342-401)."""
    if a_cfg_scale == 1.0 and r_cfg_scale == 1.0 and e_cfg_scale == 1.0:
        return fmt_forward(sd, cfg, t, x, wa, wr, we, prev_x, prev_wa, prev_we, dtype)
    z = torch.zeros_like
    pw = prev_we
    if not include_r_cfg:
        # rows: [uncond | all | audio-only]  (FMT.py:360-373)
        out = fmt_forward(sd, cfg, t, torch.cat([x, x, x]), torch.cat([z(wa), wa, wa]),
                          torch.cat([wr, wr, wr]), torch.cat([z(we), we, z(we)]),
                          torch.cat([prev_x] * 3), torch.cat([prev_wa] * 3),
                          None if pw is None else torch.cat([z(pw), pw, z(pw)]), dtype)
        u, al, au = out.chunk(3, dim=0)
        return u + a_cfg_scale * (au - u) + e_cfg_scale * (al - au)
    # 4-way with a null reference row (FMT.py:380-399)
    out = fmt_forward(sd, cfg, t, torch.cat([x] * 4), torch.cat([z(wa), z(wa), wa, wa]),
                      torch.cat([z(wr), wr, wr, wr]), torch.cat([z(we), z(we), we, z(we)]),
                      torch.cat([prev_x] * 4), torch.cat([prev_wa] * 4),
                      None if pw is None else torch.cat([z(pw), z(pw), pw, z(pw)]), dtype)
    tu, u, al, au = out.chunk(4, dim=0)
    return tu + r_cfg_scale * (u - tu) + a_cfg_scale * (au - u) + e_cfg_scale * (al - au)


def euler_grid(nfe, dtype=torch.float32):
    """torch.linspace(0, 1, nfe) (FLOAT.py:188): nfe-1 evaluations, t=1 never evaluated."""
    return torch.linspace(0, 1, nfe, dtype=dtype)


def sample_chunk(sd, cfg, x0, wa_c, wr, we_c, prev_x, prev_wa, prev_we, nfe,
                 a_cfg_scale, r_cfg_scale, e_cfg_scale, include_r_cfg=False, dtype=torch.float32, method="euler"):
    """Fixed-grid solve over one 50-frame window (FLOAT.py:229-248, nodes_adv.py:629-659).
    `method`: torchdiffeq's fixed-step list (src/nodes/__init__.py:15-23).  torchdiffeq is absent here, so its
    published step rules are restated (parity with the package unpinned): euler; midpoint; rk4 = the 3/8-rule
    `rk4_alt_step_func`; heun2 / heun3 = their Butcher tableaux."""
    ts = euler_grid(nfe, dtype)
    x = x0.to(dtype)
    P = cfg.num_prev_frames

    def f(t, y):
        return fmt_forward_cfv(sd, cfg, t.reshape(1), y, wa_c, wr, we_c, prev_x, prev_wa, prev_we,
                               a_cfg_scale, r_cfg_scale, e_cfg_scale, include_r_cfg, dtype)[:, P:]

    for i in range(nfe - 1):
        t0, t1 = ts[i], ts[i + 1]
        dt = t1 - t0
        if method == "euler":
            x = x + dt * f(t0, x)
        elif method == "midpoint":
            k1 = f(t0, x)
            x = x + dt * f(t0 + 0.5 * dt, x + k1 * (0.5 * dt))
        elif method == "rk4":
            k1 = f(t0, x)
            k2 = f(t0 + dt / 3, x + dt * k1 / 3)
            k3 = f(t0 + dt * 2 / 3, x + dt * (k2 - k1 / 3))
            k4 = f(t1, x + dt * (k1 - k2 + k3))
            x = x + (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
        elif method == "heun2":
            k1 = f(t0, x)
            k2 = f(t0 + dt, x + dt * k1)
            x = x + dt * (0.5 * k1 + 0.5 * k2)
        elif method == "heun3":
            k1 = f(t0, x)
            k2 = f(t0 + dt / 3, x + dt * k1 / 3)
            k3 = f(t0 + dt * 2 / 3, x + dt * (k2 * 2 / 3))
            x = x + dt * (0.25 * k1 + 0.75 * k3)
        else:
            raise ValueError("unknown fixed-step method %r" % (method,))
    return x


def pad_replicate(a, length):
    """F.pad(..., mode='replicate') along time (FLOAT.py:226-227)."""
    if a.shape[1] >= length:
        return a
    return torch.cat([a, a[:, -1:].expand(-1, length - a.shape[1], -1)], dim=1)


def sample_rd(sd, cfg, r_s, wa, we, noise, nfe, a_cfg_scale=2.0, r_cfg_scale=1.0, e_cfg_scale=1.0,
              include_r_cfg=False, dtype=torch.float32, method="euler"):
    """The auto-regressive chunk loop (FLOAT.py:209-253; nodes_adv.py:578-694).
    r_s (B,512); wa (B,T,512); we (B,1,7) static or (B,T,7) dynamic; noise (n_chunks,B,50,512)
    is the explicit stand-in for the sequential torch.randn draws (FLOAT.py:215).
    Returns r_d (B,T,512)."""
    B, T, _ = wa.shape
    L, P = cfg.num_frames_for_clip, cfg.num_prev_frames
    n_chunks = int(math.ceil(T / L))
    dynamic = we.shape[1] > 1
    prev_x = torch.zeros(B, P, cfg.dim_w, dtype=dtype)
    prev_wa = torch.zeros(B, P, cfg.dim_a, dtype=dtype)
    prev_we = torch.zeros(B, P, cfg.dim_e, dtype=dtype)
    out = []
    for k in range(n_chunks):
        wa_c = pad_replicate(wa[:, k * L:(k + 1) * L].to(dtype), L)
        we_c = pad_replicate(we[:, k * L:(k + 1) * L].to(dtype), L) if dynamic else we.to(dtype)
        xs = sample_chunk(sd, cfg, noise[k], wa_c, r_s.to(dtype), we_c, prev_x, prev_wa,
                          prev_we if dynamic else None, nfe, a_cfg_scale, r_cfg_scale, e_cfg_scale,
                          include_r_cfg, dtype, method)
        out.append(xs)
        prev_x, prev_wa = xs[:, -P:], wa_c[:, -P:]
        if dynamic:
            prev_we = we_c[:, -P:]
    return torch.cat(out, dim=1)[:, :T]


# ------------------------------------------------------------------------- decoder

_SQRT2 = math.sqrt(2.0)


def fir_kernel(gain=1.0, dtype=torch.float32, taps=(1.0, 3.0, 3.0, 1.0)):
    """make_kernel: outer(k, k) / sum, [1,3,3,1] x [1,3,3,1] / 64 by default (styledecoder.py:39-44); x4 when up-sampling
    (:79,:118)."""
    k1 = torch.tensor([float(t) for t in taps], dtype=dtype)
    k = k1[:, None] * k1[None, :]
    return k / k.sum() * gain


def upfirdn(x, k, up=1, pad=(0, 0)):
    """Zero-insert by `up`, pad (pad0 before, pad1 after), correlate with the flipped FIR
    (styledecoder.py:16-32).  Depth-wise; x (B,C,H,W)."""
    B, C, H, W = x.shape
    if up > 1:
        z = x.new_zeros(B, C, H * up, W * up)
        z[:, :, ::up, ::up] = x
        x = z
    x = F.pad(x, [pad[0], pad[1], pad[0], pad[1]])
    w = torch.flip(k, [0, 1]).to(x.dtype)[None, None].expand(C, 1, -1, -1)
    return F.conv2d(x, w, groups=C)


def equal_linear(x, w, b):
    """x (W/sqrt(in))^T + b  (styledecoder.py:168,177; lr_mul = 1)."""
    return x @ (w.to(x.dtype) * (1.0 / math.sqrt(w.shape[1]))).T + b.to(x.dtype)


def modulated_conv(x, style, sd, prefix, demodulate=True, upsample=False, blur_kernel=(1, 3, 3, 1)):
    """ModulatedConv2d (styledecoder.py:238-272), one explicit conv per batch item.  blur_kernel: the constructor argument of the
    up-sampling form (styledecoder.py:197,205-213); its Blur pads (pad0, pad1) = ((p + 1) // 2 + 1, p // 2 + 1), p = len - 4."""
    w = sd[prefix + ".weight"].to(x.dtype)[0]  # (Cout, Cin, k, k)
    cout, cin, ks, _ = w.shape
    s = equal_linear(style, sd[prefix + ".modulation.weight"], sd[prefix + ".modulation.bias"])
    scale = 1.0 / math.sqrt(cin * ks * ks)
    outs = []
    for b in range(x.shape[0]):
        wb = scale * w * s[b][None, :, None, None]
        if demodulate:
            wb = wb * torch.rsqrt((wb ** 2).sum(dim=(1, 2, 3)) + 1e-8)[:, None, None, None]
        if upsample:
            # conv_transpose2d stride 2 (styledecoder.py:250-257) then Blur pad (1,1), gain 4
            y = F.conv_transpose2d(x[b:b + 1], wb.transpose(0, 1), stride=2, padding=0)
            # the Blur's kernel is a registered buffer: after the strict load (nodes_vadv_loader.py:632) it is the CHECKPOINT's
            # `<conv>.blur.kernel`, the constructor's make_kernel(blur_kernel) * 4 only where the state has no such key
            k2 = sd.get(prefix + ".blur.kernel")
            k2 = fir_kernel(4.0, x.dtype, blur_kernel) if k2 is None else k2.to(x.dtype)
            p = k2.shape[0] - 2 - 2
            y = upfirdn(y, k2, pad=((p + 1) // 2 + 1, p // 2 + 1))
        else:
            y = F.conv2d(x[b:b + 1], wb, padding=ks // 2)
        outs.append(y)
    return torch.cat(outs, dim=0)


def styled_conv(x, style, sd, prefix, upsample=False, blur_kernel=(1, 3, 3, 1)):
    """StyledConv with noise=None (styledecoder.py:320-325): modconv, + bias, lrelu(0.2) sqrt2."""
    y = modulated_conv(x, style, sd, prefix + ".conv", True, upsample, blur_kernel)
    return F.leaky_relu(y + sd[prefix + ".activate.bias"].to(x.dtype), 0.2) * _SQRT2


def upsample2(x, k=None):
    """Upsample([1,3,3,1]) (styledecoder.py:74-90): zero-insert x2, pad (2,1), FIR gain 4.  k: the module's registered
    `upsample.kernel` buffer where the state holds one (the strict load, nodes_vadv_loader.py:632, puts the checkpoint's 4 x 4
    buffer over the constructor's make_kernel([1,3,3,1]) * 4; the padding stays the constructor's)."""
    return upfirdn(x, fir_kernel(4.0, x.dtype) if k is None else k.to(x.dtype), up=2, pad=(2, 1))


def to_rgb(x, sd, prefix, skip=None):
    """ToRGB (styledecoder.py:368-386): 1x1 EqualConv (no bias) -> FusedLeakyReLU(3) -> +bias
    -> + Upsample(skip)."""
    w = sd[prefix + ".conv.0.weight"].to(x.dtype)
    y = F.conv2d(x, w * (1.0 / math.sqrt(w.shape[1])))
    y = F.leaky_relu(y + sd[prefix + ".conv.1.bias"].to(x.dtype), 0.2) * _SQRT2
    y = y + sd[prefix + ".bias"].to(x.dtype)
    if skip is not None:
        y = y + upsample2(skip, sd.get(prefix + ".upsample.kernel"))
    return y


def to_flow(x, style, feat, sd, prefix, skip=None):
    """ToFlow (styledecoder.py:399-425).  Returns (feat_warp, blended, out3, grid)."""
    out = modulated_conv(x, style, sd, prefix + ".conv", demodulate=False)
    out = out + sd[prefix + ".bias"].to(x.dtype)
    if skip is not None:
        out = out + upsample2(skip, sd.get(prefix + ".upsample.kernel"))
    R = x.shape[2]
    lin = torch.linspace(-1, 1, R, dtype=torch.float64).to(torch.float32).to(x.dtype)  # np.linspace f64 -> f32
    gx = lin[None, :].expand(R, R)
    gy = lin[:, None].expand(R, R)
    ident = torch.stack([gx, gy], dim=-1)[None]  # channel 0 = x (width), 1 = y
    grid = torch.tanh(out[:, 0:2]).permute(0, 2, 3, 1) + ident
    mask = torch.sigmoid(out[:, 2:3])
    fw = F.grid_sample(feat.to(x.dtype).expand(x.shape[0], -1, -1, -1), grid, mode="bilinear",
                       padding_mode="zeros", align_corners=False) * mask
    return fw, fw + x * (1.0 - mask), out, grid


def synthesis(sd, latent, feats, dtype=torch.float32, return_all=False, blur_kernel=(1, 3, 3, 1)):
    """Synthesis.forward with alpha=None (styledecoder.py:497-534).
    latent (B,512) = s_r + r_d[:,t]; feats: 7 maps (1|B,C,R,R), R = 8..512.
    Returns rgb (B,3,S,S) (and the 64x64-level flow grid when return_all).
    blur_kernel reaches the StyledConvs only (styledecoder.py:470,486-488); ToRGB / ToFlow are built with their default."""
    latent = latent.to(dtype)
    B = latent.shape[0]
    x = sd["input.input"].to(dtype).expand(B, -1, -1, -1)
    x = styled_conv(x, latent, sd, "conv1")
    skip = None
    skip_flow = None
    flow64 = None
    inter = {}
    for li, feat in enumerate(feats):
        x = styled_conv(x, latent, sd, "convs.%d" % (2 * li), upsample=True, blur_kernel=blur_kernel)
        x = styled_conv(x, latent, sd, "convs.%d" % (2 * li + 1))
        fw, x, skip_flow, grid = to_flow(x, latent, feat, sd, "to_flows.%d" % li, skip_flow)
        skip = to_rgb(fw, sd, "to_rgbs.%d" % li, skip)
        if x.shape[2] == 64:
            flow64 = grid
        if return_all:
            inter[x.shape[2]] = dict(x=x, flow=skip_flow, rgb=skip)
    if return_all:
        return skip, flow64, inter
    return skip


def postprocess(img):
    """clamp(-1,1), (x+1)/2, CHW -> HWC (FLOAT.py:149-152)."""
    return ((img.clamp(-1, 1) + 1) / 2).permute(0, 2, 3, 1).contiguous()


def decode_frames(sd, s_r, r_d, feats, dtype=torch.float32, blur_kernel=(1, 3, 3, 1)):
    """decode_latent_into_processed_images (FLOAT.py:113-169): frame t uses latent
    s_r + r_d[:,t]; returns (T,H,W,3) fp32 in [0,1]."""
    T = r_d.shape[1]
    frames = []
    for t in range(T):
        img = synthesis(sd, s_r.to(dtype) + r_d[:, t].to(dtype), feats, dtype, blur_kernel=blur_kernel)
        frames.append(postprocess(img)[0].to(torch.float32))
    return torch.stack(frames, dim=0)


def direction(sd, lam, dtype=torch.float32):
    """Direction.forward (styledecoder.py:434-444): Q from QR(W + 1e-8); out = lam @ Q^T."""
    q, _ = torch.linalg.qr(sd["direction.weight"].to(dtype) + 1e-8)
    return lam.to(dtype) @ q.T


# ------------------------------------------------------------------------- appearance encoder
# EncoderApp / Encoder (encoder.py:183-281), once per clip.


def enc_blur(x, pad, k=None):
    """Blur([1,3,3,1], pad) of a down-sampling ConvLayer (encoder.py:59-75,160-166): upfirdn with
    up = down = 1, i.e. zero-pad by (pad0, pad1) and correlate with the FLIPPED 4x4 FIR (encoder.py:28-29) - make_kernel([1,3,3,1])
    = outer / 64, or the layer's registered `kernel` buffer `k` where the state holds one (any 4 x 4 values)."""
    return upfirdn(x, fir_kernel(1.0, x.dtype) if k is None else k.to(x.dtype), up=1, pad=pad)


def equal_conv2d(x, w, stride=1, padding=0):
    """EqualConv2d without bias (encoder.py:88-106): conv2d(x, W / sqrt(Cin k^2))."""
    w = w.to(x.dtype)
    return F.conv2d(x, w * (1.0 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3])), stride=stride, padding=padding)


def enc_conv_layer(x, sd, prefix, k, downsample=False, activate=True):
    """ConvLayer (encoder.py:146-181).  Sequential indices: plain = [conv, act]; down-sampling =
    [blur, conv, act] with pad0 = (p+1)//2, pad1 = p//2, p = (4 - 2) + (k - 1), stride 2, padding 0."""
    if downsample:
        p = (4 - 2) + (k - 1)
        x = enc_blur(x, ((p + 1) // 2, p // 2), sd.get(prefix + "0.kernel"))
        y = equal_conv2d(x, sd[prefix + "1.weight"], stride=2, padding=0)
        bias_key = prefix + "2.bias"
    else:
        y = equal_conv2d(x, sd[prefix + "0.weight"], stride=1, padding=k // 2)
        bias_key = prefix + "1.bias"
    if activate:  # FusedLeakyReLU: leaky_relu(x + b, 0.2) * sqrt(2)  (encoder.py:13-14,47-57)
        y = F.leaky_relu(y + sd[bias_key].to(x.dtype), 0.2) * _SQRT2
    return y


def enc_res_block(x, sd, prefix):
    """ResBlock (encoder.py:183-199): conv1 3x3, conv2 blur+3x3 stride 2, skip blur+1x1 stride 2 (no
    bias, no activation); (out + skip) / sqrt(2)."""
    o = enc_conv_layer(x, sd, prefix + "conv1.", 3)
    o = enc_conv_layer(o, sd, prefix + "conv2.", 3, downsample=True)
    s = enc_conv_layer(x, sd, prefix + "skip.", 1, downsample=True, activate=False)
    return (o + s) / _SQRT2


def encode_appearance(sd, img, dtype=torch.float32):
    """EncoderApp.forward + Encoder.fc (encoder.py:203-231, 234-247, 266-281 with input_target=None plus
    FLOAT.py's enc.fc call).  img (B,3,S,S) in [-1,1].
    Returns s_r (B,512), feats = res[::-1][2:] (resolutions 8..S, reference order), lambda (B,20)."""
    sd = {k: v.to(dtype) for k, v in sd.items()}
    x = img.to(dtype)
    p = "net_app.convs."
    n_res = 0
    while (p + "%d.conv1.0.weight" % (n_res + 1)) in sd:
        n_res += 1
    h = enc_conv_layer(x, sd, p + "0.", 1)
    res = [h]
    for i in range(1, n_res + 1):
        h = enc_res_block(h, sd, p + "%d." % i)
        res.append(h)
    h = equal_conv2d(h, sd[p + "%d.weight" % (n_res + 1)])  # EqualConv2d(C, 512, 4, padding=0, bias=False)
    res.append(h)
    s_r = h.squeeze(-1).squeeze(-1)
    feats = res[::-1][2:]
    lam = s_r
    i = 0
    while ("fc.%d.weight" % i) in sd:  # 4 x EqualLinear(512,512) + EqualLinear(512,20), no activation
        lam = equal_linear(lam, sd["fc.%d.weight" % i], sd["fc.%d.bias" % i])
        i += 1
    return s_r, feats, lam


# ------------------------------------------------------------------------- audio conditioning encoder
# SURVEY.md section 8f row 2.  The reference subclasses transformers' Wav2Vec2Model (wav2vec2.py:10) and only
# overrides forward to insert the interpolation (wav2vec2.py:66-68); the module arithmetic below restates the
# `transformers` package (requirements.txt:8 `transformers>=4`; 5.15.0 installed where the goldens were made):
# Wav2Vec2FeatureEncoder / GroupNormConvLayer / NoLayerNormConvLayer / FeatureProjection /
# PositionalConvEmbedding / SamePadLayer / Encoder / EncoderLayer / Attention (eager) / FeedForward, for
# feat_extract_norm="group", conv_bias=False, do_stable_layer_norm=False (model_configs/wav2vec2_base/config.json).


def _gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))  # ACT2FN["gelu"] = exact GELU


def _ln_affine(x, w, b, eps):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def wav2vec_features(sd, cfg, a, dtype=torch.float32):
    """feature_extractor(input_values) (wav2vec2.py:64-65): (B,N) -> (B,C,L)."""
    p = "wav2vec2.feature_extractor.conv_layers."
    h = a.to(dtype)[:, None]
    layer_norm = getattr(cfg, "feat_extract_norm", "group") == "layer"
    for i, s in enumerate(cfg.conv_stride):
        bias = sd[p + "%d.conv.bias" % i].to(dtype) if getattr(cfg, "conv_bias", False) else None
        h = F.conv1d(h, sd[p + "%d.conv.weight" % i].to(dtype), bias, stride=s)
        if layer_norm:  # Wav2Vec2LayerNormConvLayer: LayerNorm over the channels of each time step (eps 1e-5, affine)
            h = _ln_affine(h.transpose(1, 2), sd[p + "%d.layer_norm.weight" % i].to(dtype), sd[p + "%d.layer_norm.bias" % i].to(dtype),
                           1e-5).transpose(1, 2)
        elif i == 0:  # GroupNorm(num_groups = C): per-channel statistics over time, eps 1e-5, affine
            mu = h.mean(dim=2, keepdim=True)
            var = ((h - mu) ** 2).mean(dim=2, keepdim=True)
            h = (h - mu) / torch.sqrt(var + 1e-5) * sd[p + "0.layer_norm.weight"].to(dtype)[None, :, None] \
                + sd[p + "0.layer_norm.bias"].to(dtype)[None, :, None]
        h = _gelu(h)
    return h


def linear_interpolation(feat, seq_len):
    """wav2vec2.py:184-197: F.interpolate(features^T, size=seq_len, mode='linear', align_corners=True)^T, written
    out: src = t (L-1)/(T-1); out = (1-w) f[floor(src)] + w f[floor(src)+1].  feat (B,L,C)."""
    B, L, C = feat.shape
    if seq_len == 1:
        return feat[:, :1]
    scale = torch.tensor((L - 1) / (seq_len - 1), dtype=torch.float32)
    src = scale * torch.arange(seq_len, dtype=torch.float32)
    i0 = src.floor().long().clamp(max=L - 1)
    i1 = (i0 + 1).clamp(max=L - 1)
    w = (src - i0.float()).to(feat.dtype)[None, :, None]
    return (1 - w) * feat[:, i0] + w * feat[:, i1]


def pos_conv_weight(sd, dtype=torch.float32):
    """weight_norm(conv, dim=2): w = g * v / ||v||, the norm taken over (out, in) for every tap."""
    p = "wav2vec2.encoder.pos_conv_embed.conv."
    if p + "parametrizations.weight.original1" in sd:
        g, v = sd[p + "parametrizations.weight.original0"].to(dtype), sd[p + "parametrizations.weight.original1"].to(dtype)
    elif p + "weight_v" in sd:
        g, v = sd[p + "weight_g"].to(dtype), sd[p + "weight_v"].to(dtype)
    else:
        return sd[p + "weight"].to(dtype)
    return g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()


def wav2vec_encoder(sd, cfg, h, dtype=torch.float32):
    """Wav2Vec2Encoder.forward without mask: returns the list hidden_states[1:] (one (B,T,D) per layer)."""
    p = "wav2vec2.encoder."
    g = lambda k: sd[p + k].to(dtype)  # noqa: E731
    K = cfg.num_conv_pos_embeddings
    pos = F.conv1d(h.transpose(1, 2), pos_conv_weight(sd, dtype), g("pos_conv_embed.conv.bias"), padding=K // 2,
                   groups=cfg.num_conv_pos_embedding_groups)
    if K % 2 == 0:
        pos = pos[:, :, :-1]  # Wav2Vec2SamePadLayer
    h = h + _gelu(pos).transpose(1, 2)
    stable = getattr(cfg, "do_stable_layer_norm", False)  # Wav2Vec2EncoderStableLayerNorm: pre-LN layers, one LN at the end
    if not stable:
        h = _ln_affine(h, g("layer_norm.weight"), g("layer_norm.bias"), cfg.layer_norm_eps)
    H = cfg.num_attention_heads
    hd = cfg.hidden_size // H
    outs = []
    for l in range(cfg.num_hidden_layers):
        q = "layers.%d." % l
        lin = lambda x, n: x @ g(q + n + ".weight").T + g(q + n + ".bias")  # noqa: E731
        B, T, D = h.shape
        ln1 = lambda x: _ln_affine(x, g(q + "layer_norm.weight"), g(q + "layer_norm.bias"), cfg.layer_norm_eps)  # noqa: E731
        ln2 = lambda x: _ln_affine(x, g(q + "final_layer_norm.weight"), g(q + "final_layer_norm.bias"), cfg.layer_norm_eps)  # noqa: E731
        x = ln1(h) if stable else h
        qh = lin(x, "attention.q_proj").view(B, T, H, hd).transpose(1, 2)
        kh = lin(x, "attention.k_proj").view(B, T, H, hd).transpose(1, 2)
        vh = lin(x, "attention.v_proj").view(B, T, H, hd).transpose(1, 2)
        att = torch.softmax(qh @ kh.transpose(-1, -2) * hd ** -0.5, dim=-1) @ vh
        att = lin(att.transpose(1, 2).reshape(B, T, D), "attention.out_proj")
        ffn = lambda x: lin(_gelu(lin(x, "feed_forward.intermediate_dense")), "feed_forward.output_dense")  # noqa: E731
        if stable:  # Wav2Vec2EncoderLayerStableLayerNorm
            h = h + att
            h = h + ffn(ln2(h))
        else:       # Wav2Vec2EncoderLayer
            h = ln1(h + att)
            h = ln2(h + ffn(h))
        outs.append(h)
    if stable:
        outs[-1] = _ln_affine(outs[-1], g("layer_norm.weight"), g("layer_norm.bias"), cfg.layer_norm_eps)
    return outs


def audio_encoder_inference(sd, cfg, a, seq_len, sampling_rate=16000, fps=25.0, dtype=torch.float32):
    """AudioEncoder.inference (FLOAT.py:370-375): replicate-pad to a multiple of seq_len*sr/fps, wav2vec2 hidden
    states of every layer stacked per frame (FLOAT.py:345-352), audio_projection (FLOAT.py:338-342).
    a (B,N) normalised waveform -> wa (B,seq_len,dim_w)."""
    sd = {k: v.to(dtype) for k, v in sd.items()}
    need = int(seq_len * sampling_rate / fps)
    if a.shape[1] % need != 0:
        a = F.pad(a[:, None], (0, need - a.shape[1]), mode="replicate")[:, 0]
    f = wav2vec_features(sd, cfg, a, dtype).transpose(1, 2)
    f = linear_interpolation(f, seq_len)
    f = _ln_affine(f, sd["wav2vec2.feature_projection.layer_norm.weight"], sd["wav2vec2.feature_projection.layer_norm.bias"],
                   cfg.layer_norm_eps)
    h = f @ sd["wav2vec2.feature_projection.projection.weight"].T + sd["wav2vec2.feature_projection.projection.bias"]
    hs = wav2vec_encoder(sd, cfg, h, dtype)
    x = hs[-1] if cfg.only_last_features else torch.stack(hs, dim=1).permute(0, 2, 1, 3).reshape(h.shape[0], h.shape[1], -1)
    y = x @ sd["audio_projection.0.weight"].T + sd["audio_projection.0.bias"]
    y = _ln_affine(y, sd["audio_projection.1.weight"], sd["audio_projection.1.bias"], 1e-5)
    return y * torch.sigmoid(y)


def audio2emotion_predict(sd, cfg, a, dtype=torch.float32):
    """Audio2Emotion.predict_emotion (FLOAT.py:396-401) on Wav2Vec2ForSpeechClassification.forward
    (wav2vec2_ser.py:78-96): wav2vec2 on the un-interpolated features, mean over time (merged_strategy 'mean', :58-75),
    dense -> tanh -> out_proj (:31-38; dropout inert), softmax.  a (B,N) -> (B, num_labels)."""
    sd = {k: v.to(dtype) for k, v in sd.items()}
    f = wav2vec_features(sd, cfg, a, dtype).transpose(1, 2)
    f = _ln_affine(f, sd["wav2vec2.feature_projection.layer_norm.weight"], sd["wav2vec2.feature_projection.layer_norm.bias"],
                   cfg.layer_norm_eps)
    h = f @ sd["wav2vec2.feature_projection.projection.weight"].T + sd["wav2vec2.feature_projection.projection.bias"]
    h = wav2vec_encoder(sd, cfg, h, dtype)[-1].mean(dim=1)
    x = torch.tanh(h @ sd["classifier.dense.weight"].T + sd["classifier.dense.bias"])
    return torch.softmax(x @ sd["classifier.out_proj.weight"].T + sd["classifier.out_proj.bias"], dim=1)
