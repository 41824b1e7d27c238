#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE (imported from /root/reference).

BUILD-CONTAINER ONLY (needs /root/reference; see tools/ref_import.py for the stand-ins).
Weights come from the build's own seeded synthesiser (comfyui-float_optimized_amd/weights.py),
so fixtures hold only seeds, inputs and the reference's outputs - never reference source.

    python tools/make_goldens.py            # writes tests/golden/*.npz, prints oracle deltas
"""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_import  # noqa: E402
from tests.util import load_pkg  # noqa: E402

pkg = load_pkg()
weights = pkg.weights
config = pkg.config
from oracle import float_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
torch.manual_seed(0)
torch.set_num_threads(8)


def rnd(seed, *shape, std=1.0):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32) * std)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print("  wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max()), float(
        (a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def ref_fmt(ns, cfg, seed):
    opt = ns.base_options.BaseOptions()
    for k in ("dim_w", "dim_a", "dim_e", "dim_h", "fmt_depth", "num_heads", "mlp_ratio",
              "num_prev_frames", "attention_window"):
        setattr(opt, k, getattr(cfg, k))
    opt.wav2vec_sec = cfg.num_frames_for_clip / opt.fps
    opt.rank = "cpu"
    m = ns.FMT.FlowMatchingTransformer(opt)
    sd = weights.synth_fmt_state(cfg, seed)
    # pos_embed and alignment_mask stay the REFERENCE's own (FMT.py:15-40, built at 234-236 / 249-250): the goldens
    # must not be produced with this repo's tables injected into the reference
    own = {k: v for k, v in sd.items() if k not in ("pos_embed", "alignment_mask")}
    res = m.load_state_dict(own, strict=False)
    assert sorted(res.missing_keys) == ["alignment_mask", "pos_embed"] and not res.unexpected_keys, res
    m.eval()
    return m, sd, opt


def gen_fmt_tables(ns):
    """enc_dec_mask (FMT.py:15-19) and get_sinusoid_encoding_table (FMT.py:22-40) of the reference itself for the temporal
    structures the tests use: (tokens, attention window) = (60,2) default, (45,3), (80,1), (70,2); table width 1024 and 256."""
    print("[fmt tables]")
    arrs = {}
    for ntok, win in ((60, 2), (45, 3), (80, 1), (70, 2)):
        arrs["mask_%d_%d" % (ntok, win)] = ns.FMT.enc_dec_mask(ntok, ntok, 1, expansion=win)
    for ntok, d in ((60, 1024), (80, 1024), (45, 256), (70, 512)):
        arrs["pos_%d_%d" % (ntok, d)] = ns.FMT.get_sinusoid_encoding_table(ntok, d)
    # and as the model registers them (FMT.py:234-236, 249-250)
    m, _, _ = ref_fmt(ns, config.FmtConfig(), 1)
    arrs["model_pos_embed"] = m.pos_embed.data
    arrs["model_alignment_mask"] = m.alignment_mask
    save("fmt_tables", **arrs)


def fmt_inputs(cfg, seed, dynamic=False):
    L, P = cfg.num_frames_for_clip, cfg.num_prev_frames
    d = dict(
        x=rnd(seed + 1, 1, L, cfg.dim_w),
        wa=rnd(seed + 2, 1, L, cfg.dim_a),
        wr=rnd(seed + 3, 1, cfg.dim_w),
        prev_x=rnd(seed + 5, 1, P, cfg.dim_w),
        prev_wa=rnd(seed + 6, 1, P, cfg.dim_a),
    )
    if dynamic:
        d["we"] = torch.softmax(rnd(seed + 4, 1, L, cfg.dim_e), -1)
        d["prev_we"] = torch.softmax(rnd(seed + 7, 1, P, cfg.dim_e), -1)
    else:
        d["we"] = torch.softmax(rnd(seed + 4, 1, 1, cfg.dim_e), -1)
        d["prev_we"] = None
    return d


def gen_fmt_eval(ns, cfg, tag, seed):
    print("[fmt eval %s]" % tag)
    m, sd, _ = ref_fmt(ns, cfg, seed)
    t = torch.tensor([0.37])
    cases = {
        "nocfg": dict(a=1.0, r=1.0, e=1.0, rc=False, dyn=False),
        "cfg3": dict(a=2.0, r=1.0, e=1.0, rc=False, dyn=False),
        "cfg4": dict(a=2.0, r=1.5, e=1.3, rc=True, dyn=False),
        "cfg3dyn": dict(a=1.0, r=1.0, e=3.0, rc=False, dyn=True),
    }
    arrs = dict(seed=seed, t=t)
    for cname, c in cases.items():
        inp = fmt_inputs(cfg, seed + (100 if c["dyn"] else 0), c["dyn"])
        with torch.no_grad():
            ref = m.forward_with_cfv(t, inp["x"], inp["wa"], inp["wr"], inp["we"], inp["prev_x"], inp["prev_wa"],
                                     inp["prev_we"], a_cfg_scale=c["a"], r_cfg_scale=c["r"], e_cfg_scale=c["e"],
                                     include_r_cfg=c["rc"])
        orc = O.fmt_forward_cfv(sd, cfg, t, inp["x"], inp["wa"], inp["wr"], inp["we"], inp["prev_x"], inp["prev_wa"],
                                inp["prev_we"], c["a"], c["r"], c["e"], c["rc"])
        orc64 = O.fmt_forward_cfv(sd, cfg, t, inp["x"], inp["wa"], inp["wr"], inp["we"], inp["prev_x"],
                                  inp["prev_wa"], inp["prev_we"], c["a"], c["r"], c["e"], c["rc"],
                                  dtype=torch.float64)
        print("  %-8s oracle32-ref max|d| %.3e rel %.3e | ref-oracle64 rel %.3e | |ref| rms %.3f" % (
            (cname,) + maxdiff(orc, ref) + (maxdiff(ref, orc64)[1], float(ref.pow(2).mean().sqrt()))))
        for k, v in inp.items():
            if v is not None:
                arrs["%s_%s" % (cname, k)] = v
        arrs["%s_scales" % cname] = np.array([c["a"], c["r"], c["e"], float(c["rc"])], dtype=np.float32)
        arrs["%s_out" % cname] = ref
    save("fmt_eval_%s" % tag, **arrs)


def gen_fmt_sample(ns, cfg, tag, seed, T, nfe, dynamic, a, e, noise_seed=15, compact=False):
    print("[fmt sample %s] T=%d nfe=%d dynamic=%s" % (tag, T, nfe, dynamic))
    m, sd, opt = ref_fmt(ns, cfg, seed)
    L, P = cfg.num_frames_for_clip, cfg.num_prev_frames
    wa = rnd(seed + 11, 1, T, cfg.dim_a)
    r_s = rnd(seed + 12, 1, cfg.dim_w)
    if dynamic:
        # per-2 s windows nearest-upsampled to T (nodes_vadv.py:829-840 output format)
        nwin = int(math.ceil(T / L))
        we_w = torch.softmax(rnd(seed + 13, 1, nwin, cfg.dim_e), -1)
        idx = torch.clamp((torch.arange(T).float() * nwin / T).long(), max=nwin - 1)
        we = we_w[:, idx]
    else:
        we = torch.softmax(rnd(seed + 13, 1, 1, cfg.dim_e), -1)
    n_chunks = int(math.ceil(T / L))
    # Noise exactly as the reference draws it: sequential randn from one seeded generator
    # (FLOAT.py:203-215, nodes_adv.py:606-607), made explicit for the fixture.
    g = torch.Generator("cpu")
    g.manual_seed(noise_seed)
    noise = torch.stack([torch.randn(1, L, cfg.dim_w, generator=g) for _ in range(n_chunks)])
    g.manual_seed(noise_seed)
    with torch.no_grad():
        ref = ns.nodes_adv._perform_ode_sampling_loop(
            m, r_s, wa, we, T, P, L, cfg.dim_w, nfe, "euler", 1e-5, 1e-5, torch.device("cpu"),
            a, 1.0, e, False, g)
    orc = O.sample_rd(sd, cfg, r_s, wa, we, noise, nfe, a, 1.0, e)
    print("  oracle32-ref max|d| %.3e rel %.3e ; |ref| rms %.3f" % (maxdiff(orc, ref) + (float(ref.pow(2).mean().sqrt()),)))
    if compact:
        # full-length configs: inputs are regenerated from the seed by the tests (rnd / softmax / draw_noise above are all
        # seeded numpy / torch-CPU generators); the fixture holds the reference's r_d only
        save("fmt_sample_%s" % tag, seed=seed, T=T, nfe=nfe, a=a, e=e, dynamic=int(dynamic), noise_seed=noise_seed, r_d=ref)
    else:
        save("fmt_sample_%s" % tag, seed=seed, T=T, nfe=nfe, a=a, e=e, wa=wa, r_s=r_s, we=we, noise=noise, r_d=ref)
    return dict(m=m, sd=sd, wa=wa, r_s=r_s, we=we, noise=noise, r_d=ref)


def gen_config_frames(ns, tag, run, seed, pick):
    """Frames of a full-length configuration: the reference's decode loop (FLOAT.py:113-169) on picked frames of the
    reference's own r_d, 512x512; lattice + band + mean like dec_512."""
    d, dsd = ref_dec(ns, 512, seed)
    feats = weights.synth_feats(512, seed=seed)
    s_r = rnd(seed + 4, 1, 512)
    fake = types.SimpleNamespace(motion_autoencoder=types.SimpleNamespace(dec=d), pbar=types.SimpleNamespace(update=lambda n: None))
    with torch.no_grad():
        frames = ns.FLOAT.FLOAT.decode_latent_into_processed_images(fake, s_r, feats, run["r_d"][:, pick])
    save("frames_%s" % tag, seed=seed, pick=np.array(pick), lattice=frames[:, ::7, ::5], band=frames[:, 250:258],
         mean=frames.mean(dim=(1, 2, 3)))


def ref_dec(ns, size, seed):
    d = ns.styledecoder.Synthesis(size, 512, 20)
    sd = weights.synth_decoder_state(size, seed=seed)
    d.load_state_dict(sd, strict=True)
    d.eval()
    return d, sd


def gen_dec_units(ns, seed):
    print("[decoder unit ops]")
    S = ns.styledecoder
    arrs = dict(seed=seed)
    style = rnd(seed + 1, 2, 512)
    arrs["style"] = style
    for name, cin, cout, R, up in (("plain", 32, 16, 8, False), ("up", 32, 16, 8, True)):
        mc = S.ModulatedConv2d(cin, cout, 3, 512, upsample=up)
        sd = {"c.weight": rnd(seed + 2, 1, cout, cin, 3, 3), "c.modulation.weight": rnd(seed + 3, cin, 512),
              "c.modulation.bias": 1 + rnd(seed + 4, cin, std=0.1)}
        mc.weight.data.copy_(sd["c.weight"])
        mc.modulation.weight.data.copy_(sd["c.modulation.weight"])
        mc.modulation.bias.data.copy_(sd["c.modulation.bias"])
        x = rnd(seed + 5, 2, cin, R, R)
        with torch.no_grad():
            ref = mc(x, style)
        orc = O.modulated_conv(x, style, sd, "c", True, up)
        print("  modconv %-5s oracle-ref max|d| %.3e rel %.3e" % ((name,) + maxdiff(orc, ref)))
        arrs.update({"mc_%s_x" % name: x, "mc_%s_out" % name: ref, "mc_%s_w" % name: sd["c.weight"],
                     "mc_%s_mw" % name: sd["c.modulation.weight"], "mc_%s_mb" % name: sd["c.modulation.bias"]})
    # ToFlow with a previous flow, then ToRGB with a previous rgb
    C, R = 32, 16
    tf = S.ToFlow(C, 512)
    sd = {"f.bias": rnd(seed + 6, 1, 3, 1, 1, std=0.1), "f.conv.weight": rnd(seed + 7, 1, 3, C, 1, 1, std=0.3),
          "f.conv.modulation.weight": rnd(seed + 8, C, 512), "f.conv.modulation.bias": 1 + rnd(seed + 9, C, std=0.1)}
    tf.bias.data.copy_(sd["f.bias"])
    tf.conv.weight.data.copy_(sd["f.conv.weight"])
    tf.conv.modulation.weight.data.copy_(sd["f.conv.modulation.weight"])
    tf.conv.modulation.bias.data.copy_(sd["f.conv.modulation.bias"])
    x = rnd(seed + 10, 2, C, R, R)
    feat = rnd(seed + 11, 2, C, R, R)
    pflow = rnd(seed + 12, 2, 3, R // 2, R // 2, std=0.5)
    with torch.no_grad():
        fw, bl, o3, grid = tf(x, style, feat, pflow)
    ofw, obl, oo3, ogrid = O.to_flow(x, style, feat, sd, "f", pflow)
    print("  toflow feat_warp %.3e blend %.3e out %.3e grid %.3e (max|d|)" % (
        maxdiff(ofw, fw)[0], maxdiff(obl, bl)[0], maxdiff(oo3, o3)[0], maxdiff(ogrid, grid)[0]))
    arrs.update(tf_x=x, tf_feat=feat, tf_prev=pflow, tf_warp=fw, tf_blend=bl, tf_out=o3, tf_grid=grid,
                tf_bias=sd["f.bias"], tf_w=sd["f.conv.weight"], tf_mw=sd["f.conv.modulation.weight"],
                tf_mb=sd["f.conv.modulation.bias"])
    tr = S.ToRGB(C, 512)
    sdr = {"r.bias": rnd(seed + 13, 1, 3, 1, 1, std=0.1), "r.conv.0.weight": rnd(seed + 14, 3, C, 1, 1),
           "r.conv.1.bias": rnd(seed + 15, 1, 3, 1, 1, std=0.1)}
    tr.bias.data.copy_(sdr["r.bias"])
    tr.conv[0].weight.data.copy_(sdr["r.conv.0.weight"])
    tr.conv[1].bias.data.copy_(sdr["r.conv.1.bias"])
    prgb = rnd(seed + 16, 2, 3, R // 2, R // 2)
    with torch.no_grad():
        ref = tr(fw, prgb)
    orc = O.to_rgb(fw, sdr, "r", prgb)
    print("  torgb  oracle-ref max|d| %.3e" % maxdiff(orc, ref)[0])
    arrs.update(tr_prev=prgb, tr_out=ref, tr_bias=sdr["r.bias"], tr_w=sdr["r.conv.0.weight"], tr_b1=sdr["r.conv.1.bias"])
    # Direction (QR) - once per clip
    dw = rnd(seed + 17, 512, 20)
    dm = S.Direction(20)
    dm.weight.data.copy_(dw)
    lam = rnd(seed + 18, 2, 20)
    with torch.no_grad():
        ref = dm(lam)
    orc = O.direction({"direction.weight": dw}, lam)
    print("  direction oracle-ref max|d| %.3e" % maxdiff(orc, ref)[0])
    arrs.update(dir_w=dw, dir_lam=lam, dir_out=ref)
    save("dec_units", **arrs)


def gen_dec(ns, size, seed, n_frames, sparse):
    print("[decoder size %d]" % size)
    d, sd = ref_dec(ns, size, seed)
    feats = weights.synth_feats(size, seed=seed)
    s_r = rnd(seed + 21, 1, 512)
    r_d = rnd(seed + 22, 1, n_frames, 512, std=0.5)
    # the reference's own decode loop + post-process, called unbound (FLOAT.py:113-169)
    FL = ns.FLOAT.FLOAT
    fake = types.SimpleNamespace(motion_autoencoder=types.SimpleNamespace(dec=d),
                                 pbar=types.SimpleNamespace(update=lambda n: None))
    with torch.no_grad():
        frames = FL.decode_latent_into_processed_images(fake, s_r, feats, r_d)
        raw0, flow0 = d(s_r + r_d[:, 0], None, feats)
    orc = O.decode_frames(sd, s_r, r_d, feats)
    orc64 = O.decode_frames(sd, s_r, r_d, feats, dtype=torch.float64)
    print("  frames oracle32-ref max|d| %.3e ; ref-oracle64 max|d| %.3e ; frame mean %.3f std %.3f sat %.3f" % (
        maxdiff(orc, frames)[0], maxdiff(frames, orc64)[0], float(frames.mean()), float(frames.std()),
        float(((frames == 0) | (frames == 1)).float().mean())))
    arrs = dict(seed=seed, size=size, s_r=s_r, r_d=r_d, raw0_mean=float(raw0.mean()), raw0_std=float(raw0.std()))
    if sparse:
        # full 512x512 frames are 3 MB each: keep a strided lattice that hits every pixel parity,
        # a full-resolution band, and per-frame statistics
        arrs["lattice"] = frames[:, ::7, ::5]
        arrs["band"] = frames[:, 250:258]
        arrs["mean"] = frames.mean(dim=(1, 2, 3))
        arrs["sqmean"] = frames.pow(2).mean(dim=(1, 2, 3))
        arrs["raw0_lattice"] = raw0[0][:, ::7, ::5]
    else:
        arrs["frames"] = frames
        arrs["raw0"] = raw0
        arrs["flow0"] = flow0
    save("dec_%d" % size, **arrs)


def gen_dec_units_hip(ns, seed):
    """Unit ops at shapes the HIP kernels take (channels in multiples of 32), one case per kernel route of
    float_dec_debug_styled_conv / float_dec_debug_flow_level.  Inputs are regenerated from the seeds by the tests
    (tests/test_dec_units_gpu.py); the fixture holds the reference modules' outputs."""
    print("[decoder unit ops, kernel shapes]")
    S = ns.styledecoder
    arrs = dict(seed=seed)
    F = 2
    style = rnd(seed + 1, F, 512)
    cases = (("plain8", 64, 32, 8, False), ("plain32", 64, 64, 32, False), ("up4", 64, 32, 4, True), ("up8", 64, 32, 8, True),
             ("up32", 32, 32, 32, True))
    arrs["sc_cases"] = np.array([[c[1], c[2], c[3], int(c[4])] for c in cases])
    for i, (name, cin, cout, R, up) in enumerate(cases):
        sc = S.StyledConv(cin, cout, 3, 512, upsample=up)
        k = seed + 100 * (i + 1)
        w, mw, mb, ab = rnd(k + 2, 1, cout, cin, 3, 3), rnd(k + 3, cin, 512), 1 + rnd(k + 4, cin, std=0.1), rnd(k + 5, 1, cout, 1, 1, std=0.1)
        sc.conv.weight.data.copy_(w)
        sc.conv.modulation.weight.data.copy_(mw)
        sc.conv.modulation.bias.data.copy_(mb)
        sc.activate.bias.data.copy_(ab.reshape(sc.activate.bias.shape))
        sc.noise.weight.data.zero_()
        x = rnd(k + 6, F, cin, R, R)
        with torch.no_grad():
            ref = sc(x, style)
        sd = {"c.conv.weight": w, "c.conv.modulation.weight": mw, "c.conv.modulation.bias": mb, "c.activate.bias": ab}
        orc = O.styled_conv(x, style, sd, "c", up)
        print("  styled_conv %-8s oracle-ref max|d| %.3e rel %.3e  out std %.3f" % ((name,) + maxdiff(orc, ref) + (float(ref.std()),)))
        arrs["sc_%s_out" % name] = ref
    for j, (name, C, R, prev) in enumerate((("c32", 32, 16, True), ("c128", 128, 32, False))):
        k = seed + 1000 * (j + 1)
        tf, tr = S.ToFlow(C, 512), S.ToRGB(C, 512)
        fb, fw = rnd(k + 6, 1, 3, 1, 1, std=0.1), rnd(k + 7, 1, 3, C, 1, 1, std=0.3)
        fmw, fmb = rnd(k + 8, C, 512), 1 + rnd(k + 9, C, std=0.1)
        rb, rw, rb1 = rnd(k + 13, 1, 3, 1, 1, std=0.1), rnd(k + 14, 3, C, 1, 1), rnd(k + 15, 1, 3, 1, 1, std=0.1)
        tf.bias.data.copy_(fb)
        tf.conv.weight.data.copy_(fw)
        tf.conv.modulation.weight.data.copy_(fmw)
        tf.conv.modulation.bias.data.copy_(fmb)
        tr.bias.data.copy_(rb)
        tr.conv[0].weight.data.copy_(rw)
        tr.conv[1].bias.data.copy_(rb1.reshape(tr.conv[1].bias.shape))
        x, feat = rnd(k + 10, F, C, R, R), rnd(k + 11, 1, C, R, R)
        pflow = rnd(k + 12, F, 3, R // 2, R // 2, std=0.5) if prev else None
        prgb = rnd(k + 16, F, 3, R // 2, R // 2) if prev else None
        with torch.no_grad():
            fwarp, blend, o3, grid = tf(x, style, feat.repeat(F, 1, 1, 1), pflow)
            rgb = tr(fwarp, prgb)
        print("  flow level %-5s out std %.3f blend std %.3f rgb std %.3f" % (name, float(o3.std()), float(blend.std()), float(rgb.std())))
        arrs.update({"fl_%s_out" % name: o3, "fl_%s_blend" % name: blend, "fl_%s_rgb" % name: rgb})
    arrs["fl_cases"] = np.array([[32, 16, 1], [128, 32, 0]])
    save("dec_units_hip", **arrs)


def gen_dec_stress(ns, size, seed, kind, sparse):
    """Hard cases for the 16-bit decoder (weights.stress_decoder): the reference Synthesis on a full-range warp over white-noise
    features, and on styles of +-300 with activations of 1e2..1e4 (where an unnormalised fp16 decoder overflows)."""
    print("[decoder stress %s, size %d]" % (kind, size))
    sd, feats = weights.stress_decoder(size, seed=seed, kind=kind)
    d = ns.styledecoder.Synthesis(size, 512, 20)
    d.load_state_dict(sd, strict=True)
    d.eval()
    s_r = rnd(seed + 21, 1, 512)
    r_d = rnd(seed + 22, 1, 2, 512, std=0.5)
    raws = []
    with torch.no_grad():
        for t in range(r_d.shape[1]):
            raws.append(d(s_r + r_d[:, t], None, feats)[0])
    raw = torch.cat(raws)
    o32 = torch.cat([O.synthesis(sd, s_r + r_d[:, t], feats) for t in range(r_d.shape[1])])
    o64 = torch.cat([O.synthesis(sd, s_r + r_d[:, t], feats, dtype=torch.float64) for t in range(r_d.shape[1])])
    print("  raw std %.3e max %.3e ; oracle32-ref max|d| %.3e ; ref-oracle64 max|d| %.3e rel %.3e" % (
        float(raw.std()), float(raw.abs().max()), maxdiff(o32, raw)[0], maxdiff(raw, o64.float())[0], maxdiff(raw, o64.float())[1]))
    arrs = dict(seed=seed, size=size, s_r=s_r, r_d=r_d, raw_std=float(raw.std()), ref_vs_f64_max=maxdiff(raw, o64.float())[0],
                ref_vs_f64_rel=maxdiff(raw, o64.float())[1])
    if sparse:
        arrs["raw_lattice"] = raw[:, :, ::7, ::5]
        arrs["raw_band"] = raw[:, :, 250:258]
    else:
        arrs["raw"] = raw
    save("dec_stress_%s_%d" % (kind, size), **arrs)


def gen_dec_cm2(ns, seed=1800, size=128):
    """Synthesis with channel_multiplier=2 (styledecoder.py:447-467: 512 / 256 channels at 64 / 128 px), two frames."""
    print("[decoder channel_multiplier 2, size %d]" % size)
    d = ns.styledecoder.Synthesis(size, 512, 20, channel_multiplier=2)
    sd = weights.synth_decoder_state(size, seed=seed, channel_multiplier=2)
    d.load_state_dict(sd, strict=True)
    d.eval()
    feats = weights.synth_feats(size, seed=seed, channel_multiplier=2)
    s_r = rnd(seed + 21, 1, 512)
    r_d = rnd(seed + 22, 1, 2, 512, std=0.5)
    with torch.no_grad():
        raw = torch.cat([d(s_r + r_d[:, t], None, feats)[0] for t in range(2)])
    orc = torch.cat([O.synthesis(sd, s_r + r_d[:, t], feats) for t in range(2)])
    print("  raw std %.3f ; oracle-ref max|d| %.3e" % (float(raw.std()), maxdiff(orc, raw)[0]))
    save("dec_cm2_%d" % size, seed=seed, size=size, s_r=s_r, r_d=r_d, raw=raw)


def gen_dec_blur(ns, seed=1900, size=128):
    """Synthesis(blur_kernel=...) with other 4-tap kernels than [1,3,3,1] (the LoadFloatSynthesisModel widget,
    nodes_vadv_loader.py:567-611; styledecoder.py:448,486-488: only the StyledConvs receive it): the three up-conv kernel
    routes of float_dec_debug_styled_conv for a symmetric and an ASYMMETRIC kernel (upfirdn2d convolves with the flipped
    kernel, styledecoder.py:28-29: the asymmetric case pins the orientation), and two frames of a whole 128-px Synthesis whose
    checkpoint holds that kernel's `conv.blur.kernel` buffers (the strict load puts the checkpoint's buffers over the
    constructor's kernel, nodes_vadv_loader.py:632 - asserted below)."""
    print("[decoder blur kernels]")
    S = ns.styledecoder
    arrs = dict(seed=seed, size=size)
    style = rnd(seed + 1, 2, 512)
    kernels = ([1, 2, 2, 1], [1, 2, 4, 1])
    cases = (("up4", 64, 32, 4, 2), ("up8", 64, 32, 8, 2), ("up32", 32, 32, 32, 1))
    arrs["kernels"] = np.array(kernels)
    arrs["sc_cases"] = np.array([[c[1], c[2], c[3], c[4]] for c in cases])
    for ki, bk in enumerate(kernels):
        for i, (name, cin, cout, R, F) in enumerate(cases):
            sc = S.StyledConv(cin, cout, 3, 512, upsample=True, blur_kernel=bk)
            k = seed + 100 * (i + 1)
            w, mw, mb, ab = rnd(k + 2, 1, cout, cin, 3, 3), rnd(k + 3, cin, 512), 1 + rnd(k + 4, cin, std=0.1), rnd(k + 5, 1, cout, 1, 1, std=0.1)
            sc.conv.weight.data.copy_(w)
            sc.conv.modulation.weight.data.copy_(mw)
            sc.conv.modulation.bias.data.copy_(mb)
            sc.activate.bias.data.copy_(ab.reshape(sc.activate.bias.shape))
            sc.noise.weight.data.zero_()
            x = rnd(k + 6, F, cin, R, R)
            with torch.no_grad():
                ref = sc(x, style[:F])
            sd = {"c.conv.weight": w, "c.conv.modulation.weight": mw, "c.conv.modulation.bias": mb, "c.activate.bias": ab}
            orc = O.styled_conv(x, style[:F], sd, "c", True, blur_kernel=bk)
            dflt = O.styled_conv(x, style[:F], sd, "c", True)
            print("  k%d %-5s oracle-ref max|d| %.3e rel %.3e ; vs [1,3,3,1] rel %.3e" % ((ki, name) + maxdiff(orc, ref) + (maxdiff(dflt, ref)[1],)))
            arrs["sc_k%d_%s_out" % (ki, name)] = ref
    bk = kernels[1]
    d = S.Synthesis(size, 512, 20, blur_kernel=bk)
    sd = weights.synth_decoder_state(size, seed=seed, blur_kernel=bk)  # the buffers a checkpoint trained with this kernel holds
    d.load_state_dict(sd, strict=True)
    sd0 = weights.synth_decoder_state(size, seed=seed)
    d0 = S.Synthesis(size, 512, 20, blur_kernel=bk)
    d0.load_state_dict(sd0, strict=True)  # the strict load overwrites the constructor's kernel with the checkpoint's buffers
    assert all(torch.equal(d0.state_dict()[k], sd0[k]) for k in sd0 if k.endswith("blur.kernel"))
    d.eval()
    feats = weights.synth_feats(size, seed=seed)
    s_r = rnd(seed + 21, 1, 512)
    r_d = rnd(seed + 22, 1, 2, 512, std=0.5)
    with torch.no_grad():
        raw = torch.cat([d(s_r + r_d[:, t], None, feats)[0] for t in range(2)])
    orc = torch.cat([O.synthesis(sd, s_r + r_d[:, t], feats, blur_kernel=bk) for t in range(2)])
    dflt = torch.cat([O.synthesis(sd0, s_r + r_d[:, t], feats) for t in range(2)])
    print("  synthesis %s raw std %.3f ; oracle-ref max|d| %.3e ; vs [1,3,3,1] max|d| %.3e" % (bk, float(raw.std()), maxdiff(orc, raw)[0], maxdiff(dflt, raw)[0]))
    arrs.update(s_r=s_r, r_d=r_d, raw=raw)
    save("dec_blur", **arrs)


def gen_fir_buffers(ns, seed=2000, size=64):
    """FIR buffers that are NOT make_kernel([1,3,3,1]) outside the up-sampling StyledConvs - what a checkpoint may hold and the
    strict load (nodes_vadv_loader.py:632) puts over the constructor's kernels:
      * encoder (encoder.py:59-75,160-166): `conv2.0.kernel` of every ResBlock = make_kernel([1,2,4,1]) (asymmetric: pins the flip of
        upfirdn2d, encoder.py:28-29), `skip.0.kernel` = a seeded NON-separable positive 4 x 4 kernel of sum 1;
      * decoder (styledecoder.py:74-90,373,394): `to_rgbs.N.upsample.kernel` = make_kernel([1,2,4,1]) * 4, `to_flows.N.upsample.kernel`
        = 4 * outer([1,3,3,1], [1,2,4,1]) / 64 (rank 1, different factors per axis).
    Outputs of the reference's Encoder / Synthesis with those states loaded strictly."""
    print("[FIR buffers from the checkpoint]")
    esd, dsd = weights.fir_buffer_states(size, seed)
    enc = ns.encoder.Encoder(size, 512, 20)
    enc.load_state_dict(esd, strict=True)
    enc.eval()
    img = torch.from_numpy(np.random.RandomState(seed).rand(1, 3, size, size).astype(np.float32)) * 2 - 1
    with torch.no_grad():
        s_r, _, feats = enc(img, None)
        lam = enc.fc(s_r)
    o_s, o_f, o_l = O.encode_appearance(esd, img)
    d_s, d_f, _ = O.encode_appearance(weights.synth_encoder_state(size, seed=seed), img)
    print("  encoder oracle-ref max|d|: s_r %.3e lam %.3e feats %.3e ; default kernels vs ref: feats rel %.3e" % (
        maxdiff(o_s, s_r)[0], maxdiff(o_l, lam)[0], max(maxdiff(a, b)[0] for a, b in zip(o_f, feats)),
        max(maxdiff(a, b)[1] for a, b in zip(d_f, feats))))
    arrs = dict(seed=seed, size=size, enc_s_r=s_r, enc_lam=lam)
    for i, f in enumerate(feats):  # every 4th pixel of the larger maps (+ per-channel means of the whole map)
        st = 4 if f.shape[-1] > 16 else 1
        arrs["enc_feat%d_stride" % i] = st
        arrs["enc_feat%d" % i] = f[:, :, ::st, ::st]
        arrs["enc_feat%d_mean" % i] = f.mean(dim=(2, 3))
    d = ns.styledecoder.Synthesis(size, 512, 20)
    d.load_state_dict(dsd, strict=True)
    d.eval()
    dfeats = weights.synth_feats(size, seed=seed)
    s_r2 = rnd(seed + 21, 1, 512)
    r_d = rnd(seed + 22, 1, 2, 512, std=0.5)
    with torch.no_grad():
        raw = torch.cat([d(s_r2 + r_d[:, t], None, dfeats)[0] for t in range(2)])
    orc = torch.cat([O.synthesis(dsd, s_r2 + r_d[:, t], dfeats) for t in range(2)])
    dflt = torch.cat([O.synthesis(weights.synth_decoder_state(size, seed=seed), s_r2 + r_d[:, t], dfeats) for t in range(2)])
    print("  synthesis raw std %.3f ; oracle-ref max|d| %.3e ; default kernels vs ref rel %.3e" % (
        float(raw.std()), maxdiff(orc, raw)[0], maxdiff(dflt, raw)[1]))
    arrs.update(dec_s_r=s_r2, dec_r_d=r_d, dec_raw=raw)
    save("fir_buffers", **arrs)


def gen_e2e_config1(ns, seed=900):
    """BASELINE.json configs[0]: 1 s audio -> 25 frames, 512x512, nfe=10 (9 Euler evaluations), fp32, the
    reference's own sampler (nodes_adv.py:545-694) and decode loop (FLOAT.py:113-169) chained on CPU."""
    print("[e2e config 1]")
    cfg = config.FmtConfig()
    m, fsd, _ = ref_fmt(ns, cfg, seed)
    d, dsd = ref_dec(ns, 512, seed)
    feats = weights.synth_feats(512, seed=seed)
    T = 25
    wa = torch.nn.functional.silu(rnd(seed + 1, 1, T, cfg.dim_a))
    we = torch.softmax(rnd(seed + 2, 1, 1, cfg.dim_e), -1)
    r_s = rnd(seed + 3, 1, cfg.dim_w, std=0.5)
    s_r = rnd(seed + 4, 1, cfg.dim_w)
    g = torch.Generator("cpu")
    g.manual_seed(15)
    noise = torch.stack([torch.randn(1, 50, cfg.dim_w, generator=g)])
    g.manual_seed(15)
    with torch.no_grad():
        r_d = ns.nodes_adv._perform_ode_sampling_loop(m, r_s, wa, we, T, 10, 50, cfg.dim_w, 10, "euler", 1e-5, 1e-5,
                                                      torch.device("cpu"), 2.0, 1.0, 1.0, False, g)
        fake = types.SimpleNamespace(motion_autoencoder=types.SimpleNamespace(dec=d), pbar=types.SimpleNamespace(update=lambda n: None))
        pick = [0, 12, 24]
        frames = ns.FLOAT.FLOAT.decode_latent_into_processed_images(fake, s_r, feats, r_d[:, pick])
    orc_rd = O.sample_rd(fsd, cfg, r_s, wa, we, noise, 10, 2.0, 1.0, 1.0)
    orc = O.decode_frames(dsd, s_r, orc_rd[:, pick], feats)
    print("  r_d oracle-ref rel %.3e ; frames oracle-ref max|d| %.3e" % (maxdiff(orc_rd, r_d)[1], maxdiff(orc, frames)[0]))
    save("e2e_config1", seed=seed, wa=wa, we=we, r_s=r_s, s_r=s_r, noise=noise, r_d=r_d, pick=np.array(pick),
         lattice=frames[:, ::7, ::5], band=frames[:, 250:258], mean=frames.mean(dim=(1, 2, 3)))


def gen_encoder(ns, size, seed, sparse, gain=1.0, name=None):
    """Encoder.forward(img, None) + Encoder.fc + Direction of the reference itself (encoder.py:266-281, 242-247;
    styledecoder.py:428-444; FLOAT.py:283-291) on seeded synthetic weights.  gain != 1: every conv weight multiplied by it
    (weights.scale_encoder_convs) - the range-stress fixtures: activations grow by `gain` per conv, so the deep layers leave
    fp16's range (gain 6) or come close to it (gain 2) while the fp32 reference is untroubled."""
    print("[encoder %d gain %g]" % (size, gain))
    esd = weights.scale_encoder_convs(weights.synth_encoder_state(size, seed=seed), gain)
    enc = ns.encoder.Encoder(size, 512, 20)
    enc.load_state_dict(esd, strict=True)
    enc.eval()
    dsd = weights.synth_decoder_state(size, seed=seed)
    dirn = ns.styledecoder.Direction(20)
    dirn.load_state_dict({"weight": dsd["direction.weight"]})
    img = torch.from_numpy(np.random.RandomState(seed).rand(1, 3, size, size).astype(np.float32)) * 2 - 1
    with torch.no_grad():
        s_r, _, feats = enc(img, None)
        lam = enc.fc(s_r)
        r_s = dirn(lam)
    o_s, o_f, o_l = O.encode_appearance(esd, img)
    o_r = O.direction(dsd, o_l)
    print("  oracle-ref max|d|: s_r %.3e lam %.3e r_s %.3e feats %.3e" % (
        maxdiff(o_s, s_r)[0], maxdiff(o_l, lam)[0], maxdiff(o_r, r_s)[0], max(maxdiff(a, b)[0] for a, b in zip(o_f, feats))))
    # the image is RandomState(seed).rand(1,3,S,S)*2-1: tests regenerate it from the seed
    arrs = dict(seed=seed, size=size, gain=float(gain), s_r=s_r, lam=lam, r_s=r_s)
    for i, f in enumerate(feats):
        st = max(1, f.shape[-1] // 16) if sparse else (2 if f.shape[-1] > 16 else 1)
        arrs["feat%d_stride" % i] = st
        arrs["feat%d" % i] = f[:, :, ::st, ::st]
        arrs["feat%d_mean" % i] = f.mean(dim=(2, 3))
        arrs["feat%d_absmax" % i] = f.abs().max()
    save(name or "enc_%d" % size, **arrs)


def gen_audio(ns, tag, cfg, seed, seconds, T):
    """AudioEncoder.inference of the reference itself (FLOAT.py:304-375 on transformers' Wav2Vec2Model through
    wav2vec2.py:33-98) with seeded synthetic weights and the SURVEY 8d synthetic waveform."""
    print("[audio encoder %s]" % tag)
    sd = weights.synth_audio_state(cfg, seed=seed)
    opt = ns.base_options.BaseOptions()
    opt.dim_w, opt.only_last_features = cfg.dim_w, cfg.only_last_features
    enc = ns.FLOAT.AudioEncoder(opt, cfg.to_hf())
    r = enc.load_state_dict(sd, strict=True)
    enc.eval()
    a = weights.synth_waveform(seconds, seed=seed + 1)
    with torch.no_grad():
        wa = enc.inference(a, seq_len=T)
        feat = enc.wav2vec2.feature_extract(a if a.shape[1] % int(T * 640) == 0 else
                                            torch.nn.functional.pad(a[:, None], (0, int(T * 640) - a.shape[1]), mode="replicate")[:, 0], T)
    o = O.audio_encoder_inference(sd, cfg, a, T)
    print("  oracle-ref max|d| %.3e rel %.3e ; |wa| rms %.3f" % (maxdiff(o, wa) + (float(wa.pow(2).mean().sqrt()),)))
    save("aud_%s" % tag, seed=seed, seconds=seconds, T=T, wa=wa, feat_interp=feat[:, :, ::8])


def gen_emotion(ns, tag, cfg, seed, seconds):
    """Audio2Emotion.predict_emotion (FLOAT.py:396-401).  The reference's Wav2Vec2ForSpeechClassification cannot be
    instantiated on transformers 5.x (its init_weights() call breaks), so the golden is assembled from the
    same parts it is made of: transformers' Wav2Vec2Model, the reference's own Wav2Vec2ClassificationHead class and the
    'mean' pooling of merged_strategy (wav2vec2_ser.py:23-38,58-75,94-96), then softmax."""
    import importlib
    from transformers import Wav2Vec2Model
    print("[speech emotion %s]" % tag)
    ser = ref_import.import_ref("src.nodes.models.wav2vec2_ser")
    sd = weights.synth_audio_state(cfg, seed=seed)
    hf = cfg.to_hf()
    hf.final_dropout = 0.0
    m, head = Wav2Vec2Model(hf), ser.Wav2Vec2ClassificationHead(hf)
    m.load_state_dict({k[len("wav2vec2."):]: v for k, v in sd.items() if k.startswith("wav2vec2.")}, strict=True)
    head.load_state_dict({k[len("classifier."):]: v for k, v in sd.items() if k.startswith("classifier.")}, strict=True)
    m.eval()
    head.eval()
    a = weights.synth_waveform(seconds, seed=seed + 1)
    with torch.no_grad():
        hs = m(a).last_hidden_state
        scores = torch.softmax(head(hs.mean(dim=1)), dim=1)
    o = O.audio2emotion_predict(sd, cfg, a)
    print("  oracle-ref max|d| %.3e ; scores %s" % (maxdiff(o, scores)[0], [round(float(v), 4) for v in scores[0]]))
    save("emo_%s" % tag, seed=seed, seconds=seconds, scores=scores, pooled=hs.mean(dim=1))


def gen_node_surface(ns):
    """Widget/return contracts of the three north-star nodes and the batch/seed schedule of
    FloatProcess.floatprocess (nodes.py:189-222), captured from the reference classes themselves."""
    import contextlib
    import json
    print("[node surface]")
    import sys as _sys
    st = ref_import.stub("seconohe.torch")
    st.model_to_target = lambda logger, model: contextlib.nullcontext()
    import importlib
    nodes = ref_import.import_ref("src.nodes.nodes")
    adv = ns.nodes_adv

    def contract(cls):
        it = cls.INPUT_TYPES()
        return dict(INPUT_TYPES=it, RETURN_TYPES=list(cls.RETURN_TYPES), RETURN_NAMES=list(cls.RETURN_NAMES),
                    FUNCTION=cls.FUNCTION, CATEGORY=cls.CATEGORY, UNIQUE_NAME=cls.UNIQUE_NAME, DISPLAY_NAME=cls.DISPLAY_NAME)

    calls = []

    class FakePipe:
        rank = "cpu"

        class opt:
            cudnn_benchmark_enabled = False
            r_cfg_scale = 1.0
            fps = 25.0

        class G:
            pass

        def run_inference(self, path, img, audio, **kw):
            calls.append(dict(image_mark=float(img[0, 0, 0, 0]), audio_mark=float(audio["waveform"][0, 0, 0]),
                              seed=kw["seed"], emo=kw["emo"], no_crop=kw["no_crop"], a=kw["a_cfg_scale"], e=kw["e_cfg_scale"],
                              r=kw["r_cfg_scale"]))
            return torch.full((2, 4, 4, 3), float(len(calls)))

    img = torch.zeros(2, 4, 4, 3)
    img[0] += 10
    img[1] += 11
    wav = torch.zeros(3, 1, 8)
    wav[0] += 20
    wav[1] += 21
    wav[2] += 22
    out = nodes.FloatProcess().floatprocess(img, {"waveform": wav, "sample_rate": 16000}, FakePipe(), 2.0, 1.0, 30.0, "happy",
                                            False, 1000)
    sched = dict(calls=calls, images_shape=list(out[0].shape), images_first=[float(out[0][i, 0, 0, 0]) for i in range(out[0].shape[0])],
                 audio_shape=list(out[1]["waveform"].shape), audio_values=out[1]["waveform"][0, 0].tolist(), fps=out[2])
    fixture = dict(LoadFloatModelsOpt=contract(nodes.LoadFloatModels), FloatProcessOpt=contract(nodes.FloatProcess),
                   FloatAdvancedParameters=contract(adv.FloatAdvancedParameters), schedule=sched,
                   base_options={k: v for k, v in vars(ns.base_options.BaseOptions()).items()})
    with open(os.path.join(GOLD, "node_surface.json"), "w") as f:
        json.dump(fixture, f, indent=1, default=str)
    print("  wrote node_surface.json")


def gen_node_surface_va(ns):
    """Contracts of the very-advanced loaders / stage nodes captured from the reference classes (nodes_vadv_loader.py,
    nodes_vadv.py): widget names, types, defaults and ranges (tooltips dropped - they are prose, not contract), return
    tuples, FUNCTION / CATEGORY / names."""
    import importlib
    import json
    print("[node surface VA]")

    def strip(o):
        if isinstance(o, dict):
            return {k: strip(v) for k, v in o.items() if k != "tooltip"}
        if isinstance(o, (list, tuple)):
            return [strip(v) for v in o]
        return o
    out = {}
    for m in ("nodes_vadv_loader", "nodes_vadv"):
        mod = ref_import.import_ref("src.nodes." + m)
        for name in dir(mod):
            c = getattr(mod, name)
            if isinstance(c, type) and hasattr(c, "UNIQUE_NAME") and c.__module__ == mod.__name__:
                out[name] = dict(INPUT_TYPES=strip(c.INPUT_TYPES()), RETURN_TYPES=list(c.RETURN_TYPES), RETURN_NAMES=list(c.RETURN_NAMES),
                                 FUNCTION=c.FUNCTION, CATEGORY=c.CATEGORY, UNIQUE_NAME=c.UNIQUE_NAME, DISPLAY_NAME=c.DISPLAY_NAME,
                                 SUFFIX=getattr(mod, "SUFFIX", None))
    with open(os.path.join(GOLD, "node_surface_va.json"), "w") as f:
        json.dump(out, f, indent=1, default=str)
    print("  wrote node_surface_va.json (%d classes)" % len(out))


def gen_full_configs(ns):
    """BASELINE.json configs[1] (10 s, 250 frames, 5 windows x 50 Euler evaluations, static emotion, a=2 e=1) and configs[4]
    (30 s, 750 frames, 15 windows, per-window dynamic emotion, a=1 e=3) through the reference's own sampler
    (nodes_adv.py:545-694) and decode loop; ~25 s / ~75 s of CPU."""
    full = config.FmtConfig()
    run = gen_fmt_sample(ns, full, "config2", seed=2000, T=250, nfe=51, dynamic=False, a=2.0, e=1.0, compact=True)
    gen_config_frames(ns, "config2", run, 2000, [0, 124, 249])
    run = gen_fmt_sample(ns, full, "config5", seed=2100, T=750, nfe=51, dynamic=True, a=1.0, e=3.0, compact=True)
    gen_config_frames(ns, "config5", run, 2100, [0, 374, 749])


def main():
    ns = ref_import.load()
    if os.environ.get("GOLDENS_ONLY") == "tables":
        gen_fmt_tables(ns)
        return
    if os.environ.get("GOLDENS_ONLY") == "configs":
        gen_full_configs(ns)
        return
    if os.environ.get("GOLDENS_ONLY") == "va":
        gen_node_surface_va(ns)
        return
    if os.environ.get("GOLDENS_ONLY") == "emo":
        gen_emotion(ns, "small", config.small_emotion_config(), seed=1400, seconds=1.3)
        gen_emotion(ns, "xlsr", config.emotion_audio_config(), seed=1500, seconds=1.0)
        return
    if os.environ.get("GOLDENS_ONLY") == "aud":
        gen_audio(ns, "small", config.small_audio_config(), seed=1200, seconds=1.3, T=33)
        gen_audio(ns, "base", config.AudioConfig(), seed=1300, seconds=2.0, T=50)
        return
    if os.environ.get("GOLDENS_ONLY") == "decx":
        gen_dec_units_hip(ns, seed=1600)
        gen_dec_stress(ns, 64, 1700, "warp", sparse=False)
        gen_dec_stress(ns, 64, 1710, "range", sparse=False)
        gen_dec_stress(ns, 512, 1720, "warp_smooth", sparse=True)
        gen_dec_stress(ns, 512, 1730, "range", sparse=True)
        gen_dec_stress(ns, 512, 1740, "warp_half", sparse=True)
        gen_dec_cm2(ns)
        return
    if os.environ.get("GOLDENS_ONLY") == "warp_half":
        gen_dec_stress(ns, 512, 1740, "warp_half", sparse=True)
        return
    if os.environ.get("GOLDENS_ONLY") == "blur":
        gen_dec_blur(ns)
        return
    if os.environ.get("GOLDENS_ONLY") == "fir":
        gen_fir_buffers(ns)
        return
    if os.environ.get("GOLDENS_ONLY") == "encx":
        gen_encoder(ns, 64, seed=1010, sparse=False, gain=2.0, name="enc_stress_64_g2")
        gen_encoder(ns, 64, seed=1010, sparse=False, gain=6.0, name="enc_stress_64_g6")
        return
    if os.environ.get("GOLDENS_ONLY") == "enc":
        gen_encoder(ns, 64, seed=1000, sparse=False)
        gen_encoder(ns, 512, seed=1100, sparse=True)
        return
    if os.environ.get("GOLDENS_ONLY") != "fmt":
        gen_node_surface(ns)
        gen_node_surface_va(ns)
    if os.environ.get("GOLDENS_ONLY") == "nodes":
        return
    gen_e2e_config1(ns)
    if os.environ.get("GOLDENS_ONLY") == "e2e":
        return
    small = config.small_fmt_config()
    full = config.FmtConfig()
    gen_fmt_eval(ns, small, "small", seed=100)
    gen_fmt_eval(ns, full, "full", seed=200)
    gen_fmt_sample(ns, small, "small_static", seed=300, T=125, nfe=5, dynamic=False, a=2.0, e=1.0)
    gen_fmt_sample(ns, small, "small_dynamic", seed=400, T=125, nfe=5, dynamic=True, a=1.0, e=3.0)
    gen_fmt_sample(ns, full, "full_static", seed=500, T=25, nfe=10, dynamic=False, a=2.0, e=1.0)
    gen_fmt_tables(ns)
    if os.environ.get("GOLDENS_ONLY") == "fmt":
        return
    gen_full_configs(ns)
    gen_dec_units(ns, seed=600)
    gen_dec(ns, 64, seed=700, n_frames=3, sparse=False)
    gen_dec(ns, 512, seed=800, n_frames=2, sparse=True)
    gen_dec_units_hip(ns, seed=1600)
    gen_dec_blur(ns)
    gen_fir_buffers(ns)
    gen_dec_stress(ns, 64, 1700, "warp", sparse=False)
    gen_dec_stress(ns, 64, 1710, "range", sparse=False)
    gen_dec_stress(ns, 512, 1720, "warp_smooth", sparse=True)
    gen_dec_stress(ns, 512, 1730, "range", sparse=True)
    gen_dec_stress(ns, 512, 1740, "warp_half", sparse=True)
    gen_dec_cm2(ns)
    gen_encoder(ns, 64, seed=1000, sparse=False)
    gen_encoder(ns, 512, seed=1100, sparse=True)
    gen_encoder(ns, 64, seed=1010, sparse=False, gain=2.0, name="enc_stress_64_g2")
    gen_encoder(ns, 64, seed=1010, sparse=False, gain=6.0, name="enc_stress_64_g6")
    gen_audio(ns, "small", config.small_audio_config(), seed=1200, seconds=1.3, T=33)
    gen_audio(ns, "base", config.AudioConfig(), seed=1300, seconds=2.0, T=50)
    gen_emotion(ns, "small", config.small_emotion_config(), seed=1400, seconds=1.3)
    gen_emotion(ns, "xlsr", config.emotion_audio_config(), seed=1500, seconds=1.0)


if __name__ == "__main__":
    main()
