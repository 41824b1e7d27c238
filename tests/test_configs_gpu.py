"""BASELINE.json configs[1] and configs[4] at their full length against what the REFERENCE itself produced on CPU
(tools/make_goldens.py::gen_full_configs: nodes_adv._perform_ode_sampling_loop + FLOAT.decode_latent_into_processed_images):
  config2: 10 s, 250 frames = 5 auto-regressive windows x 50 Euler evaluations, static emotion, a=2 e=1
  config5: 30 s, 750 frames = 15 windows x 50 evaluations, per-window dynamic emotion with prev_we hand-off, a=1 e=3
The per-window rel-L2 of r_d is printed so that drift of the 16-bit chain over the AR windows is visible; three frames of each
clip are decoded at 512x512 and compared with the reference's frames.  Tolerances: fp16 operands (the headline dtype of
bench.py and the nodes) r_d <= 4e-3 rel-L2 per window and overall, frames >= 40 dB PSNR and <= 2/255 mean |d|;
bf16 operands (diagnostic) r_d <= 2e-2."""
import pytest
import torch

from tests.util import golden, load_pkg, rel_l2, sample_inputs

pkg = load_pkg()
pytestmark = pytest.mark.gpu
CFG = pkg.config.FmtConfig()
LIMIT = {"fp16": 4e-3, "bf16": 2e-2}


def _run(tag, dtype):
    g = golden("fmt_sample_" + tag)
    inp = sample_inputs(CFG, g["seed"], g["T"], bool(g["dynamic"]), g["noise_seed"])
    sd = pkg.weights.synth_fmt_state(CFG, g["seed"])
    fmt = pkg.fmt.FlowMatchingTransformerHIP(sd, CFG, "cuda:0", dtype)
    r_d = fmt.sample(inp["r_s"], inp["wa"], inp["we"], inp["noise"], g["nfe"], g["a"], 1.0, g["e"]).cpu()
    ref = g["r_d"]
    assert r_d.shape == ref.shape == (1, g["T"], CFG.dim_w)
    L = CFG.num_frames_for_clip
    per_window = [rel_l2(r_d[:, k:k + L], ref[:, k:k + L]) for k in range(0, g["T"], L)]
    print("%s %s r_d rel-L2 overall %.3e; per window: %s" % (tag, dtype, rel_l2(r_d, ref), " ".join("%.2e" % e for e in per_window)))
    return g, r_d, ref, per_window


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("tag", ["config2", "config5"])
def test_full_length_chain_vs_reference(tag, dtype):
    g, r_d, ref, per_window = _run(tag, dtype)
    assert rel_l2(r_d, ref) < LIMIT[dtype]
    assert max(per_window) < LIMIT[dtype] * 1.5, per_window


@pytest.mark.parametrize("tag", ["config2", "config5"])
def test_full_length_frames_vs_reference(tag):
    g, r_d, ref, _ = _run(tag, "fp16")
    f = golden("frames_" + tag)
    pick = [int(i) for i in f["pick"]]
    dec = pkg.decoder.SynthesisHIP(pkg.weights.synth_decoder_state(512, seed=f["seed"]), 512, 512, "cuda:0", "fp16", max_frames=4)
    from tests.util import seeded_normal
    s_r = seeded_normal(f["seed"] + 4, 1, 512)
    frames = dec.decode_latent_into_processed_images(s_r, r_d[:, pick], pkg.weights.synth_feats(512, seed=f["seed"])).cpu()
    d = frames[:, ::7, ::5] - f["lattice"]
    psnr = [float(-10 * torch.log10((d[i] ** 2).mean())) for i in range(len(pick))]
    band = float((frames[:, 250:258] - f["band"]).abs().mean())
    mean_err = float((frames.mean(dim=(1, 2, 3)) - f["mean"]).abs().max())
    print("%s frames %s: PSNR %s dB, band mean|d| %.2e, frame-mean err %.2e" % (tag, pick, " ".join("%.1f" % p for p in psnr), band, mean_err))
    assert min(psnr) >= 40.0 and band <= 2.0 / 255 and mean_err < 2e-3
    assert dec.saturation() == 0


def test_sixty_second_clip_full_size():
    """BASELINE configs[2] at its size on one GPU: 60 s = 1500 frames = 30 AR windows x 50 Euler evaluations, 512x512.
    The AR chain is causal and the input generators are sequential, so the first 250 frames of this clip ARE config2's:
    they must equal the HIP 250-frame run bit for bit and the reference's r_d (fmt_sample_config2.npz) within tolerance,
    frames 0 / 124 / 249 must match the reference's frames, and a frame shard of the long clip must equal the same frames
    of the whole (what `bench.py --mode shard` relies on across GPUs)."""
    g = golden("fmt_sample_config2")
    short = sample_inputs(CFG, g["seed"], 250, False, g["noise_seed"])
    long = sample_inputs(CFG, g["seed"], 1500, False, g["noise_seed"])
    assert torch.equal(long["wa"][:, :250], short["wa"]) and torch.equal(long["noise"][:5], short["noise"])
    sd = pkg.weights.synth_fmt_state(CFG, g["seed"])
    fmt = pkg.fmt.FlowMatchingTransformerHIP(sd, CFG, "cuda:0", "fp16")
    r_long = fmt.sample(long["r_s"], long["wa"], long["we"], long["noise"], g["nfe"], g["a"], 1.0, g["e"])
    r_short = fmt.sample(short["r_s"], short["wa"], short["we"], short["noise"], g["nfe"], g["a"], 1.0, g["e"])
    assert r_long.shape == (1, 1500, CFG.dim_w) and torch.isfinite(r_long).all()
    assert torch.equal(r_long[:, :250], r_short)
    assert rel_l2(r_long[:, :250].cpu(), g["r_d"]) < LIMIT["fp16"]
    # windows 6..30 have no reference run (75 s of CPU each 10 s); their statistics must look like the first five's
    rms = r_long.reshape(30, 50, -1).pow(2).mean(dim=(1, 2)).sqrt()
    assert float(rms.max() / rms.min()) < 1.5, rms
    f = golden("frames_config2")
    pick = [int(i) for i in f["pick"]]
    from tests.util import seeded_normal
    dec = pkg.decoder.SynthesisHIP(pkg.weights.synth_decoder_state(512, seed=f["seed"]), 512, 512, "cuda:0", "fp16", max_frames=32)
    dec.set_feats(pkg.weights.synth_feats(512, seed=f["seed"]))
    s_r = seeded_normal(f["seed"] + 4, 1, 512)
    frames = dec.decode_latent_into_processed_images(s_r, r_long[:, :250])
    d = frames[pick].cpu()[:, ::7, ::5] - f["lattice"]
    psnr = [float(-10 * torch.log10((d[i] ** 2).mean())) for i in range(len(pick))]
    assert min(psnr) >= 40.0, psnr
    t0, t1 = pkg.distributed.frame_shard(1500, 8, 7)  # the last of 8 ranks
    shard = dec.decode_latent_into_processed_images(s_r, r_long[:, t0:t1])
    whole_tail = dec.decode_latent_into_processed_images(s_r, r_long[:, 1250:1500])
    assert (t0, t1) == (1313, 1500) and torch.equal(shard, whole_tail[t0 - 1250:])
    assert dec.saturation() == 0  # 2000 frames decoded, nothing clamped at fp16's range
    print("60 s clip: prefix == 10 s clip bitwise; frames %s PSNR %s dB; window rms %.3f..%.3f" % (
        pick, " ".join("%.1f" % p for p in psnr), float(rms.min()), float(rms.max())))
