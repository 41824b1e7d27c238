"""Edge cases of the hot path on the GPU, against the live oracle: ragged / minimal lengths, the
degenerate Euler grids, 4-way CFG through the whole loop, dynamic emotion with a ragged tail, and the
largest sizes of BASELINE.json (60 s clip) through size-independent properties."""
import pytest
import torch

from oracle import float_oracle as O
from tests.util import load_pkg, rel_l2

pkg = load_pkg()
pytestmark = pytest.mark.gpu
CFG = pkg.config.FmtConfig()
C, W = pkg.config, pkg.weights


@pytest.fixture(scope="module")
def fmt():
    sd = pkg.weights.synth_fmt_state(CFG, seed=31)
    return sd, pkg.fmt.FlowMatchingTransformerHIP(sd, CFG, "cuda:0", "bf16")


@pytest.mark.parametrize("T,nfe,dynamic", [(1, 3, False), (49, 2, False), (50, 3, True), (51, 3, True), (100, 1, False)])
def test_ragged_lengths_and_degenerate_grids(fmt, T, nfe, dynamic):
    sd, m = fmt
    c = pkg.pipeline.synth_conditions(CFG, T, seed=T, dynamic_we=dynamic)
    noise = pkg.fmt.draw_noise((T + 49) // 50, 1, CFG, seed=15)
    a, e = (1.0, 3.0) if dynamic else (2.0, 1.0)
    got = m.sample(c["r_s"], c["wa"], c["we"], noise, nfe, a, 1.0, e).cpu()
    ref = O.sample_rd(sd, CFG, c["r_s"], c["wa"], c["we"], noise, nfe, a, 1.0, e)
    assert got.shape == (1, T, 512)
    if nfe == 1:  # no evaluation: the sample is the noise itself, bit for bit
        assert torch.equal(got, noise.reshape(1, -1, 512)[:, :T])
    else:
        assert rel_l2(got, ref) < 2e-2


def test_four_way_cfg_through_the_loop(fmt):
    sd, m = fmt
    T = 70
    c = pkg.pipeline.synth_conditions(CFG, T, seed=9)
    noise = pkg.fmt.draw_noise(2, 1, CFG, seed=15)
    got = m.sample(c["r_s"], c["wa"], c["we"], noise, 4, 2.0, 1.5, 1.2, include_r_cfg=True).cpu()
    ref = O.sample_rd(sd, CFG, c["r_s"], c["wa"], c["we"], noise, 4, 2.0, 1.5, 1.2, include_r_cfg=True)
    assert rel_l2(got, ref) < 2e-2


def test_batch_items_are_independent(fmt):
    """B > 1 at the host mirror = the reference's per-item loop (nodes.py:189-209)."""
    sd, m = fmt
    c0, c1 = pkg.pipeline.synth_conditions(CFG, 60, seed=1), pkg.pipeline.synth_conditions(CFG, 60, seed=2)
    noise = pkg.fmt.draw_noise(2, 2, CFG, seed=15)
    both = m.sample(torch.cat([c0["r_s"], c1["r_s"]]), torch.cat([c0["wa"], c1["wa"]]), torch.cat([c0["we"], c1["we"]]), noise, 3)
    one = m.sample(c1["r_s"], c1["wa"], c1["we"], noise[:, 1:2], 3)
    assert torch.equal(both[1:2], one)


def test_sixty_second_clip_properties():
    """BASELINE configs[2] size (1500 frames, 30 windows) with a short grid: prefix property of the AR chain
    (the first 10 s of the 60 s clip equal the 10 s clip), finite output, frames in range."""
    fmt_sd = pkg.weights.synth_fmt_state(CFG, seed=33)
    dec_sd = pkg.weights.synth_decoder_state(64, seed=33)
    hp = pkg.pipeline.FloatHotPath(fmt_sd, dec_sd, CFG, "cuda:0", 64, max_frames=32)
    c = pkg.pipeline.synth_conditions(CFG, 1500, seed=4)
    noise = pkg.fmt.draw_noise(30, 1, CFG, seed=15)
    feats = pkg.weights.synth_feats(64, seed=33)
    frames, r_d = hp.generate(c["r_s"], c["wa"], c["we"], c["s_r"], feats, 3, noise=noise, return_rd=True)
    short = hp.sample(c["r_s"], c["wa"][:, :250], c["we"], 3, noise=noise[:5])
    torch.cuda.synchronize()
    assert frames.shape == (1500, 64, 64, 3) and torch.isfinite(frames).all()
    assert float(frames.min()) >= 0 and float(frames.max()) <= 1
    assert torch.equal(r_d[:, :250], short)


@pytest.mark.parametrize("method", ["midpoint", "rk4", "heun2", "heun3"])
def test_other_fixed_step_solvers(method):
    """The other entries of the reference's solver dropdown (src/nodes/__init__.py:15-23) against the oracle's
    restatement of the same published step rules; two windows so the AR hand-off is covered."""
    cfg = pkg.config.small_fmt_config()
    sd = pkg.weights.synth_fmt_state(cfg, seed=41)
    m = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", "fp16")
    m.set_method(method)
    T = 80
    c = pkg.pipeline.synth_conditions(cfg, T, seed=6)
    noise = pkg.fmt.draw_noise(2, 1, cfg, seed=15)
    got = m.sample(c["r_s"], c["wa"], c["we"], noise, 4, 2.0, 1.0, 1.0).cpu()
    ref = O.sample_rd(sd, cfg, c["r_s"], c["wa"], c["we"], noise, 4, 2.0, 1.0, 1.0, method=method)
    eul = O.sample_rd(sd, cfg, c["r_s"], c["wa"], c["we"], noise, 4, 2.0, 1.0, 1.0)
    err = rel_l2(got, ref)
    print(method, "rel-L2 %.3e (vs euler %.3e)" % (err, rel_l2(got, eul)))
    assert err < 4e-3 and rel_l2(got, eul) > 5 * err   # matches its own scheme, and is not Euler
    m.set_method("euler")
    back = m.sample(c["r_s"], c["wa"], c["we"], noise, 4, 2.0, 1.0, 1.0).cpu()
    assert rel_l2(back, eul) < 4e-3
    with pytest.raises(ValueError):
        m.set_method("dopri5")


@pytest.mark.parametrize("n_prev,n_cur,window", [(5, 40, 3), (10, 60, 2), (2, 50, 1), (10, 70, 2)])
def test_other_temporal_structures(n_prev, n_cur, window):
    """LoadFMTModel lets fps / wav2vec_sec / num_prev_frames / attention_window reshape the model
    (nodes_vadv_loader.py:791-840): windows of n_cur = int(wav2vec_sec * fps) frames with n_prev of context and a band
    of +-window keys.  Up to 80 tokens per window (e.g. fps 30: 60 + 10), 3- and 4-way CFG."""
    cfg = C.small_fmt_config()
    cfg.num_prev_frames, cfg.num_frames_for_clip, cfg.attention_window = n_prev, n_cur, window
    sd = W.synth_fmt_state(cfg, seed=61)
    f = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", "fp16")
    g = torch.Generator().manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    T = n_cur + 7
    r_s, wa, we = r(1, cfg.dim_w) * 0.5, torch.nn.functional.silu(r(1, T, cfg.dim_a)), torch.softmax(r(1, 1, cfg.dim_e), -1)
    noise = pkg.fmt.draw_noise(2, 1, cfg, 15)
    got = f.sample(r_s, wa, we, noise, 4, 2.0, 1.0, 1.0).cpu()
    want = O.sample_rd(sd, cfg, r_s, wa, we, noise, 4, 2.0, 1.0, 1.0)
    assert got.shape == (1, T, cfg.dim_w) and rel_l2(got, want) < 4e-3
    got4 = f.sample(r_s, wa, we, noise, 4, 2.0, 1.5, 1.0, include_r_cfg=True).cpu()  # up to 4 x 70 = 280 rows
    assert rel_l2(got4, O.sample_rd(sd, cfg, r_s, wa, we, noise, 4, 2.0, 1.5, 1.0, True)) < 4e-3


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("B,dynamic,rcfg", [(2, False, False), (4, True, False), (3, False, True), (4, False, False), (8, False, False),
                                            (13, False, False), (16, False, False), (10, False, True), (16, True, False)])
def test_batched_sampling_equals_per_clip(B, dynamic, rcfg, dtype):
    """float_fmt_sample_batch: B clips stacked along the rows of ONE launch chain (nodes_vadv.py:618-735 takes batches).  Each
    clip must be what the one-clip chain gives for it - within rounding, the GEMM tilings depend on the row count - and
    within the operand type's tolerance of the oracle; 70 frames = 2 windows, so the per-clip AR hand-off, replicate pad and
    (dynamic) prev_we are covered.  B = 13 / 16 (3-way CFG: 2 340 / 2 880 rows) and B = 10 with 4-way CFG (2 400 rows) run
    tier 3 of `pick_rb` (csrc/fmt_api.hip: 192 x 128 tiles for all four layers, fc2 as 2 K slices) - the default cap of
    FLOAT Process batches (pipeline.py) and bench.py's value_batch16."""
    sd = pkg.weights.synth_fmt_state(CFG, seed=41)
    one = pkg.fmt.FlowMatchingTransformerHIP(sd, CFG, "cuda:0", dtype)
    many = pkg.fmt.FlowMatchingTransformerHIP(sd, CFG, "cuda:0", dtype, max_batch=B)
    T = 70
    cs = [pkg.pipeline.synth_conditions(CFG, T, seed=50 + q, dynamic_we=dynamic) for q in range(B)]
    cat = lambda k: torch.cat([c[k] for c in cs])  # noqa: E731
    noise = pkg.fmt.draw_noise(2, B, CFG, seed=15)
    a, r, e = (1.0, 1.0, 3.0) if dynamic else ((2.0, 1.5, 1.2) if rcfg else (2.0, 1.0, 1.0))
    got = many.sample(cat("r_s"), cat("wa"), cat("we"), noise, 5, a, r, e, include_r_cfg=rcfg).cpu()
    assert got.shape == (B, T, 512)
    # the stacked chain (row-blocked LDS-DMA GEMM tiles from 300 rows on) sums in a fixed order: bitwise run to run
    assert torch.equal(got, many.sample(cat("r_s"), cat("wa"), cat("we"), noise, 5, a, r, e, include_r_cfg=rcfg).cpu())
    assert many.saturation() == 0
    tol = 2e-2 if dtype == "bf16" else 4e-3
    for q in range(B):
        alone = one.sample(cs[q]["r_s"], cs[q]["wa"], cs[q]["we"], noise[:, q:q + 1], 5, a, r, e, include_r_cfg=rcfg).cpu()
        assert rel_l2(got[q:q + 1], alone) < 0.25 * tol, (q, rel_l2(got[q:q + 1], alone))
    q = B - 1
    ref = O.sample_rd(sd, CFG, cs[q]["r_s"], cs[q]["wa"], cs[q]["we"], noise[:, q:q + 1], 5, a, r, e, include_r_cfg=rcfg)
    assert rel_l2(got[q:q + 1], ref) < tol
    with pytest.raises(ValueError, match="max_batch"):
        pkg.native.check(pkg.native.lib().float_fmt_sample_batch(
            one._h, 2, None, None, T, None, 1, None, 5, a, r, e, 0, None, None))


def test_batched_sampling_runge_kutta():
    sd = pkg.weights.synth_fmt_state(CFG, seed=42)
    one = pkg.fmt.FlowMatchingTransformerHIP(sd, CFG, "cuda:0", "fp16")
    many = pkg.fmt.FlowMatchingTransformerHIP(sd, CFG, "cuda:0", "fp16", max_batch=2)
    one.set_method("heun3")
    many.set_method("heun3")
    cs = [pkg.pipeline.synth_conditions(CFG, 60, seed=60 + q) for q in range(2)]
    noise = pkg.fmt.draw_noise(2, 2, CFG, seed=15)
    got = many.sample(torch.cat([c["r_s"] for c in cs]), torch.cat([c["wa"] for c in cs]), torch.cat([c["we"] for c in cs]), noise, 3).cpu()
    for q in range(2):
        alone = one.sample(cs[q]["r_s"], cs[q]["wa"], cs[q]["we"], noise[:, q:q + 1], 3).cpu()
        assert rel_l2(got[q:q + 1], alone) < 1e-3


@pytest.mark.parametrize("method,nfe", [("euler", 131), ("rk4", 20), ("euler", 66)])
def test_long_grids_cross_the_modulation_batch(method, nfe):
    """More than 64 evaluations per window: the hoisted adaLN projection then runs in several batches of 64 evaluations
    (130 = 64 + 64 + 2; rk4 at nfe 20 = 76; 65 = 64 + 1), each reusing the modulation slab.  Small model, fp32 operands against
    the oracle at 1e-4, and fp16 at its own tolerance; the per-evaluation projection (FLOAT_FMT_HOIST=0) is covered bitwise
    by tests/test_variants_gpu.py."""
    cfg = C.small_fmt_config()
    sd = W.synth_fmt_state(cfg, seed=51)
    g = torch.Generator().manual_seed(3)
    T = 60
    r_s, wa = torch.randn(1, cfg.dim_w, generator=g), torch.randn(1, T, cfg.dim_a, generator=g)
    we = torch.softmax(torch.randn(1, 1, cfg.dim_e, generator=g), -1)
    noise = pkg.fmt.draw_noise(2, 1, cfg, seed=15)
    ref = O.sample_rd(sd, cfg, r_s, wa, we, noise, nfe, 2.0, 1.0, 1.0, method=method)
    for dtype, tol in (("fp32", 1e-4), ("fp16", 4e-3)):
        m = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", dtype)
        m.set_method(method)
        got = m.sample(r_s, wa, we, noise, nfe, 2.0, 1.0, 1.0).cpu()
        assert rel_l2(got, ref) < tol, (dtype, rel_l2(got, ref))


def test_batched_handle_is_sized_for_the_free_hbm(monkeypatch):
    """FloatHotPath.batched_fmt: a stacked-clip handle costs 3.1 GB of workspace per clip; with little HBM free (another model
    resident) it is sized for what fits - the chain then runs the batch in chunks - instead of failing inside float_fmt_create."""
    cfg = CFG
    hp = pkg.pipeline.FloatHotPath(W.synth_fmt_state(cfg, seed=1), W.synth_decoder_state(64, seed=1), cfg, "cuda:0", 64, max_frames=8)
    real = torch.cuda.mem_get_info
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (int(9.5 * 2**30), real(dev)[1]))
    small = hp.batched_fmt(16)
    assert small.max_batch == 2          # (9.5 - 2) GB // 3.3 GB
    monkeypatch.setattr(torch.cuda, "mem_get_info", real)
    assert hp.batched_fmt(2) is small    # cached
    cs = [pkg.pipeline.synth_conditions(cfg, 60, seed=q) for q in range(3)]
    noise = pkg.fmt.draw_noise(2, 3, cfg, seed=15)
    cat = lambda k: torch.cat([c[k] for c in cs])  # noqa: E731
    got = small.sample(cat("r_s"), cat("wa"), cat("we"), noise, 3).cpu()   # 3 clips on a 2-clip handle: chunks of 2 + 1
    one = hp.fmt.sample(cs[2]["r_s"], cs[2]["wa"], cs[2]["we"], noise[:, 2:3], 3).cpu()
    assert got.shape == (3, 60, cfg.dim_w) and rel_l2(got[2:3], one) < 1e-3
