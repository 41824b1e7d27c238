"""The two tables the FMT builds instead of loading - enc_dec_mask (FMT.py:15-19, registered at 234-236) and
get_sinusoid_encoding_table (FMT.py:22-40, copied into pos_embed at 249-250) - pinned to what the REFERENCE's own functions
produced (tests/golden/fmt_tables.npz, tools/make_goldens.py::gen_fmt_tables; the model there is built WITHOUT this repo's
tables injected).  CPU part: the host-side generators and the oracle's, bit for bit.  GPU part: the table the HIP operator
adds (given or regenerated in C) and the band its attention kernel implements."""
import pytest
import torch

from oracle import float_oracle as O
from tests.util import golden, load_pkg

pkg = load_pkg()
W, C = pkg.weights, pkg.config
G = golden("fmt_tables")
MASKS = [(60, 2), (45, 3), (80, 1), (70, 2)]
TABLES = [(60, 1024), (80, 1024), (45, 256), (70, 512)]


@pytest.mark.parametrize("ntok,win", MASKS)
def test_band_mask_is_the_reference_mask(ntok, win):
    ref = G["mask_%d_%d" % (ntok, win)].bool()
    assert torch.equal(W.band_mask(ntok, win), ref)
    assert torch.equal(O.alignment_mask(ntok, win), ref)


@pytest.mark.parametrize("ntok,d", TABLES)
def test_sinusoid_table_is_the_reference_table_bitwise(ntok, d):
    ref = G["pos_%d_%d" % (ntok, d)]
    assert torch.equal(W.sinusoid_table(ntok, d), ref)
    assert torch.equal(O.sinusoid_table(ntok, d), ref)


def test_tables_as_the_reference_model_registers_them():
    cfg = C.FmtConfig()
    assert torch.equal(G["model_pos_embed"], W.sinusoid_table(cfg.n_tokens, cfg.dim_h)[None])
    assert torch.equal(G["model_alignment_mask"].bool(), W.band_mask(cfg.n_tokens, cfg.attention_window))
    sd = W.synth_fmt_state(cfg, seed=1)
    assert torch.equal(sd["pos_embed"], G["model_pos_embed"])


def _probe_cfg(ntok, win):
    # smallest shape the operator accepts (head_dim 128): the band only depends on (tokens, window)
    return C.FmtConfig(dim_w=128, dim_a=128, dim_e=7, dim_h=256, fmt_depth=1, num_heads=2, mlp_ratio=4.0,
                       num_prev_frames=10, num_frames_for_clip=ntok - 10, attention_window=win)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("ntok,win", MASKS)
def test_attention_kernel_band_is_the_reference_mask(ntok, win, dtype):
    """q = k = 0 makes every visible key equally likely; v row j = e_j (in both heads) makes output[i, j] the probability
    query i gives key j: positive exactly where enc_dec_mask leaves the pair open, 1 / (#open keys of row i) there."""
    cfg = _probe_cfg(ntok, win)
    fmt = pkg.fmt.FlowMatchingTransformerHIP(W.synth_fmt_state(cfg, seed=2), cfg, "cuda:0", dtype)
    D = cfg.dim_h
    qkv = torch.zeros(ntok, 3 * D)
    for j in range(ntok):
        qkv[j, 2 * D + j] = 1.0
        qkv[j, 2 * D + 128 + j] = 1.0
    out = fmt.attention_probe(qkv).cpu()
    blocked = G["mask_%d_%d" % (ntok, win)].bool()
    for h in range(2):
        p = out[:, h * 128:h * 128 + ntok]
        assert torch.equal(p > 0, ~blocked), "head %d sees a different band than enc_dec_mask" % h
        want = (~blocked).float() / (~blocked).sum(dim=1, keepdim=True)
        assert float((p - want).abs().max()) < 2e-3
        assert float(out[:, h * 128 + ntok:(h + 1) * 128].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("ntok,d", [(60, 1024), (80, 1024), (45, 256), (70, 512)])
def test_operator_positional_table(ntok, d):
    """With `pos_embed` in the checkpoint the operator uses it verbatim; without it (the VA loaders drop it,
    nodes_vadv_loader.py:822-840) the C side regenerates the reference's table (libm sin / cos: <= 1 ulp of 1.0)."""
    cfg = C.FmtConfig(dim_w=128, dim_a=128, dim_e=7, dim_h=d, fmt_depth=1, num_heads=d // 128, mlp_ratio=1.0,
                      num_prev_frames=10, num_frames_for_clip=ntok - 10, attention_window=2)
    sd = W.synth_fmt_state(cfg, seed=3)
    ref = G["pos_%d_%d" % (ntok, d)]
    given = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", "bf16").pos_embed_in_use().cpu()
    assert torch.equal(given, ref)
    bare = {k: v for k, v in sd.items() if k != "pos_embed"}
    regen = pkg.fmt.FlowMatchingTransformerHIP(bare, cfg, "cuda:0", "bf16").pos_embed_in_use().cpu()
    assert float((regen - ref).abs().max()) <= 1.2e-7


@pytest.mark.gpu
def test_eval_golden_without_checkpoint_tables():
    """The reference-made evaluation golden through an operator that was given neither table."""
    from tests.util import rel_l2
    g = golden("fmt_eval_full")
    cfg = C.FmtConfig()
    sd = {k: v for k, v in W.synth_fmt_state(cfg, g["seed"]).items() if k not in ("pos_embed", "alignment_mask")}
    fmt = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", "fp16")
    out = fmt.forward_with_cfv(g["t"], g["cfg3_x"], g["cfg3_wa"], g["cfg3_wr"], g["cfg3_we"], g["cfg3_prev_x"],
                               g["cfg3_prev_wa"], None, a_cfg_scale=2.0, e_cfg_scale=1.0).cpu()
    assert rel_l2(out, g["cfg3_out"]) < 4e-3
