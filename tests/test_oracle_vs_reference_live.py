"""Build-container only (marker `reference`): run the imported reference itself next to the oracle on
fresh seeds (not the golden ones), so the oracle is pinned beyond the committed fixtures.  Skipped on
the GPU box, where /root/reference does not exist."""
import os
import sys

import pytest
import torch

from oracle import float_oracle as O
from tests.util import ROOT, load_pkg, rel_l2

pkg = load_pkg()
pytestmark = pytest.mark.reference


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_import
    if not ref_import.available():
        pytest.skip("reference tree not present")
    return ref_import.load()


def test_fmt_eval_and_window_live(ref):
    cfg = pkg.config.small_fmt_config()
    sd = pkg.weights.synth_fmt_state(cfg, seed=77)
    opt = ref.base_options.BaseOptions()
    for k in ("dim_w", "dim_a", "dim_e", "dim_h", "fmt_depth", "num_heads", "mlp_ratio", "num_prev_frames", "attention_window"):
        setattr(opt, k, getattr(cfg, k))
    opt.rank = "cpu"
    m = ref.FMT.FlowMatchingTransformer(opt)
    m.load_state_dict(sd, strict=True)
    m.eval()
    g = torch.Generator().manual_seed(9)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    x, wa, wr, we = r(2, 50, 128), r(2, 50, 128), r(2, 128), torch.softmax(r(2, 1, 7), -1)
    px, pwa = r(2, 10, 128), r(2, 10, 128)
    t = torch.tensor([0.81])
    with torch.no_grad():
        want = m.forward_with_cfv(t, x, wa, wr, we, px, pwa, None, a_cfg_scale=1.7, r_cfg_scale=1.0, e_cfg_scale=2.2)
    got = O.fmt_forward_cfv(sd, cfg, t, x, wa, wr, we, px, pwa, None, 1.7, 1.0, 2.2)
    assert rel_l2(got, want) < 1e-5


def test_synthesis_live(ref):
    sd = pkg.weights.synth_decoder_state(64, seed=78)
    feats = pkg.weights.synth_feats(64, seed=78)
    d = ref.styledecoder.Synthesis(64, 512, 20)
    d.load_state_dict(sd, strict=True)
    d.eval()
    g = torch.Generator().manual_seed(10)
    lat = torch.randn(2, 512, generator=g)
    with torch.no_grad():
        want, flow = d(lat, None, [f.expand(2, -1, -1, -1) for f in feats])
    got, gflow, _ = O.synthesis(sd, lat, feats, return_all=True)
    assert float((got - want).abs().max()) < 1e-4 and float((gflow - flow).abs().max()) < 1e-5


def test_encoder_oracle_live(ref):
    sd = pkg.weights.synth_encoder_state(64, seed=79)
    e = ref.encoder.Encoder(64, 512, 20)
    e.load_state_dict(sd, strict=True)
    e.eval()
    img = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(3)) * 2 - 1
    with torch.no_grad():
        s_r, _, feats = e(img, None)
        lam = e.fc(s_r)
    s2, f2, l2 = O.encode_appearance(sd, img)
    assert float((s_r - s2).abs().max()) < 1e-5 and float((lam - l2).abs().max()) < 1e-5
    assert all(float((a - b).abs().max()) < 1e-5 for a, b in zip(feats, f2))
