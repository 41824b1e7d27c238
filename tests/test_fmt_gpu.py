"""GPU parity of the FMT operator against the CPU oracle and the committed goldens, through the
C ABI.  bf16 operands / fp32 accumulation: tolerance 2e-2 rel-L2 on the velocity and on r_d after a
full window; fp16 operands are held to 4e-3."""
import pytest
import torch

from oracle import float_oracle as O
from tests.util import golden, load_pkg, rel_l2

pkg = load_pkg()
W, C = pkg.weights, pkg.config
pytestmark = pytest.mark.gpu

TOL = {"bf16": 2e-2, "fp16": 4e-3}


def _fmt(cfg, seed, dtype, use_graph=True):
    sd = W.synth_fmt_state(cfg, seed)
    return sd, pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", dtype=dtype, use_graph=2 if use_graph is True else int(use_graph))


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("tag", ["small", "full"])
def test_eval_golden(tag, dtype):
    g = golden("fmt_eval_" + tag)
    cfg = C.small_fmt_config() if tag == "small" else C.FmtConfig()
    sd, fmt = _fmt(cfg, g["seed"], dtype)
    for case in ("nocfg", "cfg3", "cfg4", "cfg3dyn"):
        a, r, e, rc = [float(v) for v in g[case + "_scales"]]
        out = fmt.forward_with_cfv(g["t"], g[case + "_x"], g[case + "_wa"], g[case + "_wr"], g[case + "_we"],
                                   g[case + "_prev_x"], g[case + "_prev_wa"], g.get(case + "_prev_we"),
                                   a_cfg_scale=a, r_cfg_scale=r, e_cfg_scale=e, include_r_cfg=bool(rc)).cpu()
        err = rel_l2(out, g[case + "_out"])
        print(tag, dtype, case, "rel-L2 %.3e" % err)
        assert err < TOL[dtype], (case, err)
    assert fmt.saturation() == 0  # float_fmt_saturation: no 16-bit activation store was clamped


@pytest.mark.parametrize("tag", ["small_static", "small_dynamic", "full_static"])
def test_sample_golden(tag):
    g = golden("fmt_sample_" + tag)
    cfg = C.FmtConfig() if tag.startswith("full") else C.small_fmt_config()
    for use_graph in (0, 2, 1):  # eager, hipGraph replay (1 and 2 are the same since the adaLN GEMM left the step)
        sd, fmt = _fmt(cfg, g["seed"], "bf16", use_graph)
        r_d = fmt.sample(g["r_s"], g["wa"], g["we"], g["noise"], g["nfe"], g["a"], 1.0, g["e"]).cpu()
        assert r_d.shape == g["r_d"].shape
        err = rel_l2(r_d, g["r_d"])
        print(tag, "graph%d" % use_graph if use_graph else "eager", "rel-L2 %.3e" % err)
        assert err < TOL["bf16"], err
        if use_graph:
            assert torch.equal(r_d, r_eager), "graph replay must be bitwise identical to eager launches"
        r_eager = r_d


def test_window_50_steps_vs_oracle():
    """One full-size window at the headline setting: nfe = 51 (50 evaluations), a=2, e=1."""
    cfg = C.FmtConfig()
    sd, fmt = _fmt(cfg, 11, "bf16")
    gen = torch.Generator().manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=gen)  # noqa: E731
    x0, wa, wr, we = r(1, 50, 512), r(1, 50, 512), r(1, 512), torch.softmax(r(1, 1, 7), -1)
    px, pwa = r(1, 10, 512), r(1, 10, 512)
    got = fmt.sample_chunk(x0, wa, wr, we, px, pwa, None, nfe=51, a_cfg_scale=2.0, e_cfg_scale=1.0).cpu()
    ref = O.sample_chunk(sd, cfg, x0, wa, wr, we, px, pwa, None, 51, 2.0, 1.0, 1.0)
    err = rel_l2(got, ref)
    print("50-step window rel-L2 %.3e" % err)
    assert err < TOL["bf16"]
    # determinism: same inputs, same bits
    again = fmt.sample_chunk(x0, wa, wr, we, px, pwa, None, nfe=51, a_cfg_scale=2.0, e_cfg_scale=1.0).cpu()
    assert torch.equal(got, again)


def test_linearity_of_cfg_scales():
    """Size-independent property: v(a,e) is affine in the scales (FMT.py:379)."""
    cfg = C.FmtConfig()
    sd, fmt = _fmt(cfg, 5, "bf16")
    g = golden("fmt_eval_full")
    args = [g["t"]] + [g["cfg3_" + k] for k in ("x", "wa", "wr", "we", "prev_x", "prev_wa")]
    v = lambda a, e: fmt.forward_with_cfv(*args, None, a_cfg_scale=a, e_cfg_scale=e).cpu()  # noqa: E731
    v11, v21, v31, v12 = v(1.0001, 1.0), v(2.0, 1.0), v(3.0, 1.0), v(1.0001, 2.0)
    assert rel_l2(v31 - v21, v21 - v11) < 2e-3
    assert float((v12 - v11).abs().max()) > 0


def test_error_behaviour():
    cfg = C.small_fmt_config()
    sd, fmt = _fmt(cfg, 1, "bf16")
    z = torch.zeros
    with pytest.raises(ValueError, match="prev_we"):
        fmt.forward_with_cfv(torch.tensor([0.1]), z(1, 50, 128), z(1, 50, 128), z(1, 128), z(1, 50, 7), z(1, 10, 128),
                             z(1, 10, 128), None)
    with pytest.raises(ValueError):
        fmt.forward_with_cfv(torch.tensor([0.1]), z(1, 49, 128), z(1, 50, 128), z(1, 128), z(1, 1, 7), z(1, 10, 128),
                             z(1, 10, 128), None)
    bad = dict(sd)
    del bad["blocks.0.attn.qkv.weight"]
    with pytest.raises(KeyError):
        pkg.fmt.FlowMatchingTransformerHIP(bad, cfg, "cuda:0")


def test_sample_is_capturable_by_the_caller():
    """include/float_hip.h: run-time calls may be issued while the caller captures `stream` - the chain is launched straight
    into that capture (no nested graph launch, no host memory read by the stream: evaluation times are formed on the device).
    A torch CUDA graph around `sample` must replay to the bits of the plain call, also after the inputs changed in place."""
    cfg = C.small_fmt_config()
    sd, fmt = _fmt(cfg, 9, "fp16")
    T = 70
    c = pkg.pipeline.synth_conditions(cfg, T, seed=4, device="cuda:0")
    # small config: dim_w = dim_a = 128
    g = torch.Generator().manual_seed(0)
    r_s = torch.randn(1, cfg.dim_w, generator=g).cuda()
    wa = torch.randn(1, T, cfg.dim_a, generator=g).cuda()
    we = c["we"]
    noise = pkg.fmt.draw_noise(2, 1, cfg, seed=15).cuda()
    plain = fmt.sample(r_s, wa, we, noise, 5, 2.0, 1.0, 1.0).clone()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fmt.sample(r_s, wa, we, noise, 5, 2.0, 1.0, 1.0)  # warm-up on the capture stream
        torch.cuda.current_stream().synchronize()
        with torch.cuda.graph(graph, stream=side):
            captured = fmt.sample(r_s, wa, we, noise, 5, 2.0, 1.0, 1.0)
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(captured, plain)
    wa.mul_(0.5)  # same buffers, new contents: the replay must follow
    want = fmt.sample(r_s, wa, we, noise, 5, 2.0, 1.0, 1.0).clone()
    graph.replay()
    torch.cuda.synchronize()
    assert not torch.equal(want, plain)
    d = (captured - want).abs().amax(dim=(0, 2))
    assert torch.equal(captured, want), ("frames that differ", torch.nonzero(d).flatten().tolist()[:10], float(d.max()),
                                         "captured == old result" if torch.equal(captured, plain) else "")


def test_graph_cache_is_bounded():
    """At most 8 window graphs per handle (least recently used evicted): sweeping a CFG scale neither leaks executables nor
    changes results when an evicted key comes back."""
    cfg = C.small_fmt_config()
    sd, fmt = _fmt(cfg, 10, "fp16")
    g = torch.Generator().manual_seed(1)
    r_s, wa = torch.randn(1, cfg.dim_w, generator=g), torch.randn(1, 50, cfg.dim_a, generator=g)
    we = torch.softmax(torch.randn(1, 1, cfg.dim_e, generator=g), -1)
    noise = pkg.fmt.draw_noise(1, 1, cfg, seed=15)
    first = fmt.sample(r_s, wa, we, noise, 4, 1.5, 1.0, 1.0).cpu()
    for i in range(12):
        assert torch.isfinite(fmt.sample(r_s, wa, we, noise, 4, 2.0 + 0.1 * i, 1.0, 1.0)).all()
    assert torch.equal(fmt.sample(r_s, wa, we, noise, 4, 1.5, 1.0, 1.0).cpu(), first)


@pytest.mark.parametrize("hpw", [1, 2])
@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
def test_fused_attention_proj_launch(hpw, dtype, monkeypatch):
    """FLOAT_FMT_ATTNPROJ=1|2: banded attention and attn.proj in ONE launch (split-K over the heads, folded by the next
    LayerNorm launch; FMT.py:71-89) against the same reference goldens as the default two-launch chain, and bitwise
    reproducible from run to run."""
    monkeypatch.setenv("FLOAT_FMT_ATTNPROJ", str(hpw))
    g = golden("fmt_eval_full")
    cfg = C.FmtConfig()
    sd, fmt = _fmt(cfg, g["seed"], dtype)
    tol = 2e-5 if dtype == "fp32" else TOL[dtype]
    for case in ("cfg3", "cfg4"):
        a, r, e, rc = [float(v) for v in g[case + "_scales"]]
        call = lambda: fmt.forward_with_cfv(g["t"], g[case + "_x"], g[case + "_wa"], g[case + "_wr"], g[case + "_we"],  # noqa: E731
                                            g[case + "_prev_x"], g[case + "_prev_wa"], g.get(case + "_prev_we"),
                                            a_cfg_scale=a, r_cfg_scale=r, e_cfg_scale=e, include_r_cfg=bool(rc)).cpu()
        out = call()
        err = rel_l2(out, g[case + "_out"])
        print("fused hpw=%d" % hpw, dtype, case, "rel-L2 %.3e" % err)
        assert err < tol, (case, err)
        assert torch.equal(out, call())
    assert fmt.saturation() == 0
    gs = golden("fmt_sample_full_static")
    sd, fmt = _fmt(cfg, gs["seed"], dtype)
    r_d = fmt.sample(gs["r_s"], gs["wa"], gs["we"], gs["noise"], gs["nfe"], gs["a"], 1.0, gs["e"]).cpu()
    assert rel_l2(r_d, gs["r_d"]) < tol


def test_fmt_saturation_counter_fires():
    """float_fmt_saturation: x_embedder input rows beyond fp16's range (|x| = 1e6 > 65504) are clamped AND counted in the fp16
    handle; the bf16 handle (fp32's exponent range) reports 0.  Without this, `saturation() == 0` elsewhere would prove nothing."""
    g = golden("fmt_eval_small")
    cfg = C.small_fmt_config()
    args = lambda x: (g["t"], x, g["cfg3_wa"], g["cfg3_wr"], g["cfg3_we"], g["cfg3_prev_x"], g["cfg3_prev_wa"], None)  # noqa: E731
    for dtype, expect in (("fp16", True), ("bf16", False)):
        sd, fmt = _fmt(cfg, g["seed"], dtype)
        fmt.forward_with_cfv(*args(g["cfg3_x"]), a_cfg_scale=2.0, e_cfg_scale=1.0)
        assert fmt.saturation() == 0
        fmt.forward_with_cfv(*args(g["cfg3_x"] * 1e6), a_cfg_scale=2.0, e_cfg_scale=1.0)
        n = fmt.saturation(reset=True)
        print(dtype, "clamped-store threads with |x| = 1e6:", n)
        assert (n > 0) == expect and fmt.saturation() == 0


def test_persistent_evaluation_kernel(monkeypatch):
    """FLOAT_FMT_MEGA=1: the 59 stages of an evaluation as ONE persistent kernel with in-kernel grid barriers (write-through
    stores, coherent loads, no fences) instead of 59 dependent launches.  Same stage bodies, so it agrees with the launch chain
    to rounding (x-embed splits K over 8 waves instead of 4), holds the goldens, is bitwise reproducible, and its barrier
    watchdog stays silent (float_fmt_saturation returns an error if a barrier timed out)."""
    g = golden("fmt_sample_full_static")
    cfg = C.FmtConfig()
    sd, chain = _fmt(cfg, g["seed"], "fp16")
    ref = chain.sample(g["r_s"], g["wa"], g["we"], g["noise"], g["nfe"], g["a"], 1.0, g["e"]).cpu()
    monkeypatch.setenv("FLOAT_FMT_MEGA", "1")
    sd, mega = _fmt(cfg, g["seed"], "fp16")
    a = mega.sample(g["r_s"], g["wa"], g["we"], g["noise"], g["nfe"], g["a"], 1.0, g["e"]).cpu()
    b = mega.sample(g["r_s"], g["wa"], g["we"], g["noise"], g["nfe"], g["a"], 1.0, g["e"]).cpu()
    print("persistent kernel vs launch chain rel-L2 %.3e, vs reference %.3e" % (rel_l2(a, ref), rel_l2(a, g["r_d"])))
    assert torch.equal(a, b) and rel_l2(a, ref) < 5e-4 and rel_l2(a, g["r_d"]) < TOL["fp16"]
    assert mega.saturation() == 0  # also raises if the barrier watchdog fired
