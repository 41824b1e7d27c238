"""The non-Euler fixed-step solvers of the reference's dropdown (src/nodes/__init__.py:15-23, options/base_options.py:50) live in
torchdiffeq, which is absent from /root/reference and from this image: the oracle restates the published step rules
(oracle/float_oracle.py::sample_chunk) and parity with the package itself stays unpinned.  What CAN be pinned without the package
is that each restated rule is a consistent Runge-Kutta scheme of its published ORDER: on the FMT's own vector field (tiny
configuration, fp64) the error against a 64-step rk4 reference must fall by 2^p when the step is halved - p = 1 euler,
2 midpoint / heun2, 3 heun3, 4 rk4 (3/8 rule).  A wrong coefficient or stage time drops the observed order to 1."""
import math

import pytest
import torch

from oracle import float_oracle as O
from tests.util import load_pkg

pkg = load_pkg()

ORDER = {"euler": 1, "midpoint": 2, "heun2": 2, "heun3": 3, "rk4": 4}


@pytest.fixture(scope="module")
def problem():
    cfg = pkg.config.small_fmt_config()
    sd = {k: v.double() for k, v in pkg.weights.synth_fmt_state(cfg, seed=9).items()}
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)  # noqa: E731
    L, P = cfg.num_frames_for_clip, cfg.num_prev_frames
    args = dict(x0=r(1, L, cfg.dim_w), wa=r(1, L, cfg.dim_a), wr=r(1, cfg.dim_w), we=torch.softmax(r(1, 1, cfg.dim_e), -1),
                px=r(1, P, cfg.dim_w), pwa=r(1, P, cfg.dim_a))

    def solve(method, nfe):
        return O.sample_chunk(sd, cfg, args["x0"], args["wa"], args["wr"], args["we"], args["px"], args["pwa"], None, nfe,
                              2.0, 1.0, 1.0, dtype=torch.float64, method=method)
    return solve, solve("rk4", 65)


@pytest.mark.parametrize("method", list(ORDER))
def test_restated_step_rule_has_its_published_order(problem, method):
    solve, ref = problem
    e1 = float((solve(method, 5) - ref).norm() / ref.norm())   # 4 steps
    e2 = float((solve(method, 9) - ref).norm() / ref.norm())   # 8 steps
    p = math.log2(e1 / e2)
    print("%-8s error %.3e (4 steps) -> %.3e (8 steps): observed order %.2f, published %d" % (method, e1, e2, p, ORDER[method]))
    assert abs(p - ORDER[method]) < 0.5, (method, e1, e2, p)
