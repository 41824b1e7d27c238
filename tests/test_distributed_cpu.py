"""world_size-2 gloo tests of the multi-GPU host logic (no GPU): frame/window partitioning, the
boundary all_gather + re-solve of the window-parallel sampler, and the r_d broadcast.  The FMT is a
CPU test double built on the oracle with the same sample_chunk signature as the HIP mirror."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import float_oracle as O
from tests.util import load_pkg

pkg = load_pkg()
D = pkg.distributed


class OracleFmt:
    def __init__(self, sd, cfg):
        self.sd, self.cfg = sd, cfg

    def sample_chunk(self, x0, wa, wr, we, prev_x, prev_wa, prev_we=None, nfe=10, a_cfg_scale=1.0, r_cfg_scale=1.0,
                     e_cfg_scale=1.0, include_r_cfg=False):
        return O.sample_chunk(self.sd, self.cfg, x0, wa, wr, we, prev_x, prev_wa, prev_we, nfe, a_cfg_scale, r_cfg_scale,
                              e_cfg_scale, include_r_cfg)


def _problem(dynamic):
    cfg = pkg.config.small_fmt_config()
    sd = pkg.weights.synth_fmt_state(cfg, seed=4)
    T = 170  # 4 windows, last one padded
    g = torch.Generator().manual_seed(2)
    wa = torch.randn(1, T, cfg.dim_a, generator=g)
    r_s = torch.randn(1, cfg.dim_w, generator=g)
    we = torch.softmax(torch.randn(1, T if dynamic else 1, cfg.dim_e, generator=g), -1)
    noise = pkg.fmt.draw_noise(4, 1, cfg, seed=15)
    return cfg, sd, T, wa, r_s, we, noise


def test_shards_cover_everything():
    for T in (1, 7, 250, 1500):
        for world in (1, 2, 3, 8):
            r = [D.frame_shard(T, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == T
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_sample_range_equals_oracle_loop():
    for dynamic in (False, True):
        cfg, sd, T, wa, r_s, we, noise = _problem(dynamic)
        ref = O.sample_rd(sd, cfg, r_s, wa, we, noise, 4, 2.0, 1.0, 1.0)
        xs, _ = D.sample_range(OracleFmt(sd, cfg), cfg, r_s, wa, we, noise, 0, 4, 4, 2.0, 1.0, 1.0)
        assert torch.allclose(xs[:, :T], ref, atol=1e-6)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        cfg, sd, T, wa, r_s, we, noise = _problem(True)
        fmt = OracleFmt(sd, cfg)
        ref = O.sample_rd(sd, cfg, r_s, wa, we, noise, 4, 1.0, 1.0, 3.0)
        # one round: rank 0 exact, rank 1 re-solved from rank 0's exact tail -> exact for world = 2
        loc, (t0, t1), rep = D.sample_window_parallel(fmt, cfg, r_s, wa, we, noise, 4, 1.0, 1.0, 3.0, iters=1)
        err1 = float((loc - ref[:, t0:t1]).abs().max())
        # zero rounds: rank 1 starts from zero history -> differs from the sequential chain
        loc0, _, _ = D.sample_window_parallel(fmt, cfg, r_s, wa, we, noise, 4, 1.0, 1.0, 3.0, iters=0)
        err0 = float((loc0 - ref[:, t0:t1]).abs().max())
        # broadcast of r_d
        r_d = ref.clone() if rank == 0 else torch.zeros_like(ref)
        D.broadcast_latents(r_d, 0)
        q.put((rank, t0, t1, err1, err0, rep["seam_rel_change"], bool(torch.equal(r_d, ref))))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_window_parallel_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, a0, b0, e1_0, e0_0, seam0, bc0), (r1, a1, b1, e1_1, e0_1, seam1, bc1) = res
    assert (a0, b0, a1, b1) == (0, 100, 100, 170)
    assert e1_0 < 1e-6 and e1_1 < 1e-5, (e1_0, e1_1)       # one round is exact for two ranks
    assert e0_0 < 1e-6 and e0_1 > 1e-3                       # without the exchange rank 1 is visibly off
    assert seam0 == 0.0 and seam1 > 0.0                      # and the seam report says so
    assert bc0 and bc1
