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
from tests.util import ROOT, load_pkg

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
        xs, tail = D.sample_range(OracleFmt(sd, cfg), cfg, r_s, wa, we, noise, 0, 4, 4, 2.0, 1.0, 1.0)
        # the contract of both implementations: rows trimmed to the clip; behind the clip's trimmed last window prev_x is zeros
        assert xs.shape[1] == T and torch.allclose(xs, ref, atol=1e-6)
        assert float(tail[0].abs().max()) == 0.0 and tail[0].shape == (1, cfg.num_prev_frames, cfg.dim_w)
        mid, tail = D.sample_range(OracleFmt(sd, cfg), cfg, r_s, wa, we, noise, 0, 2, 4, 2.0, 1.0, 1.0)
        L, P = cfg.num_frames_for_clip, cfg.num_prev_frames
        assert mid.shape[1] == 2 * L and torch.allclose(tail[0], ref[:, 2 * L - P:2 * L], atol=1e-6)


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
        q.put((rank, t0, t1, err1, err0, float(rep["seam_rel_change"]), bool(torch.equal(r_d, ref))))
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


class OracleHotPath:
    """CPU double of pipeline.FloatHotPath (sample / decode / generate) on the oracle: 64-px decoder, small FMT."""

    def __init__(self, cfg, fmt_sd, dec_sd, feats):
        self.cfg, self.fmt_sd, self.dec_sd, self.feats = cfg, fmt_sd, dec_sd, feats
        self.device = torch.device("cpu")
        self.sample_calls = 0

    def sample(self, r_s, wa, we, nfe, a, r, e, seed=15, noise=None):
        self.sample_calls += 1
        return O.sample_rd(self.fmt_sd, self.cfg, r_s, wa, we, noise, nfe, a, r, e)

    def decode(self, s_r, feats, r_d, frame_range=None):
        rd = r_d if frame_range is None else r_d[:, frame_range[0]:frame_range[1]]
        return O.decode_frames(self.dec_sd, s_r, rd, self.feats)

    def generate(self, r_s, wa, we, s_r, feats, nfe, a, r, e, seed=15, noise=None, frame_range=None):
        return self.decode(s_r, feats, self.sample(r_s, wa, we, nfe, a, r, e, noise=noise), frame_range)


def _shard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        cfg = pkg.config.small_fmt_config()
        cfg.dim_w = cfg.dim_a = 512  # the decoder's style width
        fmt_sd = pkg.weights.synth_fmt_state(cfg, seed=6)
        dec_sd = pkg.weights.synth_decoder_state(64, seed=6)
        feats = pkg.weights.synth_feats(64, seed=6)
        T = 13
        c = pkg.pipeline.synth_conditions(cfg, T, seed=1)
        noise = pkg.fmt.draw_noise(1, 1, cfg, seed=15)
        hp = OracleHotPath(cfg, fmt_sd, dec_sd, feats)
        whole = hp.generate(c["r_s"], c["wa"], c["we"], c["s_r"], None, 3, 2.0, 1.0, 1.0, noise=noise)
        hp.sample_calls = 0
        out = {}
        for chain in ("replicate", "broadcast"):
            fr, (t0, t1) = D.generate_frame_sharded(hp, c["r_s"], c["wa"], c["we"], c["s_r"], None, 3, noise=noise, chain=chain)
            out[chain] = (bool(torch.equal(fr, whole[t0:t1])), t0, t1)
        q.put((rank, out, hp.sample_calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_frame_shard_replicate_and_broadcast_two_ranks_gloo():
    """Exact multi-GPU rendering: each rank's shard equals the same frames of the single-process clip, with the latent
    chain replicated (no communication) and with rank 0 sampling and broadcasting r_d (only rank 0 samples)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, out, calls in res:
        for chain in ("replicate", "broadcast"):
            ok, t0, t1 = out[chain]
            assert ok and (t0, t1) == D.frame_shard(13, 2, rank), (rank, chain)
        assert calls == (2 if rank == 0 else 1)  # replicate: both sample; broadcast: rank 0 only


def _exact_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        # history must reach a window's LAST 10 frames for a wrong tail to matter downstream: the band lets information move
        # attention_window tokens per layer, so 2 layers x 5 tokens x 6 evaluations = 60 tokens >= the 50-frame window (with
        # the default window of 2 and a short grid the tail of a window does not depend on its history at all)
        cfg, _, T, wa, r_s, we, noise = _problem(False)
        cfg.attention_window = 5
        sd = pkg.weights.synth_fmt_state(cfg, seed=4)
        fmt = OracleFmt(sd, cfg)
        nfe = 7
        ref = O.sample_rd(sd, cfg, r_s, wa, we, noise, nfe, 2.0, 1.0, 1.0)
        loc, (t0, t1), rep = D.sample_window_parallel(fmt, cfg, r_s, wa, we, noise, nfe, 2.0, 1.0, 1.0, iters=world - 1)
        under, _, rep_u = D.sample_window_parallel(fmt, cfg, r_s, wa, we, noise, nfe, 2.0, 1.0, 1.0, iters=world - 2)
        q.put((rank, t0, t1, bool(torch.equal(loc, ref[:, t0:t1])), float(rep["seam_rel_change"]),
               bool(torch.equal(under, ref[:, t0:t1]))))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_window_parallel_world_minus_one_rounds_is_the_sequential_chain():
    """--window-iters = world-1 reproduces the sequential AR chain bit for bit on every rank (rank k's history is exact after
    k rounds); one round fewer leaves the last rank off, and the last round's seam change of converged ranks is 0."""
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exact_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [(a, b) for _, a, b, _, _, _ in res] == [(0, 100), (100, 150), (150, 170)]
    assert all(exact for _, _, _, exact, _, _ in res)
    assert res[0][4] == 0.0 and res[1][4] == 0.0 and res[2][4] > 0.0  # only the last rank still moved in the last round
    assert res[0][5] and res[1][5] and not res[2][5]


def _seam_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        cfg = pkg.config.small_fmt_config()
        cfg.attention_window = 5  # history must be able to reach a window's last frames (see _exact_worker)
        sd = pkg.weights.synth_fmt_state(cfg, seed=4)
        n_win = 2 * world  # two windows per rank
        T = n_win * cfg.num_frames_for_clip
        g = torch.Generator().manual_seed(2)
        wa = torch.randn(1, T, cfg.dim_a, generator=g)
        r_s = torch.randn(1, cfg.dim_w, generator=g)
        we = torch.softmax(torch.randn(1, 1, cfg.dim_e, generator=g), -1)
        noise = pkg.fmt.draw_noise(n_win, 1, cfg, seed=15)
        fmt = OracleFmt(sd, cfg)
        nfe = 7
        ref = O.sample_rd(sd, cfg, r_s, wa, we, noise, nfe, 2.0, 1.0, 1.0)
        seam_, (t0, t1), rs = D.sample_window_parallel(fmt, cfg, r_s, wa, we, noise, nfe, 2.0, 1.0, 1.0, iters=1, resolve_chunks=1)
        full_, _, rf = D.sample_window_parallel(fmt, cfg, r_s, wa, we, noise, nfe, 2.0, 1.0, 1.0, iters=1, resolve_chunks=0)
        L = cfg.num_frames_for_clip
        err = lambda x: float((x - ref[:, t0:t0 + x.shape[1]]).norm() / ref[:, t0:t0 + x.shape[1]].norm())  # noqa: E731
        q.put((rank, rs["windows_solved"], rf["windows_solved"], float(rs["seam_rel_change"]), float(rs["seam_next_rel_change"]),
               err(seam_), err(full_), err(seam_[:, :L]), err(full_[:, :L]), bool(torch.equal(seam_[:, :L], full_[:, :L]))))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4])
def test_window_parallel_seam_resolve(world):
    """`resolve_chunks=1` re-solves only the rank's first window after the boundary all_gather - (n + 1) / n of
    the rank's windows instead of 2n / n - and reports the change of the hand-off frames at the NEXT boundary as the error
    metric.  The re-solved window is exactly what a full re-solve computes for it; rank 0 never re-solves."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_seam_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, n_seam, n_full, seam, seam_next, e_seam, e_full, e1_seam, e1_full, same_first in res:
        print("rank %d: windows solved %d (seam) vs %d (full); seam change %.2e, next-boundary change %.2e; error vs the "
              "sequential chain %.2e (seam) / %.2e (full)" % (rank, n_seam, n_full, seam, seam_next, e_seam, e_full))
        assert (n_seam, n_full) == ((2, 2) if rank == 0 else (3, 4))      # cost (n + k) / n against 2n / n
        assert same_first                                                # the seam window itself: identical to the full re-solve
        if rank == 0:
            assert seam == 0.0 and seam_next == 0.0 and e_seam < 1e-6
        else:
            assert seam > 0.0 and seam_next > 0.0
            # the windows behind the seam keep their first history: not better than the full re-solve, and the metric is of the
            # size of the error it stands for (rank 1's full re-solve is exact after one round)
            assert e_seam >= e_full - 1e-7
            if rank == 1:
                assert e_full < 1e-5 and e1_seam < 1e-5


def test_bench_rendezvous_names_the_missing_ranks():
    """bench.py's roll call in front of init_process_group: with one of three ranks absent, rank 0 exits 3 within the timeout and
    names the missing rank, the rank that did arrive exits 3 too; with everybody present all get a store."""
    import socket
    import subprocess
    import sys

    def free_port():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        p = s.getsockname()[1]
        s.close()
        return p
    code = ("import os, sys; sys.path.insert(0, %r); import bench; "
            "st = bench.rendezvous_store(int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])); st.set('ok', '1'); print('joined')" % ROOT)

    def run(world, present, timeout):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(world),
                   FLOAT_BENCH_RDZV_TIMEOUT=str(timeout))
        ps = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
              for r in present]
        return [(p.wait(timeout=120), p.stdout.read().decode(), p.stderr.read().decode()) for p in ps]
    res = run(3, [0, 1], 6)
    assert [r[0] for r in res] == [3, 3], res
    assert "rank(s) [2] of 3 never arrived" in res[0][2] and "no go-ahead" in res[1][2]
    res = run(2, [0, 1], 60)
    assert [r[0] for r in res] == [0, 0] and all("joined" in r[1] for r in res), res
