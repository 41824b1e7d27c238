"""Multi-GPU use of the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on MI355X; "gloo" in the CPU tests).  Nothing here is in the reference (it has no distributed
code); SURVEY.md section 8e defines the three modes.

  replicas         independent clips, one per rank - no communication (BASELINE configs[3]).
  frame shard      ONE long clip, exact: the latent chain is sequential (window k needs the last 10
                   sampled frames of window k-1, FLOAT.py:217-222), so every rank runs the identical,
                   bitwise-deterministic chain (or rank 0 runs it and broadcasts r_d: T*512*4 bytes,
                   3 MB for 60 s) and decodes only its contiguous frame range.  Frames never cross
                   xGMI - each rank copies its own shard to the host.
  window parallel  ONE long clip, approximate (the north-star wording): rank w owns a contiguous
                   range of 50-frame windows and first samples them from zero history, like window 0;
                   one all_gather of the boundary latents (last 10 frames of x, wa, we: ~41 KB per
                   rank, latency-bound on xGMI) gives every rank its predecessor's tail, and the rank
                   re-solves from that history - its whole range (`iters` such rounds make ranks
                   0..iters exact; world-1 rounds reproduce the sequential chain bit for bit), or only
                   its first `resolve_chunks` windows, the seam (cost n + k instead of 2n windows per
                   rank and round); the residual seam error is reported, never hidden.
"""
import math

import torch
import torch.distributed as dist


def frame_shard(T, world, rank):
    """Contiguous balanced frame range of `rank`."""
    base, rem = divmod(T, world)
    t0 = rank * base + min(rank, rem)
    return t0, t0 + base + (1 if rank < rem else 0)


def window_shard(n_windows, world, rank):
    return frame_shard(n_windows, world, rank)


def _pad_rep(a, n):
    if a.shape[1] >= n:
        return a
    return torch.cat([a, a[:, -1:].expand(-1, n - a.shape[1], -1)], dim=1)


def sample_range(fmt, cfg, r_s, wa, we, noise, w0, w1, nfe, a_cfg, r_cfg, e_cfg, hist=None):
    """The AR window loop (FLOAT.py:209-253) over windows [w0, w1) starting from `hist`
    (prev_x, prev_wa, prev_we) - zeros when None, as for window 0.  `fmt` needs
    sample_chunk(x0, wa, wr, we, prev_x, prev_wa, prev_we, nfe=, a_cfg_scale=, r_cfg_scale=, e_cfg_scale=).
    ONE contract for both implementations (the host loop below and the operator's own window loop, picked for a HIP handle
    with one clip): returns (samples (B, min(T, w1 L) - w0 L, dim_w) - the range's rows TRIMMED to the clip - and
    tail = (prev_x, prev_wa, prev_we), the history window w1 would start from.  When window w1 - 1 is the clip's last, trimmed
    window nobody samples after it and prev_x is zeros (prev_wa / prev_we are the tails of the replicate-padded window)."""
    L, P = cfg.num_frames_for_clip, cfg.num_prev_frames
    B = wa.shape[0]
    dev, dt = wa.device, wa.dtype
    dynamic = we.shape[1] > 1
    if hasattr(fmt, "_h") and B == 1 and P > 0:
        return _sample_range_native(fmt, cfg, r_s, wa, we, noise, w0, w1, nfe, a_cfg, r_cfg, e_cfg, hist)
    if hist is None:
        hist = (torch.zeros(B, P, cfg.dim_w, device=dev, dtype=dt), torch.zeros(B, P, cfg.dim_a, device=dev, dtype=dt),
                torch.zeros(B, P, cfg.dim_e, device=dev, dtype=dt))
    prev_x, prev_wa, prev_we = hist
    out = []
    for k in range(w0, w1):
        wa_c = _pad_rep(wa[:, k * L:(k + 1) * L], L)
        we_c = _pad_rep(we[:, k * L:(k + 1) * L], L) if dynamic else we
        xs = fmt.sample_chunk(noise[k], wa_c, r_s, we_c, prev_x, prev_wa, prev_we if dynamic else None, nfe=nfe,
                              a_cfg_scale=a_cfg, r_cfg_scale=r_cfg, e_cfg_scale=e_cfg)
        xs = xs.to(dev, dt)
        out.append(xs)
        prev_x, prev_wa = xs[:, -P:], wa_c[:, -P:]
        if dynamic:
            prev_we = we_c[:, -P:]
    T = wa.shape[1]
    xs = torch.cat(out, dim=1)[:, :min(T, w1 * L) - w0 * L]
    if w1 * L > T:  # the clip's last, trimmed window: same tail as the native loop gives (nobody samples after it)
        prev_x = torch.zeros_like(prev_x)
    return xs, (prev_x, prev_wa, prev_we)


def _sample_range_native(fmt, cfg, r_s, wa, we, noise, w0, w1, nfe, a_cfg, r_cfg, e_cfg, hist):
    """sample_range on the HIP operator's own window loop (float_fmt_sample_begin_range / _next): every window is one
    hipGraph replay with the hand-off done on the device, instead of one sample_chunk call with host-side slicing per window.
    Returns the range's rows of r_d trimmed to the clip (the last window of the clip has T - k L rows) and the tail."""
    from .fmt import WindowSampler
    L, P = cfg.num_frames_for_clip, cfg.num_prev_frames
    T = wa.shape[1]
    dynamic = we.shape[1] > 1
    ws = WindowSampler(fmt, r_s, wa, we, noise, nfe, a_cfg, r_cfg, e_cfg, windows=(w0, w1), hist=hist)
    while ws.left > 0:
        ws.next()
    t0, t1 = w0 * L, min(T, w1 * L)
    xs = ws.r_d[:, t0:t1]
    fmt._last_job = ws  # keeps the job's tensors alive until the stream has run it
    if w1 * L <= T:
        prev_x = xs[:, -P:]
    else:  # the clip's last, trimmed window: nobody samples after it
        prev_x = torch.zeros(1, P, cfg.dim_w, device=xs.device, dtype=xs.dtype)
    k = w1 - 1
    prev_wa = _pad_rep(wa[:, k * L:(k + 1) * L], L)[:, -P:].to(xs.device, xs.dtype)
    prev_we = (_pad_rep(we[:, k * L:(k + 1) * L], L)[:, -P:] if dynamic else torch.zeros(1, P, cfg.dim_e)).to(xs.device, xs.dtype)
    return xs, (prev_x, prev_wa, prev_we)


def sample_window_parallel(fmt, cfg, r_s, wa, we, noise, nfe, a_cfg=2.0, r_cfg=1.0, e_cfg=1.0, iters=1, group=None,
                           resolve_chunks=0, exchange_at_world_1=False):
    """Window-parallel sampling of one clip.  Every rank passes the FULL wa/we/noise (they are tiny);
    returns (r_d_local, (t0, t1), report).

    Each of the `iters` rounds is one all_gather of the boundary latents followed by a re-solve from the predecessor's tail:
      resolve_chunks = 0  the rank's WHOLE window range (cost: (1 + iters) x its windows).  Rank k is exact after k rounds, so
                          iters = world - 1 reproduces the sequential chain bit for bit: the exact mode.
      resolve_chunks = k  only the rank's first k windows - the seam - (cost: n + iters * k windows for a rank of n; SURVEY 8e's
                          first-chunk re-solve).  The windows behind them keep the history they were solved with; whether
                          that matters is what `seam_next_rel_change` measures.
    report (0-dim tensors on the latents' device - the caller converts them when it reports, not inside the timed loop):
      seam_rel_change       rel-L2 change of the re-solved latents caused by the last exchange (0 once the rank's history has
                            converged to the sequential chain's);
      seam_next_rel_change  the error metric of the seam re-solve: rel-L2 change, in the last round, of the hand-off frames
                            (last num_prev_frames frames) of the last re-solved window - the history the NEXT, not re-solved
                            window was computed from.  0 means the rest of the rank's range is what a full re-solve would give;
      windows_solved        windows this rank solved in total (the cost).
    exchange_at_world_1: run the all_gather rounds even in a one-rank group (nothing is re-solved: rank 0 has no predecessor) -
    how a single-GPU box exercises the collective on the backend that N > 1 will use."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    L, P = cfg.num_frames_for_clip, cfg.num_prev_frames
    T = wa.shape[1]
    n_win = int(math.ceil(T / L))
    w0, w1 = window_shard(n_win, world, rank)
    xs, tail = sample_range(fmt, cfg, r_s, wa, we, noise, w0, w1, nfe, a_cfg, r_cfg, e_cfg, None)
    solved = w1 - w0
    seam = torch.zeros((), device=xs.device, dtype=torch.float32)  # stays on the device: no host sync per round
    seam_next = torch.zeros((), device=xs.device, dtype=torch.float32)
    k = (w1 - w0) if resolve_chunks <= 0 else min(int(resolve_chunks), w1 - w0)
    if resolve_chunks > 0 and iters > 1:
        # a partial re-solve never changes a rank's own tail (its LAST window), so every later round would gather the same
        # boundaries and re-solve the same windows to the same result: wasted work and a seam change of 0 that reads as
        # converged.  Only the whole-range re-solve (resolve_chunks = 0) iterates.
        raise ValueError("iters > 1 needs resolve_chunks = 0 (a seam re-solve of %d window(s) is a single round)" % resolve_chunks)
    for _ in range(iters if (world > 1 or (exchange_at_world_1 and dist.is_initialized())) else 0):
        # boundary latents of every rank: [x tail | wa tail | we tail] flattened, one all_gather
        mine = torch.cat([t.reshape(t.shape[0], -1) for t in tail], dim=1).contiguous()
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine, group=group)
        if rank > 0 and w1 > w0:
            g = gathered[rank - 1]
            B = g.shape[0]
            nx, na = P * cfg.dim_w, P * cfg.dim_a
            hist = (g[:, :nx].reshape(B, P, cfg.dim_w), g[:, nx:nx + na].reshape(B, P, cfg.dim_a),
                    g[:, nx + na:].reshape(B, P, cfg.dim_e))
            new_xs, new_tail = sample_range(fmt, cfg, r_s, wa, we, noise, w0, w0 + k, nfe, a_cfg, r_cfg, e_cfg, hist)
            solved += k
            old = xs[:, :k * L]
            seam = ((new_xs - old).norm() / (new_xs.norm() + 1e-30)).float()
            seam_next = ((new_xs[:, -P:] - old[:, -P:]).norm() / (new_xs[:, -P:].norm() + 1e-30)).float()
            if k == w1 - w0:
                xs, tail = new_xs, new_tail
            else:  # the seam windows replace their first versions; the rank's own tail (its LAST window) is unchanged
                xs = torch.cat([new_xs, xs[:, k * L:]], dim=1)
    t0, t1 = w0 * L, min(T, w1 * L)
    return xs[:, :t1 - t0], (t0, t1), {"seam_rel_change": seam, "seam_next_rel_change": seam_next, "windows": (w0, w1),
                                       "rounds": iters, "resolve_chunks": k, "windows_solved": solved}


def broadcast_latents(r_d, src=0, group=None, at_world_1=False):
    """Exact mode, variant 2: rank `src` sampled the chain, everybody else receives r_d (B,T,512)."""
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or at_world_1):
        dist.broadcast(r_d, src=src, group=group)
    return r_d


def generate_frame_sharded(hot_path, r_s, wa, we, s_r, feats, nfe, a_cfg=2.0, r_cfg=1.0, e_cfg=1.0, noise=None, seed=15,
                           chain="replicate", group=None):
    """Exact multi-GPU rendering of one clip: returns (frames of this rank's range, (t0, t1)).
    chain="replicate": every rank runs the deterministic latent chain itself (zero communication);
    chain="broadcast": rank 0 runs it, one broadcast of r_d."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    T = wa.shape[1]
    t0, t1 = frame_shard(T, world, rank)
    if chain == "replicate" or world == 1:
        frames = hot_path.generate(r_s, wa, we, s_r, feats, nfe, a_cfg, r_cfg, e_cfg, seed=seed, noise=noise, frame_range=(t0, t1))
        return frames, (t0, t1)
    if chain != "broadcast":
        raise ValueError("chain must be 'replicate' or 'broadcast'")
    if rank == 0:
        r_d = hot_path.sample(r_s, wa, we, nfe, a_cfg, r_cfg, e_cfg, seed=seed, noise=noise)
    else:
        r_d = torch.empty(wa.shape[0], T, hot_path.cfg.dim_w, device=hot_path.device, dtype=torch.float32)
    broadcast_latents(r_d, 0, group)
    return hot_path.decode(s_r, feats, r_d, (t0, t1)), (t0, t1)
