"""Host mirror of the reference's FlowMatchingTransformer sampling interface, running on the HIP
operator (`float_fmt_*`, include/float_hip.h).  Names and argument meaning follow
reference FMT.py:342-345 (forward_with_cfv) and nodes_adv.py:545-572 (_perform_ode_sampling_loop)."""
import ctypes as C
import math

import torch

from . import native
from .config import FmtConfig


@native.rebuildable
class FlowMatchingTransformerHIP:
    """Holds packed weights + workspace on one GPU.  `sample` stacks up to `max_batch` clips per launch chain
    (float_fmt_sample_batch); the single-evaluation / single-window calls loop batch items on the host, like the
    reference's FloatProcess (nodes.py:189-209)."""

    def __init__(self, state_dict, cfg: FmtConfig = None, device="cuda:0", dtype="fp16", use_graph=2, max_batch=1):
        """max_batch: clips `sample` may run through one launch chain (float_fmt_sample_batch); larger batches are cut
        into groups of that size.  Sizes the workspace (3.1 GB of modulation slab per clip at the default shape)."""
        self.cfg = cfg or FmtConfig()
        self.device = torch.device(device)
        self.dtype = dtype = native.canon_dtype(dtype)
        self.max_batch = max(1, int(max_batch))
        L = native.lib()
        c = self.cfg
        ncfg = native.FmtCfg(c.dim_w, c.dim_a, c.dim_e, c.dim_h, c.fmt_depth, c.num_heads,
                             int(c.dim_h * c.mlp_ratio), c.num_prev_frames, c.num_frames_for_clip,
                             c.attention_window, native.DTYPES[dtype], int(use_graph), self.max_batch)
        sd = {k[4:] if k.startswith("fmt.") else k: v for k, v in state_dict.items()
              if k not in ("alignment_mask", "fmt.alignment_mask")}
        arr, keep = native.tensor_table(sd)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            native.check(L.float_fmt_create(C.byref(ncfg), arr, len(sd), C.byref(h)))
        self._h = h
        del keep

    def set_method(self, name):
        """torchdiffeq fixed-grid method name: euler | midpoint | rk4 | heun2 | heun3 (base_options.py:50)."""
        if name not in native.ODE_METHODS:
            raise ValueError("unknown fixed-step ODE method %r (known: %s)" % (name, ", ".join(native.ODE_METHODS)))
        native.check(native.lib().float_fmt_set_method(self._h, native.ODE_METHODS[name]))
        self.method = name

    # ------------------------------------------------------------------ test hooks (float_fmt_debug)
    @torch.no_grad()
    def pos_embed_in_use(self):
        """(n_tokens, dim_h): the positional table the operator adds (checkpoint's or regenerated, FMT.py:22-40,249-250)."""
        out = torch.empty(self.cfg.n_tokens, self.cfg.dim_h, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            native.check(native.lib().float_fmt_debug(self._h, 0, None, native.dev_ptr(out), native.stream_ptr(self.device)))
        return out

    @torch.no_grad()
    def attention_probe(self, qkv):
        """qkv (n_tokens, 3*dim_h) = [q | k | v] -> (n_tokens, dim_h): the chain's banded attention kernel on caller data."""
        qkv = self._f(qkv)
        if qkv.shape != (self.cfg.n_tokens, 3 * self.cfg.dim_h):
            raise ValueError("qkv must be (%d,%d)" % (self.cfg.n_tokens, 3 * self.cfg.dim_h))
        out = torch.empty(self.cfg.n_tokens, self.cfg.dim_h, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            native.check(native.lib().float_fmt_debug(self._h, 1, native.dev_ptr(qkv), native.dev_ptr(out),
                                                      native.stream_ptr(self.device)))
        return out

    def saturation(self, reset=False):
        """Threads with a clamped (+-65504) or non-finite 16-bit activation store since create / the last reset
        (float_fmt_saturation): 0 unless the checkpoint leaves fp16's range - then the result is not the reference's within the
        stated tolerance; run it with dtype="fp32" (or "bf16").  Always 0 for bf16 / fp32 handles.  Synchronises the current stream."""
        with torch.cuda.device(self.device):
            return native.saturation("float_fmt_saturation", self._h, self.device, reset)

    def close(self):
        if getattr(self, "_h", None) and native is not None:  # `native` is None during interpreter shutdown
            native.lib().float_fmt_destroy(self._h)
            self._h = None

    __del__ = close

    # ------------------------------------------------------------------ helpers
    def _f(self, t):
        return None if t is None else t.to(self.device, torch.float32).contiguous()

    def _check_shapes(self, x, wa, wr, we, prev_x, prev_wa, prev_we):
        c = self.cfg
        B = x.shape[0]
        if x.shape != (B, c.num_frames_for_clip, c.dim_w):
            raise ValueError("x must be (B,%d,%d), got %s" % (c.num_frames_for_clip, c.dim_w, tuple(x.shape)))
        if wa.shape != (B, c.num_frames_for_clip, c.dim_a):
            raise ValueError("wa must be (B,%d,%d), got %s" % (c.num_frames_for_clip, c.dim_a, tuple(wa.shape)))
        if wr.shape != (B, c.dim_w):
            raise ValueError("wr must be (B,%d), got %s" % (c.dim_w, tuple(wr.shape)))
        if prev_x is None or prev_wa is None:
            raise ValueError("prev_x was provided, but prev_wa was not." if prev_x is not None
                             else "prev_x / prev_wa are required")
        if we.dim() != 3 or we.shape[2] != c.dim_e:
            raise ValueError("we must be (B,1|%d,%d)" % (c.num_frames_for_clip, c.dim_e))
        if we.shape[1] > 1 and prev_we is None:
            raise ValueError("`we` is dynamic (T>1), but prev_we was not provided with prev_x/prev_wa.")
        if we.shape[1] not in (1, c.num_frames_for_clip):
            raise ValueError("Dynamic emotion latent `we` time dimension (%d) does not match audio latent `wa` "
                             "time dimension (%d)." % (we.shape[1] + c.num_prev_frames,
                                                       c.num_frames_for_clip + c.num_prev_frames))

    # ------------------------------------------------------------------ reference-shaped API
    @torch.no_grad()
    def forward_with_cfv(self, t, x, wa, wr, we, prev_x, prev_wa, prev_we=None, a_cfg_scale=1.0, r_cfg_scale=1.0,
                         e_cfg_scale=1.0, include_r_cfg=False, **kwargs):
        """FMT.py:342-401.  Returns (B, n_prev+n_cur, dim_w) on the GPU."""
        x, wa, wr, we, prev_x, prev_wa, prev_we = map(self._f, (x, wa, wr, we, prev_x, prev_wa, prev_we))
        self._check_shapes(x, wa, wr, we, prev_x, prev_wa, prev_we)
        c, L = self.cfg, native.lib()
        tval = float(torch.as_tensor(t).reshape(-1)[0])
        B = x.shape[0]
        out = torch.empty(B, c.n_tokens, c.dim_w, device=self.device, dtype=torch.float32)
        dynamic = we.shape[1] > 1
        with torch.cuda.device(self.device):
            s = native.stream_ptr(self.device)
            for b in range(B):
                native.check(L.float_fmt_eval(
                    self._h, tval, native.dev_ptr(x[b]), native.dev_ptr(wa[b]), native.dev_ptr(wr[b]),
                    native.dev_ptr(we[b]), we.shape[1], native.dev_ptr(prev_x[b]), native.dev_ptr(prev_wa[b]),
                    native.dev_ptr(prev_we[b]) if dynamic else None, a_cfg_scale, r_cfg_scale, e_cfg_scale,
                    1 if include_r_cfg else 0, native.dev_ptr(out[b]), s))
        return out

    @torch.no_grad()
    def sample_chunk(self, x0, wa, wr, we, prev_x, prev_wa, prev_we=None, nfe=10, a_cfg_scale=1.0, r_cfg_scale=1.0,
                     e_cfg_scale=1.0, include_r_cfg=False):
        """odeint(euler) over linspace(0,1,nfe) for one window (FLOAT.py:229-248): (B, n_cur, dim_w)."""
        x0, wa, wr, we, prev_x, prev_wa, prev_we = map(self._f, (x0, wa, wr, we, prev_x, prev_wa, prev_we))
        self._check_shapes(x0, wa, wr, we, prev_x, prev_wa, prev_we)
        c, L = self.cfg, native.lib()
        B = x0.shape[0]
        out = torch.empty(B, c.num_frames_for_clip, c.dim_w, device=self.device, dtype=torch.float32)
        dynamic = we.shape[1] > 1
        with torch.cuda.device(self.device):
            s = native.stream_ptr(self.device)
            for b in range(B):
                native.check(L.float_fmt_sample_chunk(
                    self._h, native.dev_ptr(x0[b]), native.dev_ptr(wa[b]), native.dev_ptr(wr[b]),
                    native.dev_ptr(we[b]), we.shape[1], native.dev_ptr(prev_x[b]), native.dev_ptr(prev_wa[b]),
                    native.dev_ptr(prev_we[b]) if dynamic else None, int(nfe), a_cfg_scale, r_cfg_scale, e_cfg_scale,
                    1 if include_r_cfg else 0, native.dev_ptr(out[b]), s))
        return out

    @torch.no_grad()
    def sample(self, r_s, wa, we, noise, nfe=10, a_cfg_scale=2.0, r_cfg_scale=1.0, e_cfg_scale=1.0,
               include_r_cfg=False):
        """The AR window loop (FLOAT.py:209-253; nodes_adv.py:578-694).
        r_s (B,dim_w); wa (B,T,dim_a); we (B,1|T,dim_e); noise (n_chunks,B,n_cur,dim_w) explicit.
        Returns r_d (B,T,dim_w) on the GPU."""
        r_s, wa, we, noise = map(self._f, (r_s, wa, we, noise))
        c, L = self.cfg, native.lib()
        B, T = wa.shape[0], wa.shape[1]
        n_chunks = int(math.ceil(T / c.num_frames_for_clip))
        if noise.shape != (n_chunks, B, c.num_frames_for_clip, c.dim_w):
            raise ValueError("noise must be (%d,%d,%d,%d), got %s" % (n_chunks, B, c.num_frames_for_clip, c.dim_w,
                                                                      tuple(noise.shape)))
        if we.shape[1] not in (1, T):
            raise ValueError("Dynamic emotion latent `we` time dimension (%d) does not match audio latent `wa` "
                             "time dimension (%d)." % (we.shape[1], T))
        r_d = torch.empty(B, T, c.dim_w, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            s = native.stream_ptr(self.device)
            for b0 in range(0, B, self.max_batch):
                nb = min(self.max_batch, B - b0)
                nz = noise[:, b0:b0 + nb].contiguous()
                native.check(L.float_fmt_sample_batch(
                    self._h, nb, native.dev_ptr(r_s[b0:b0 + nb]), native.dev_ptr(wa[b0:b0 + nb]), T,
                    native.dev_ptr(we[b0:b0 + nb]), we.shape[1], native.dev_ptr(nz), int(nfe), a_cfg_scale, r_cfg_scale,
                    e_cfg_scale, 1 if include_r_cfg else 0, native.dev_ptr(r_d[b0:b0 + nb]), s))
        return r_d


class WindowSampler:
    """Incremental form of FlowMatchingTransformerHIP.sample for ONE clip (B = 1): each next() enqueues
    one 50-frame window on the current stream, so the caller can decode window k on another stream
    while window k+1 is being sampled."""

    def __init__(self, fmt, r_s, wa, we, noise, nfe, a_cfg_scale, r_cfg_scale, e_cfg_scale, include_r_cfg=False,
                 windows=None, hist=None, r_d=None):
        """windows=(w0, w1): only that window range of the clip, starting from hist = (prev_x, prev_wa, prev_we) (each
        (1, n_prev, dim) or None = zeros) - float_fmt_sample_begin_range, the multi-GPU window shard.  r_d: a (1, T, dim_w)
        device tensor to write into (else a new one; rows outside the job's windows are left as they are)."""
        self.fmt = fmt
        c = fmt.cfg
        f = fmt._f
        self.r_s, self.wa, self.we = f(r_s).reshape(-1), f(wa).reshape(-1, c.dim_a), f(we).reshape(-1, c.dim_e)
        self.T = self.wa.shape[0]
        self.n_chunks = int(math.ceil(self.T / c.num_frames_for_clip))
        self.noise = f(noise).reshape(self.n_chunks, c.num_frames_for_clip, c.dim_w)
        self.r_d = torch.empty(1, self.T, c.dim_w, device=fmt.device, dtype=torch.float32) if r_d is None else r_d
        w0, w1 = windows if windows is not None else (0, self.n_chunks)
        self.left = w1 - w0
        args = (fmt._h, native.dev_ptr(self.r_s), native.dev_ptr(self.wa), self.T, native.dev_ptr(self.we),
                self.we.shape[0], native.dev_ptr(self.noise), int(nfe), a_cfg_scale, r_cfg_scale, e_cfg_scale,
                1 if include_r_cfg else 0, native.dev_ptr(self.r_d))
        with torch.cuda.device(fmt.device):
            if windows is None and hist is None:
                native.check(native.lib().float_fmt_sample_begin(*args))
            else:
                # the history tensors must stay alive until the first next() has run on the stream
                self._hist = [None if t is None else f(t).reshape(c.num_prev_frames, -1) for t in (hist or (None, None, None))]
                native.check(native.lib().float_fmt_sample_begin_range(
                    *args, int(w0), int(w1), *[native.dev_ptr(t) if t is not None else None for t in self._hist]))

    def next(self):
        """Enqueue the next window on the current stream; returns (window index, frame range)."""
        k, left = C.c_int32(0), C.c_int32(0)
        with torch.cuda.device(self.fmt.device):
            native.check(native.lib().float_fmt_sample_next(self.fmt._h, native.stream_ptr(self.fmt.device),
                                                            C.byref(k), C.byref(left)))
        self.left = left.value
        L = self.fmt.cfg.num_frames_for_clip
        return k.value, (k.value * L, min(self.T, (k.value + 1) * L))


def draw_noise(n_chunks, batch, cfg, seed, device="cpu"):
    """The reference's noise stream made explicit: sequential randn(B, n_cur, dim_w) draws from one
    generator seeded once per clip (FLOAT.py:203-215)."""
    g = torch.Generator(device)
    g.manual_seed(int(seed))
    return torch.stack([torch.randn(batch, cfg.num_frames_for_clip, cfg.dim_w, generator=g, device=device)
                        for _ in range(n_chunks)])
