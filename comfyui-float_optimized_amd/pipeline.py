"""The hot path as one object: (conditioning tensors in HBM) -> r_d -> frames.
Mirrors FLOAT.sample + decode_latent_into_processed_images (reference FLOAT.py:172-253, 113-169)
with the noise stream explicit."""
import logging
import math
import os
import warnings

import torch

from . import native
from .config import FmtConfig
from .decoder import SynthesisHIP
from .fmt import FlowMatchingTransformerHIP, WindowSampler, draw_noise


class Fp16RangeError(OverflowError):
    """A 16-bit operator clamped (or stored inf for) activations beyond fp16's range: the result is not the reference's."""


def report_range(counts, where, mode=None):
    """counts: {operator name: float_*_saturation total}.  Any non-zero entry means a checkpoint / input left fp16's range in
    that operator (the frames may hold wrong or black regions).  FLOAT_AMD_RANGE = warn (default: RuntimeWarning + log) |
    raise (Fp16RangeError) | off."""
    bad = {k: v for k, v in counts.items() if v}
    mode = (mode or os.environ.get("FLOAT_AMD_RANGE", "warn")).lower()
    if not bad or mode == "off":
        return bad
    msg = ("%s: fp16 range exceeded in %s - the frames are NOT the reference's within the stated tolerance (they may show wrong "
           "or black regions). Run this checkpoint with dtype fp32 (decoder / encoder) or bf16 / fp32 (FMT / audio)."
           % (where, ", ".join("%s (%d stores)" % kv for kv in sorted(bad.items()))))
    if mode == "raise":
        raise Fp16RangeError(msg)
    logging.getLogger("float_amd").error(msg)
    warnings.warn(msg, RuntimeWarning, stacklevel=3)
    return bad


class FloatHotPath:
    def __init__(self, fmt_state, dec_state, cfg: FmtConfig = None, device="cuda:0", size=512, fmt_dtype="fp16",
                 dec_dtype="fp16", max_frames=32, use_graph=2, max_batch=1):
        self.cfg = cfg or FmtConfig()
        self.device = torch.device(device)
        self.size = size
        self._fmt_state, self._use_graph = fmt_state, use_graph
        self.fmt = FlowMatchingTransformerHIP(fmt_state, self.cfg, device, fmt_dtype, use_graph, max_batch)
        self.dec = SynthesisHIP(dec_state, size, self.cfg.dim_w, device, dec_dtype, max_frames)

    def n_chunks(self, T):
        return int(math.ceil(T / self.cfg.num_frames_for_clip))

    def batched_fmt(self, n_clips):
        """An FMT handle whose workspace holds `n_clips` stacked clips (float_fmt_sample_batch), built on first use from the
        same weights and cached: the default handle is sized for ONE clip (3.1 GB of workspace per clip).  Up to
        FLOAT_AMD_FMT_MAX_BATCH (default 16, the operator's limit: 50 GB of workspace of the 288) clips per chain; larger
        batches run in chunks of that size.  Per clip the chain costs 79 / 52 / 36 / 28 / 24 ms at 1 / 2 / 4 / 8 / 16 clips."""
        cap = max(1, min(16, int(os.environ.get("FLOAT_AMD_FMT_MAX_BATCH", "16"))))
        mb = min(cap, max(1, int(n_clips)))
        if mb > self.fmt.max_batch and mb not in self.__dict__.get("_fmt_batched", {}):
            # 3.1 GB of workspace per stacked clip + the weights once more: size the handle for what the device has free (another
            # model resident in ComfyUI, a smaller GPU) instead of failing with out-of-memory inside the create call
            free, _ = torch.cuda.mem_get_info(self.device)
            fit = int((free - (2 << 30)) // int(3.3 * 2**30))
            if fit < mb:
                logging.getLogger("float_amd").warning("batched FMT handle: %.1f GB of HBM free, stacking %d clips per chain instead of %d",
                                                       free / 2**30, max(1, fit), mb)
                mb = max(1, fit)
        if mb <= self.fmt.max_batch:
            return self.fmt
        cache = self.__dict__.setdefault("_fmt_batched", {})
        if mb not in cache:
            for k in list(cache):
                cache.pop(k).close()
            cache[mb] = FlowMatchingTransformerHIP(self._fmt_state, self.cfg, self.device, self.fmt.dtype, self._use_graph, mb)
            cache[mb].set_method(getattr(self.fmt, "method", "euler"))
        return cache[mb]

    def range_counts(self, reset=True):
        """{operator: float_*_saturation total} of this object's 16-bit handles (synchronises the current stream)."""
        out = {}
        if self.fmt.dtype == "fp16":
            out["fmt"] = self.fmt.saturation(reset) + sum(f.saturation(reset) for f in self.__dict__.get("_fmt_batched", {}).values())
        if self.dec.dtype == "fp16":
            out["decoder"] = self.dec.saturation(reset)
        return out

    @torch.no_grad()
    def sample(self, r_s, wa, we, nfe, a_cfg_scale=2.0, r_cfg_scale=1.0, e_cfg_scale=1.0, seed=15, noise=None,
               include_r_cfg=False):
        """r_d (B,T,512).  `noise` (n_chunks,B,50,512) overrides the seeded CPU-generator draw."""
        if noise is None:
            noise = draw_noise(self.n_chunks(wa.shape[1]), wa.shape[0], self.cfg, seed)
        return self.fmt.sample(r_s, wa, we, noise, nfe, a_cfg_scale, r_cfg_scale, e_cfg_scale, include_r_cfg)

    @torch.no_grad()
    def decode(self, s_r, feats, r_d, frame_range=None):
        """(T,H,W,3) fp32 in [0,1] on the GPU for batch item 0; frame_range=(t0,t1) decodes a shard."""
        if feats is not None:
            self.dec.set_feats(feats)
        rd = r_d[0] if r_d.dim() == 3 else r_d
        if frame_range is not None:
            rd = rd[frame_range[0]:frame_range[1]]
        return self.dec.decode_latent_into_processed_images(s_r, rd)

    def staging(self, n_frames):
        """Device-side frame buffer of float_dec_frames_host, cached per clip length (786 MB for 250 frames at 512 px;
        sized for 288 GB).  The host side is NOT cached: callers get a fresh pinned tensor (torch's caching host allocator
        hands the block of a released earlier result back without a new hipHostMalloc), because ComfyUI keeps node outputs
        alive across executions and a re-used buffer would silently overwrite them."""
        cache = self.__dict__.setdefault("_staging", {})
        shape = (n_frames, self.size, self.size, 3)
        if shape not in cache:
            cache.clear()  # one clip length at a time
            cache[shape] = torch.empty(shape, device=self.device, dtype=torch.float32)
        return cache[shape]

    @torch.no_grad()
    def decode_to_host(self, s_r, r_d, feats=None, frame_range=None, out=None):
        """Frames of one clip (r_d (T,512)) into pinned host memory through float_dec_frames_host: the frames of batch i cross
        PCIe inside the launches of batch i+1.  Returns the host tensor (T,H,W,3); it is complete once the current stream has
        been synchronised (the callers that hand it to the user do that)."""
        if feats is not None:
            self.dec.set_feats(feats)
        rd = r_d[0] if r_d.dim() == 3 else r_d
        if frame_range is not None:
            rd = rd[frame_range[0]:frame_range[1]]
        n = rd.shape[0]
        # The copy workgroups / hipMemcpyAsync of the PREVIOUS call may still be writing its host tensor, and torch's caching
        # host allocator knows nothing about writes it did not issue: this object keeps a reference to that tensor until its
        # event has completed, so the block cannot be handed out again (e.g. as `out` below) while the GPU stores into it.
        # No host wait here: a batch of clips queues its decodes back to back.  Entries leave the list once their event has
        # completed (query, not synchronize); the device-side staging buffer is shared, which is safe in stream order.
        inflight = self.__dict__.setdefault("_host_inflight", [])
        inflight[:] = [(t, e) for t, e in inflight if not e.query()]
        if out is None:
            out = torch.empty((n, self.size, self.size, 3), dtype=torch.float32, pin_memory=True)
        # (the frames by hipMemcpyAsync on a second stream instead of copy workgroups inside the next batch's launches: 121.4-122.1 vs
        # 105.7-106.8 ms per clip on the round-6 kernels, as in round 3 - decoder.decode_into_host(copy_stream=) keeps the form)
        self.dec.decode_into_host(s_r, rd, out, self.staging(n))
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        inflight.append((out, ev))
        return out

    def release_host_inflight(self):
        """Drop the references decode_to_host keeps on host tensors whose copies have completed (call after the stream has been
        synchronised: a batch of 16 clips would otherwise pin 12.6 GB here until the next decode)."""
        inflight = self.__dict__.get("_host_inflight")
        if inflight:
            inflight[:] = [(t, e) for t, e in inflight if not e.query()]

    @torch.no_grad()
    def generate_to_host(self, r_s, wa, we, s_r, feats, nfe, a_cfg_scale=2.0, r_cfg_scale=1.0, e_cfg_scale=1.0, seed=15,
                         noise=None, frame_range=None, out=None, return_rd=False):
        """The product's hot path for one clip (B = 1): conditioning tensors in HBM -> frames in pinned host memory, the
        reference's destination (FLOAT.py:139,157-167).  This is what InferenceAgent.run_inference, FloatProcess and bench.py run."""
        if feats is not None:
            self.dec.set_feats(feats)
        if noise is None:
            noise = draw_noise(self.n_chunks(wa.shape[1]), 1, self.cfg, seed)
        r_d = self.sample(r_s, wa, we, nfe, a_cfg_scale, r_cfg_scale, e_cfg_scale, seed, noise)
        host = self.decode_to_host(s_r, r_d, None, frame_range, out)
        return (host, r_d) if return_rd else host

    def _overlap_streams(self, mode):
        """(chain stream, decoder stream) of the stage-overlapped form.  mode "prio": two streams of one device queue set, the
        chain on the higher priority; "cu:N": disjoint CU sets (hipExtStreamCreateWithCUMask), the decoder on the last N CUs."""
        cache = self.__dict__.setdefault("_ov_streams", {})
        if mode not in cache:
            dev = self.device
            if mode.startswith("cu:"):
                n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
                n_dec = max(8, min(n_cu - 8, int(mode[3:])))
                with torch.cuda.device(dev):
                    cache[mode] = (native.cu_range_stream(0, n_cu - n_dec, dev), native.cu_range_stream(n_cu - n_dec, n_cu, dev))
            elif mode in ("prio", "plain", "hi", "lo"):
                # (least, greatest) priority: the greater priority is the SMALLER number; 0 is the default.  "hi" / "lo" raise the
                # chain / lower the decoder only (tools/probes/overlap_check.py)
                lo, hi = torch.cuda.Stream.priority_range()
                pf = hi if mode in ("prio", "hi") else 0
                pd = lo if mode in ("prio", "lo") else 0
                cache[mode] = (torch.cuda.Stream(dev, priority=pf), torch.cuda.Stream(dev, priority=pd))
            else:
                raise ValueError("overlap mode must be 'prio' or 'cu:N', got %r" % (mode,))
        return cache[mode]

    @torch.no_grad()
    def generate_to_host_overlap(self, r_s, wa, we, s_r, nfe, a_cfg_scale=2.0, r_cfg_scale=1.0, e_cfg_scale=1.0, noise=None,
                                 out=None, mode="prio", return_rd=False):
        """generate_to_host with the two stages pipelined (FLOAT.py runs 209-253 then 113-169; here window k is decoded and
        handed to the host on a second stream while the chain samples window k + 1).  Same kernels on the same operands as
        the sequential order, per-window decode batches (50 frames = 32 + 18 instead of 250 = 7 x 32 + 26): frames bitwise
        equal to generate_to_host (decode batching does not change a frame, tests/test_dec_gpu.py).  FLOAT_AMD_OVERLAP
        selects it in InferenceAgent.infer_device; what it measures against the sequential order: DESIGN.md section 7."""
        T = wa.shape[1]
        dev = self.device
        if noise is None:
            noise = draw_noise(self.n_chunks(T), 1, self.cfg, 15)
        s_fmt, s_dec = self._overlap_streams(mode)
        cur = torch.cuda.current_stream(dev)
        s_fmt.wait_stream(cur)
        s_dec.wait_stream(cur)
        inflight = self.__dict__.setdefault("_host_inflight", [])
        inflight[:] = [(t, e) for t, e in inflight if not e.query()]
        if out is None:
            out = torch.empty((T, self.size, self.size, 3), dtype=torch.float32, pin_memory=True)
        staging = self.staging(T)
        s_r_d = s_r.to(dev, torch.float32).reshape(-1).contiguous()
        with torch.cuda.stream(s_fmt):
            ws = WindowSampler(self.fmt, r_s, wa, we, noise, nfe, a_cfg_scale, r_cfg_scale, e_cfg_scale)
        while ws.left > 0:
            with torch.cuda.stream(s_fmt):
                _, (f0, f1) = ws.next()
                ev = torch.cuda.Event()
                ev.record(s_fmt)
            with torch.cuda.stream(s_dec):
                s_dec.wait_event(ev)
                self.dec.decode_into_host(s_r_d, ws.r_d[0, f0:f1], out[f0:f1], staging[f0:f1])
        cur.wait_stream(s_dec)
        cur.wait_stream(s_fmt)
        done = torch.cuda.Event()
        done.record(cur)
        inflight.append((out, done))
        self._last_job = ws  # the job's tensors were allocated on the side streams: alive until the caller has synchronised
        return (out, ws.r_d) if return_rd else out

    @torch.no_grad()
    def generate(self, r_s, wa, we, s_r, feats, nfe, a_cfg_scale=2.0, r_cfg_scale=1.0, e_cfg_scale=1.0, seed=15,
                 noise=None, overlap=False, frame_range=None, return_rd=False):
        """Whole hot path for one clip (B = 1): frames (T,H,W,3) on the GPU.

        overlap=True pipelines the two stages on two HIP streams: the FMT chain of window k+1 runs while
        the decoder renders the 50 frames of window k.  Results are identical to the sequential order
        (same kernels, same operands).  Measured on MI355X it does NOT pay (r01: 183 vs 174 ms per 10 s
        clip): the decoder's grids own every CU, so each of the chain's ~3000 tiny dependent kernels per
        window queues behind running decoder workgroups; kept as an option for CU-partitioned streams.
        frame_range=(t0,t1) decodes only that shard of the clip (multi-GPU frame sharding)."""
        if feats is not None:
            self.dec.set_feats(feats)
        T = wa.shape[1]
        if noise is None:
            noise = draw_noise(self.n_chunks(T), 1, self.cfg, seed)
        if not overlap:
            r_d = self.sample(r_s, wa, we, nfe, a_cfg_scale, r_cfg_scale, e_cfg_scale, seed, noise)
            frames = self.decode(s_r, None, r_d, frame_range)
            return (frames, r_d) if return_rd else frames
        t0, t1 = frame_range if frame_range is not None else (0, T)
        dev = self.device
        if not hasattr(self, "_s_fmt"):
            split = getattr(self, "cu_split", None)
            if split:  # disjoint CU sets: chain on CUs [0, split), decoder on [split, n_cu)
                n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
                with torch.cuda.device(dev):
                    self._s_fmt = native.cu_range_stream(0, split, dev)
                    self._s_dec = native.cu_range_stream(split, n_cu, dev)
            else:
                self._s_fmt = torch.cuda.Stream(dev, priority=getattr(self, "fmt_stream_priority", -1))
                self._s_dec = torch.cuda.Stream(dev, priority=0)
        cur = torch.cuda.current_stream(dev)
        self._s_fmt.wait_stream(cur)
        self._s_dec.wait_stream(cur)
        out = torch.empty(t1 - t0, self.size, self.size, 3, device=dev, dtype=torch.float32)
        s_r_d = s_r.to(dev, torch.float32).reshape(-1).contiguous()
        with torch.cuda.stream(self._s_fmt):
            ws = WindowSampler(self.fmt, r_s, wa, we, noise, nfe, a_cfg_scale, r_cfg_scale, e_cfg_scale)
        L = native.lib()
        while ws.left > 0:
            with torch.cuda.stream(self._s_fmt):
                _, (f0, f1) = ws.next()
                ev = torch.cuda.Event()
                ev.record(self._s_fmt)
            a, b = max(f0, t0), min(f1, t1)
            if a >= b:
                continue
            with torch.cuda.stream(self._s_dec):
                self._s_dec.wait_event(ev)
                rd = ws.r_d[0, a:b]
                native.check(L.float_dec_frames(self.dec._h, native.dev_ptr(s_r_d), native.dev_ptr(rd), b - a,
                                                native.dev_ptr(out[a - t0:b - t0]), native.stream_ptr(dev)))
        cur.wait_stream(self._s_dec)
        cur.wait_stream(self._s_fmt)
        # keep the job's tensors alive until the streams have been joined
        self._last_job = ws
        return (out, ws.r_d) if return_rd else out


def synth_conditions(cfg: FmtConfig, T, seed=0, dynamic_we=False, device="cpu"):
    """Synthetic stand-ins for the off-path encoders' outputs (formats of SURVEY.md section 8a):
    wa ~ SiLU(LayerNorm-ed projection) (FLOAT.py:338-342), we softmax scores, r_s, s_r."""
    g = torch.Generator().manual_seed(1000 + seed)
    wa = torch.nn.functional.silu(torch.randn(1, T, cfg.dim_a, generator=g))
    if dynamic_we:
        nwin = int(math.ceil(T / cfg.num_frames_for_clip))
        w = torch.softmax(torch.randn(1, nwin, cfg.dim_e, generator=g), -1)
        idx = torch.clamp((torch.arange(T).float() * nwin / T).long(), max=nwin - 1)  # nearest upsample, nodes_vadv.py:835-838
        we = w[:, idx]
    else:
        we = torch.softmax(torch.randn(1, 1, cfg.dim_e, generator=g), -1)
    r_s = torch.randn(1, cfg.dim_w, generator=g) * 0.5
    s_r = torch.randn(1, cfg.dim_w, generator=g)
    return dict(wa=wa.to(device), we=we.to(device), r_s=r_s.to(device), s_r=s_r.to(device))
