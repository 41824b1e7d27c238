"""Host mirror of the motion-autoencoder decoder (reference styledecoder.py Synthesis and the
decode loop FLOAT.py:113-169), running on the HIP operator (`float_dec_*`, include/float_hip.h)."""
import ctypes as C

import torch

from . import native


def blur_taps(blur_kernel):
    """The loader's `blur_kernel` as the operator's optional `blur_kernel` tensor: None for the default [1,3,3,1], else a
    (4,) fp32 tensor.  Other lengths change the Blur's padding (styledecoder.py:209-213) and are not implemented."""
    if blur_kernel is None:
        return None
    k = torch.as_tensor(blur_kernel, dtype=torch.float32).reshape(-1)
    if k.numel() != 4:
        raise ValueError("the HIP decoder implements 4-tap blur kernels (got %d taps: %s)" % (k.numel(), list(blur_kernel)))
    if abs(float(k.sum())) < 1e-12:
        raise ValueError("blur_kernel sums to zero: %s" % (list(blur_kernel),))
    return None if k.tolist() == [1.0, 3.0, 3.0, 1.0] else k


@native.rebuildable
class SynthesisHIP:
    def __init__(self, state_dict, size=512, style_dim=512, device="cuda:0", dtype="fp16", max_frames=16, blur_kernel=None):
        """blur_kernel: Synthesis(blur_kernel=...) of the reference (styledecoder.py:448, handed to the StyledConvs only, :486-488;
        ToRGB / ToFlow keep their default [1,3,3,1]).  None = [1,3,3,1]; any 4-tap kernel is accepted (ValueError otherwise)."""
        if native.DTYPES.get(dtype) not in (native.FLOAT_DT_FP16, native.FLOAT_DT_FP32):
            raise ValueError("the decoder runs fp16 operands (fp32 = verification mode) only (got %r): bf16 left the 512-px "
                             "frames at the 40 dB limit" % (dtype,))
        self.size, self.style_dim = size, style_dim
        self.device = torch.device(device)
        self.dtype = dtype = native.canon_dtype(dtype)
        L = native.lib()
        pref = "motion_autoencoder.dec."
        sd = {(k[len(pref):] if k.startswith(pref) else k): v for k, v in state_dict.items()}
        self.blur_kernel = blur_taps(blur_kernel)
        if self.blur_kernel is not None:
            sd["blur_kernel"] = self.blur_kernel
        arr, keep = native.tensor_table(sd)
        cfg = native.DecCfg(size, style_dim, native.DTYPES[dtype], max_frames)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            native.check(L.float_dec_create(C.byref(cfg), arr, len(sd), C.byref(h)))
        self._h = h
        self._feats = None
        del keep

    def close(self):
        if getattr(self, "_h", None) and native is not None:
            native.lib().float_dec_destroy(self._h)
            self._h = None

    __del__ = close

    def feat_shapes(self):
        """[(C, R)] of the skip features the operator reads (float_dec_feat_shape), reference order 8..size."""
        out, i = [], 0
        c, r = C.c_int32(0), C.c_int32(0)
        while native.lib().float_dec_feat_shape(self._h, i, C.byref(c), C.byref(r)) == 0:
            out.append((c.value, r.value))
            i += 1
        return out

    def set_feats(self, feats):
        """feats: the encoder's skip list (encoder.py:220-231), 7 tensors (1,C,R,R), R = 8..size.  Every tensor is checked
        against the decoder's level table first: the repack kernel reads C*R*R floats from each pointer, so a map from a
        differently sized or channelled encoder (the VA nodes let users wire any loader to any loader) must not reach it."""
        want = self.feat_shapes()
        if len(feats) != len(want):
            raise ValueError("expected %d feature maps (8..%d), got %d" % (len(want), self.size, len(feats)))
        for i, (f, (c, r)) in enumerate(zip(feats, want)):
            if f.dim() not in (3, 4) or tuple(f.shape[-3:]) != (c, r, r) or (f.dim() == 4 and f.shape[0] != 1):
                raise ValueError("feats[%d] must be (1,%d,%d,%d) for a %d-px decoder, got %s"
                                 % (i, c, r, r, self.size, tuple(f.shape)))
        fs = [f.to(self.device, torch.float32).reshape(f.shape[-3], f.shape[-2], f.shape[-1]).contiguous() for f in feats]
        ptrs = (C.c_void_p * len(fs))(*[f.data_ptr() for f in fs])
        with torch.cuda.device(self.device):
            native.check(native.lib().float_dec_set_feats(self._h, ptrs, len(fs), native.stream_ptr(self.device)))
        self._feats = fs  # keep alive until the async repack has run

    def set_feats16(self, bufs, dtype=None):
        """Skip features as flat NHWC device buffers of the decoder's element type (EncoderHIP.export_feats16)."""
        ptrs = (C.c_void_p * len(bufs))(*[b.data_ptr() for b in bufs])
        with torch.cuda.device(self.device):
            native.check(native.lib().float_dec_set_feats16(self._h, ptrs, len(bufs), native.DTYPES[dtype or self.dtype],
                                                            native.stream_ptr(self.device)))

    def _run(self, fn, s_r, r_d, shape):
        s_r = s_r.to(self.device, torch.float32).reshape(-1).contiguous()
        r_d = r_d.to(self.device, torch.float32).reshape(-1, self.style_dim).contiguous()
        if s_r.numel() != self.style_dim:
            raise ValueError("s_r must have %d elements (decoder batch is 1, FLOAT.py:140)" % self.style_dim)
        T = r_d.shape[0]
        out = torch.empty((T,) + shape, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            native.check(fn(self._h, native.dev_ptr(s_r), native.dev_ptr(r_d), T, native.dev_ptr(out),
                            native.stream_ptr(self.device)))
        return out

    @torch.no_grad()
    def decode_latent_into_processed_images(self, s_r, r_d, s_r_feats=None):
        """FLOAT.py:113-169: (T, H, W, 3) fp32 in [0,1]; stays on the GPU (the caller copies out)."""
        if s_r_feats is not None:
            self.set_feats(s_r_feats)
        return self._run(native.lib().float_dec_frames, s_r, r_d, (self.size, self.size, 3))

    def saturation(self, reset=False, per_site=False):
        """(thread, tile) groups of 16-bit activation stores that held an inf / NaN since create / the last reset
        (float_dec_saturation; the decoder stores an out-of-range value as inf - loud - instead of clamping it):
        0 unless the checkpoint leaves fp16's range - then the frames are not the reference's and dtype="fp32" is the
        way to run it.  per_site: also the 40 per-layer counters.  Synchronises the current stream."""
        tot = C.c_uint64(0)
        sites = (C.c_uint64 * native.DEC_SAT_SITES)()
        with torch.cuda.device(self.device):
            native.check(native.lib().float_dec_saturation(self._h, C.byref(tot), sites, 1 if reset else 0,
                                                           native.stream_ptr(self.device)))
        return (tot.value, list(sites)) if per_site else tot.value

    @torch.no_grad()
    def decode_into_host(self, s_r, r_d, host, staging=None, copy_stream=None):
        """decode_latent_into_processed_images with the reference's destination (a pre-allocated CPU tensor, FLOAT.py:139):
        `host` (T, size, size, 3) fp32.  Pinned (`pin_memory()`): the frames of batch i are stored into it by copy
        workgroups inside the launches of batch i+1 (or, with `copy_stream`, copied on that stream while the next batch
        renders - see include/float_hip.h for why that form does not pay on MI355X).  Pageable: accepted, one staged
        hipMemcpyAsync behind each batch (the operator asks the runtime what `host` is, nothing is assumed).  Returns the
        device staging tensor; the frames are in `host` once the current stream has been synchronised."""
        s_r = s_r.to(self.device, torch.float32).reshape(-1).contiguous()
        r_d = r_d.to(self.device, torch.float32).reshape(-1, self.style_dim).contiguous()
        T = r_d.shape[0]
        shape = (T, self.size, self.size, 3)
        if tuple(host.shape) != shape or host.dtype != torch.float32 or host.is_cuda or not host.is_contiguous():
            raise ValueError("host must be a contiguous CPU float32 tensor of shape %s" % (shape,))
        if staging is None or tuple(staging.shape) != shape:
            staging = torch.empty(shape, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            native.check(native.lib().float_dec_frames_host(
                self._h, native.dev_ptr(s_r), native.dev_ptr(r_d), T, native.dev_ptr(staging), C.c_void_p(host.data_ptr()),
                native.stream_ptr(self.device), C.c_void_p(copy_stream.cuda_stream) if copy_stream is not None else None))
        return staging

    @torch.no_grad()
    def direction(self, lam):
        """Direction.forward (styledecoder.py:428-444): (B, motion_dim) -> (B, style_dim) = lam @ Q^T."""
        lam = lam.to(self.device, torch.float32).contiguous()
        if lam.dim() != 2:
            raise ValueError("direction input must be (B, motion_dim), got %s" % (tuple(lam.shape),))
        out = torch.empty(lam.shape[0], self.style_dim, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            for b in range(lam.shape[0]):
                native.check(native.lib().float_dec_direction(self._h, native.dev_ptr(lam[b]), native.dev_ptr(out[b]),
                                                              native.stream_ptr(self.device)))
        return out

    @torch.no_grad()
    def synthesis_raw(self, s_r, r_d):
        """Un-clamped Synthesis.forward output (T, 3, H, W) (styledecoder.py:532-534)."""
        return self._run(native.lib().float_dec_frames_raw, s_r, r_d, (3, self.size, self.size))


def _unit(dtype, cin, cout, res, upsample, n_frames, style_dim, flags=0):
    if native.DTYPES.get(dtype) not in (native.FLOAT_DT_FP16, native.FLOAT_DT_FP32):
        raise ValueError("unit ops run fp16 or fp32 operands (got %r)" % (dtype,))
    return native.DecUnit(native.DTYPES[dtype], cin, cout, res, 1 if upsample else 0, n_frames, style_dim, flags)


@torch.no_grad()
def debug_styled_conv(state, x, style, upsample=False, dtype="fp16", device="cuda:0", style_norm=True, blur_kernel=None):
    """Test hook (float_dec_debug_styled_conv): StyledConv.forward(x, style) of the reference (styledecoder.py:302-325, noise
    weight 0) through the production kernels.  state: `conv.weight` (1,cout,cin,3,3), `conv.modulation.weight|bias`,
    `activate.bias`; x (F,cin,R,R), style (F,style_dim).  Returns (out (F,cout,R',R') on the GPU, values clamped at fp16's range)."""
    dev = torch.device(device)
    F, cin, R, _ = x.shape
    cout = state["conv.weight"].shape[1]
    sd = {"sc." + k: v for k, v in state.items()}
    if blur_taps(blur_kernel) is not None:
        sd["blur_kernel"] = blur_taps(blur_kernel)
    arr, keep = native.tensor_table(sd)
    u = _unit(dtype, cin, cout, R, upsample, F, style.shape[-1], 0 if style_norm else 1)
    xd, sdv = x.to(dev, torch.float32).contiguous(), style.to(dev, torch.float32).contiguous()
    Ro = 2 * R if upsample else R
    out = torch.empty(F, cout, Ro, Ro, device=dev, dtype=torch.float32)
    sat = C.c_uint64(0)
    with torch.cuda.device(dev):
        native.check(native.lib().float_dec_debug_styled_conv(C.byref(u), arr, len(sd), native.dev_ptr(xd), native.dev_ptr(sdv),
                                                              native.dev_ptr(out), C.byref(sat), native.stream_ptr(dev)))
    del keep
    return out, sat.value


@torch.no_grad()
def debug_flow_level(state, x, feat, style, prev_flow=None, prev_rgb=None, dtype="fp16", device="cuda:0"):
    """Test hook (float_dec_debug_flow_level): ToFlow (styledecoder.py:399-425) + ToRGB (:368-386) of one level through
    dec_flow_kernel.  state: `to_flow.*`, `to_rgb.*` with the reference modules' key names.  Returns (flow `out` (F,3,R,R),
    blend (F,C,R,R), rgb (F,3,R,R)) on the GPU."""
    dev = torch.device(device)
    F, Cc, R, _ = x.shape
    arr, keep = native.tensor_table(state)
    u = _unit(dtype, Cc, 0, R, False, F, style.shape[-1])
    f32 = lambda t: None if t is None else t.to(dev, torch.float32).contiguous()  # noqa: E731
    xd, fd, sdv, pf, pr = f32(x), f32(feat.reshape(Cc, R, R)), f32(style), f32(prev_flow), f32(prev_rgb)
    of = torch.empty(F, 3, R, R, device=dev)
    ob = torch.empty(F, Cc, R, R, device=dev)
    org = torch.empty(F, 3, R, R, device=dev)
    with torch.cuda.device(dev):
        native.check(native.lib().float_dec_debug_flow_level(
            C.byref(u), arr, len(state), native.dev_ptr(xd), native.dev_ptr(fd), native.dev_ptr(sdv), native.dev_ptr(pf),
            native.dev_ptr(pr), native.dev_ptr(of), native.dev_ptr(ob), native.dev_ptr(org), native.stream_ptr(dev)))
    del keep
    return of, ob, org
