"""Host mirror of the motion-autoencoder encoder (reference encoder.py Encoder / EncoderApp and
FLOAT.encode_image_into_latent, FLOAT.py:283-291), running on the HIP operator (`float_enc_*`,
include/float_hip.h).  Once per clip: image -> s_r, skip features, motion coefficients, r_s."""
import ctypes as C
import math

import torch

from . import native


@native.rebuildable
class EncoderHIP:
    """Encoder(size, dim, dim_motion) (encoder.py:234-281).  `state_dict` uses the reference keys
    (`net_app.convs.*`, `fc.*`; a `motion_autoencoder.enc.` prefix is stripped); pass the decoder's
    `direction.weight` (styledecoder.py:431) as `direction_weight` to get r_s from the same launch chain."""

    def __init__(self, state_dict, size=512, dim=512, dim_motion=20, device="cuda:0", dtype="fp16", direction_weight=None):
        self.size, self.dim, self.dim_motion = size, dim, dim_motion
        self.device = torch.device(device)
        self.dtype = dtype = native.canon_dtype(dtype)
        self.n_feats = int(math.log2(size)) - 2
        pref = "motion_autoencoder.enc."
        # the Blur buffers (`*.kernel`, encoder.py:59-71) travel too: the operator has [1,3,3,1] in its code and refuses others
        sd = {(k[len(pref):] if k.startswith(pref) else k): v for k, v in state_dict.items()}
        if direction_weight is not None:
            sd["direction.weight"] = direction_weight
        self.has_direction = "direction.weight" in sd
        arr, keep = native.tensor_table(sd)
        cfg = native.EncCfg(size, dim, dim_motion, native.DTYPES[dtype])
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            native.check(native.lib().float_enc_create(C.byref(cfg), arr, len(sd), C.byref(h)))
        self._h = h
        del keep

    def saturation(self, reset=False):
        """Threads with a clamped (+-65504) or non-finite 16-bit activation store since create / the last reset
        (float_enc_saturation): 0 unless the checkpoint leaves fp16's range - then the result is not the reference's within the
        stated tolerance; run it with dtype="fp32".  Always 0 for bf16 / fp32 handles.  Synchronises the current stream."""
        with torch.cuda.device(self.device):
            return native.saturation("float_enc_saturation", self._h, self.device, reset)

    def close(self):
        if getattr(self, "_h", None) and native is not None:
            native.lib().float_enc_destroy(self._h)
            self._h = None

    __del__ = close

    def feat_shapes(self):
        from .weights import ENC_CHANNELS
        return [(ENC_CHANNELS[8 << i], 8 << i, 8 << i) for i in range(self.n_feats)]

    @torch.no_grad()
    def encode_image_into_latent(self, x, want_feats=True):
        """FLOAT.py:283-291 for one image: x (1,3,S,S) or (3,S,S) in [-1,1].
        Returns s_r (1,dim), r_s_lambda (1,dim_motion), feats [(1,C,R,R)] (reference order), r_s (1,dim)|None."""
        if x.numel() != 3 * self.size * self.size:
            raise ValueError("encoder input must be one (3,%d,%d) image, got %s" % (self.size, self.size, tuple(x.shape)))
        x = x.to(self.device, torch.float32).reshape(3, self.size, self.size).contiguous()
        dev = self.device
        s_r = torch.empty(1, self.dim, device=dev)
        lam = torch.empty(1, self.dim_motion, device=dev)
        r_s = torch.empty(1, self.dim, device=dev) if self.has_direction else None
        feats = [torch.empty((1,) + shp, device=dev) for shp in self.feat_shapes()] if want_feats else []
        ptrs = (C.c_void_p * max(len(feats), 1))(*[f.data_ptr() for f in feats])
        with torch.cuda.device(dev):
            native.check(native.lib().float_enc_forward(
                self._h, native.dev_ptr(x), native.dev_ptr(s_r), native.dev_ptr(lam), native.dev_ptr(r_s), ptrs, len(feats),
                native.stream_ptr(dev)))
        return s_r, lam, feats, r_s

    def forward(self, input_source, input_target=None, h_start=None):
        """Encoder.forward with input_target=None (encoder.py:277-281): (h_source, None, feats)."""
        if input_target is not None:
            raise NotImplementedError("the inference path only encodes the source image (FLOAT.py:283-286)")
        s_r, _, feats, _ = self.encode_image_into_latent(input_source)
        return s_r, None, feats

    __call__ = forward

    def export_feats16(self, bufs=None):
        """Copies of the NHWC skip features of the last forward in the operator's element type (float_enc_export_feats16), one
        flat device tensor per map - what `SynthesisHIP.set_feats16` takes back.  A batch of portraits keeps one set per item,
        so that item i can be decoded after item j was encoded without a second encoder pass."""
        eb = 4 if self.dtype == "fp32" else 2
        if bufs is None:
            bufs = [torch.empty(c * r * r * eb, dtype=torch.uint8, device=self.device) for c, r, _ in self.feat_shapes()]
        ptrs = (C.c_void_p * len(bufs))(*[b.data_ptr() for b in bufs])
        with torch.cuda.device(self.device):
            native.check(native.lib().float_enc_export_feats16(self._h, ptrs, len(bufs), native.stream_ptr(self.device)))
        return bufs

    def hand_feats_to(self, dec):
        """Give the NHWC 16-bit skip features of the last forward to a SynthesisHIP of the same dtype
        without the fp32 NCHW round trip (float_enc_feats16 -> float_dec_set_feats16)."""
        n = self.n_feats
        ptrs = (C.c_void_p * n)()
        ch = (C.c_int32 * n)()
        got = C.c_int32(0)
        L = native.lib()
        native.check(L.float_enc_feats16(self._h, ptrs, ch, n, C.byref(got)))
        with torch.cuda.device(self.device):
            native.check(L.float_dec_set_feats16(dec._h, ptrs, got.value, native.DTYPES[self.dtype], native.stream_ptr(self.device)))
