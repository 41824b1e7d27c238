"""Host mirror of the audio conditioning encoder (reference FLOAT.AudioEncoder, FLOAT.py:304-375; its wav2vec2
backbone src/nodes/models/wav2vec2.py), running on the HIP operator (`float_aud_*`, include/float_hip.h)."""
import ctypes as C

import torch
import torch.nn.functional as F

from . import native
from .config import AudioConfig


class AudioEncoderHIP:
    """state_dict keys: `wav2vec2.*`, `audio_projection.{0,1}.*` (an `audio_encoder.` prefix is stripped) - the
    reference's AudioEncoder.state_dict()."""

    def __init__(self, state_dict, cfg: AudioConfig = None, device="cuda:0", dtype="fp16", sampling_rate=16000, fps=25.0):
        self.cfg = cfg or AudioConfig()
        self.device = torch.device(device)
        self.dtype = dtype
        self.sampling_rate, self.fps = sampling_rate, fps
        pref = "audio_encoder."
        sd = {(k[len(pref):] if k.startswith(pref) else k): v for k, v in state_dict.items()
              if not k.endswith("masked_spec_embed")}  # only used when mask_time_indices is given (never at inference)
        c = self.cfg
        n = len(c.conv_dim)
        if n > 8 or len(c.conv_kernel) != n or len(c.conv_stride) != n:
            raise ValueError("feature extractor must have <= 8 layers with matching kernel/stride lists")
        ncfg = native.AudCfg()
        ncfg.n_conv = n
        for i in range(n):
            ncfg.conv_dim[i], ncfg.conv_kernel[i], ncfg.conv_stride[i] = c.conv_dim[i], c.conv_kernel[i], c.conv_stride[i]
        ncfg.hidden, ncfg.layers, ncfg.heads, ncfg.intermediate = c.hidden_size, c.num_hidden_layers, c.num_attention_heads, c.intermediate_size
        ncfg.pos_k, ncfg.pos_groups = c.num_conv_pos_embeddings, c.num_conv_pos_embedding_groups
        ncfg.dim_w, ncfg.only_last, ncfg.dtype, ncfg.ln_eps = c.dim_w, int(c.only_last_features), native.DTYPES[dtype], c.layer_norm_eps
        arr, keep = native.tensor_table(sd)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            native.check(native.lib().float_aud_create(C.byref(ncfg), arr, len(sd), C.byref(h)))
        self._h = h
        del keep

    def close(self):
        if getattr(self, "_h", None) and native is not None:
            native.lib().float_aud_destroy(self._h)
            self._h = None

    __del__ = close

    @torch.no_grad()
    def inference(self, a, seq_len):
        """AudioEncoder.inference (FLOAT.py:370-375): a (B,N) normalised 16 kHz waveform -> wa (B,seq_len,dim_w)."""
        a = a.to(self.device, torch.float32)
        if a.dim() == 1:
            a = a[None]
        need = int(seq_len * self.sampling_rate / self.fps)
        if a.shape[1] % need != 0:  # FLOAT.py:371-373
            a = F.pad(a[:, None], (0, need - a.shape[1]), mode="replicate")[:, 0]
        a = a.contiguous()
        out = torch.empty(a.shape[0], seq_len, self.cfg.dim_w, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            for b in range(a.shape[0]):
                native.check(native.lib().float_aud_inference(self._h, native.dev_ptr(a[b]), a.shape[1], int(seq_len),
                                                              native.dev_ptr(out[b]), native.stream_ptr(self.device)))
        return out
