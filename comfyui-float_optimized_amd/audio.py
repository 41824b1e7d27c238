"""Host mirror of the audio conditioning encoder (reference FLOAT.AudioEncoder, FLOAT.py:304-375; its wav2vec2
backbone src/nodes/models/wav2vec2.py), running on the HIP operator (`float_aud_*`, include/float_hip.h)."""
import ctypes as C

import torch
import torch.nn.functional as F

from . import native
from .config import AudioConfig


@native.rebuildable
class AudioEncoderHIP:
    """state_dict keys: `wav2vec2.*`, `audio_projection.{0,1}.*` (an `audio_encoder.` prefix is stripped) - the
    reference's AudioEncoder.state_dict()."""

    def __init__(self, state_dict, cfg: AudioConfig = None, device="cuda:0", dtype="fp16", sampling_rate=16000, fps=25.0):
        self.cfg = cfg or AudioConfig()
        self.device = torch.device(device)
        self.dtype = dtype = native.canon_dtype(dtype)
        self.sampling_rate, self.fps = sampling_rate, fps
        sd = {}
        for k, v in state_dict.items():
            for pref in ("audio_encoder.", "emotion_encoder.wav2vec2_for_emotion."):
                if k.startswith(pref):
                    k = k[len(pref):]
            if not k.endswith("masked_spec_embed"):  # only used when mask_time_indices is given (never at inference)
                sd[k] = v
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
        if c.feat_extract_norm not in ("group", "layer"):
            raise ValueError("feat_extract_norm must be 'group' or 'layer', got %r" % (c.feat_extract_norm,))
        ncfg.feat_norm_layer, ncfg.stable_ln = int(c.feat_extract_norm == "layer"), int(c.do_stable_layer_norm)
        ncfg.conv_bias, ncfg.num_labels = int(c.conv_bias), int(c.num_labels)
        arr, keep = native.tensor_table(sd)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            native.check(native.lib().float_aud_create(C.byref(ncfg), arr, len(sd), C.byref(h)))
        self._h = h
        self._reserved = (0, 0)  # a new handle has no workspace yet (also after a rebuild: native.rebuildable)
        del keep

    def saturation(self, reset=False):
        """Threads with a clamped (+-65504) or non-finite 16-bit activation store since create / the last reset
        (float_aud_saturation): 0 unless the checkpoint leaves fp16's range - then the result is not the reference's within the
        stated tolerance; run it with dtype="fp32" (or "bf16").  Always 0 for bf16 / fp32 handles.  Synchronises the current stream."""
        with torch.cuda.device(self.device):
            return native.saturation("float_aud_saturation", self._h, self.device, reset)

    def close(self):
        if getattr(self, "_h", None) and native is not None:
            native.lib().float_aud_destroy(self._h)
            self._h = None

    __del__ = close

    def reserve(self, n_samples, seq_len=0):
        """Workspace for clips of up to n_samples samples / seq_len frames (float_aud_reserve): the operator's run-time calls
        never allocate, so this mirror grows the reservation (a stream synchronise + hipMalloc) before a longer clip."""
        cap = getattr(self, "_reserved", (0, 0))
        frames = int(seq_len) if seq_len > 0 else int(n_samples) // 320 + 1
        if n_samples <= cap[0] and frames <= cap[1]:
            return
        with torch.cuda.device(self.device):
            native.check(native.lib().float_aud_reserve(self._h, int(n_samples), int(seq_len), native.stream_ptr(self.device)))
        self._reserved = (max(cap[0], int(n_samples)), max(cap[1], frames))

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
        self.reserve(a.shape[1], seq_len)
        out = torch.empty(a.shape[0], seq_len, self.cfg.dim_w, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            for b in range(a.shape[0]):
                native.check(native.lib().float_aud_inference(self._h, native.dev_ptr(a[b]), a.shape[1], int(seq_len),
                                                              native.dev_ptr(out[b]), native.stream_ptr(self.device)))
        return out


@native.rebuildable
class Audio2EmotionHIP(AudioEncoderHIP):
    """Speech-to-emotion (reference Audio2Emotion, FLOAT.py:378-401, on Wav2Vec2ForSpeechClassification,
    wav2vec2_ser.py:41-118): the wav2vec2-large variant of the same operator with the classification head.
    state_dict keys: `wav2vec2.*`, `classifier.{dense,out_proj}.*` (prefix `emotion_encoder.wav2vec2_for_emotion.` stripped)."""

    id2label = {0: "angry", 1: "disgust", 2: "fear", 3: "happy", 4: "neutral", 5: "sad", 6: "surprise"}  # FLOAT.py:390

    def __init__(self, state_dict, cfg: AudioConfig = None, device="cuda:0", dtype="fp16"):
        from .config import emotion_audio_config
        cfg = cfg or emotion_audio_config()
        if not cfg.num_labels:
            raise ValueError("Audio2EmotionHIP needs a config with num_labels > 0")
        super().__init__(state_dict, cfg, device, dtype)

    def inference(self, a, seq_len):
        raise TypeError("the speech-emotion model has no audio projection; use predict_emotion")

    @torch.no_grad()
    def predict_emotion(self, a, prev_a=None):
        """FLOAT.py:396-401: a (B,N) normalised waveform -> softmax scores (B, num_labels)."""
        if prev_a is not None:
            a = torch.cat([prev_a, a], dim=1)
        a = a.to(self.device, torch.float32)
        if a.dim() == 1:
            a = a[None]
        a = a.contiguous()
        self.reserve(a.shape[1], 0)
        out = torch.empty(a.shape[0], self.cfg.num_labels, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            for b in range(a.shape[0]):
                native.check(native.lib().float_aud_classify(self._h, native.dev_ptr(a[b]), a.shape[1], native.dev_ptr(out[b]),
                                                             native.stream_ptr(self.device)))
        return out
