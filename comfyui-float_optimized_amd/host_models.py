"""Host-side (PyTorch on ROCm or CPU) producers of the hot path's inputs.  These run ONCE per
clip and are plumbing, not part of the HIP hot path ("OUT OF SCOPE for HIP").
The appearance encoder and Direction are NOT here any more: they are the HIP operator `float_enc_*`
(encoder.py in this package).

  * audio encoder       wav2vec2-base hidden states, interpolated to T frames *before* the
                        transformer, 12x768 -> 512 projection + LayerNorm + SiLU (FLOAT.py:304-375,
                        wav2vec2.py:33-98,184-197)
  * image / audio pre-processing of the simple node (generate.py:29-39, 69-73)

Sub-module names are the reference's checkpoint keys so the split checkpoint layout
(audio_projections/projection.safetensors, audio/wav2vec2-base-960h) loads unchanged.
"""
import math

import torch
import torch.nn.functional as F


def preprocess_image(image_hwc, size=512):
    """(H,W,3) float in [0,1] (ComfyUI IMAGE item) -> (1,3,size,size) in [-1,1].  The reference resizes
    with cv2.INTER_AREA on uint8 (generate.py:34-39); cv2 is not a dependency here, so area-averaging
    is done with adaptive_avg_pool2d (identical for integer shrink factors) / bilinear when enlarging."""
    x = image_hwc[..., :3].permute(2, 0, 1)[None].float()
    x = (x * 255.0).round().clamp(0, 255)
    if x.shape[-1] != size or x.shape[-2] != size:
        if x.shape[-1] >= size and x.shape[-2] >= size:
            x = F.adaptive_avg_pool2d(x, (size, size))
        else:
            x = F.interpolate(x, size=(size, size), mode="bilinear", align_corners=False)
    return x / 127.5 - 1.0


def preprocess_audio(waveform, sample_rate, target_rate=16000):
    """ComfyUI AUDIO item (C,N) -> mono 16 kHz, zero-mean / unit-variance like
    Wav2Vec2FeatureExtractor(do_normalize=True) (generate.py:69-73).  Resampling uses linear
    interpolation when the rate differs (the reference uses librosa soxr_hq; benchmarks feed 16 kHz)."""
    w = waveform.float()
    if w.dim() == 2:
        w = w.mean(dim=0)
    if sample_rate != target_rate:
        n = int(round(w.shape[-1] * target_rate / sample_rate))
        w = F.interpolate(w[None, None], size=n, mode="linear", align_corners=False)[0, 0]
    return ((w - w.mean()) / torch.sqrt(w.var(unbiased=False) + 1e-7))[None]


class AudioEncoderHost(torch.nn.Module):
    """wav2vec2 with the feature sequence interpolated to T frames before the transformer
    (wav2vec2.py:66-68) + the 12x768 -> 512 projection head (FLOAT.py:338-342, 345-352)."""

    def __init__(self, config=None, dim_w=512, only_last_features=False):
        super().__init__()
        from transformers import Wav2Vec2Config, Wav2Vec2Model
        self.config = config or Wav2Vec2Config()
        self.wav2vec2 = Wav2Vec2Model(self.config)
        self.only_last = only_last_features
        d_in = self.config.hidden_size * (1 if only_last_features else self.config.num_hidden_layers)
        self.audio_projection = torch.nn.Sequential(torch.nn.Linear(d_in, dim_w), torch.nn.LayerNorm(dim_w), torch.nn.SiLU())
        self.eval()

    @torch.no_grad()
    def inference(self, a, seq_len, sampling_rate=16000, fps=25.0):
        """a (B,N) normalised waveform -> wa (B,seq_len,512) (FLOAT.py:370-375)."""
        need = int(seq_len * sampling_rate / fps)
        if a.shape[1] % need != 0:
            a = F.pad(a[:, None], (0, need - a.shape[1]), mode="replicate")[:, 0]
        m = self.wav2vec2
        f = m.feature_extractor(a).transpose(1, 2)
        f = F.interpolate(f.transpose(1, 2), size=seq_len, align_corners=True, mode="linear").transpose(1, 2)
        h, _ = m.feature_projection(f)
        out = m.encoder(h, output_hidden_states=not self.only_last, return_dict=True)
        if self.only_last:
            x = out.last_hidden_state
        else:
            x = torch.stack(out.hidden_states[1:], dim=1).permute(0, 2, 1, 3)
            x = x.reshape(x.shape[0], x.shape[1], -1)
        return self.audio_projection(x)


class EmotionHost(torch.nn.Module):
    """Speech-emotion recogniser (Audio2Emotion, FLOAT.py:378-401; wav2vec2_ser.py:52-118): wav2vec2 encoder,
    mean-pool over time, Linear -> tanh -> Linear head, softmax over the 7 labels.  Sub-module names
    (`wav2vec2`, `classifier.dense`, `classifier.out_proj`) are the checkpoint keys under
    `emotion_encoder.wav2vec2_for_emotion.`."""

    def __init__(self, config=None, num_labels=7):
        super().__init__()
        from transformers import Wav2Vec2Config, Wav2Vec2Model
        # wav2vec2-large-xlsr shape of the bundled emotion_ser config when none is given
        self.config = config or Wav2Vec2Config(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                                               intermediate_size=4096, feat_extract_norm="layer",
                                               do_stable_layer_norm=True)
        self.wav2vec2 = Wav2Vec2Model(self.config)
        self.classifier = torch.nn.Module()
        self.classifier.dense = torch.nn.Linear(self.config.hidden_size, self.config.hidden_size)
        self.classifier.out_proj = torch.nn.Linear(self.config.hidden_size, num_labels)
        self.eval()

    @torch.no_grad()
    def predict_emotion(self, a):
        """a (B,N) normalised waveform -> softmax scores (B,7) (FLOAT.py:396-401)."""
        h = self.wav2vec2(a, return_dict=True).last_hidden_state.mean(dim=1)
        logits = self.classifier.out_proj(torch.tanh(self.classifier.dense(h)))
        return torch.softmax(logits, dim=1)


EMOTION_LABELS = ["angry", "disgust", "fear", "happy", "neutral", "sad", "surprise"]  # FLOAT.py:390


def emotion_one_hot(name, device="cpu"):
    """One-hot `we` (1,1,7) for a named emotion (FLOAT.py:196-200; float like nodes_adv.py:533-536)."""
    idx = EMOTION_LABELS.index(str(name).lower())
    return F.one_hot(torch.tensor(idx, device=device), num_classes=len(EMOTION_LABELS)).float()[None, None]
