"""Host-side (PyTorch on ROCm or CPU) producers of the hot path's inputs.  These run ONCE per
clip and are plumbing, not part of the HIP hot path ("OUT OF SCOPE for HIP").
The appearance encoder + Direction and the audio encoder (wav2vec2-base + projection) are NOT here any more:
they are the HIP operators `float_enc_*` / `float_aud_*` (encoder.py / audio.py in this package, SURVEY.md
section 8f rows 1-2).  What is left:

  * image / audio pre-processing of the simple node (generate.py:29-39, 69-73)
  * the speech-emotion classifier used only by emotion="none" (wav2vec2-large-xlsr, FLOAT.py:378-401)
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
