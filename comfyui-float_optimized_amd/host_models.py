"""Host-side (PyTorch on ROCm or CPU) producers of the hot path's inputs.  These run ONCE per
clip and are plumbing, not part of the HIP hot path ("OUT OF SCOPE for HIP").
The appearance encoder + Direction, the audio encoder (wav2vec2-base + projection) and the speech-emotion classifier
(wav2vec2-large + head) are NOT here: they are the HIP operators `float_enc_*` / `float_aud_*` (encoder.py / audio.py in
this package).  What is left is tensor plumbing:

  * image / audio pre-processing of the simple node (generate.py:29-39, 69-73)
  * the one-hot emotion vector of a named emotion (FLOAT.py:196-200)
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


EMOTION_LABELS = ["angry", "disgust", "fear", "happy", "neutral", "sad", "surprise"]  # FLOAT.py:390


def emotion_one_hot(name, device="cpu"):
    """One-hot `we` (1,1,7) for a named emotion (FLOAT.py:196-200; float like nodes_adv.py:533-536)."""
    idx = EMOTION_LABELS.index(str(name).lower())
    return F.one_hot(torch.tensor(idx, device=device), num_classes=len(EMOTION_LABELS)).float()[None, None]
