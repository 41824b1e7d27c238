"""Host-side (PyTorch on ROCm or CPU) producers of the hot path's inputs.  These run ONCE per
clip and are plumbing, not part of the HIP hot path ("OUT OF SCOPE for HIP").
The appearance encoder + Direction, the audio encoder (wav2vec2-base + projection) and the speech-emotion classifier
(wav2vec2-large + head) are NOT here: they are the HIP operators `float_enc_*` / `float_aud_*` (encoder.py / audio.py in
this package).  What is left is tensor plumbing:

  * image / audio pre-processing of the simple node (generate.py:29-39, 69-73), the face-aligned crop (utils/image.py:135-180;
    detector optional) and the band-limited resampler in front of wav2vec2
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


def resample_sinc(w, orig_rate, new_rate, zeros=24, rolloff=0.945):
    """Band-limited polyphase resampling of a 1-D waveform (Hann-windowed sinc, `zeros` zero crossings per side, cut-off at
    `rolloff` x the lower Nyquist): the anti-alias low-pass the reference gets from librosa's soxr_hq
    (utils/audio.py -> librosa.resample), which plain linear interpolation lacks - without it everything between 8 kHz and
    the source Nyquist folds into the wav2vec2 band when ComfyUI hands over 44.1 / 48 kHz audio."""
    orig_rate, new_rate = int(orig_rate), int(new_rate)
    if orig_rate == new_rate:
        return w
    g = math.gcd(orig_rate, new_rate)
    up, down = new_rate // g, orig_rate // g
    if up * (2 * int(math.ceil(zeros * max(up, down) / (min(up, down) * rolloff))) + down) > 2e7:
        # near-coprime rates (44099 -> 16000: up 16000, 44k taps per phase = 5.6 GB of kernel): quantise the source rate to
        # the nearest multiple of 50 Hz by a linear-interpolation pre-pass (error band far above the wav2vec2 band's needs:
        # a ratio change of <= 0.06 %), then run the polyphase filter at the friendly ratio
        snapped = max(50, int(round(orig_rate / 50.0)) * 50)
        if snapped != orig_rate:
            n_mid = max(1, int(round(w.shape[-1] * snapped / float(orig_rate))))
            w = F.interpolate(w[None, None].double(), size=n_mid, mode="linear", align_corners=False)[0, 0].to(w.dtype)
            return resample_sinc(w, snapped, new_rate, zeros, rolloff)
        # the SOURCE rate is friendly already, the TARGET is not (44100 -> 16001): filter to the nearest multiple of 50 Hz of
        # the target, then a linear-interpolation post-pass of <= 0.16 % (the signal is band-limited below it by then)
        tgt = max(50, int(round(new_rate / 50.0)) * 50)
        if tgt == new_rate:
            raise ValueError("cannot build a polyphase resampler for %d -> %d Hz" % (orig_rate, new_rate))
        y = resample_sinc(w, orig_rate, tgt, zeros, rolloff)
        n_out = max(1, int(math.ceil(w.shape[-1] * new_rate / float(orig_rate))))
        return F.interpolate(y[None, None].double(), size=n_out, mode="linear", align_corners=False)[0, 0].to(w.dtype)
    base = min(up, down) * rolloff          # cut-off in units of the common rate's Nyquist / max(up, down)
    width = int(math.ceil(zeros * down / base))
    # kernel[phase p of `up`, tap k]: output sample n * up + p reads input samples around n * down + p * down / up
    idx = torch.arange(-width, width + down, dtype=torch.float64)[None, :] / down
    ph = -torch.arange(up, dtype=torch.float64)[:, None] / up
    t = (idx + ph) * base
    t = t.clamp(-zeros, zeros)
    win = torch.cos(t * math.pi / zeros / 2) ** 2
    tpi = t * math.pi
    ker = torch.where(tpi == 0, torch.ones_like(tpi), torch.sin(tpi) / tpi) * win * (base / down)
    ker = ker.to(w.dtype)[:, None, :]
    n = w.shape[-1]
    x = F.pad(w[None, None], (width, width + down))
    y = F.conv1d(x, ker, stride=down)       # (1, up, frames)
    y = y.transpose(1, 2).reshape(-1)
    return y[: int(math.ceil(n * up / down))]


def preprocess_audio(waveform, sample_rate, target_rate=16000, device=None):
    """ComfyUI AUDIO item (C,N) -> mono 16 kHz, zero-mean / unit-variance like
    Wav2Vec2FeatureExtractor(do_normalize=True) (generate.py:69-73).  A different source rate goes through a band-limited
    resampler (the reference: librosa soxr_hq)."""
    w = waveform
    if device is not None and sample_rate == target_rate:
        w = w.to(device, non_blocking=True)  # nothing to resample: mono mix and normalisation run where the encoder runs
    w = w.float()
    if w.dim() == 2:
        w = w.mean(dim=0)
    if sample_rate != target_rate:
        w = resample_sinc(w, sample_rate, target_rate)  # on the waveform's own device (the host for a ComfyUI AUDIO item)
    if device is not None:
        w = w.to(device, non_blocking=True)
    return ((w - w.mean()) / torch.sqrt(w.var(unbiased=False) + 1e-7))[None]


def process_img(img_hwc, input_size, margin=1.6, index=1, logger=None):
    """The reference's face-aligned crop (utils/image.py:135-180) on an (H,W,3) float image in [0,1]: detect faces on a
    360-px-high copy, take box `index`, crop a square of `margin` x the larger half side around its centre from the
    zero-bordered image, resize to input_size.  The detector (`face_alignment`, SFD) is an optional dependency: without it
    - or when it finds no face, exactly like the reference (utils/image.py:151-158) - the centre square is cropped and a
    warning is logged.  Returns (crop (S,S,3) float in [0,1], bbox (x, y, w, h))."""
    H, Wd = int(img_hwc.shape[0]), int(img_hwc.shape[1])
    mult = 360.0 / H
    bboxes = None
    fa = None
    try:  # ANY failure to obtain a detector (package absent, a stub or broken install, model files unreachable) takes the
        fa = _face_detector()  # reference's no-face branch below instead of failing the node at its default face_align=True
    except Exception as e:  # noqa: BLE001
        if logger is not None:
            logger.warning("face_align=True, but no face detector is available (%s: %s): no face detection, the centre square "
                           "of the image is used (install face_alignment for the reference's crop)" % (type(e).__name__, e))
    if fa is not None:
        import numpy as np
        small = F.interpolate(img_hwc.permute(2, 0, 1)[None], scale_factor=mult, mode="area" if mult < 1.0 else "bicubic")
        small = (small[0].permute(1, 2, 0).clamp(0, 1) * 255).round().to(torch.uint8).cpu().numpy()
        det = fa.face_detector.detect_from_image(np.ascontiguousarray(small))
        bboxes = [(int(x1 / mult), int(y1 / mult), int(x2 / mult), int(y2 / mult), sc) for (x1, y1, x2, y2, sc) in (det or []) if sc > 0.95]
    if not bboxes:
        if bboxes is not None and logger is not None:
            logger.warning("Failed to detect any face in the image, no face align performed")
        my, mx = H // 2, Wd // 2
        bs = min(mx, my)
        bbox_r = (mx - bs, my - bs, 2 * bs, 2 * bs)
        img = img_hwc
    else:
        if index > len(bboxes):
            if logger is not None:
                logger.warning("Only %d detected, using the first one" % len(bboxes))
            index = 1
        b = bboxes[index - 1]
        bsy, bsx = int((b[3] - b[1]) / 2), int((b[2] - b[0]) / 2)
        my, mx = int((b[1] + b[3]) / 2), int((b[0] + b[2]) / 2)
        bs = int(max(bsy, bsx) * margin)
        img = F.pad(img_hwc.permute(2, 0, 1), (bs, bs, bs, bs)).permute(1, 2, 0)  # cv2.copyMakeBorder(..., value=0)
        bbox_r = (mx - bs, my - bs, 2 * bs, 2 * bs)
        my, mx = my + bs, mx + bs
    crop = img[my - bs:my + bs, mx - bs:mx + bs]
    if crop.shape[0] != input_size or crop.shape[1] != input_size:
        c = crop.permute(2, 0, 1)[None].float()
        if mult < 1.0 and c.shape[-1] >= input_size:
            c = F.adaptive_avg_pool2d(c, (input_size, input_size))      # cv2.INTER_AREA
        else:
            c = F.interpolate(c, size=(input_size, input_size), mode="bicubic", align_corners=False).clamp(0, 1)  # INTER_CUBIC
        crop = c[0].permute(1, 2, 0)
    return crop, bbox_r


_FA = None


def _face_detector():
    global _FA
    if _FA is None:
        import face_alignment
        _FA = face_alignment.FaceAlignment(face_alignment.LandmarksType.TWO_D, flip_input=False)
    return _FA


EMOTION_LABELS = ["angry", "disgust", "fear", "happy", "neutral", "sad", "surprise"]  # FLOAT.py:390


def emotion_index(name):
    """label2id.get(str(emo).lower(), None) of the reference (FLOAT.py:196): None for anything that is not one of the seven
    labels - None itself, 'none', 'S2E' (run_inference's default) ... - which means "predict the scores from the audio"."""
    name = str(name).lower()
    return EMOTION_LABELS.index(name) if name in EMOTION_LABELS else None


def emotion_one_hot(name, device="cpu"):
    """One-hot `we` (1,1,7) for a named emotion (FLOAT.py:196-200; float like nodes_adv.py:533-536)."""
    idx = emotion_index(name)
    if idx is None:
        raise ValueError("%r is not one of %s" % (name, EMOTION_LABELS))
    return F.one_hot(torch.tensor(idx, device=device), num_classes=len(EMOTION_LABELS)).float()[None, None]
