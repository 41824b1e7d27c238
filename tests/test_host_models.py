"""Host-side plumbing in front of the HIP operators (host_models.py): the band-limited resampler that replaces the
reference's librosa soxr_hq path, the face-aligned crop with its no-detector / no-face fallback (utils/image.py:135-180),
and the emotion-label rule of FLOAT.py:196 (anything that is not one of the seven labels = predict from the audio)."""
import numpy as np
import pytest
import torch

from tests.util import load_pkg

pkg = load_pkg()
hm = pkg.host_models


@pytest.mark.parametrize("src", [48000, 44100, 22050, 8000])
def test_resampler_matches_polyphase_reference_in_band(src):
    from math import gcd
    from scipy.signal import resample_poly
    t = np.arange(src) / src
    x = (0.5 * np.sin(2 * np.pi * 440 * t) + 0.3 * np.sin(2 * np.pi * 2500 * t)).astype(np.float32)
    y = hm.resample_sinc(torch.from_numpy(x), src, 16000).numpy()
    g = gcd(src, 16000)
    ref = resample_poly(x.astype(np.float64), 16000 // g, src // g)
    assert len(y) == 16000 == len(ref)
    assert np.abs(y[300:-300] - ref[300:-300]).max() < 2e-3


def test_resampler_suppresses_what_would_alias():
    """An 11 kHz tone at 48 kHz lies above the 8 kHz Nyquist of the target: linear interpolation folds it to 5 kHz at full
    amplitude (rms 0.35), the band-limited resampler removes it."""
    t = np.arange(48000) / 48000
    hi = torch.from_numpy((0.5 * np.sin(2 * np.pi * 11000 * t)).astype(np.float32))
    y = hm.resample_sinc(hi, 48000, 16000)
    assert float(y[300:-300].pow(2).mean().sqrt()) < 1e-3
    a = hm.preprocess_audio(torch.stack([hi, hi]), 48000)
    assert a.shape == (1, 16000) and abs(float(a.mean())) < 1e-4


def test_face_crop_fallback_is_the_reference_no_face_branch():
    """Without the optional detector (or without a face) the reference crops the centre square (utils/image.py:151-158)."""
    img = torch.rand(720, 800, 3)  # taller than 360 px: the reference shrinks with INTER_AREA (mult < 1)
    crop, bbox = hm.process_img(img, 360)
    assert bbox == (40, 0, 720, 720) and crop.shape == (360, 360, 3)
    want = torch.nn.functional.adaptive_avg_pool2d(img[:, 40:760].permute(2, 0, 1)[None], (360, 360))[0].permute(1, 2, 0)
    try:
        import face_alignment  # noqa: F401
    except ImportError:
        assert torch.allclose(crop, want, atol=1e-6)
    same, bbox = hm.process_img(torch.rand(64, 64, 3), 64)
    assert same.shape == (64, 64, 3) and bbox == (0, 0, 64, 64)


def test_face_crop_falls_back_when_the_detector_cannot_be_built(monkeypatch):
    """An importable but unusable `face_alignment` (a stub left in sys.modules, a broken install, unreachable model files) must
    take the reference's no-face branch with a warning (utils/image.py:151-158), not crash FLOAT Process at its default
    face_align=True."""
    import logging
    import sys
    import types
    monkeypatch.setitem(sys.modules, "face_alignment", types.ModuleType("face_alignment"))  # no FaceAlignment attribute
    monkeypatch.setattr(hm, "_FA", None)
    seen = []

    class H(logging.Handler):
        def emit(self, rec):
            seen.append(rec.getMessage())
    log = logging.getLogger("test_face_crop")
    log.addHandler(H())
    img = torch.rand(720, 800, 3)
    crop, bbox = hm.process_img(img, 360, logger=log)
    want = torch.nn.functional.adaptive_avg_pool2d(img[:, 40:760].permute(2, 0, 1)[None], (360, 360))[0].permute(1, 2, 0)
    assert bbox == (40, 0, 720, 720) and torch.allclose(crop, want, atol=1e-6)
    assert seen and "no face detector is available" in seen[0]


def test_emotion_labels_follow_label2id_get():
    assert hm.emotion_index("Happy") == 3 and hm.emotion_index("neutral") == 4
    for other in (None, "none", "S2E", "s2e", "joyful"):
        assert hm.emotion_index(other) is None  # -> speech-to-emotion prediction (FLOAT.py:196-198)
    assert hm.emotion_one_hot("sad").tolist() == [[[0, 0, 0, 0, 0, 1, 0]]]
    with pytest.raises(ValueError):
        hm.emotion_one_hot("S2E")


def test_resampler_near_coprime_rate_is_bounded():
    """a near-coprime source rate (44099 Hz: up 16000 x 44k taps = 5.6 GB of kernel) must not blow up; it snaps to
    the nearest multiple of 50 Hz first and stays a clean resampler."""
    import math
    import time
    import torch
    H = pkg.host_models
    t = torch.arange(44099) / 44099.0
    w = torch.sin(2 * math.pi * 440 * t)
    t0 = time.time()
    y = H.resample_sinc(w, 44099, 16000)
    assert time.time() - t0 < 5.0 and y.shape == (16000,)
    ref = torch.sin(2 * math.pi * 440 * torch.arange(16000) / 16000.0)
    assert float((y[200:-200] - ref[200:-200]).abs().max()) < 2e-3


def test_resampler_unfriendly_target_rate_terminates():
    """a source rate that already is a multiple of 50 Hz with a target that is not (44100 -> 16001 via
    opt.sampling_rate) used to recurse without end; now the filter runs at the nearest friendly target and a linear
    post-pass lands on the requested length."""
    t = np.arange(44100) / 44100.0
    x = torch.from_numpy(np.sin(2 * np.pi * 440 * t).astype(np.float32))
    y = hm.resample_sinc(x, 44100, 16001)
    assert y.shape[0] == 16001
    tt = np.arange(16001) / 16001.0
    assert float(np.abs(y.numpy()[200:-200] - np.sin(2 * np.pi * 440 * tt)[200:-200]).max()) < 2e-2


def test_range_report_modes(monkeypatch):
    """pipeline.report_range: what the product calls do with the fp16 range counters (float_*_saturation totals)."""
    import warnings
    P = pkg.pipeline
    assert P.report_range({"fmt": 0, "decoder": 0}, "x") == {}
    monkeypatch.setenv("FLOAT_AMD_RANGE", "warn")
    with pytest.warns(RuntimeWarning, match="fp16 range exceeded in decoder \\(3 stores\\)"):
        assert P.report_range({"fmt": 0, "decoder": 3}, "clip") == {"decoder": 3}
    monkeypatch.setenv("FLOAT_AMD_RANGE", "raise")
    with pytest.raises(P.Fp16RangeError, match="encoder"):
        P.report_range({"encoder": 1}, "clip")
    assert issubclass(P.Fp16RangeError, OverflowError)
    monkeypatch.setenv("FLOAT_AMD_RANGE", "off")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert P.report_range({"encoder": 1}, "clip") == {"encoder": 1}
