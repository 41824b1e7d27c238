"""Audio-operator timing probe: wav2vec2-base + projection on a 10-s clip (float_aud_inference) and the wav2vec2-large
speech-emotion classifier (float_aud_classify), ms per call.  FLOAT_AUD_ATTN_MFMA=0 selects the one-wave-per-query attention."""
import os
import sys

import torch

sys.path.insert(0, ".")
from tests.util import load_pkg  # noqa: E402

pkg = load_pkg()
acfg = pkg.config.AudioConfig()
aud = pkg.audio.AudioEncoderHIP(pkg.weights.synth_audio_state(acfg, seed=1), acfg, "cuda:0", "fp16")
wav = pkg.weights.synth_waveform(10.0, seed=1).cuda()


def ev_ms(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


wa = aud.inference(wav, seq_len=250)
t_a = ev_ms(lambda: aud.inference(wav, seq_len=250))
msg = "audio encoder %.3f ms (|wa| %.6f)" % (t_a, float(wa.abs().mean()))
if os.environ.get("AUD_SER", "1") != "0":
    ecfg = pkg.config.emotion_audio_config()
    ser = pkg.audio.Audio2EmotionHIP(pkg.weights.synth_audio_state(ecfg, seed=2), ecfg, "cuda:0", "fp16")
    sc = ser.predict_emotion(wav)
    msg += "; speech emotion %.3f ms (scores %s)" % (ev_ms(lambda: ser.predict_emotion(wav)), [round(float(v), 5) for v in sc.flatten()[:3]])
print(msg, "[FLOAT_AUD_ATTN_MFMA=%s]" % os.environ.get("FLOAT_AUD_ATTN_MFMA", "1"))
