"""Does a second stream slow the FMT chain that follows because its hardware queue stays mapped?  The audio encoder on a side
stream beside the appearance encoder: none / a persistent side stream / a stream created and destroyed per step."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.util import load_pkg
pkg = load_pkg()
N = pkg.native
cfg = pkg.config.FmtConfig()
dev = torch.device("cuda:0")
T, size = 250, 512
fmt_sd = pkg.weights.synth_fmt_state(cfg, seed=1)
dec_sd = pkg.weights.synth_decoder_state(size, seed=1)
hp = pkg.pipeline.FloatHotPath(fmt_sd, dec_sd, cfg, dev, size, "fp16", "fp16", 32)
enc = pkg.encoder.EncoderHIP(pkg.weights.synth_encoder_state(size, seed=1), size, cfg.dim_w, 20, dev, "fp16", direction_weight=dec_sd["direction.weight"])
acfg = pkg.config.AudioConfig()
aud = pkg.audio.AudioEncoderHIP(pkg.weights.synth_audio_state(acfg, seed=1), acfg, dev, "fp16")
img = (torch.from_numpy(np.random.RandomState(0).rand(1, 3, size, size).astype("float32")) * 2 - 1).to(dev)
wav = pkg.weights.synth_waveform(10.0, seed=1).to(dev)
cond = pkg.pipeline.synth_conditions(cfg, T, seed=0, device=dev)
noise = pkg.fmt.draw_noise(5, 1, cfg, seed=15).to(dev)
host = torch.empty(T, size, size, 3, dtype=torch.float32, pin_memory=True)
staging = torch.empty(T, size, size, 3, dtype=torch.float32, device=dev)
n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
persistent = torch.cuda.Stream(dev)
cpu_ms = []


def step(mode):
    cur = torch.cuda.current_stream(dev)
    side, raw = None, None
    if mode == "persistent":
        side = persistent
    elif mode == "fresh":
        side = N.cu_range_stream(0, n_cu, dev)
    if side is not None:
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            wa = aud.inference(wav, seq_len=T)
    s_r, _, _, r_s = enc.encode_image_into_latent(img, want_feats=False)
    enc.hand_feats_to(hp.dec)
    if side is not None:
        cur.wait_stream(side)
        wa.record_stream(cur)
    else:
        wa = aud.inference(wav, seq_len=T)
    if mode == "fresh":
        side.synchronize()
        N.check(N.lib().float_stream_destroy(side.cuda_stream))
    c0 = time.perf_counter()
    r_d = hp.sample(r_s, wa, cond["we"], 51, 2.0, 1.0, 1.0, noise=noise)
    cpu_ms.append((time.perf_counter() - c0) * 1e3)
    hp.dec.decode_into_host(s_r, r_d[0], host, staging)


for mode in os.environ.get("MODES", "none,persistent,fresh,none").split(","):
    for _ in range(2):
        step(mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 6
    del cpu_ms[:]
    for _ in range(n):
        step(mode)
    torch.cuda.synchronize()
    print("%-10s %.2f ms per clip; host time inside hp.sample %.2f ms" % (mode, (time.perf_counter() - t0) * 1e3 / n, sum(cpu_ms) / n), flush=True)
