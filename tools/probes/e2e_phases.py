"""Where an end-to-end step (bench.py) spends its wall-clock: events between the phases of consecutive steps."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.util import load_pkg
pkg = load_pkg()
cfg = pkg.config.FmtConfig()
dev = torch.device("cuda:0")
T, size = 250, 512
fmt_sd = pkg.weights.synth_fmt_state(cfg, seed=1)
dec_sd = pkg.weights.synth_decoder_state(size, seed=1)
hp = pkg.pipeline.FloatHotPath(fmt_sd, dec_sd, cfg, dev, size, "fp16", "fp16", 32, use_graph=int(os.environ.get("GRAPH", "2")))
enc = pkg.encoder.EncoderHIP(pkg.weights.synth_encoder_state(size, seed=1), size, cfg.dim_w, 20, dev, "fp16", direction_weight=dec_sd["direction.weight"])
acfg = pkg.config.AudioConfig()
aud = pkg.audio.AudioEncoderHIP(pkg.weights.synth_audio_state(acfg, seed=1), acfg, dev, "fp16")
img = (torch.from_numpy(np.random.RandomState(0).rand(1, 3, size, size).astype("float32")) * 2 - 1).to(dev)
wav = pkg.weights.synth_waveform(10.0, seed=1).to(dev)
cond = pkg.pipeline.synth_conditions(cfg, T, seed=0, device=dev)
noise = pkg.fmt.draw_noise(5, 1, cfg, seed=15).to(dev)
host = torch.empty(T, size, size, 3, dtype=torch.float32, pin_memory=True)
staging = torch.empty(T, size, size, 3, dtype=torch.float32, device=dev)
mode = os.environ.get("MODE", "pipe")
side = torch.cuda.Stream(dev)
def step(ev=None):
    def mark(i):
        if ev is not None:
            ev[i].record()
    mark(0)
    s_r, _, _, r_s = enc.encode_image_into_latent(img, want_feats=False)
    enc.hand_feats_to(hp.dec)
    mark(1)
    wa = aud.inference(wav, seq_len=T)
    mark(2)
    r_d = hp.sample(r_s, wa, cond["we"], 51, 2.0, 1.0, 1.0, noise=noise)
    mark(3)
    if mode == "pipe":
        hp.dec.decode_into_host(s_r, r_d[0], host, staging, copy_stream=side)
    elif mode == "inorder":
        hp.dec.decode_into_host(s_r, r_d[0], host, staging)
    elif mode == "serial":
        host.copy_(hp.decode(s_r, None, r_d), non_blocking=True)
    else:
        hp.decode(s_r, None, r_d)
    mark(4)
dummy = torch.zeros(1024, device=dev)
_step = step
def step(ev=None):
    if os.environ.get("DUMMY_SIDE"):
        with torch.cuda.stream(side):
            dummy.add_(1.0)
        if os.environ.get("DUMMY_SIDE") == "wait":
            torch.cuda.current_stream().wait_stream(side)
    _step(ev)
    if os.environ.get("SIDE_COPY"):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            host.copy_(staging, non_blocking=True)
        if os.environ.get("SIDE_COPY") == "wait":
            torch.cuda.current_stream().wait_stream(side)
        else:
            side.synchronize()
    if os.environ.get("FULL_SYNC"):
        torch.cuda.synchronize()
if os.environ.get("MAIN_STREAM"):
    main = torch.cuda.Stream(dev)
    torch.cuda.set_stream(main)
for _ in range(2):
    step()
torch.cuda.synchronize()
n = 5
evs = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(n)]
t0 = time.perf_counter()
cpu = []
for i in range(n):
    c0 = time.perf_counter()
    step(evs[i])
    cpu.append((time.perf_counter() - c0) * 1e3)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3 / n
print("mode %s: wall %.2f ms/step; cpu enqueue per step %s" % (mode, wall, " ".join("%.1f" % c for c in cpu)))
for i in range(n):
    print("  step %d: enc %.2f aud %.2f fmt %.2f dec %.2f | gap to next step start %.2f" % (
        i, evs[i][0].elapsed_time(evs[i][1]), evs[i][1].elapsed_time(evs[i][2]), evs[i][2].elapsed_time(evs[i][3]),
        evs[i][3].elapsed_time(evs[i][4]), evs[i][4].elapsed_time(evs[i + 1][0]) if i + 1 < n else 0.0))
