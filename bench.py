#!/usr/bin/env python3
"""Benchmark of the FLOAT hot path on MI355X: FMT Euler sampling loop + Synthesis decoder.

One "step" = one clip: `--seconds` of audio at 25 fps (default 10 s -> 250 frames, BASELINE.json
configs[1]) sampled with `--nfe` grid points (default 51 = 50 Euler evaluations per 50-frame
window, 3-way CFG a=2,e=1) and decoded to 512x512 frames that stay in HBM.  Conditioning tensors
(wa, we, r_s, s_r, feats) and the noise are synthetic and resident in HBM before the timed region.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, each decoding its own clip (BASELINE.json configs[3], replicas, weak
scaling); `--mode shard` instead shards ONE N x `--seconds` clip by audio window (configs[2]).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from tests.util import load_pkg  # noqa: E402

# SURVEY.md section 8(d) algorithmic work per unit
FMT_WEIGHT_BYTES_PER_EVAL = 313.4e6   # bf16 weights streamed once per evaluation
FMT_FLOP_PER_EVAL_CFG3 = 55.87e9
DEC_FLOP_PER_FRAME = 37.79e9
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0             # dense bf16/fp16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--nfe", type=int, default=51, help="Euler grid points; evaluations = nfe-1")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--fmt-dtype", default="bf16")
    ap.add_argument("--dec-dtype", default="fp16")
    ap.add_argument("--max-frames", type=int, default=32)
    ap.add_argument("--mode", default="replicas", choices=["replicas", "shard", "window"],
                    help="N>1: replicas = one clip per GPU (exact); shard = one N x clip, latent chain replicated, frames "
                         "sharded (exact); window = one N x clip, windows sharded, RCCL all_gather of boundary latents + "
                         "re-solve (approximate unless --window-iters = N-1)")
    ap.add_argument("--window-iters", type=int, default=1)
    ap.add_argument("--dynamic-we", action="store_true", help="BASELINE configs[4]: per-window emotion")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--graph-mode", type=int, default=2, help="2: single-branch hipGraph per window, 1: adaLN GEMM on a parallel branch")
    ap.add_argument("--fmt-priority", type=int, default=-1)
    ap.add_argument("--cu-split", type=int, default=0, help="with --overlap: FMT chain on CUs [0,N), decoder on the rest")
    ap.add_argument("--overlap", action="store_true", help="pipeline FMT sampling of window k+1 with the decode of window k on two streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--d2h", action="store_true", help="also time the copy of the frames to host memory")
    return ap.parse_args()


def cpu_baseline(pkg, cfg, fmt_sd, dec_sd, feats, cond, nfe_evals):
    """The oracle (a port of the reference's torch-CPU path) on this host's cores, bounded sample:
    a few CFG evaluations of the FMT and a few decoded frames; fps = 1 / (evals/frame * t_eval + t_frame)."""
    from oracle import float_oracle as O
    n_threads = torch.get_num_threads()
    L = cfg.num_frames_for_clip
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, L, cfg.dim_w, generator=g)
    wa, we, r_s, s_r = cond["wa"][:, :L].cpu(), cond["we"][:, :1].cpu(), cond["r_s"].cpu(), cond["s_r"].cpu()
    px = torch.zeros(1, cfg.num_prev_frames, cfg.dim_w)
    n_eval, n_frames = 6, 4
    O.fmt_forward_cfv(fmt_sd, cfg, torch.tensor([0.5]), x, wa, r_s, we, px, px, None, 2.0, 1.0, 1.0)  # warm-up
    t0 = time.perf_counter()
    for i in range(n_eval):
        O.fmt_forward_cfv(fmt_sd, cfg, torch.tensor([i / n_eval]), x, wa, r_s, we, px, px, None, 2.0, 1.0, 1.0)
    t_eval = (time.perf_counter() - t0) / n_eval
    cfeats = [f.cpu() for f in feats]
    t0 = time.perf_counter()
    O.decode_frames(dec_sd, s_r, x[:, :n_frames] * 0.5, cfeats)
    t_frame = (time.perf_counter() - t0) / n_frames
    evals_per_frame = nfe_evals / float(L)
    fps = 1.0 / (evals_per_frame * t_eval + t_frame)
    return {"value": round(fps, 4), "unit": "frames/s", "cores": n_threads, "kind": "port",
            "sample": "%d CFG-3 FMT evaluations (%.3f s each) + %d decoded 512x512 frames (%.3f s each), fp32 oracle"
                      % (n_eval, t_eval, n_frames, t_frame)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("FLOAT_BENCH_BACKEND", "nccl")  # "gloo": ranks may share one GPU (smoke test of the N>1 path)
    n_dev = max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    pkg = load_pkg()
    cfg = pkg.config.FmtConfig()
    fps_video = 25.0
    T = int(math.ceil(args.seconds * fps_video))
    fmt_sd = pkg.weights.synth_fmt_state(cfg, seed=1)
    dec_sd = pkg.weights.synth_decoder_state(args.size, seed=1)
    hp = pkg.pipeline.FloatHotPath(fmt_sd, dec_sd, cfg, dev, args.size, args.fmt_dtype, args.dec_dtype,
                                   args.max_frames, use_graph=0 if args.no_graph else args.graph_mode)
    hp.fmt_stream_priority = args.fmt_priority
    hp.cu_split = args.cu_split
    feats = [f.to(dev) for f in pkg.weights.synth_feats(args.size, seed=1 + rank)]
    hp.dec.set_feats(feats)

    if args.mode in ("shard", "window") and world > 1:
        # one long clip of world*T frames; every rank runs the identical (deterministic) latent chain,
        # then decodes its contiguous frame range - see comfyui-float_optimized_amd/distributed.py
        T_total = T * world
        cond = pkg.pipeline.synth_conditions(cfg, T_total, seed=0, dynamic_we=args.dynamic_we, device=dev)
    else:
        T_total = T
        cond = pkg.pipeline.synth_conditions(cfg, T, seed=rank, dynamic_we=args.dynamic_we, device=dev)
    noise = pkg.fmt.draw_noise(hp.n_chunks(T_total), 1, cfg, seed=15).to(dev)
    a_cfg, e_cfg = (1.0, 3.0) if args.dynamic_we else (2.0, 1.0)

    seam = {}

    def step():
        if args.mode == "window" and world > 1:
            r_loc, (t0, t1), rep = pkg.distributed.sample_window_parallel(
                hp.fmt, cfg, cond["r_s"], cond["wa"], cond["we"], noise, args.nfe, a_cfg, 1.0, e_cfg, iters=args.window_iters)
            seam.update(rep)
            if world > 1:  # report the largest seam change over the ranks
                import torch.distributed as dist
                t = torch.tensor([rep["seam_rel_change"]], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                seam["seam_rel_change"] = float(t.item())
            return hp.decode(cond["s_r"], None, r_loc)
        fr = (rank * T, (rank + 1) * T) if (args.mode == "shard" and world > 1) else None
        return hp.generate(cond["r_s"], cond["wa"], cond["we"], cond["s_r"], None, args.nfe, a_cfg, 1.0, e_cfg, noise=noise,
                           overlap=args.overlap, frame_range=fr)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        frames = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frames = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert frames.shape[1:] == (args.size, args.size, 3) and (args.mode == "window" or frames.shape[0] == T)
    total_frames = T * world * args.steps
    fps = total_frames / elapsed

    extra = {}
    if rank == 0:
        # stage split (one more step, not part of `value`)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
        r_d = hp.sample(cond["r_s"], cond["wa"], cond["we"], args.nfe, a_cfg, 1.0, e_cfg, noise=noise)
        ev[1].record()
        fr = hp.decode(cond["s_r"], None, r_d, (0, T))
        ev[2].record()
        torch.cuda.synchronize()
        extra["stage_ms"] = {"fmt_sample": round(ev[0].elapsed_time(ev[1]), 3), "decode": round(ev[1].elapsed_time(ev[2]), 3)}
        try:
            # once-per-clip host-side stage (PyTorch-ROCm, not part of `value`): appearance encoder + Direction +
            # wav2vec2-base audio encoder with random weights on synthetic image/audio (SURVEY.md 8d inputs)
            enc = pkg.encoder.EncoderHIP(pkg.weights.synth_encoder_state(args.size, seed=1), args.size, cfg.dim_w, 20, dev,
                                         args.dec_dtype, direction_weight=dec_sd["direction.weight"])
            acfg = pkg.config.AudioConfig()
            aud = pkg.audio.AudioEncoderHIP(pkg.weights.synth_audio_state(acfg, seed=1), acfg, dev, args.dec_dtype)
            img = torch.rand(1, 3, args.size, args.size, device=dev) * 2 - 1
            wav = pkg.weights.synth_waveform(args.seconds, seed=1).to(dev)

            def timed(fn):
                fn()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                return round((time.perf_counter() - t1) * 1e3, 3)

            def enc_stage():  # image -> s_r, r_s, skip features into the decoder (HIP operator float_enc_*)
                enc.encode_image_into_latent(img, want_feats=False)
                enc.hand_feats_to(hp.dec)

            def aud_stage():  # waveform -> wa (T, 512): wav2vec2-base + audio projection (HIP operator float_aud_*)
                aud.inference(wav, seq_len=T)
            extra["stage_ms"]["appearance_encoder_hip"] = timed(enc_stage)
            extra["stage_ms"]["audio_encoder_hip"] = timed(aud_stage)

            # SURVEY.md 8(d) wall-clock definition, reported beside `value` (never as it): (image, waveform) in HBM ->
            # frames in pinned host memory through every operator of the path, one clip
            host = torch.empty(T, args.size, args.size, 3, dtype=torch.float32, pin_memory=True)

            def end_to_end():
                s_r, _, _, r_s = enc.encode_image_into_latent(img, want_feats=False)
                enc.hand_feats_to(hp.dec)
                wa = aud.inference(wav, seq_len=T)
                r_d2 = hp.sample(r_s, wa, cond["we"], args.nfe, a_cfg, 1.0, e_cfg, noise=noise)
                host.copy_(hp.decode(s_r, None, r_d2, (0, T)))
            e2e_ms = timed(end_to_end)
            extra["end_to_end"] = {"ms_per_clip": e2e_ms, "frames_per_s": round(T / (e2e_ms * 1e-3), 1),
                                   "includes": "appearance encoder + audio encoder + FMT sampling + decode + D2H of the frames (pinned)"}
            hp.dec.set_feats(feats)  # restore the bench's synthetic features
            del aud, enc, host
        except Exception as e:  # conditioning is plumbing; never fail the bench for it
            extra["stage_ms"]["conditioning"] = "n/a (%s: %s)" % (type(e).__name__, e)
        if args.d2h:
            host = torch.empty(fr.shape, dtype=torch.float32, pin_memory=True)
            t1 = time.perf_counter()
            host.copy_(fr)
            torch.cuda.synchronize()
            extra["stage_ms"]["d2h"] = round((time.perf_counter() - t1) * 1e3, 3)
            extra["fps_incl_d2h"] = round(T / ((extra["stage_ms"]["fmt_sample"] + extra["stage_ms"]["decode"] + extra["stage_ms"]["d2h"]) * 1e-3), 2)

    roof = None
    if rank == 0 and not args.no_roofline:
        # Per-launch kernel durations: one more identical step with hipEvents around every launch of
        # the two dominant kernel classes, recorded on the stream they are launched on (eager launches;
        # the timed region above replays the same kernels from a hipGraph).
        n_chunks_rank = hp.n_chunks(T_total)
        pkg.native.set_profiling(True)
        r_d = hp.sample(cond["r_s"], cond["wa"], cond["we"], args.nfe, a_cfg, 1.0, e_cfg, noise=noise)
        hp.decode(cond["s_r"], None, r_d, (0, T))
        torch.cuda.synchronize()
        g_ms, g_n = pkg.native.profile_ms(0)
        c_ms, c_n = pkg.native.profile_ms(1)
        pkg.native.set_profiling(False)
        n_eval = n_chunks_rank * (args.nfe - 1)
        gemm_total_ms, conv_total_ms = g_ms * g_n, c_ms * c_n
        gemm_bytes = FMT_WEIGHT_BYTES_PER_EVAL * n_eval
        conv_flop = DEC_FLOP_PER_FRAME * T
        gemm_roof = {"kernel": "fmt_gemm_kernel", "bound": "hbm", "achieved": round(gemm_bytes / (gemm_total_ms * 1e-3) / 1e9, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "launches": g_n, "avg_launch_us": round(g_ms * 1e3, 2),
                     "algorithmic_bytes_per_launch": round(gemm_bytes / max(g_n, 1)), "traffic": None}
        gemm_roof["frac"] = round(gemm_roof["achieved"] / HBM_PEAK_GBS, 4)
        conv_roof = {"kernel": "dec_conv_kernel", "bound": "mfma", "achieved": round(conv_flop / (conv_total_ms * 1e-3) / 1e12, 2),
                     "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "launches": c_n, "avg_launch_us": round(c_ms * 1e3, 2),
                     "algorithmic_flop_per_launch": round(conv_flop / max(c_n, 1)), "traffic": None}
        conv_roof["frac"] = round(conv_roof["achieved"] / MFMA_PEAK_TFLOPS, 4)
        # HBM traffic per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
        # separate passes, tools/profile_hotpath.py); null when the summary is not there
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            gemm_roof["traffic"] = pmc["fmt_gemm"]["hbm_bytes_per_launch"]
            conv_roof["traffic"] = pmc["dec_conv"]["hbm_bytes_per_launch"]
        except Exception:
            pass
        try:  # MFMA pipe utilisation of the two classes from the committed SQ counter pass (tools/make_mfma_json.py)
            mf = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_mfma.json")))
            gemm_roof["mfma_util_pmc"] = mf["fmt_gemm"]["mfma_util"]
            conv_roof["mfma_util_pmc"] = mf["dec_conv"]["mfma_util"]
        except Exception:
            pass
        roof = (conv_roof, gemm_roof) if conv_total_ms >= gemm_total_ms else (gemm_roof, conv_roof)
        extra["kernel_class_ms"] = {"fmt_gemm": round(gemm_total_ms, 2), "dec_conv": round(conv_total_ms, 2)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(pkg, cfg, fmt_sd, dec_sd, feats, cond, args.nfe - 1)

    if rank == 0:
        out = {
            "metric": "512x512 frames/sec end-to-end audio->video @50 ODE steps (hot path: FMT sampling + decode)",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "%s+%s" % (args.fmt_dtype, args.dec_dtype) if args.fmt_dtype != args.dec_dtype else args.fmt_dtype,
            "data": "synthetic",
            "config": {"workload": "configs[1]: %.0f s audio -> %d frames %dx%d, %d Euler evaluations/window, CFG a=%.1f e=%.1f%s"
                                   % (args.seconds, T, args.size, args.size, args.nfe - 1, a_cfg, e_cfg,
                                      ", dynamic per-window emotion" if args.dynamic_we else ""),
                       "nfe": args.nfe, "frames_per_clip": T, "fmt_dtype": args.fmt_dtype, "dec_dtype": args.dec_dtype,
                       "decode_batch": args.max_frames, "hip_graph": not args.no_graph,
                       "stage_overlap": args.overlap,
                       "parallelism": ("replicas x%d (one clip per GPU)" % world) if args.mode == "replicas" or world == 1
                       else ("shard: one %d-frame clip, latent chain replicated, frames sharded x%d" % (T_total, world)
                             if args.mode == "shard" else
                             "window: one %d-frame clip, windows sharded x%d, boundary all_gather x%d round(s), max seam change %.3e"
                             % (T_total, world, args.window_iters, seam.get("seam_rel_change", 0.0)))},
        }
        out.update(extra)
        if roof:
            out["roofline"] = roof[0]
            out["roofline_secondary"] = roof[1]
        if cpu:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
