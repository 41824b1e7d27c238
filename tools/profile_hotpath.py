#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 counter passes (PMC passes serialise kernels, so the whole
bench would take minutes): `--what dec` decodes 16 frames at 512x512 twice, `--what fmt` runs one
50-frame window with 10 Euler evaluations (eager launches), `--what fmtb` the same window for `--batch` stacked clips (default 16:
2 880 rows, tier 3 of the row-blocked LDS-DMA GEMM tiles - what bench.py's roofline_batch prices; 4 = 720 rows).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/profile_hotpath.py --what dec
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import load_pkg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--what", default="dec", choices=["dec", "fmt", "fmtb"])
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--batch", type=int, default=16)
args = ap.parse_args()
pkg = load_pkg()
cfg = pkg.config.FmtConfig()
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
if args.what == "dec":
    sd = pkg.weights.synth_decoder_state(512, seed=1)
    dec = pkg.decoder.SynthesisHIP(sd, 512, 512, dev, "fp16", max_frames=max(16, min(args.frames, 32)))
    dec.set_feats(pkg.weights.synth_feats(512, seed=1))
    s_r, r_d = torch.randn(1, 512, generator=g), torch.randn(1, args.frames, 512, generator=g) * 0.5
    for _ in range(args.reps):
        out = dec.decode_latent_into_processed_images(s_r, r_d)
    torch.cuda.synchronize()
    print("decoded", tuple(out.shape), float(out.mean()))
elif args.what == "fmtb":
    B = args.batch
    sd = pkg.weights.synth_fmt_state(cfg, seed=1)
    fmt = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, dev, "fp16", use_graph=0, max_batch=B)
    cs = [pkg.pipeline.synth_conditions(cfg, 50, seed=q) for q in range(B)]
    cat = lambda k: torch.cat([c[k] for c in cs])  # noqa: E731
    noise = pkg.fmt.draw_noise(1, B, cfg, 15)
    for _ in range(args.reps):
        r_d = fmt.sample(cat("r_s"), cat("wa"), cat("we"), noise, 11, 2.0, 1.0, 1.0)
    torch.cuda.synchronize()
    print("sampled", tuple(r_d.shape), float(r_d.abs().mean()))
else:
    sd = pkg.weights.synth_fmt_state(cfg, seed=1)
    fmt = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, dev, "fp16", use_graph=0)
    c = pkg.pipeline.synth_conditions(cfg, 50, seed=0)
    noise = pkg.fmt.draw_noise(1, 1, cfg, 15)
    for _ in range(args.reps):
        r_d = fmt.sample(c["r_s"], c["wa"], c["we"], noise, 11, 2.0, 1.0, 1.0)
    torch.cuda.synchronize()
    print("sampled", tuple(r_d.shape), float(r_d.abs().mean()))
