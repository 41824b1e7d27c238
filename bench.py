#!/usr/bin/env python3
"""Benchmark of the FLOAT audio -> talking-portrait path on MI355X.

One "step" = one clip end to end through the PRODUCT's own call, `InferenceAgent.infer_device` (what `run_inference` and the
FLOAT Process node run after their host-side image / audio plumbing): a 512x512 portrait and `--seconds` of 16 kHz audio, both
resident in HBM, through every operator of the path - appearance encoder, wav2vec2 audio encoder, noise draw, FMT sampling
(`--nfe` grid points: default 51 = 50 Euler evaluations per 50-frame window, 3-way CFG a=2 e=1), Synthesis decoder - to `T`
fp32 frames in a freshly allocated pinned HOST tensor (the reference's destination, FLOAT.py:139; the frames of decode batch i
leave inside the launches of batch i+1), stream synchronised.  `value` = frames of all ranks / wall-clock of the timed steps.
The FMT sampling + decode part alone (frames left in HBM: round 1's headline) is reported beside it as `hot_path`, and the same
step with bf16 FMT operands (BASELINE.json's wording for configs[1]) as `value_bf16`.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1 without a launcher: this process starts N ranks itself (one per GPU, before it touches a GPU) and relays rank 0's line.
Modes for N > 1: `replicas` one clip per GPU (BASELINE.json configs[3], weak scaling, no communication); `shard` ONE
N x `--seconds` clip, latent chain replicated, frames sharded (exact); `window` windows sharded with an RCCL all_gather of
boundary latents (configs[2]; approximate unless --window-iters = N-1).  Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# an fp16 operator that leaves its range during a timed clip invalidates the number: make it an error (pipeline.report_range)
os.environ["FLOAT_AMD_RANGE"] = "raise"

# SURVEY.md section 8(d) algorithmic work per unit
FMT_WEIGHT_BYTES_PER_EVAL = 313.4e6    # every 16-bit weight of the FMT once
FMT_ADALN_WEIGHT_BYTES = 51200 * 1024 * 2.0  # of which the fused adaLN projection: now read once per WINDOW, not per evaluation
FMT_FLOP_PER_EVAL_CFG3 = 55.87e9
DEC_FLOP_PER_FRAME = 37.79e9
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0              # dense bf16/fp16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="default: >= 10 s of timed GPU work")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--nfe", type=int, default=51, help="Euler grid points; evaluations = nfe-1")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--fmt-dtype", default="fp16", help="MFMA operand type of the FMT (fp32 accumulate): fp16 holds the stated "
                    "tolerance (>= 40 dB frames vs the reference); bf16 = BASELINE.json's wording, same speed, 34 dB")
    ap.add_argument("--dec-dtype", default="fp16")
    ap.add_argument("--max-frames", type=int, default=32)
    ap.add_argument("--mode", default="replicas", choices=["replicas", "shard", "window"])
    ap.add_argument("--window-iters", type=int, default=1)
    ap.add_argument("--window-resolve", type=int, default=1, help="windows of a rank re-solved after each boundary all_gather "
                    "(the seam); 0 = the rank's whole range (with --window-iters N-1: the exact mode)")
    ap.add_argument("--dynamic-we", action="store_true", help="BASELINE configs[4]: per-window emotion, a=1 e=3")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the stage split / hot-path side numbers")
    ap.add_argument("--no-bf16", action="store_true", help="skip the second measurement with bf16 FMT operands")
    ap.add_argument("--no-s2e", action="store_true", help="skip the measurement with the speech-emotion model in the step "
                    "(emotion='none', the reference's default widget: wav2vec2-large classifier on the clip's audio)")
    ap.add_argument("--no-variants", action="store_true", help="skip the literal nfe=50 run, the run from host inputs and the batches")
    ap.add_argument("--no-clock-sample", action="store_true", help="skip the rocm-smi sample of the delivered clock / power (3 s of extra steps)")
    ap.add_argument("--overlap", default=None, help="prio | cu:N: decode window k on a second stream beside the FMT chain of window "
                    "k + 1 (FLOAT_AMD_OVERLAP; pipeline.generate_to_host_overlap).  Default: the product's default (sequential)")
    ap.add_argument("--quick", action="store_true", help="only the headline measurement (A/B runs): no extras / roofline / "
                    "CPU baseline / bf16 / s2e / variants")
    ap.add_argument("--batches", default="4,8,16", help="stacked-clip throughput runs (value_batchB), comma separated; empty = none")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------- N > 1 self-launch
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_threads(world):
    """Intra-op host threads of one rank: the cores this process may use, divided by the ranks of the node (at least 1).  Eight
    ranks each bringing torch's default pool (one thread per core) oversubscribe the host 8x while they synthesise and pack
    600 M parameters and draw noise per clip."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, cores // max(1, world))


def launch_ranks(n):
    """Start n fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) BEFORE this process
    makes any GPU call, wait for all of them, relay rank 0's output.  Exit code != 0 if any rank failed or fewer than n
    joined (a rank that cannot join the rendezvous exits non-zero by itself)."""
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(n))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # host threads per rank: the ranks share the host's cores (rank_threads); no NUMA pinning - the step's host work is a noise
    # draw and a few hundred launches, the inputs cross PCIe once
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        env.setdefault(var, str(rank_threads(n)))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    if any(rcs):
        sys.stderr.write("bench.py: rank exit codes %s\n" % rcs)
        sys.exit(1)
    line = [l for l in out0.splitlines() if l.startswith("{")]
    if not line or json.loads(line[-1]).get("n_gpus") != n:
        sys.stderr.write("bench.py: expected a result line with n_gpus=%d\n" % n)
        sys.exit(1)
    sys.exit(0)


def rendezvous_store(rank, world):
    """The job's TCPStore (rank 0 hosts it unless a torchrun agent already does) with a roll call in front of
    init_process_group: every rank signs in, rank 0 waits FLOAT_BENCH_RDZV_TIMEOUT seconds (default 180) for all of them and
    otherwise says WHICH ranks never arrived and exits 3; the others exit 3 when rank 0's go-ahead does not come.  A hung
    rendezvous of an 8-GPU run is then one readable line in the log, not a driver timeout."""
    import datetime
    import torch.distributed as dist
    timeout = float(os.environ.get("FLOAT_BENCH_RDZV_TIMEOUT", "180"))
    agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
    try:
        store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world, is_master=(rank == 0 and not agent_store),
                              timeout=datetime.timedelta(seconds=timeout), wait_for_workers=False)
    except Exception as e:  # noqa: BLE001
        sys.stderr.write("bench.py: rank %d could not reach the rendezvous store at %s:%s within %.0f s: %s\n"
                         % (rank, os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"), timeout, str(e).splitlines()[0][:200]))
        sys.exit(3)
    store.set("bench/here/%d" % rank, socket.gethostname())
    if rank == 0:
        t_end = time.time() + timeout
        missing = list(range(world))
        while missing and time.time() < t_end:
            missing = [r for r in missing if not store.check(["bench/here/%d" % r])]
            if missing:
                time.sleep(0.2)
        if missing:
            sys.stderr.write("bench.py: rendezvous timed out after %.0f s: rank(s) %s of %d never arrived\n" % (timeout, missing, world))
            store.set("bench/go", "abort")
            sys.exit(3)
        store.set("bench/go", "go")
    else:
        try:
            store.wait(["bench/go"], datetime.timedelta(seconds=timeout))
            ok = store.get("bench/go") == b"go"
        except Exception:  # noqa: BLE001
            ok = False
        if not ok:
            sys.stderr.write("bench.py: rank %d: no go-ahead from rank 0 within %.0f s (other ranks missing, or rank 0 died)\n" % (rank, timeout))
            sys.exit(3)
    return dist.PrefixStore("pg", store)


class stdout_to_stderr:
    """RCCL prints a banner (ROCm version / hostname / library path) on the C-level stdout when the first communicator comes
    up; the driver reads ONE JSON line from this process's stdout, so file descriptor 1 points at stderr meanwhile."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:  # the banner sits in libc's buffer (stdout is a pipe: fully buffered): push it out while fd 1 still is stderr
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def source_hash():
    """Identity of the kernels a profile summary belongs to: sha256 over the HIP sources of the profiled kernel classes - the
    FMT and decoder operators and what they share (the GPU box has no .git; the encoder / audio operators have no counters)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "comfyui-float_optimized_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")) and f.startswith(("fmt_", "dec_", "common")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load_profile_json(name, warnings):
    """A committed rocprofv3 summary (profiles/<name>), or None when absent or taken on other kernel sources."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None
    if d.get("source_hash") != source_hash():
        warnings.append("profiles/%s was taken on kernel sources %s, this build is %s: counter-derived fields are null"
                        % (name, d.get("source_hash"), source_hash()))
        return None
    return d


def cpu_baseline(pkg, torch, cfg, fmt_sd, dec_sd, feats, cond, nfe_evals):
    """The oracle (a port of the reference's torch-CPU path) on this host's cores, bounded sample: a few CFG evaluations of
    the FMT and a few decoded frames; fps = 1 / (evals/frame * t_eval + t_frame)."""
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
            "sample": "%d CFG-3 FMT evaluations (%.3f s each) + %d decoded 512x512 frames (%.3f s each), fp32 oracle; "
                      "encoders not included" % (n_eval, t_eval, n_frames, t_frame)}


def main():
    args = parse()
    if args.quick:
        args.no_extras = args.no_roofline = args.no_cpu_baseline = args.no_bf16 = args.no_s2e = args.no_variants = True
    if args.overlap is not None:
        os.environ["FLOAT_AMD_OVERLAP"] = args.overlap
    if os.environ.get("FLOAT_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["FLOAT_BENCH_WATCHDOG"]), exit=True)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)  # never returns

    import numpy as np
    import torch
    from tests.util import load_pkg

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    host_threads = rank_threads(world)
    if world > 1:  # also under torchrun, which sets OMP_NUM_THREADS=1 by itself: the pool follows the policy, not the launcher
        torch.set_num_threads(host_threads)
    backend = os.environ.get("FLOAT_BENCH_BACKEND", "nccl")  # "gloo": ranks may share one GPU (smoke test of the N>1 path)
    n_dev = max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(dev)
    dist = None
    rccl_ranks, rccl_note = None, None
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ  # torchrun / launch_ranks: env rendezvous
    if world > 1 or (launched and backend == "nccl"):
        # (also a one-rank launch by torchrun: its agent owns the store, a private tcp:// rendezvous would wait for ever)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        store = rendezvous_store(rank, world)  # exits non-zero, naming the ranks that never arrived, instead of hanging
        with stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", store=store, rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, store=store, rank=rank, world_size=world)
        if dist.get_world_size() != world:
            raise RuntimeError("rendezvous gave %d ranks, expected %d" % (dist.get_world_size(), world))
    elif backend == "nccl" and os.environ.get("FLOAT_BENCH_RCCL1", "1") != "0":
        # N = 1: still bring RCCL up (a one-rank communicator on this GPU) so that the collective calls of the N > 1 modes have
        # run on the box at least once; a failure here is reported on the line, it does not fail the single-GPU measurement
        try:
            import torch.distributed as dist
            with stdout_to_stderr():
                dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % free_port(), rank=0, world_size=1, device_id=dev)
        except Exception as e:  # noqa: BLE001
            rccl_note = "RCCL one-rank communicator failed: %s" % (str(e).splitlines()[0][:200],)
            dist = None
    if dist is not None and backend == "nccl":
        # ranks as RCCL itself counts them: all_reduce of ones on the device, plus the other two collectives of
        # distributed.py (all_gather of boundary latents, broadcast of the chain) on tensors of their real sizes
        with stdout_to_stderr():  # the communicator (and its banner) comes up with the first collective
            ones = torch.ones(1, device=dev)
            dist.all_reduce(ones)
            tail = torch.zeros(2, 10, 512, device=dev)
            gathered = [torch.empty_like(tail) for _ in range(dist.get_world_size())]
            dist.all_gather(gathered, tail)
            dist.broadcast(tail, src=0)
            torch.cuda.synchronize()
        rccl_ranks = int(ones.item())

    pkg = load_pkg()
    cfg = pkg.config.FmtConfig()
    fps_video = 25.0
    T = int(math.ceil(args.seconds * fps_video))
    fmt_sd = pkg.weights.synth_fmt_state(cfg, seed=1)
    dec_sd = pkg.weights.synth_decoder_state(args.size, seed=1)
    # the product object: the agent the Load FLOAT Models node builds (src/nodes/generate.py), on seeded weights of the
    # checkpoint's shapes (wav2vec2-base audio encoder; no speech-emotion model: the clip's emotion is a label)
    import importlib
    gen = importlib.import_module(pkg.__name__ + ".src.nodes.generate")
    opt = importlib.import_module(pkg.__name__ + ".src.nodes.options.base_options").BaseOptions()
    opt.nfe, opt.input_size, opt.fps, opt.rank = args.nfe, args.size, fps_video, dev
    acfg = pkg.config.AudioConfig()
    parts = dict(enc=pkg.weights.synth_encoder_state(args.size, seed=1), dec=dec_sd, fmt=fmt_sd,
                 audio_encoder=(pkg.weights.synth_audio_state(acfg, seed=1), acfg))
    with_s2e = rank == 0 and world == 1 and not args.no_s2e and not args.dynamic_we
    if with_s2e:  # the speech-emotion model at its checkpoint shape (wav2vec2-large, 24 layers: wav2vec2_ser.py:23-96)
        ecfg = pkg.config.emotion_audio_config()
        parts["emotion_encoder"] = (pkg.weights.synth_audio_state(ecfg, seed=2), ecfg)
    agent = gen.InferenceAgent(opt, parts, dev, max_frames=args.max_frames, use_graph=0 if args.no_graph else 2,
                               fmt_dtype=args.fmt_dtype, dec_dtype=args.dec_dtype)
    hp, enc, aud = agent.G, agent.enc, agent.audio_encoder

    one_clip = args.mode in ("shard", "window") and world > 1
    T_total = T * world if one_clip else T
    # inputs of the step, resident in HBM before the timed region: portrait, waveform, emotion scores, noise
    img = (torch.from_numpy(np.random.RandomState(0 if one_clip else rank).rand(1, 3, args.size, args.size).astype("float32"))
           * 2 - 1).to(dev)
    wav = pkg.weights.synth_waveform(args.seconds * (world if one_clip else 1), seed=1 if one_clip else 1 + rank).to(dev)
    cond = pkg.pipeline.synth_conditions(cfg, T_total, seed=0 if one_clip else rank, dynamic_we=args.dynamic_we, device=dev)
    we = cond["we"]
    noise = pkg.fmt.draw_noise(hp.n_chunks(T_total), 1, cfg, seed=15).to(dev)
    a_cfg, e_cfg = (1.0, 3.0) if args.dynamic_we else (2.0, 1.0)
    t0f, t1f = pkg.distributed.frame_shard(T_total, world, rank) if (args.mode == "shard" and world > 1) else (0, T_total)
    if args.mode == "window" and world > 1:
        L = cfg.num_frames_for_clip
        w0, w1 = pkg.distributed.window_shard(hp.n_chunks(T_total), world, rank)
        t0f, t1f = w0 * L, min(T_total, w1 * L)
    n_local = t1f - t0f
    seam = {}
    product = world == 1 or args.mode == "replicas"  # the agent's call; shard / window drive the operators themselves
    if args.dynamic_we:
        product = False  # per-window emotion comes from the VA sampler node, not from run_inference
    last = {}

    def conditioning():
        s_r, _, _, r_s = enc.encode_image_into_latent(img, want_feats=False)
        enc.hand_feats_to(hp.dec)
        wa = aud.inference(wav, seq_len=T_total)
        return s_r, r_s, wa

    emo_of_step = ["neutral"]  # a label: the emotion is given.  None = the reference's default widget 'none' (speech-to-emotion)

    def step():
        """(image, waveform) in HBM -> this rank's frames in pinned host memory."""
        last.pop("host", None)  # the previous result is released first, as a caller that consumed it would have
        if product:
            last["host"] = agent.infer_device(img, wav, a_cfg, 1.0, e_cfg, emo=emo_of_step[0], seed=15)
            return
        s_r, r_s, wa = conditioning()
        noise = agent._noise_to_device(hp.n_chunks(T_total), 15)  # the clip's draw, into the pinned buffer, as infer_device does
        if args.mode == "window" and world > 1:
            r_loc, _, rep = pkg.distributed.sample_window_parallel(hp.fmt, cfg, r_s, wa, we, noise, args.nfe, a_cfg, 1.0, e_cfg,
                                                                   iters=args.window_iters, resolve_chunks=args.window_resolve)
            seam.update(rep)
            last["host"] = hp.decode_to_host(s_r, r_loc[0])
        else:
            r_d = hp.sample(r_s, wa, we, args.nfe, a_cfg, 1.0, e_cfg, noise=noise)
            last["host"] = hp.decode_to_host(s_r, r_d[0, t0f:t1f])
        torch.cuda.current_stream(dev).synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if args.mode == "window":
            for key in ("seam_rel_change", "seam_next_rel_change"):
                t = torch.as_tensor(seam.get(key, 0.0)).reshape(1).to(dev if backend == "nccl" else "cpu", torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                seam[key] = float(t.item())
    host = last["host"]
    per_rank = None
    if world > 1:
        # Every rank's own stage times (hipEvents, one more pass outside the timed region) and the latency of the job's one data-path
        # collective - the boundary all_gather of 2 x 10 x 512 floats - so that the first multi-GPU run reads rank by rank
        def ev_ms1(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn()
            e1.record()
            torch.cuda.synchronize()
            return out, round(e0.elapsed_time(e1), 3)
        (s_r_, r_s_, wa_), t_cond = ev_ms1(conditioning)
        r_d_, t_chain = ev_ms1(lambda: hp.sample(r_s_, wa_, we, args.nfe, a_cfg, 1.0, e_cfg, noise=noise))
        _, t_dec = ev_ms1(lambda: hp.dec.decode_into_host(s_r_, r_d_[0, t0f:t1f], host, hp.staging(n_local)))
        cdev = dev if backend == "nccl" else "cpu"
        tail = torch.zeros(2, 10, 512, device=cdev)
        gathered = [torch.empty_like(tail) for _ in range(world)]
        dist.all_gather(gathered, tail)
        barrier()
        t_ag = time.perf_counter()
        for _ in range(10):
            dist.all_gather(gathered, tail)
        if backend == "nccl":
            torch.cuda.synchronize()
        t_ag = (time.perf_counter() - t_ag) / 10
        mine = {"rank": rank, "device": torch.cuda.get_device_properties(dev).name, "local_rank": local_rank, "frames": n_local,
                "stage_ms": {"encoders": t_cond, "chain": t_chain, "decode_and_hand_over": t_dec},
                "boundary_all_gather_us": round(t_ag * 1e6, 1), "host_threads": torch.get_num_threads()}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    assert host.shape[0] == n_local and host.is_pinned()
    assert float(host[0].min()) >= 0.0 and float(host[-1].max()) <= 1.0 and float(host.mean()) > 0.0
    # fp16 range: every timed clip went through agent.check_range in "raise" mode (product path); the operator-driven
    # modes are checked here
    range_hits = agent.range_counts(reset=True)
    assert not any(range_hits.values()), "an fp16 operator left its range: frames are not the reference's: %s" % range_hits
    frames_per_step = T_total if one_clip else T * world
    fps = frames_per_step * args.steps / elapsed
    staging = hp.staging(n_local)

    extra, warnings = {}, []
    if rank == 0 and not args.no_extras:
        def ev_ms(fn, reps=3):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return round(e0.elapsed_time(e1) / reps, 3)

        s_r, r_s, wa = conditioning()
        keep = {}

        def f_sample():
            keep["r_d"] = hp.sample(r_s, wa, we, args.nfe, a_cfg, 1.0, e_cfg, noise=noise)
        st = {"appearance_encoder": ev_ms(lambda: (enc.encode_image_into_latent(img, want_feats=False), enc.hand_feats_to(hp.dec))),
              "audio_encoder": ev_ms(lambda: aud.inference(wav, seq_len=T_total)),
              "fmt_sample": ev_ms(f_sample)}
        if with_s2e:
            st["speech_emotion"] = ev_ms(lambda: agent.emotion_predictor(wav))
        rd_loc = keep["r_d"][0, t0f:t1f]
        st["decode"] = ev_ms(lambda: hp.decode(s_r, None, rd_loc))
        st["decode_and_d2h"] = ev_ms(lambda: hp.dec.decode_into_host(s_r, rd_loc, host, staging))
        st["d2h_alone"] = ev_ms(lambda: host.copy_(staging, non_blocking=True))
        extra["stage_ms"] = st
        hot = st["fmt_sample"] + st["decode"]
        extra["hot_path"] = {"ms_per_clip": round(hot, 3), "frames_per_s": round(n_local / (hot * 1e-3), 1),
                             "what": "FMT sampling + decode of this rank's frames, conditioning pre-staged, frames left in HBM"}

    roof = None
    if rank == 0 and not args.no_roofline:
        # Per-launch kernel durations: one more identical pass with hipEvents around every launch of the dominant kernel
        # classes, recorded on the stream they are launched on (eager launches; the timed region replays the same kernels
        # from a hipGraph).
        s_r, r_s, wa = conditioning()
        pkg.native.set_profiling(True)
        r_d = hp.sample(r_s, wa, we, args.nfe, a_cfg, 1.0, e_cfg, noise=noise)
        hp.decode(s_r, None, r_d[0, t0f:t1f])
        torch.cuda.synchronize()
        g_ms, g_n = pkg.native.profile_ms(0)
        c_ms, c_n = pkg.native.profile_ms(1)
        m_ms, m_n = pkg.native.profile_ms(2)
        pkg.native.set_profiling(False)
        n_win = hp.n_chunks(T_total)
        n_eval = n_win * (args.nfe - 1)
        gemm_total_ms, conv_total_ms, mod_total_ms = g_ms * g_n, c_ms * c_n, m_ms * max(m_n, 0)
        gemm_bytes = (FMT_WEIGHT_BYTES_PER_EVAL - FMT_ADALN_WEIGHT_BYTES) * n_eval
        conv_flop = DEC_FLOP_PER_FRAME * n_local
        mod_flop = 2.0 * (3 * cfg.n_tokens) * 51200 * 1024 * n_eval
        gemm_roof = {"kernel": "fmt_gemm_kernel (step chain: qkv / proj / fc1 / fc2 / x-embed / head)", "bound": "hbm",
                     "achieved": round(gemm_bytes / (gemm_total_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "launches": g_n, "avg_launch_us": round(g_ms * 1e3, 2),
                     "algorithmic_bytes_per_launch": round(gemm_bytes / max(g_n, 1)), "traffic": None}
        gemm_roof["frac"] = round(gemm_roof["achieved"] / HBM_PEAK_GBS, 4)
        conv_roof = {"kernel": "dec_conv_kernel", "bound": "mfma", "achieved": round(conv_flop / (conv_total_ms * 1e-3) / 1e12, 2),
                     "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "launches": c_n, "avg_launch_us": round(c_ms * 1e3, 2),
                     "algorithmic_flop_per_launch": round(conv_flop / max(c_n, 1)), "traffic": None}
        conv_roof["frac"] = round(conv_roof["achieved"] / MFMA_PEAK_TFLOPS, 4)
        mod_roof = None
        if m_n > 0:
            mod_roof = {"kernel": "fmt_gemm_big4_kernel (adaLN projection of all evaluations of a window; FLOAT_FMT_BIG=0: fmt_gemm_dma_kernel)", "bound": "mfma",
                        "achieved": round(mod_flop / (mod_total_ms * 1e-3) / 1e12, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "launches": m_n, "avg_launch_us": round(m_ms * 1e3, 2), "algorithmic_flop_per_launch": round(mod_flop / m_n),
                        "traffic": None}
            mod_roof["frac"] = round(mod_roof["achieved"] / MFMA_PEAK_TFLOPS, 4)
        # HBM traffic per launch / MFMA pipe utilisation from the committed rocprofv3 --pmc passes (tools/profile_round.sh),
        # only when they were taken on these kernel sources
        pmc = load_profile_json("r06_pmc_traffic.json", warnings)
        if pmc:
            gemm_roof["traffic"] = pmc.get("fmt_gemm", {}).get("hbm_bytes_per_launch")
            conv_roof["traffic"] = pmc.get("dec_conv", {}).get("hbm_bytes_per_launch")
        mf = load_profile_json("r06_pmc_mfma.json", warnings)
        if mf:
            gemm_roof["mfma_util_pmc"] = mf.get("fmt_gemm", {}).get("mfma_util")
            conv_roof["mfma_util_pmc"] = mf.get("dec_conv", {}).get("mfma_util")
        # Peaks MEASURED on this box beside the spec-sheet ones: streaming read bandwidth and the dense fp16
        # MFMA rate of register-operand loops (float_probe_peaks).  `peak` / `frac` stay the guide's figures.
        try:
            peaks = pkg.native.probe_peaks(dev)
            props = torch.cuda.get_device_properties(dev)
            extra["device"] = dict(peaks, name=props.name, arch=getattr(props, "gcnArchName", None),
                                   hbm_GiB=round(props.total_memory / 2**30, 1))
            mfma_meas = max(peaks["mfma_f16_16x16x32_TFLOPs"], peaks["mfma_f16_32x32x16_TFLOPs"])
            for r_, pk in ((gemm_roof, peaks["hbm_read_GBps"]), (conv_roof, mfma_meas), (mod_roof, mfma_meas)):
                if r_ is not None and pk > 0:
                    r_["peak_measured"] = pk
                    r_["frac_measured"] = round(r_["achieved"] / pk, 4)
        except Exception as e:  # the probe needs 4 GiB of scratch; the line is valid without it
            warnings.append("float_probe_peaks failed: %s" % e)
        # Shader clock and socket power the board DELIVERS under the step (outside the timed region: a few more steps run while
        # `rocm-smi` is asked from a side thread).  2.5 PFLOP/s is the MFMA rate at 2.4 GHz; profiles/r05_clock_power.txt has the
        # same sample under each kernel class (the adaLN GEMM alone sits at the socket's power cap, 1.92-1.97 GHz).
        if product and not args.no_clock_sample:
            try:
                import re
                import subprocess
                import threading
                samples, stop = [], threading.Event()
                # rocm-smi numbers the physical GPUs: the bench device through HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES if they remap it
                smi_index = dev.index or 0
                vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or ""
                vis = [v.strip() for v in vis.split(",") if v.strip()]
                if smi_index < len(vis) and vis[smi_index].isdigit():
                    smi_index = int(vis[smi_index])

                def sampler():
                    while not stop.is_set():
                        try:
                            out = subprocess.run(["rocm-smi", "-d", str(smi_index), "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
                        except Exception:  # noqa: BLE001
                            return
                        m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
                        w = re.search(r"Power \(W\): ([\d.]+)", out)
                        if m:
                            samples.append((int(m.group(1)), float(w.group(1)) if w else None))
                th = threading.Thread(target=sampler, daemon=True)
                t_end = time.perf_counter() + 3.0
                step()
                th.start()
                while time.perf_counter() < t_end:
                    step()
                stop.set()
                th.join(timeout=15)
                samples = [x for x in samples if x[0] > 500]  # a sample that fell between two steps reads the idle clock
                if not samples:
                    warnings.append("clock sample: rocm-smi -d %d gave no sample under the step (not installed, or no busy sample in 3 s)" % smi_index)
                if samples and "device" in extra:
                    extra["device"]["sclk_MHz_under_step"] = [min(x[0] for x in samples), max(x[0] for x in samples)]
                    pw = [x[1] for x in samples if x[1] is not None]
                    if pw:
                        extra["device"]["power_W_under_step"] = [min(pw), max(pw)]
            except Exception as e:  # noqa: BLE001  (no rocm-smi on the box: the line is valid without the sample)
                warnings.append("clock sample failed: %s" % (str(e).splitlines()[0][:120],))
        roofs = sorted([(gemm_total_ms, gemm_roof), (conv_total_ms, conv_roof)] + ([(mod_total_ms, mod_roof)] if mod_roof else []),
                       key=lambda x: -x[0])
        roof = [r for _, r in roofs]
        extra["kernel_class_ms"] = {"fmt_gemm": round(gemm_total_ms, 2), "dec_conv": round(conv_total_ms, 2),
                                    "fmt_adaln_gemm": round(mod_total_ms, 2)}

    # the same step with bf16 FMT operands (BASELINE.json names bf16 for configs[1]; it fails the 40 dB frame tolerance -
    # tests/test_pipeline_gpu.py - so the headline type is fp16, same MFMA rate): measured in the same run
    bf16 = None
    if rank == 0 and world == 1 and product and not args.no_bf16 and args.fmt_dtype != "bf16":
        fmt_keep = hp.fmt
        hp.fmt = pkg.fmt.FlowMatchingTransformerHIP(fmt_sd, cfg, dev, "bf16", 0 if args.no_graph else 2, 1)
        k = max(3, min(20, args.steps))
        step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        barrier()
        el = time.perf_counter() - t0
        bf16 = {"value_bf16": round(T * k / el, 3), "ms_per_step_bf16": round(el / k * 1e3, 3), "steps_bf16": k}
        hp.fmt = fmt_keep

    def timed(k):
        step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        barrier()
        return time.perf_counter() - t0

    variants = {}
    batch_done = 0
    if rank == 0 and world == 1 and product:
        k = max(3, min(20, args.steps))
        if with_s2e:
            # the reference's DEFAULT workflow: emotion = 'none' -> the speech-emotion model scores the clip's audio
            # (nodes.py:146-160, FLOAT.py:196-198); same step, one more operator inside the timed region
            emo_of_step[0] = None
            el = timed(k)
            emo_of_step[0] = "neutral"
            variants.update({"value_s2e": round(T * k / el, 3), "ms_per_step_s2e": round(el / k * 1e3, 3), "steps_s2e": k})
        if not args.no_variants:
            # the literal "nfe = 50" grid of the reference (49 Euler evaluations per window; SURVEY.md section 8 "N ODE steps")
            nfe_keep = opt.nfe
            opt.nfe = args.nfe - 1
            el = timed(k)
            opt.nfe = nfe_keep
            variants.update({"value_nfe%d" % (args.nfe - 1): round(T * k / el, 3), "ms_per_step_nfe%d" % (args.nfe - 1): round(el / k * 1e3, 3)})
            # from HOST inputs, as FloatProcess hands them over (generate.py:139-148): (H,W,3) image in [0,1] and a 16 kHz mono
            # waveform in pageable host memory -> resize / normalise / H2D (host_inputs) -> the step above
            ref_img = ((img[0].permute(1, 2, 0) + 1) * 0.5).cpu()[None]
            ref_audio = {"waveform": wav.reshape(1, 1, -1).cpu(), "sample_rate": 16000}

            def from_host():
                last.pop("host", None)
                last["host"] = agent.run_inference(None, ref_img, ref_audio, a_cfg, 1.0, e_cfg, emo="neutral", no_crop=True, seed=15)
            from_host()
            barrier()
            t0 = time.perf_counter()
            for _ in range(k):
                from_host()
            barrier()
            el = time.perf_counter() - t0
            t1 = time.perf_counter()
            for _ in range(k):
                agent.host_inputs(ref_img, ref_audio, True)
            torch.cuda.synchronize()
            variants.update({"value_from_host_inputs": round(T * k / el, 3), "ms_per_step_from_host_inputs": round(el / k * 1e3, 3),
                             "host_inputs_ms": round((time.perf_counter() - t1) / k * 1e3, 3)})
            # THROUGHPUT form of the same product path: B clips of equal length through ONE stacked FMT chain
            # (InferenceAgent.infer_device_batch -> float_fmt_sample_batch; what FLOAT Process runs for a batch of portraits,
            # BASELINE configs[3] on one GPU): every weight is read once per evaluation for all of them
            def batch_run(nb, reps):
                items = [(torch.roll(img, i, dims=-1), wav) for i in range(nb)]

                def batch_step():
                    last.pop("hosts", None)
                    last["hosts"] = agent.infer_device_batch(items, a_cfg, 1.0, e_cfg, "neutral", [15 + i for i in range(nb)])
                batch_step()
                barrier()
                t0 = time.perf_counter()
                for _ in range(reps):
                    batch_step()
                barrier()
                el = time.perf_counter() - t0
                last.pop("hosts", None)
                return el

            last.pop("host", None)
            for nb in [int(b) for b in args.batches.split(",") if b]:
                try:
                    reps = max(2, k // nb)
                    el = batch_run(nb, reps)
                    variants.update({"value_batch%d" % nb: round(nb * T * reps / el, 3), "ms_per_step_batch%d" % nb: round(el / reps * 1e3, 3)})
                    batch_done = nb
                except Exception as e:  # noqa: BLE001  (host or device memory for the largest batches)
                    warnings.append("batch of %d clips failed: %s" % (nb, str(e).splitlines()[0][:200]))
                    break
            variants["batch_what"] = ("value_batchB: B clips of %.0f s stacked in ONE FMT chain (InferenceAgent.infer_device_batch -> "
                                      "float_fmt_sample_batch), decoded back to back, B x %d frames per step" % (args.seconds, T))
            # the stacked chain's GEMMs under hipEvents (class 3: the row-blocked LDS-DMA tile) at the largest batch that ran
            if batch_done and not args.no_roofline:
                fmtb = hp.batched_fmt(batch_done)
                cs = [conditioning() for _ in range(batch_done)]
                r_sb = torch.cat([c[1].reshape(1, -1) for c in cs])
                wab = torch.cat([c[2].reshape(1, T_total, -1) for c in cs])
                web = torch.cat([we.reshape(1, 1, -1)] * batch_done)
                nz = agent._noise_batch_to_device(hp.n_chunks(T_total), [15 + i for i in range(batch_done)])
                fmtb.sample(r_sb, wab, web, nz, args.nfe, a_cfg, 1.0, e_cfg)
                torch.cuda.synchronize()
                pkg.native.set_profiling(True)
                t0 = time.perf_counter()
                fmtb.sample(r_sb, wab, web, nz, args.nfe, a_cfg, 1.0, e_cfg)
                torch.cuda.synchronize()
                b_ms, b_n = pkg.native.profile_ms(3)
                bm_ms, bm_n = pkg.native.profile_ms(2)
                pkg.native.set_profiling(False)
                n_eval = hp.n_chunks(T_total) * (args.nfe - 1)
                rows = batch_done * 3 * cfg.n_tokens
                blk_flop = 2.0 * rows * 1024 * (3072 + 1024 + 4096 + 4096) * 8 * n_eval  # qkv + proj + fc1 + fc2 of the 8 blocks
                if b_n > 0:
                    rb = {"kernel": "fmt_gemm_rbs_kernel (qkv / proj / fc1 / fc2 of a stacked-clip step chain)", "batch": batch_done,
                          "rows": rows, "bound": "mfma", "achieved": round(blk_flop / (b_ms * b_n * 1e-3) / 1e12, 2),
                          "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "launches": b_n, "avg_launch_us": round(b_ms * 1e3, 2),
                          "algorithmic_flop_per_launch": round(blk_flop / b_n), "traffic": None,
                          "note": "one tile per CU fetches its operands once by LDS-DMA: fetch-rate-bound below ~2 900 rows "
                                  "(DESIGN.md section 6), hence far from the MFMA peak it is priced against"}
                    rb["frac"] = round(rb["achieved"] / MFMA_PEAK_TFLOPS, 4)
                    if bm_n > 0:
                        mflop = 2.0 * rows * 51200 * 1024 * n_eval
                        rb["adaln_gemm"] = {"launches": bm_n, "avg_launch_us": round(bm_ms * 1e3, 2),
                                            "achieved": round(mflop / (bm_ms * bm_n * 1e-3) / 1e12, 2), "unit": "TFLOP/s"}
                        rb["adaln_gemm"]["frac"] = round(rb["adaln_gemm"]["achieved"] / MFMA_PEAK_TFLOPS, 4)
                    pmcb = load_profile_json("r06_pmc_traffic.json", [])  # the fmtb pass runs 16 stacked clips (2 880 rows)
                    if pmcb and batch_done == 16:
                        rb["traffic"] = pmcb.get("fmt_gemm_rb", {}).get("hbm_bytes_per_launch")
                    variants["roofline_batch"] = rb

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(pkg, torch, cfg, fmt_sd, dec_sd, pkg.weights.synth_feats(args.size, seed=1), cond, args.nfe - 1)

    if rank == 0:
        par = ("replicas x%d (one clip per GPU)" % world) if not one_clip else (
            "shard: one %d-frame clip, latent chain replicated, frames sharded x%d" % (T_total, world) if args.mode == "shard" else
            "window: one %d-frame clip, windows sharded x%d, boundary all_gather x%d round(s) re-solving %s, max seam change %.3e, "
            "max change of the hand-off frames at the next boundary %.3e"
            % (T_total, world, args.window_iters, ("the first %d window(s) of a rank" % args.window_resolve) if args.window_resolve > 0
               else "the rank's whole range", seam.get("seam_rel_change", 0.0), seam.get("seam_next_rel_change", 0.0)))
        out = {
            "metric": "512x512 frames/sec end-to-end audio->video @50 ODE steps",
            "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak",  # per-GPU work is fixed as N grows: one clip per GPU, or one N-times-longer clip
            "vs_baseline": None,
            "dtype": "%s+%s" % (args.fmt_dtype, args.dec_dtype) if args.fmt_dtype != args.dec_dtype else args.fmt_dtype,
            "data": "synthetic",
            "config": {"workload": ("configs[%d]: %.0f s audio + %dx%d portrait in HBM -> %d frames in pinned host memory; appearance "
                                    "encoder + wav2vec2 audio encoder + FMT sampling (%d Euler evaluations/window, CFG a=%.1f e=%.1f%s) "
                                    "+ decode + D2H"
                                    % (4 if args.dynamic_we else 1, args.seconds * (world if one_clip else 1), args.size, args.size,
                                       T_total, args.nfe - 1, a_cfg, e_cfg, ", dynamic per-window emotion" if args.dynamic_we else ""))
                                   + (", through InferenceAgent.infer_device (the product call)" if product else ""),
                       "nfe": args.nfe, "frames_per_clip": T_total, "fmt_dtype": args.fmt_dtype, "dec_dtype": args.dec_dtype,
                       "decode_batch": args.max_frames, "hip_graph": not args.no_graph, "parallelism": par},
        }
        out.update(extra)
        out["config"]["stage_overlap"] = os.environ.get("FLOAT_AMD_OVERLAP", "") or "off"
        out["frames_sha1"] = hashlib.sha1(host[::7].contiguous().numpy().tobytes()).hexdigest()[:16]  # bitwise identity of A/B runs
        if per_rank:
            out["ranks"] = per_rank  # one entry per rank: its stage times, its all_gather latency (diagnosis of an N > 1 run)
        out["host_threads_per_rank"] = host_threads if world > 1 else torch.get_num_threads()
        out["rccl_ranks"] = rccl_ranks  # ranks as counted by an RCCL all_reduce of ones (None: no RCCL communicator in this run)
        out["fp16_range_hits"] = sum(range_hits.values())
        if rccl_note:
            warnings.append(rccl_note)
        if bf16:
            out.update(bf16)
        out.update(variants)
        if roof:
            out["roofline"] = roof[0]
            out["roofline_secondary"] = roof[1]
            if len(roof) > 2:
                out["roofline_tertiary"] = roof[2]
        if cpu:
            out["cpu_baseline"] = cpu
        if warnings:
            out["warnings"] = warnings
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
