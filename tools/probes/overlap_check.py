"""Which part of the overlapped hand-over differs from the sequential order at full size (512 px, 250 frames, 50 evaluations)?
r_d and per-window frames of generate_to_host_overlap against generate_to_host, per mode, several repetitions."""
import sys, time
import torch
sys.path.insert(0, ".")
from tests.util import load_pkg
pkg = load_pkg()
cfg = pkg.config.FmtConfig()
hp = pkg.pipeline.FloatHotPath(pkg.weights.synth_fmt_state(cfg, seed=1), pkg.weights.synth_decoder_state(512, seed=1), cfg, "cuda:0", 512, max_frames=32, use_graph=int(__import__("os").environ.get("OVC_GRAPH", "2")))
feats = pkg.weights.synth_feats(512, seed=1)
T = 250
c = pkg.pipeline.synth_conditions(cfg, T, seed=0, device="cuda:0")
noise = pkg.fmt.draw_noise(5, 1, cfg, 15).to("cuda:0")
a = (c["r_s"], c["wa"], c["we"], c["s_r"])
nfe = int(sys.argv[1]) if len(sys.argv) > 1 else 51
print("torch.cuda.Stream.priority_range() =", torch.cuda.Stream.priority_range())
seq, rd = hp.generate_to_host(*a, feats, nfe, noise=noise, return_rd=True)
torch.cuda.synchronize()
seq, rd = seq.clone(), rd.clone()
modes = [m for m in sys.argv[2:] if m not in ("solo", "torchload", "events", "waits")]
for mode in modes or ([] if ("solo" in sys.argv or "torchload" in sys.argv) else ["prio", "plain", "cu:64"]):
    for rep in range(3):
        t0 = time.perf_counter()
        o, r = hp.generate_to_host_overlap(*a, nfe, noise=noise, mode=mode, return_rd=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        bad_rd = [k for k in range(5) if not torch.equal(r[0, k * 50:(k + 1) * 50], rd[0, k * 50:(k + 1) * 50])]
        bad_fr = [k for k in range(5) if not torch.equal(o[k * 50:(k + 1) * 50], seq[k * 50:(k + 1) * 50])]
        nbad = [int((o[k * 50:(k + 1) * 50] != seq[k * 50:(k + 1) * 50]).reshape(50, -1).any(1).sum()) for k in range(5)]
        print(mode, rep, "%.1f ms" % ms, "r_d windows differing:", bad_rd, "frame windows differing:", bad_fr, "frames per window:", nbad, flush=True)

# The chain ALONE on a high-priority stream (no decoder anywhere): does the priority by itself change r_d?  (FLOAT_FMT_PRIO_GUARD=0
# lets run_mod_all keep the persistent adaLN kernel on that stream - the configuration that failed beside the decoder.)
if "solo" in sys.argv:
    lo, hi = torch.cuda.Stream.priority_range()
    s_hi = torch.cuda.Stream("cuda:0", priority=hi)
    for rep in range(3):
        torch.cuda.synchronize()
        evs = []
        with torch.cuda.stream(s_hi):
            ws = pkg.fmt.WindowSampler(hp.fmt, *a[:3], noise, nfe, 2.0, 1.0, 1.0)
            while ws.left > 0:
                ws.next()
                if "events" in sys.argv:  # what the overlapped pipeline does after every window: an event for the decoder's stream
                    ev = torch.cuda.Event()
                    ev.record(s_hi)
                    evs.append(ev)
                    if "waits" in sys.argv:
                        s_other = globals().setdefault("_s_other", torch.cuda.Stream("cuda:0"))
                        s_other.wait_event(ev)
        torch.cuda.synchronize()
        bad = [k for k in range(5) if not torch.equal(ws.r_d[0, k * 50:(k + 1) * 50], rd[0, k * 50:(k + 1) * 50])]
        print("solo chain on a priority %d stream%s, rep %d: r_d windows differing: %s" % (hi, " + event records" if evs else "", rep, bad), flush=True)

# The chain on the high-priority stream beside ANOTHER workload than the decoder on a default-priority stream (plain torch
# elementwise kernels over 1 GiB): is it the decoder, or any concurrent queue?
if "torchload" in sys.argv:
    lo, hi = torch.cuda.Stream.priority_range()
    s_hi, s_lo = torch.cuda.Stream("cuda:0", priority=hi), torch.cuda.Stream("cuda:0", priority=0)
    big = torch.zeros(1 << 28, device="cuda:0")
    for rep in range(3):
        torch.cuda.synchronize()
        with torch.cuda.stream(s_hi):
            ws = pkg.fmt.WindowSampler(hp.fmt, *a[:3], noise, nfe, 2.0, 1.0, 1.0)
        k = 0
        while ws.left > 0:
            with torch.cuda.stream(s_hi):
                ws.next()
            if k >= 1:
                with torch.cuda.stream(s_lo):
                    for _ in range(12):
                        big.add_(1.0)
            k += 1
        torch.cuda.synchronize()
        bad = [k for k in range(5) if not torch.equal(ws.r_d[0, k * 50:(k + 1) * 50], rd[0, k * 50:(k + 1) * 50])]
        print("chain on priority %d beside torch elementwise kernels, rep %d: r_d windows differing: %s" % (hi, rep, bad), flush=True)
