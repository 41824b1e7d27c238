"""The multi-GPU host logic on the HIP operator and on RCCL, as far as ONE GPU allows:
  * distributed.sample_range on float_fmt_sample_begin_range / _next (graph replay per window, hand-off on the device) is the
    sequential chain bit for bit - whole clip, and a middle window range started from the chain's own history;
  * a one-rank `nccl` (= RCCL) process group on this GPU runs the exact collective calls of the N > 1 modes on device
    tensors: the boundary all_gather of sample_window_parallel, the r_d broadcast of the frame shard, bench.py's all_reduce.
The N = 2 / 3 / 4 semantics are covered on gloo in tests/test_distributed_cpu.py."""
import os
import subprocess
import sys

import pytest
import torch

from tests.util import ROOT, load_pkg, rel_l2, sample_inputs

pkg = load_pkg()
D = pkg.distributed
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dynamic", [False, True])
def test_native_window_range_is_the_sequential_chain(dynamic):
    cfg = pkg.config.FmtConfig()
    L, P = cfg.num_frames_for_clip, cfg.num_prev_frames
    sd = pkg.weights.synth_fmt_state(cfg, seed=3)
    fmt = pkg.fmt.FlowMatchingTransformerHIP(sd, cfg, "cuda:0", "fp16")
    T = 5 * L - 13  # five windows, the last one replicate-padded and trimmed
    inp = {k: v.cuda() for k, v in sample_inputs(cfg, 40, T, dynamic).items()}
    wa, r_s, we, noise = inp["wa"], inp["r_s"], inp["we"], inp["noise"]
    ref = fmt.sample(r_s, wa, we, noise, 6, 2.0, 1.0, 1.0)
    xs, _ = D.sample_range(fmt, cfg, r_s, wa, we, noise, 0, 5, 6, 2.0, 1.0, 1.0)
    assert xs.shape == ref.shape and torch.equal(xs, ref)
    # windows [2, 5) from the history the chain itself had after window 1
    pad = lambda a, k: D._pad_rep(a[:, k * L:(k + 1) * L], L)  # noqa: E731
    hist = (ref[:, 2 * L - P:2 * L], pad(wa, 1)[:, -P:], pad(we, 1)[:, -P:] if dynamic else torch.zeros(1, P, cfg.dim_e).cuda())
    part, tail = D.sample_range(fmt, cfg, r_s, wa, we, noise, 2, 5, 6, 2.0, 1.0, 1.0, hist)
    assert torch.equal(part, ref[:, 2 * L:])
    # and a middle range: its tail is what the next window would start from
    mid, tail = D.sample_range(fmt, cfg, r_s, wa, we, noise, 2, 4, 6, 2.0, 1.0, 1.0, hist)
    assert torch.equal(mid, ref[:, 2 * L:4 * L]) and torch.equal(tail[0], ref[:, 4 * L - P:4 * L])
    with pytest.raises(ValueError, match="window range"):
        pkg.fmt.WindowSampler(fmt, r_s, wa, we, noise, 6, 2.0, 1.0, 1.0, windows=(3, 9))

    # the host loop (one sample_chunk call per window: what an object without a HIP handle gets) obeys the same contract on
    # the clip's trimmed last window: same trimmed rows, zero prev_x, same conditioning tails
    class HostLoop:
        sample_chunk = staticmethod(fmt.sample_chunk)
    h_part, h_tail = D.sample_range(HostLoop(), cfg, r_s, wa, we, noise, 2, 5, 6, 2.0, 1.0, 1.0, hist)
    n_part, n_tail = D.sample_range(fmt, cfg, r_s, wa, we, noise, 2, 5, 6, 2.0, 1.0, 1.0, hist)
    assert h_part.shape == n_part.shape and rel_l2(h_part, n_part) < 1e-3
    assert all(a.shape == b.shape for a, b in zip(h_tail, n_tail))
    assert float(h_tail[0].abs().max()) == 0.0 and float(n_tail[0].abs().max()) == 0.0 and torch.equal(h_tail[1], n_tail[1])


_RCCL_SCRIPT = r'''
import os, sys, socket
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from tests.util import load_pkg, sample_inputs
pkg = load_pkg(); D = pkg.distributed
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%%d" %% port, rank=0, world_size=1, device_id=dev)
cfg = pkg.config.small_fmt_config()
fmt = pkg.fmt.FlowMatchingTransformerHIP(pkg.weights.synth_fmt_state(cfg, seed=4), cfg, dev, "fp16")
T = 170
inp = {k: v.to(dev) for k, v in sample_inputs(cfg, 7, T, False).items()}
ref = fmt.sample(inp["r_s"], inp["wa"], inp["we"], inp["noise"], 4, 2.0, 1.0, 1.0)
loc, (t0, t1), rep = D.sample_window_parallel(fmt, cfg, inp["r_s"], inp["wa"], inp["we"], inp["noise"], 4, 2.0, 1.0, 1.0,
                                              iters=1, resolve_chunks=1, exchange_at_world_1=True)   # all_gather on RCCL
assert (t0, t1) == (0, T) and torch.equal(loc, ref), "window-parallel at world 1 must be the sequential chain"
r_d = D.broadcast_latents(ref.clone(), 0, None, at_world_1=True)                                      # broadcast on RCCL
assert torch.equal(r_d, ref)
t = torch.tensor([1.5], device=dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)    # bench.py's reduction
ones = torch.ones(1, device=dev); dist.all_reduce(ones)
dist.barrier(); torch.cuda.synchronize()
assert float(t.item()) == 1.5 and int(ones.item()) == dist.get_world_size() == 1
print("RCCL_OK backend=%%s" %% dist.get_backend())
dist.destroy_process_group()
'''


def test_rccl_collectives_of_the_multi_gpu_modes_at_world_1():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT % {"root": ROOT}], cwd=ROOT, env=env, capture_output=True, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and "RCCL_OK backend=nccl" in out, (out[-1000:], p.stderr.decode()[-3000:])
