"""bench.py as the driver runs it: one JSON line with the contract's fields, `value` = the end-to-end rate, and `--gpus N`
without a launcher starting N ranks itself (here two gloo ranks sharing the one GPU of the test box)."""
import json
import os
import subprocess
import sys

import pytest

from tests.util import ROOT

pytestmark = pytest.mark.gpu
SHORT = ["--seconds", "2", "--nfe", "6", "--steps", "1", "--warmup", "1", "--no-bf16"]


def _run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), lines  # ONE JSON line and nothing else on stdout (RCCL's banner goes to stderr)
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    r = _run(SHORT + ["--batches", "4,8"])  # every leg of the default run: roofline pass, CPU baseline, speech-to-emotion, nfe / host-input variants
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "stage_ms", "value_s2e", "value_nfe5", "value_from_host_inputs",
              "host_inputs_ms", "rccl_ranks", "fp16_range_hits", "value_batch4", "value_batch8", "roofline_batch", "host_threads_per_rank"):
        assert k in r, k
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "frames/s" and "sample" in cb
    assert r["rccl_ranks"] == 1 and r["fp16_range_hits"] == 0 and "speech_emotion" in r["stage_ms"]
    assert r["value_s2e"] > 0 and r["value_from_host_inputs"] > 0 and r["value_nfe5"] > 0
    assert r["n_gpus"] == 1 and r["unit"] == "frames/s" and r["value"] > 0 and r["vs_baseline"] is None
    assert "workload" in r["config"] and "pinned host memory" in r["config"]["workload"]
    # value is the end-to-end rate: frames / wall of the timed steps
    assert abs(r["value"] - 50 * r["steps"] / (r["ms_per_step"] * r["steps"] * 1e-3)) < 0.01 * r["value"]
    clk = r["device"].get("sclk_MHz_under_step")  # rocm-smi from a side thread, outside the timed region; absent without rocm-smi
    assert clk is None or (500 < clk[0] <= clk[1] < 3000), clk
    rb = r["roofline_batch"]
    assert rb["bound"] == "mfma" and rb["batch"] == 8 and rb["launches"] > 0 and abs(rb["frac"] - rb["achieved"] / rb["peak"]) < 1e-3
    ro = r["roofline"]
    assert ro["bound"] in ("hbm", "mfma") and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3 and "traffic" in ro
    # the peaks measured on this box travel with the spec-sheet ones: a streaming read of an MI355X sits between 2 and
    # 8 TB/s, a register-operand fp16 MFMA loop between 1 and 2.6 PFLOP/s; the spec-sheet figures stay `peak`
    d = r["device"]
    assert 2000 < d["hbm_read_GBps"] <= 8000 and 2000 < d["hbm_copy_GBps"] <= 8000 and d["compute_units"] >= 64
    assert 1000 < max(d["mfma_f16_16x16x32_TFLOPs"], d["mfma_f16_32x32x16_TFLOPs"]) <= 2600
    assert ro["peak"] in (8000.0, 2500.0) and ro["peak_measured"] > 0 and abs(ro["frac_measured"] - ro["achieved"] / ro["peak_measured"]) < 1e-3


@pytest.mark.parametrize("mode", ["replicas", "window", "shard"])
def test_gpus_2_launches_its_own_ranks(mode):
    r = _run(SHORT + ["--gpus", "2", "--mode", mode, "--no-roofline", "--no-extras", "--no-cpu-baseline"], {"FLOAT_BENCH_BACKEND": "gloo"})
    assert r["n_gpus"] == 2
    assert r["config"]["frames_per_clip"] == (50 if mode == "replicas" else 100)
    if mode == "window":
        assert "seam change" in r["config"]["parallelism"]


def test_torchrun_launch_at_world_1():
    """The driver starts N > 1 through `python -m torch.distributed.run ... bench.py --gpus N`: the same launch with one rank (the
    agent owns the rendezvous store; a private tcp:// rendezvous inside the rank would wait for ever) comes up on RCCL."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SHORT + [
               "--no-cpu-baseline", "--no-roofline", "--no-extras", "--no-s2e", "--no-variants"]
    p = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, FLOAT_BENCH_WATCHDOG="400"), capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    r = json.loads(lines[-1])
    assert r["n_gpus"] == 1 and r["rccl_ranks"] == 1 and r["value"] > 0


def test_torchrun_launch_at_world_2_gloo():
    """The driver's own N > 1 launch line - `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 2` - with two gloo ranks sharing this GPU: the roll call runs against the agent's store, one
    line comes back with n_gpus = 2 and every rank's stage times."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SHORT + [
               "--no-cpu-baseline", "--no-roofline", "--no-extras", "--no-s2e", "--no-variants"]
    p = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, FLOAT_BENCH_WATCHDOG="500", FLOAT_BENCH_BACKEND="gloo"), capture_output=True, timeout=800)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] > 0 and [e["rank"] for e in r["ranks"]] == [0, 1]
    assert all(e["boundary_all_gather_us"] > 0 for e in r["ranks"])
