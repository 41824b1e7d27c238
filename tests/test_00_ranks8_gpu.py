"""`bench.py --gpus 8` as far as ONE GPU can rehearse it: eight gloo ranks on this GPU.

This module sorts FIRST in the suite on purpose and conftest.py does not create a HIP context while collecting: a GPU hosts eight
compute processes at a time (the driver keeps eight compute VMIDs); a ninth process with a context - the pytest process itself
once any in-process GPU test has run - makes the scheduler swap whole processes in and out, and a 60-ms step of the eight ranks
was measured at 100 ms beside an idle ninth context and at 6.5 s beside this process after tests/test_aud_gpu.py.  On an 8-GPU
node every rank has a GPU to itself."""
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
    assert len(lines) == 1 and lines[0].startswith("{"), lines
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["replicas", "window"])
def test_gpus_8_ranks_share_the_host(mode):
    """What one GPU can say about `--gpus 8`: eight gloo ranks on this GPU (8 x 6.5 GB), each with cores / 8 host threads
    (bench.rank_threads).  One result line with n_gpus = 8, the whole launch well inside the driver's budget, and the eight
    ranks together - time-slicing ONE GPU - deliver at least 1 / 1.3 of what one rank alone delivers: host work (weight
    synthesis and packing, noise draws, a few thousand launches per clip) of eight ranks does not collapse on shared cores."""
    import time
    common = ["--no-roofline", "--no-extras", "--no-cpu-baseline", "--no-s2e", "--no-variants", "--steps", "8"]
    env = {"FLOAT_BENCH_BACKEND": "gloo"}
    one = _run(SHORT + common, env)
    limit = 1.3 if mode == "replicas" else 2.0
    for attempt in range(2):  # 8 steps of 11 ms: one more try if a hiccup of the box (the suite has run for minutes) spoilt the first
        t0 = time.time()
        r = _run(SHORT + common + ["--gpus", "8", "--mode", mode], env)
        wall = time.time() - t0
        if r["value"] >= one["value"] / limit:
            break
    assert r["n_gpus"] == 8 and r["host_threads_per_rank"] >= 1
    # the line carries every rank's own stage times and its boundary all_gather latency (what the first 8-GPU run is read by)
    assert [e["rank"] for e in r["ranks"]] == list(range(8))
    for e in r["ranks"]:
        assert e["frames"] == 50 and e["boundary_all_gather_us"] > 0
        assert set(e["stage_ms"]) == {"encoders", "chain", "decode_and_hand_over"} and all(v > 0 for v in e["stage_ms"].values())
    assert wall < 300, wall
    assert r["config"]["frames_per_clip"] == (50 if mode == "replicas" else 400)
    # 8 ranks x 50 frames per step on one GPU against 50 frames per step of one rank.  In window mode a rank samples its one
    # window twice (from zero history, then the seam re-solve behind the all_gather - on gloo a host round trip): its floor is
    # ~1.2x the work of the one-rank step at these sizes, hence the wider limit there
    assert r["value"] >= one["value"] / limit, (r["value"], one["value"], r["ms_per_step"], one["ms_per_step"])
