"""N > 1 with the PRODUCT kernels under the collective: two ranks share cuda:0 (GCM_SINGLE_DEVICE=1)
over gloo, each runs DenseGCM on the HIP path on its shard, GradBucket does the one all-reduce;
result = the single-process global batch.  Also `python bench.py --gpus 2` spawning its own ranks.
Needs an MI355X (one is enough)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_dp_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(selector, world, path):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GCM_SINGLE_DEVICE="1", GCM_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, WORKER, selector, path], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    return [torch.load(f"{path}.{r}") for r in range(world)]


def test_sharded_product_rollout_equals_global_batch(tmp_path):
    got = _launch("temporal", 2, str(tmp_path / "dp"))
    one = _launch("temporal", 1, str(tmp_path / "one"))[0]      # the global batch in one process
    torch.testing.assert_close(torch.cat([g["out"] for g in got], dim=1), one["out"], rtol=1e-6, atol=1e-7)
    for r in range(2):
        assert got[r]["n_params"] == one["n_params"] == 6
        for g, w in zip(got[r]["grads"], one["grads"]):
            torch.testing.assert_close(g, w, rtol=1e-5, atol=1e-6 * float(w.abs().max()))


def test_bucket_covers_selector_parameters(tmp_path):
    """cfg5's shape of job: LearnedEdge's edge network is part of the all-reduced bucket and both
    ranks end with identical gradients (the sampled edges differ per rank: no comparison with a
    global batch here)."""
    got = _launch("learned", 2, str(tmp_path / "le"))
    assert got[0]["n_params"] == 6 + 10
    for a, b in zip(got[0]["grads"], got[1]["grads"]):
        assert torch.equal(a, b)
        assert torch.isfinite(a).all()
    assert any(float(g.abs().max()) > 0 for g in got[0]["grads"][6:])


def test_bench_spawns_its_own_ranks():
    env = dict(os.environ, GCM_SINGLE_DEVICE="1", GCM_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--T", "16", "--no-cpu-baseline"], env=env, capture_output=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
