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


def _launch(selector, world, path, mode=""):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GCM_SINGLE_DEVICE="1", GCM_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, WORKER, selector, path, mode], env=env))
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


def test_bucket_aliased_gradients(tmp_path):
    """GradBucket(alias_grads=True): after the first call the parameters' .grad tensors are slices of
    the flat bucket, the next backward accumulates into it in place and all_reduce_mean is the
    collective (+ the weight) alone; same gradients as the copying form."""
    got = _launch("temporal", 2, str(tmp_path / "al"), "alias")
    ref = _launch("temporal", 2, str(tmp_path / "cp"))
    for r in range(2):
        for g, w in zip(got[r]["grads"], ref[r]["grads"]):
            torch.testing.assert_close(g, w, rtol=1e-6, atol=1e-8)


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


@pytest.mark.parametrize("mode", ["", "obs_grad", "big"])
def test_sharded_euclidean_edge_equals_global_batch(tmp_path, mode):
    """SURVEY 8e "Exception": EuclideanEdge's mean runs over every graph of the batch (distance.py:48-49).
    EuclideanEdge(shard_group=...) all-gathers the ranks' current nodes ahead of the distance kernel:
    two ranks with 4 graphs each == one process with the 8 graphs (edge decisions bit exact, beliefs,
    gradients) - on the live-row path and (obs with gradient) on the fused one."""
    got = _launch("euclid", 2, str(tmp_path / "eu"), mode)
    one = _launch("euclid", 1, str(tmp_path / "one"), mode)[0]
    assert float(one["adj"].sum()) > one["adj"].shape[0] * 4     # the selector did connect nodes
    assert torch.equal(torch.cat([g["adj"] for g in got], dim=0), one["adj"])
    torch.testing.assert_close(torch.cat([g["out"] for g in got], dim=1), one["out"], rtol=1e-6, atol=1e-7)
    for r in range(2):
        for g, w in zip(got[r]["grads"], one["grads"]):
            torch.testing.assert_close(g, w, rtol=1e-5, atol=1e-6 * float(w.abs().max()))


def test_bench_cfg5_two_ranks():
    """bench.py --config cfg5 --gpus 2 (LearnedEdge, GradBucket over GNN + edge network, per-rank device
    RNG seeds) through the launcherless spawn, two ranks sharing one GPU over gloo."""
    env = dict(os.environ, GCM_SINGLE_DEVICE="1", GCM_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    argv = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cfg5", "--gpus", "2",
            "--steps", "2", "--warmup", "1", "--T", "12", "--repeats", "1", "--no-cpu-baseline"]
    p = subprocess.run(argv, env=env, capture_output=True, timeout=600)
    if p.returncode != 0:
        # (this child failed once inside a full-suite run: bench.py then still traced its T = 2N graph in process, and
        #  kineto's stop_trace segfaulted there in 5 of 25 cfg5 runs - that trace is gone from bench.py; the retry stays
        #  for whatever else a two-rank child on one GPU may meet.  The first failure is printed)
        print(p.stderr.decode()[-2000:])
        p = subprocess.run(argv, env=env, capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "cfg5" in line["config"]["workload"]
    assert line["roofline"]["frac"] > 0 and line["value_spread"]["blocks"] == 2


def test_bench_spawns_its_own_ranks():
    env = dict(os.environ, GCM_SINGLE_DEVICE="1", GCM_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--T", "16", "--no-cpu-baseline"], env=env, capture_output=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"


# --------------------------------------------------------------------------------------------------
# RCCL itself (backend "nccl"): one rank per GPU.  On a one-GPU box: a one-rank communicator (set-up, the
# collectives' code paths, the bucket, the gathered rows); with >= 2 GPUs visible also the product kernels under
# the real collective and bench.py's sharded configuration - the first thing an 8-GPU driver run would hit.
# --------------------------------------------------------------------------------------------------
def _launch_rccl(world, argv, timeout=600):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("GCM_SINGLE_DEVICE", None)
        env.pop("GCM_DIST_BACKEND", None)
        procs.append(subprocess.Popen([sys.executable] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate(timeout=timeout)
    assert procs[0].returncode == 0
    for p in procs[1:]:
        assert p.wait(timeout=timeout) == 0
    return out.decode()


@pytest.mark.parametrize("world", [1, 2])
def test_rccl_collectives(world):
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    out = _launch_rccl(world, [os.path.join(ROOT, "tests", "_rccl_worker.py")])
    info = json.loads(out.strip().splitlines()[-1])
    assert info["world"] == world and len(info["rccl"]) >= 2


def test_sharded_product_rollout_over_rccl(tmp_path):
    """two ranks, two GPUs, backend nccl: the product kernels under the real all-reduce == the global batch"""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    path = str(tmp_path / "rccl")
    _launch_rccl(2, [WORKER, "temporal", path, ""])
    got = [torch.load(f"{path}.{r}") for r in range(2)]
    one = _launch("temporal", 1, str(tmp_path / "one"))[0]
    torch.testing.assert_close(torch.cat([g["out"] for g in got], dim=1), one["out"], rtol=1e-6, atol=1e-7)
    for r in range(2):
        for g, w in zip(got[r]["grads"], one["grads"]):
            torch.testing.assert_close(g, w, rtol=1e-5, atol=1e-6 * float(w.abs().max()))
    path = str(tmp_path / "rccl_eu")
    _launch_rccl(2, [WORKER, "euclid", path, "big"])          # all_gather_into_tensor ahead of the distance kernel
    got = [torch.load(f"{path}.{r}") for r in range(2)]
    one = _launch("euclid", 1, str(tmp_path / "one_eu"), "big")[0]
    assert torch.equal(torch.cat([g["adj"] for g in got], dim=0), one["adj"])


def test_bench_cfg5_two_gpus_over_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "GCM_SINGLE_DEVICE", "GCM_DIST_BACKEND"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cfg5", "--gpus", "2",
                        "--steps", "2", "--warmup", "1", "--T", "12", "--repeats", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["runtime"]["world_size_seen"] == 2 and line["runtime"]["rccl_version"]
