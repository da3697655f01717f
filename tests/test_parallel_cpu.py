"""world_size-2 gloo tests of the N>1 path (sharding + the single flat-bucket gradient
all-reduce).  The model under the collective is the CPU oracle: the product kernels cannot
run without a GPU, the DP plumbing can."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "graph-conv-memory_amd")):
        sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from gcm import parallel
    from oracle import dense as od

    parallel.init_from_env(backend="gloo")
    torch.manual_seed(0)
    Bg, N, F, H, T = 6, 8, 4, 5, 5
    gnn = od.canonical_gnn(F, H)                 # same seed -> replicated parameters
    obs = torch.rand(T, Bg, F)
    lo, hi = parallel.shard_bounds(Bg, rank, world)
    out, _ = od.dense_rollout(obs[:, lo:hi], None, gnn, graph_size=N,
                              edge_selectors=od.TemporalBackedge([1, 2]))
    out.mean().backward()
    bucket = parallel.GradBucket(gnn)
    bucket.all_reduce_mean((hi - lo) / Bg)
    # by value (numpy), not as shared-memory tensors: those need the sender alive at receive time
    q.put((rank, [p.grad.numpy().copy() for p in gnn.parameters()], out.detach().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_rollout_equals_global_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    got = [(r, [torch.from_numpy(g) for g in gs], torch.from_numpy(o)) for r, gs, o in got]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from oracle import dense as od
    torch.manual_seed(0)
    Bg, N, F, H, T = 6, 8, 4, 5, 5
    gnn = od.canonical_gnn(F, H)
    obs = torch.rand(T, Bg, F)
    out, _ = od.dense_rollout(obs, None, gnn, graph_size=N, edge_selectors=od.TemporalBackedge([1, 2]))
    out.mean().backward()
    # forward needs no communication: shard outputs concatenate to the global output
    torch.testing.assert_close(torch.cat([g[2] for g in got], dim=1), out.detach())
    # one all-reduce reproduces the global-batch gradient on every rank
    for r in range(world):
        for g, p in zip(got[r][1], gnn.parameters()):
            torch.testing.assert_close(g, p.grad, rtol=1e-5, atol=1e-7)


def test_shard_bounds_cover_batch():
    from gcm import parallel
    for total in (1, 7, 256, 2048):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
