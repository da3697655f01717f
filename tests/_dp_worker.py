"""One rank of the data-parallel GPU test (tests/test_parallel_gpu.py): DenseGCM on the HIP path on
its shard of the global batch, GradBucket over the GNN (and the selector's parameters when it has
any), one all-reduce; gradients and outputs go to a file."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graph-conv-memory_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def build(selector, N, F, H, dev):
    from gcm import nn as G
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.temporal import TemporalBackedge
    from gcm.edge_selectors.learned import LearnedEdge
    torch.manual_seed(0)
    gnn = G.Sequential("x, adj, weights, B, N", [
        (G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
        (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(dev)
    sel = TemporalBackedge([1, 2]) if selector == "temporal" else LearnedEdge(F).to(dev)
    return DenseGCM(gnn, edge_selectors=sel, graph_size=N), gnn, sel


def run(mem, obs):
    hidden, outs = None, []
    for t in range(obs.shape[0]):
        mx, hidden = mem(obs[t], hidden)
        outs.append(mx)
    return torch.stack(outs)


def main():
    from gcm import parallel
    selector, out_path = sys.argv[1], sys.argv[2]
    rank, local_rank, world = parallel.init_from_env()
    dev = torch.device("cuda", local_rank)
    Bg, N, F, H, T = 8, 16, 8, 16, 12
    mem, gnn, sel = build(selector, N, F, H, dev)
    torch.manual_seed(1)
    obs = torch.rand(T, Bg, F)
    lo, hi = parallel.shard_bounds(Bg, rank, world)
    out = run(mem, obs[:, lo:hi].contiguous().to(dev))
    out.mean().backward()
    mods = [gnn] + ([sel] if any(True for _ in sel.parameters()) else [])
    bucket = parallel.GradBucket(*mods)
    bucket.all_reduce_mean((hi - lo) / Bg)
    mem.check_flags()
    torch.save({"out": out.detach().cpu(), "grads": [p.grad.cpu() for p in bucket.params],
                "n_params": len(bucket.params)}, f"{out_path}.{rank}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
