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
    if selector == "euclid":       # cross-batch mean: sharded ranks all-gather their current nodes
        from gcm.edge_selectors.distance import EuclideanEdge
        sel = EuclideanEdge(2.0, shard_group=True if dist.is_initialized() else None)
    else:
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
    mode = sys.argv[3] if len(sys.argv) > 3 else ""
    Bg, N, F, H, T = (64 if "big" in mode else 8), 16, 8, 16, 12     # big: >= 32 current rows, the MFMA kernel
    mem, gnn, sel = build(selector, N, F, H, dev)
    torch.manual_seed(1)
    obs = torch.rand(T, Bg, F)
    if selector == "euclid":       # clustered: a wide margin around the threshold
        centres = 4.0 * torch.randn(3, F)
        obs = centres[torch.arange(T) % 3][:, None, :] + 0.05 * torch.randn(T, Bg, F)
    lo, hi = parallel.shard_bounds(Bg, rank, world)
    x = obs[:, lo:hi].contiguous().to(dev)
    if "obs_grad" in mode:         # a gradient w.r.t. the observations: the fused (non live-row) kernels
        x.requires_grad_(True)
    hidden, outs = None, []
    for t in range(T):
        mx, hidden = mem(x[t], hidden)
        outs.append(mx)
    out = torch.stack(outs)
    adj_final = hidden[1].detach().cpu()
    out.mean().backward()
    mods = [gnn] + ([sel] if any(True for _ in sel.parameters()) else [])
    bucket = parallel.GradBucket(*mods, alias_grads="alias" in mode)
    bucket.all_reduce_mean((hi - lo) / Bg)
    if "alias" in mode:            # a second pass accumulates straight into the bucket: no gather, no copy back
        first = [p.grad.clone() for p in bucket.params]
        bucket.zero()
        hidden, outs = None, []
        for t in range(T):
            mx, hidden = mem(x[t], hidden)
            outs.append(mx)
        torch.stack(outs).mean().backward()
        bucket.all_reduce_mean((hi - lo) / Bg)
        assert bucket.launches == (1 if (hi - lo) != Bg else 0), bucket.launches    # (gloo: the weight is one launch)
        for a, b in zip(first, bucket.params):
            torch.testing.assert_close(b.grad, a, rtol=1e-6, atol=1e-8)
    mem.check_flags()
    torch.save({"out": out.detach().cpu(), "grads": [p.grad.cpu() for p in bucket.params],
                "n_params": len(bucket.params), "adj": adj_final}, f"{out_path}.{rank}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
