"""The layered path at cfg2's shapes (B = 256, N = 128, F = H = 32): a user GNN the fused step does not cover - three
DenseGraphConv layers, and the canonical two with pooled=True - per-step loop fwd + bwd, per-kernel durations by the
in-process profiler (bench.profile_kernels).  Also used by bench.py's `layered` variant."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-conv-memory_amd")]
import torch  # noqa: E402

import bench  # noqa: E402


def build(kind, device, c):
    from gcm import nn as G
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.temporal import TemporalBackedge
    F, H, N = c["F"], c["H"], c["N"]
    torch.manual_seed(0)
    if kind == "three_layer":
        gnn = G.Sequential("x, adj, weights, B, N", [
            (G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
            (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh(),
            (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(device)
        mem = DenseGCM(gnn, edge_selectors=TemporalBackedge(bench.HOPS), graph_size=N)
    else:       # pooled: the GNN's whole output is the belief (gcm.py:309-311); here mean over the nodes
        class Pool(torch.nn.Module):
            def forward(self, x):
                return x.mean(dim=1)
        gnn = G.Sequential("x, adj, weights, B, N", [
            (G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
            (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh(), (Pool(), "x -> x")]).to(device)
        mem = DenseGCM(gnn, edge_selectors=TemporalBackedge(bench.HOPS), graph_size=N, pooled=True)
    return mem, gnn


def run(kind, T=16, reps=2, device=None):
    device = device or torch.device("cuda", 0)
    c = dict(bench.CONFIGS["cfg2"])
    mem, gnn = build(kind, device, c)
    obs = torch.rand(T, c["B"], c["F"], device=device)

    def call():
        bench.rollout(mem, obs)
        gnn.zero_grad(set_to_none=True)

    for _ in range(2):
        call()
    prof = bench.profile_kernels(call, reps=reps)
    return prof, T, c


if __name__ == "__main__":
    for kind in ("three_layer", "pooled"):
        prof, T, c = run(kind)
        rows, total = bench.kernel_table(prof, top=14)
        print("== %s: %.1f us of kernels per step (T = %d steps per rollout)" % (kind, total / T, T))
        for r in rows:
            print("  %-100s n/step=%5.2f avg=%8.2f us share=%.3f" % (r["kernel"][:100], r["launches_per_step"] / T,
                                                                    r["avg_us"], r["share_of_gpu_time"]))
