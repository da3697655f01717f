#!/usr/bin/env python3
"""Throughput of the non-headline BASELINE.json configs on one MI355X (they are parity-test
cases, not bench lines): cfg1, cfg3 (EuclideanEdge), cfg4 (SparseGCM), cfg5 per-GPU share
(LearnedEdge).  Prints one JSON object per config.  Dev / reporting tool."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch  # noqa: E402
from gcm import nn as G  # noqa: E402
from gcm.gcm import DenseGCM  # noqa: E402
from gcm.sparse_gcm import SparseGCM  # noqa: E402
from gcm.edge_selectors.temporal import TemporalBackedge  # noqa: E402
from gcm.edge_selectors.distance import EuclideanEdge  # noqa: E402
from gcm.edge_selectors.learned import LearnedEdge  # noqa: E402
from gcm.sparse_edge_selectors.temporal import TemporalEdge  # noqa: E402

dev = "cuda:0"
ONLY = sys.argv[1] if len(sys.argv) > 1 else ""   # substring filter on the config name


def dense_gnn(F, H):
    return G.Sequential("x, adj, weights, B, N", [
        (G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
        (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(dev)


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def run_dense(name, B, N, F, H, T, sel, obs, iters=5, pre=None, **kw):
    if ONLY not in name:
        return
    torch.manual_seed(0)
    gnn = dense_gnn(pre.out_features if pre is not None else F, H)
    mem = DenseGCM(gnn, edge_selectors=sel, graph_size=N, preprocessor=pre, **kw)
    params = list(gnn.parameters()) + (list(sel.parameters()) if sel is not None else []) + \
        (list(pre.parameters()) if pre is not None else [])

    def loop():
        hid, outs = None, []
        for t in range(T):
            mx, hid = mem(obs[t], hid)
            outs.append(mx)
        torch.stack(outs).mean().backward()
        for p in params:
            p.grad = None

    def roll():
        out, _ = mem.rollout(obs)
        out.mean().backward()
        for p in params:
            p.grad = None

    dt, dr = timeit(loop, iters), timeit(roll, iters)
    # the same per-step loop + backward captured once in a HIP graph and replayed
    dg = None
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            loop()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loop()
        dg = timeit(graph.replay, 4 * iters)
        mem.check_flags()
    except Exception as e:     # a path with host syncs cannot be captured
        dg = None
        print("graph capture failed:", type(e).__name__, str(e)[:200], file=sys.stderr)
        torch.cuda.synchronize()
    print(json.dumps({"config": name, "B": B, "N": N, "F": F, "H": H, "T": T,
                      "per_step_api_states_per_s": B * T / dt, "rollout_api_states_per_s": B * T / dr,
                      "per_step_api_graph_replay_states_per_s": (B * T / dg) if dg else None,
                      "ms_per_rollout": dt * 1e3, "ms_per_rollout_rollout_api": dr * 1e3,
                      "ms_per_rollout_graph_replay": dg * 1e3 if dg else None,
                      "fused": mem._structure() is not None}))


torch.manual_seed(0)
# cfg1
run_dense("cfg1 TemporalBackedge([1])", 4, 32, 8, 32, 32, TemporalBackedge([1]), torch.rand(32, 4, 8, device=dev))
# cfg3: clustered observations
B, N, F, H, T = 256, 128, 64, 32, 128
c = 4 * torch.randn(8, F)
obs3 = (c[torch.arange(T) % 8][:, None, :] + 0.05 * torch.randn(T, B, F)).to(dev)
run_dense("cfg3 EuclideanEdge(2.0) cross-batch", B, N, F, H, T, EuclideanEdge(2.0), obs3, iters=3)
run_dense("cfg3 EuclideanEdge(2.0) cross-batch, donated state", B, N, F, H, T, EuclideanEdge(2.0), obs3, iters=3,
          donate_state=True)
# cfg2's shape behind the RLlib model's default preprocessor (ray_gcm.py:117): folded into the step
B, N, F, H, T = 256, 128, 32, 32, 128
run_dense("cfg2 + Linear(32,32) preprocessor (folded), donated", B, N, F, H, T, TemporalBackedge([1, 2, 4]),
          torch.rand(T, B, F, device=dev), pre=torch.nn.Linear(F, 32).to(dev), donate_state=True)
# cfg5 per-GPU share
B, N, F, H, T = 256, 128, 32, 32, 64
run_dense("cfg5/GPU LearnedEdge(32)", B, N, F, H, T, LearnedEdge(32).to(dev), torch.rand(T, B, F, device=dev), iters=10)

# cfg4 sparse
if not ('cfg4'.startswith(ONLY) or ONLY.startswith('cfg4')):
    sys.exit(0)
B, N, F, H = 512, 512, 32, 32
torch.manual_seed(0)
g = G.Sequential("x, edges, weights", [(G.GraphConv(F, H), "x, edges, weights -> x"), torch.nn.Tanh(),
                                       (G.GraphConv(H, H), "x, edges, weights -> x"), torch.nn.Tanh()]).to(dev)
mem = SparseGCM(g, edge_selectors=TemporalEdge([1]), graph_size=N)
x = torch.rand(B, N, F, device=dev)
taus = torch.full((B,), N, dtype=torch.long, device=dev)


def oneshot():
    out, _ = mem(x, taus, None)
    out.mean().backward()
    for p in g.parameters():
        p.grad = None


dt = timeit(oneshot, 5)
print(json.dumps({"config": "cfg4 SparseGCM TemporalEdge([1]) one-shot taus=512", "B": B, "N": N, "F": F,
                  "states_per_s": B * N / dt, "ms_per_call": dt * 1e3}))
if 'oneshot' in ONLY:
    sys.exit(0)
one = torch.ones(B, dtype=torch.long, device=dev)


def stepwise(steps=64):
    hid, outs = None, []
    for t in range(steps):
        o, hid = mem(x[:, t:t + 1], one, hid)
        outs.append(o)
    torch.cat(outs, 1).mean().backward()
    for p in g.parameters():
        p.grad = None


dt = timeit(stepwise, 2, warm=1)
print(json.dumps({"config": "cfg4 SparseGCM stepwise taus=1 x64", "B": B, "N": N, "F": F,
                  "states_per_s": B * 64 / dt, "ms_per_64_steps": dt * 1e3}))
