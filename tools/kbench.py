#!/usr/bin/env python3
"""Micro-benchmark of individual C-ABI kernels at the cfg2 shapes (B=256, N=128, F=H=32).
Times back-to-back launches with events on the launch stream.  Dev tool (not part of tests)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch  # noqa: E402
from gcm import _hip, _ops  # noqa: E402

B, N, F, H = (int(v) for v in (sys.argv[1:5] + [256, 128, 32, 32][len(sys.argv) - 1:]))
dev = "cuda:0"
lib = _hip.lib()
torch.manual_seed(0)
nodes = torch.rand(B, N, F, device=dev)
adj = (torch.rand(B, N, N, device=dev) < 0.03).float()
x = torch.rand(B, F, device=dev)
cnt = torch.randint(0, N, (B,), device=dev)
flags = torch.zeros(1, dtype=torch.int32, device=dev)
W = [torch.randn(H, F, device=dev) * 0.1, torch.randn(H, device=dev) * 0.1, torch.randn(H, F, device=dev) * 0.1,
     torch.randn(H, H, device=dev) * 0.1, torch.randn(H, device=dev) * 0.1, torch.randn(H, H, device=dev) * 0.1]
p = _hip.ptr
st = _hip.stream()


def timeit(name, fn, iters=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name:34s} {a.elapsed_time(b) / iters * 1e3:8.2f} us/launch")


nodes_out, adj_out = torch.empty_like(nodes), torch.empty_like(adj)
cur, nxt = torch.empty_like(cnt), torch.empty_like(cnt)
timeit("state_advance_fwd", lambda: lib.gcm_state_advance_fwd(
    p(nodes), p(adj), None, p(cnt), p(x), p(nodes_out), p(adj_out), None, p(cur), p(nxt), p(flags), B, N, F, st))
full = torch.full((B,), N, device=dev, dtype=torch.long)
timeit("state_advance_fwd (all wrap)", lambda: lib.gcm_state_advance_fwd(
    p(nodes), p(adj), None, p(full), p(x), p(nodes_out), p(adj_out), None, p(cur), p(nxt), p(flags), B, N, F, st))
lib.gcm_state_advance_fwd(p(nodes), p(adj), None, p(cnt), p(x), p(nodes_out), p(adj_out), None, p(cur), p(nxt),
                          p(flags), B, N, F, st)
mx = torch.empty(B, H, device=dev)
h1 = torch.empty(B, N, H, device=dev)
agg1 = torch.empty(B, N, F, device=dev)
agg2 = torch.empty(B, H, device=dev)
timeit("gnn2_row_fwd", lambda: lib.gcm_dense_gnn2_row_fwd(
    p(nodes_out), p(adj_out), p(cur), p(W[0]), p(W[1]), p(W[2]), 1, p(W[3]), p(W[4]), p(W[5]), 1,
    p(mx), p(h1), p(agg1), p(agg2), p(flags), B, N, F, H, H, st))
timeit("gnn2_row_fwd (no saves)", lambda: lib.gcm_dense_gnn2_row_fwd(
    p(nodes_out), p(adj_out), p(cur), p(W[0]), p(W[1]), p(W[2]), 1, p(W[3]), p(W[4]), p(W[5]), 1,
    p(mx), None, None, None, p(flags), B, N, F, H, H, st))
P = lib.gcm_dense_gnn2_param_count(F, H, H)
g_mx = torch.randn(B, H, device=dev)
g_no = torch.randn(B, N, F, device=dev)
g_ni = torch.empty(B, N, F, device=dev)
g_obs = torch.empty(B, F, device=dev)
slabs = torch.zeros(B, P, device=dev)
timeit("gnn2_row_bwd", lambda: lib.gcm_dense_gnn2_row_bwd(
    p(g_mx), p(g_no), p(nodes_out), p(adj_out), p(cur), p(cnt), p(W[0]), p(W[1]), p(W[2]), 1, p(W[3]), p(W[4]),
    p(W[5]), 1, p(mx), p(h1), p(agg1), p(agg2), p(g_ni), p(g_obs), p(slabs), 0, B, N, F, H, H, st))
flat = torch.empty(P, device=dev)
timeit("sum_slabs", lambda: lib.gcm_sum_slabs(p(slabs), B, P, p(flat), st))
out = torch.empty(B, N, H, device=dev)
timeit("dense_graphconv_fwd (layered)", lambda: lib.gcm_dense_graphconv_fwd(
    p(nodes_out), p(adj_out), p(W[0]), p(W[1]), p(W[2]), p(out), p(agg1), B, N, F, H, 1, st))
hops = (__import__("ctypes").c_int32 * 3)(1, 2, 4)
timeit("edge_temporal", lambda: lib.gcm_edge_temporal(p(adj_out), p(cur), __import__("ctypes").addressof(hops), 3, 1, B, N, st))
big = torch.empty(256 * 1024 * 1024 // 4, device=dev)
big2 = torch.empty_like(big)
timeit("torch copy 256MiB (HBM ref)", lambda: big2.copy_(big), iters=20)
