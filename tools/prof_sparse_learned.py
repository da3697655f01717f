#!/usr/bin/env python3
"""SparseGCM + LearnedEdge at the reference's smoke size (tests/test_sparse_gcm.py:822-852: B=8, N=256, F=32),
a few calls forward + backward - for `rocprofv3 --kernel-trace --stats` (profiles/r03_sparse_learned_kernel_stats.csv:
the edge network runs on gcm_rows_linear / gcm_skinny_wgrad / gcm_relu_layernorm_bwd, no library GEMM)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch  # noqa: E402
from gcm import nn as G  # noqa: E402
from gcm.sparse_gcm import SparseGCM  # noqa: E402
from gcm.sparse_edge_selectors.learned import LearnedEdge  # noqa: E402

dev = torch.device("cuda", 0)
B, N, F, tau = 8, 256, 32, 32
torch.manual_seed(0)
g = G.Sequential("x, edges, weights", [(G.GraphConv(F, F), "x, edges, weights -> x"), torch.nn.Tanh(),
                                       (G.GraphConv(F, F), "x, edges, weights -> x"), torch.nn.Tanh()]).to(dev)
sel = LearnedEdge(F, num_edge_samples=4, window=64, log_stats=False, store_grads=False).to(dev)
mem = SparseGCM(g, edge_selectors=sel, graph_size=N)
for it in range(5):
    hidden, outs = None, []
    for _ in range(N // tau):
        out, hidden = mem(torch.randn(B, tau, F, device=dev), torch.full((B,), tau, device=dev), hidden)
        outs.append(out)
    torch.cat(outs, 1).mean().backward()
    g.zero_grad(set_to_none=True)
    sel.zero_grad(set_to_none=True)
torch.cuda.synchronize()
print("done")
