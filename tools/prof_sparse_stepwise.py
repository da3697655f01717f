import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graph-conv-memory_amd")):
    sys.path.insert(0, p)
import torch
from gcm import nn as G
from gcm.sparse_gcm import SparseGCM
from gcm.sparse_edge_selectors.temporal import TemporalEdge
dev = torch.device("cuda", 0)
B, N, F, H = 512, 512, 32, 32
torch.manual_seed(0)
g = G.Sequential("x, edges, weights", [(G.GraphConv(F, H), "x, edges, weights -> x"), torch.nn.Tanh(),
                                       (G.GraphConv(H, H), "x, edges, weights -> x"), torch.nn.Tanh()]).to(dev)
mem = SparseGCM(g, edge_selectors=TemporalEdge([1]), graph_size=N)
x = torch.rand(B, N, F, device=dev)
one = torch.ones(B, dtype=torch.long, device=dev)
def stepwise():
    hid, outs = None, []
    t0 = time.perf_counter()
    for t in range(N):
        o, hid = mem(x[:, t:t + 1], one, hid)
        outs.append(o)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    torch.cat(outs, 1).mean().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    g.zero_grad(set_to_none=True)
    return t1 - t0, t2 - t1
stepwise()
f, b = stepwise()
print("forward %.1f ms (%.0f us/step), backward %.1f ms (%.0f us/step)" % (f * 1e3, f / N * 1e6, b * 1e3, b / N * 1e6))
