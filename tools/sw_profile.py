"""cfg4 stepwise leg alone (x [B,1,F] per call x 512 + one backward), for rocprofv3 --kernel-trace --stats:
python tools/sw_profile.py [rollouts]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "graph-conv-memory_amd"))
import torch
from gcm import nn as G
from gcm.sparse_gcm import SparseGCM
from gcm.sparse_edge_selectors.temporal import TemporalEdge

B, N, F, H = 512, 512, 32, 32
dev = "cuda:0"
torch.manual_seed(0)
g = G.Sequential("x, edges, weights", [(G.GraphConv(F, H), "x, edges, weights -> x"), torch.nn.Tanh(),
                                       (G.GraphConv(H, H), "x, edges, weights -> x"), torch.nn.Tanh()]).to(dev)
mem = SparseGCM(g, edge_selectors=TemporalEdge([1]), graph_size=N)
if os.environ.get("GCM_SW_CACHE") == "0":
    mem.stepwise_cache = False
if os.environ.get("GCM_SW_FLAGS") == "off":
    mem.finite_check = "off"
x = torch.rand(B, N, F, device=dev)
xs = [x[:, t:t + 1].contiguous() for t in range(N)]
one = torch.ones(B, dtype=torch.long, device=dev)


def rollout():
    hid, outs = None, []
    for t in range(N):
        o, hid = mem(xs[t], one, hid)
        outs.append(o)
    t1 = time.perf_counter()
    torch.cat(outs, 1).mean().backward()
    g.zero_grad(set_to_none=True)
    return t1


n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(2):
    rollout()
torch.cuda.synchronize()
t0 = time.perf_counter()
fw = 0.0
for _ in range(n):
    ta = time.perf_counter()
    t1 = rollout()
    fw += t1 - ta
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("stepwise: %.1f us per call (forward loop %.1f us), %.2f M states/s" % (dt / n / N * 1e6, fw / n / N * 1e6, B * N * n / dt / 1e6))
