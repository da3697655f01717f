"""Forward / backward split of DenseGCM.rollout on cfg2 (HIP events)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
from gcm.gcm import DenseGCM
from gcm import nn as G
from gcm.edge_selectors.temporal import TemporalBackedge

B, N, F, H = 256, 128, 32, 32
T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = "cuda"
g = G.Sequential("x, adj, weights, B, N", [(G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
                                           (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(dev)
mem = DenseGCM(g, edge_selectors=TemporalBackedge([1, 2, 4]), graph_size=N)
obs = torch.rand(T, B, F, device=dev)
ev = lambda: torch.cuda.Event(enable_timing=True)
for it in range(6):
    g.zero_grad(set_to_none=True)
    e0, e1, e2 = ev(), ev(), ev()
    e0.record()
    out, hid = mem.rollout(obs)
    loss = out.mean()
    e1.record()
    loss.backward()
    e2.record()
    torch.cuda.synchronize()
    print(f"iter {it}: fwd {e0.elapsed_time(e1):.3f} ms  bwd {e1.elapsed_time(e2):.3f} ms")
