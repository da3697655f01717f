import os, sys, torch
sys.path.insert(0, "graph-conv-memory_amd")
from gcm.gcm import DenseGCM
from gcm import nn as G
from gcm.edge_selectors.temporal import TemporalBackedge
from gcm.edge_selectors.dense import DenseEdge
DEV="cuda"
def run(B,N,F,H1,H2,T,sel, grad):
    torch.manual_seed(1)
    obs = torch.rand(T, B, F, device=DEV)
    g = G.Sequential("x, adj, weights, B, N", [(G.DenseGraphConv(F, H1), "x, adj -> x"), torch.nn.Tanh(), (G.DenseGraphConv(H1, H2), "x, adj -> x"), torch.nn.Tanh()]).to(DEV)
    mem = DenseGCM(g, edge_selectors=sel, graph_size=N)
    with torch.set_grad_enabled(grad):
        out, hid = mem.rollout(obs)
    with torch.no_grad():
        h, outs = None, []
        for t in range(T):
            mx, h = mem(obs[t], h)
            outs.append(mx)
    ref = torch.stack(outs)
    err = (out - ref).abs().amax(-1).amax(-1)
    bad = (err > 1e-5).nonzero().flatten().tolist()
    print((B,N,F,H1,H2,T,type(sel).__name__, grad), "max err", float(err.max()), "first bad t", bad[:5], "state eq", [torch.equal(a,b) for a,b in zip(hid,h)])
for grad in (False, True):
    run(3,64,32,64,64,80,DenseEdge(),grad)
    run(3,64,32,64,64,80,TemporalBackedge([1,2,4]),grad)
    run(3,64,32,32,32,80,DenseEdge(),grad)
    run(3,64,64,64,64,80,DenseEdge(),grad)
