#!/usr/bin/env python3
"""cProfile of the layered LearnedEdge step (cfg5 per-GPU share): where the host time goes.  Dev tool."""
import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch
from gcm.gcm import DenseGCM
from gcm import nn as G
from gcm.edge_selectors.learned import LearnedEdge
dev = "cuda"
B, N, F, H, T = 256, 128, 32, 32, 32
g = G.Sequential("x, adj, weights, B, N", [(G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
                                           (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(dev)
mem = DenseGCM(g, edge_selectors=LearnedEdge(F).to(dev), graph_size=N)
obs = torch.rand(T, B, F, device=dev)
def run():
    hid, outs = None, []
    for t in range(T):
        mx, hid = mem(obs[t], hid)
        outs.append(mx)
    torch.stack(outs).mean().backward()
for _ in range(2):
    run()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
run()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
