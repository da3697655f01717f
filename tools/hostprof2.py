#!/usr/bin/env python3
"""cProfile of the grad-mode forward loop (per-step host cost).  Dev tool."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch
import bench
dev = torch.device("cuda", 0)
mem, gnn = bench.build_memory(dev)
T = 128
obs = torch.rand(T, bench.B, bench.F).to(dev)
def fwd():
    hid, outs = None, []
    for t in range(T):
        mx, hid = mem(obs[t], hid)
        outs.append(mx)
    return outs
for _ in range(3):
    torch.stack(fwd()).mean().backward(); gnn.zero_grad(set_to_none=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
keep = []
pr.enable()
for _ in range(3):
    keep.append(fwd())
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(14)
