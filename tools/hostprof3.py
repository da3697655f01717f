#!/usr/bin/env python3
"""torch.profiler (CPU side) of the backward pass of the per-step loop.  Dev tool."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
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
    return torch.stack(outs).mean()
for _ in range(3):
    fwd().backward(); gnn.zero_grad(set_to_none=True)
torch.cuda.synchronize()
loss = fwd()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    loss.backward()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=18, max_name_column_width=60))
