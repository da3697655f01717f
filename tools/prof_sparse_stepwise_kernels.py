#!/usr/bin/env python3
"""Kernels and host time of ONE stepwise SparseGCM call (x [B, 1, F], taus = 1) in the middle of a chain at cfg4's shape:
launch order, durations, gaps (torch.profiler), and the host's issue time per call.  Dev tool."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-conv-memory_amd")]
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
from gcm import nn as G  # noqa: E402
from gcm.sparse_gcm import SparseGCM  # noqa: E402
from gcm.sparse_edge_selectors.temporal import TemporalEdge  # noqa: E402

dev = torch.device("cuda", 0)
B, N, F, H = 512, 512, 32, 32
torch.manual_seed(0)
g = G.Sequential("x, edges, weights", [(G.GraphConv(F, H), "x, edges, weights -> x"), torch.nn.Tanh(),
                                       (G.GraphConv(H, H), "x, edges, weights -> x"), torch.nn.Tanh()]).to(dev)
mem = SparseGCM(g, edge_selectors=TemporalEdge([1]), graph_size=N)
x = torch.rand(B, N, F, device=dev)
one = torch.ones(B, dtype=torch.long, device=dev)
hid = None
for t in range(200):
    o, hid = mem(x[:, t:t + 1], one, hid)
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(200, 300):
    o, hid = mem(x[:, t:t + 1], one, hid)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("100 calls: issue %.1f us/call, done %.1f us/call" % ((t1 - t0) * 1e4, (t2 - t0) * 1e4))
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for t in range(300, 304):
        o, hid = mem(x[:, t:t + 1], one, hid)
    torch.cuda.synchronize()
ev = sorted([e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
for e in ev[-12:]:
    print("%8.1f us  %-90s" % (e.time_range.elapsed_us(), e.name[:90]))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=60))
