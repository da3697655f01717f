#!/usr/bin/env python3
"""Pure host cost per step: a tiny workload (GPU never the bottleneck), same code path.  Dev tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch
from gcm.gcm import DenseGCM
from gcm import nn as G, _ops, _ext
from gcm.edge_selectors.temporal import TemporalBackedge
dev = "cuda"
B, N, F, H, T = 4, 32, 32, 32, 128
g = G.Sequential("x, adj, weights, B, N", [(G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
                                           (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(dev)
mem = DenseGCM(g, edge_selectors=TemporalBackedge([1, 2, 4]), graph_size=N)
obs = torch.rand(T, B, F, device=dev)
xs = [obs[t] for t in range(T)]

def timed(label, fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{label:56s} {dt / T * 1e6:6.2f} us/step")

def fwd():
    hid, outs = None, []
    for t in range(T):
        mx, hid = mem(xs[t], hid)
        outs.append(mx)
    return outs
def fb():
    torch.stack(fwd()).mean().backward()
    g.zero_grad(set_to_none=True)
with torch.no_grad():
    timed("forward, no grad", fwd)
timed("forward, grad", fwd)
timed("forward + backward", fb)
with torch.no_grad():
    _, h = mem(xs[0], None); _, h = mem(xs[1], h)
    nodes, adj, w, nn_ = h
    link = nodes._gcm_link
    cfg, flags = link[2], link[3]
    packed = mem._packed_params(cfg)
    ext, handle = _ext.module(), cfg.cpp_handle()
    stream = torch._C._cuda_getCurrentRawStream(0)
    timed("ext.fused_step, no grad", lambda: [ext.fused_step(xs[t], nodes, packed, adj, nn_, flags, handle, stream) for t in range(T)])
    e = torch.empty(1, device=dev)
    timed("torch.empty(30MB-ish) alloc only", lambda: [torch.empty(7_500_000, device=dev) for t in range(T)])
    timed("a trivial kernel launch (x.add_(1))", lambda: [e.add_(1) for t in range(T)])
pg = mem._packed_params(cfg)
timed("ext.fused_step, grad (graph dropped)", lambda: [ext.fused_step(xs[t], nodes, pg, adj, nn_, flags, handle, stream) for t in range(T)])
