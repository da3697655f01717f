#!/usr/bin/env python3
"""Where the per-step forward host time goes, layer by layer.  Dev tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch
import bench
from gcm import _ops, _ext
dev = torch.device("cuda", 0)
mem, gnn = bench.build_memory(dev)
T = 128
obs = torch.rand(T, bench.B, bench.F).to(dev)
xs = [obs[t] for t in range(T)]

def timed(label, fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    print(f"{label:60s} {dt / T * 1e6:6.2f} us/step")

def full():
    hid = None
    for t in range(T):
        mx, hid = mem(xs[t], hid)
def full_fwd():
    hid = None
    for t in range(T):
        mx, hid = mem.forward(xs[t], hid)
with torch.no_grad():
    timed("mem(x, hid)  [no grad]", full)
    timed("mem.forward(x, hid)  [no grad]", full_fwd)
    _, h = mem(xs[0], None); _, h = mem(xs[1], h)
    nodes, adj, w, nn_ = h
    link = nodes._gcm_link
    cfg, flags = link[2], link[3]
    packed = mem._packed_params(cfg)
    ext = _ext.module()
    handle = cfg.cpp_handle()
    stream = torch._C._cuda_getCurrentRawStream(0)
    timed("mem._forward_fused(...) same hidden", lambda: [mem._forward_fused(xs[t], nodes, adj, w, nn_, cfg, flags, link) for t in range(T)])
    timed("_ops.fused_step(...)", lambda: [_ops.fused_step(xs[t], nodes, packed, adj, nn_, flags, cfg) for t in range(T)])
    timed("ext.fused_step(...)", lambda: [ext.fused_step(xs[t], nodes, packed, adj, nn_, flags, handle, stream) for t in range(T)])
    timed("mem._packed_params(cfg)", lambda: [mem._packed_params(cfg) for t in range(T)])
    timed("mem._poll(flags)", lambda: [mem._poll(flags) for t in range(T)])
    timed("getattr(nodes, '_gcm_link')", lambda: [getattr(nodes, "_gcm_link", None) for t in range(T)])
    def setl():
        for t in range(T):
            nodes._gcm_link = link
    timed("nodes._gcm_link = link", setl)
    timed("x.shape == link[6]", lambda: [xs[t].shape == link[6] for t in range(T)])
packed_g = None
timed("mem(x, hid)  [grad, graph dropped each step]", lambda: [mem(xs[t], None) for t in range(2)] and full())
