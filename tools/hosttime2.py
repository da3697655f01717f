#!/usr/bin/env python3
"""Host cost of one fused-step node, C++ vs Python, grad vs no_grad.  Dev tool."""
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
h = mem.get_initial_hidden_state(obs[0])
cfg = mem._fused_plan(h[0], h[1], h[2], bench.F)
flags = torch.zeros(1, dtype=torch.int32, device=dev)
packed_g = mem._packed_params(cfg)
packed_n = packed_g.detach()
ext = _ext.module()
stream = torch.cuda.current_stream().cuda_stream

def timed(label, fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / n
    print(f"{label:46s} host {t_host/T*1e6:6.1f} us/step   host+gpu {t_all/T*1e6:6.1f} us/step")
    return r

def loop(fn, packed, keep):
    def run():
        outs = []
        for t in range(T):
            r = fn(obs[t], h[0], packed, h[1], h[3], flags)
            if keep:
                outs.append(r)
        return outs
    return run
cpp = lambda o, n, p, a, c, f: ext.fused_step(o, n, p, a, c, f, cfg.cpp_handle(), stream)
py = lambda o, n, p, a, c, f: _ops._FusedStep.apply(o, n, p, a, c, f, cfg)
timed("C++ node, no grad", loop(cpp, packed_n, False))
timed("C++ node, grad, outputs dropped", loop(cpp, packed_g, False))
timed("C++ node, grad, outputs kept (128 x 30 MB)", loop(cpp, packed_g, True))
timed("Python node, no grad", loop(py, packed_n, False))
timed("Python node, grad, outputs kept", loop(py, packed_g, True))
timed("obs[t] only", lambda: [obs[t] for t in range(T)])
timed("torch._C._cuda_getCurrentRawStream", lambda: [torch._C._cuda_getCurrentRawStream(0) for t in range(T)])
timed("torch.cuda.current_stream().cuda_stream", lambda: [torch.cuda.current_stream().cuda_stream for t in range(T)])
