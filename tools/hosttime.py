#!/usr/bin/env python3
"""Where does the per-step host time go?  Dev tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch
import bench
from gcm import parallel, _ops

dev = torch.device("cuda", 0)
mem, gnn = bench.build_memory(dev)
obs = torch.rand(128, bench.B, bench.F).to(dev)
T = 128

def fwd(grad):
    ctx = torch.enable_grad() if grad else torch.no_grad()
    with ctx:
        hid, outs = None, []
        for t in range(T):
            mx, hid = mem(obs[t], hid)
            outs.append(mx)
    return outs

for _ in range(3):
    torch.stack(fwd(True)).mean().backward(); gnn.zero_grad(set_to_none=True)
torch.cuda.synchronize()

def timed(label, fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / n
    print(f"{label:38s} host {t_host*1e3:7.2f} ms  ({t_host/T*1e6:6.1f} us/step)   host+gpu {t_all*1e3:7.2f} ms")
    return r

timed("forward loop, no_grad", lambda: fwd(False))
outs = timed("forward loop, grad", lambda: fwd(True))
FW = []
def fb():
    torch.cuda.synchronize()
    tf = time.perf_counter()
    o = fwd(True)
    tf1 = time.perf_counter() - tf
    torch.cuda.synchronize()
    FW.append((tf1, time.perf_counter() - tf))
    t0 = time.perf_counter()
    torch.stack(o).mean().backward()
    t1 = time.perf_counter() - t0
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t0
    gnn.zero_grad(set_to_none=True)
    return t1, t2
rs = [fb() for _ in range(5)]
print(f"{'forward (memory recycled by backward)':38s} host {sum(r[0] for r in FW)/5*1e3:7.2f} ms  ({sum(r[0] for r in FW)/5/T*1e6:6.1f} us/step)   host+gpu {sum(r[1] for r in FW)/5*1e3:7.2f} ms")
print(f"{'backward only':38s} host {sum(r[0] for r in rs)/5*1e3:7.2f} ms  ({sum(r[0] for r in rs)/5/T*1e6:6.1f} us/step)   host+gpu {sum(r[1] for r in rs)/5*1e3:7.2f} ms")
# raw C call cost
cfg = mem._fused_plan(*mem.get_initial_hidden_state(obs[0])[:3], bench.F)
packed = mem._packed_params(cfg).detach()
h = mem.get_initial_hidden_state(obs[0])
flags = torch.zeros(1, dtype=torch.int32, device=dev)
def raw():
    for t in range(T):
        _ops._FusedStep.apply(obs[t], h[0], packed, h[1], h[3], flags, cfg)
timed("128 x _FusedStep.apply (no module)", raw)
