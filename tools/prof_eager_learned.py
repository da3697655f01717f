#!/usr/bin/env python3
"""Host cost of the eager per-step loop with LearnedEdge (cfg5): wall per step and a cProfile of the forward loop."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS["cfg5"])
donate = not (len(sys.argv) > 1 and sys.argv[1] == "functional")
mem, gnn, sel = bench.build_memory(dev, donate=donate, selector="learned", cfg=c)
obs = bench.make_obs(c, 0, dev)
T = c["T"]


def fwd():
    hidden = None
    outs = []
    for t in range(T):
        mx, hidden = mem(obs[t], hidden)
        outs.append(mx)
    return outs


def full():
    outs = fwd()
    torch.stack(outs).mean().backward()
    for m in (gnn, sel):
        m.zero_grad(set_to_none=True)


for _ in range(5):
    full()
torch.cuda.synchronize()
for name, fn in (("forward loop (graph recorded)", fwd), ("forward + backward", full)):
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    host = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20
    print(f"{name}: host {host / T * 1e6:.2f} us per step, wall {wall / T * 1e6:.2f} us per step "
          f"= {c['B'] * T / wall / 1e6:.1f} M belief-states/s")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    full()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
