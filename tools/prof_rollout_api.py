#!/usr/bin/env python3
"""Where a `rollout()` + backward call spends its time (any dense bench config): wall per call, GPU-busy time, and the
kernels of one call.  usage: prof_rollout_api.py [cfg5]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS[name])
sel = {"cfg2": "temporal", "cfg3": "euclid", "cfg5": "learned"}[name]
mem, gnn, selm = bench.build_memory(dev, donate=False, selector=sel, cfg=c)
mods = [m for m in (gnn, selm) if isinstance(m, torch.nn.Module)]
obs = bench.make_obs(c, 0, dev)


def call():
    bench.rollout_api(mem, obs)
    for m in mods:
        m.zero_grad(set_to_none=True)


for _ in range(5):
    call()
torch.cuda.synchronize()
R = 20
t0 = time.perf_counter()
for _ in range(R):
    call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / R
print(f"{name}: {dt * 1e6:.1f} us per rollout + backward = {c['B'] * c['T'] / dt / 1e6:.2f} M belief-states/s")
t0 = time.perf_counter()
for _ in range(R):
    call()
host = (time.perf_counter() - t0) / R
torch.cuda.synchronize()
print(f"host time per call (no sync): {host * 1e6:.1f} us")
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    call()
    torch.cuda.synchronize()
ev = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
print(f"GPU busy: {sum(e.device_time_total for e in ev):.1f} us in {sum(e.count for e in ev)} launches / copies")
for e in ev[:14]:
    print(f"   {e.key[:90]:90s} n={e.count:4d} total={e.device_time_total:9.1f} us")

# host side: where the call's CPU time goes
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    call()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
