#!/usr/bin/env python3
"""cfg5 with chains longer than graph_size (every step beyond N works on full graphs): time of rollout + backward.
Run twice to A/B a library-level switch (environment variable read by the library)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
for T in (64, 256):
    c = dict(bench.CONFIGS["cfg5"])
    c["T"] = T
    mem, gnn, sel = bench.build_memory(dev, donate=True, selector="learned", cfg=c)
    type(mem).did_warn = True
    obs = bench.make_obs(c, 0, dev)
    for it in range(3):
        bench.rollout(mem, obs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 5
    for it in range(R):
        bench.rollout(mem, obs)
        for m in (gnn, sel):
            m.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    print(f"T={T}: {dt * 1e3:.3f} ms per rollout+backward, {c['B'] * T / dt / 1e6:.2f} M belief-states/s")
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        bench.rollout(mem, obs)
        torch.cuda.synchronize()
    for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:6]:
        print(f"   {e.key[:70]:70s} n={e.count:4d} total={e.device_time_total:10.1f} us")
