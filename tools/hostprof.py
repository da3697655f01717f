#!/usr/bin/env python3
"""cProfile of the per-step DenseGCM loop (host overhead hunt).  Dev tool."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from gcm import parallel  # noqa: E402

dev = torch.device("cuda", 0)
mem, gnn = bench.build_memory(dev)
bucket = parallel.GradBucket(gnn)
grad_obs = len(sys.argv) > 1 and sys.argv[1] == "grad"
obs = torch.rand(128, bench.B, bench.F).to(dev).requires_grad_(grad_obs)
for _ in range(3):
    bench.rollout(mem, obs, bucket, 1.0)
    gnn.zero_grad(set_to_none=True)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(3):
    bench.rollout(mem, obs, bucket, 1.0)
    gnn.zero_grad(set_to_none=True)
torch.cuda.synchronize()
print("ms per rollout", (time.perf_counter() - t0) / 3 * 1e3, "obs.requires_grad", grad_obs)
pr = cProfile.Profile()
pr.enable()
bench.rollout(mem, obs, bucket, 1.0)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
