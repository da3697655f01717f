#!/usr/bin/env python3
"""cfg2 rollouts through the eager per-step loop (donated state), forward + backward - for
`rocprofv3 --kernel-trace --stats` when looking at the step kernels alone.  Dev tool."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
c = bench.CONFIGS["cfg2"]
mem, gnn, _ = bench.build_memory(dev, donate=True)
obs = bench.make_obs(c, 0, dev)
for _ in range(6):
    bench.rollout(mem, obs)
    gnn.zero_grad(set_to_none=True)
torch.cuda.synchronize()
print("cached steps:", mem.rows_cached_steps_taken())
