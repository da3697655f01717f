#!/usr/bin/env python3
"""Short cfg2 workload for rocprofv3 --pmc passes (HBM traffic of the dominant kernels):
2 per-step rollouts fwd+bwd and 2 rollout-API calls, T=128.  Run once per counter:
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_run.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_run.py
then tools/pmc_summarise.py writes profiles/traffic.json."""
import os
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
obs = torch.rand(128, bench.B, bench.F).to(dev)
for _ in range(2):
    bench.rollout(mem, obs, bucket, 1.0)
    gnn.zero_grad(set_to_none=True)
# the rollout entry: persistent forward, time-parallel BPTT, reverse scan
for _ in range(2):
    bench.rollout_api(mem, obs, bucket, 1.0)
    gnn.zero_grad(set_to_none=True)
# the non-advance forward kernel too (what bench.py's roofline block times)
bench.time_dominant_kernels(mem, obs, reps=20)
torch.cuda.synchronize()
print("done")
