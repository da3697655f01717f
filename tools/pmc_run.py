#!/usr/bin/env python3
"""Short cfg2 workload for rocprofv3 --pmc passes (HBM traffic of the dominant kernels): 2 eager
per-step rollouts fwd+bwd with donated state (k_step_rows<...,false>, k_bptt_rows), 2 with functional
state (k_step_rows<...,true>) and 2 rollout-API calls, T=128.  Run once per counter:
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_run.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_run.py
then tools/pmc_summarise.py writes profiles/<tag>_traffic_detail.json and profiles/traffic.json."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
obs = torch.rand(128, bench.CONFIGS['cfg2']['B'], bench.CONFIGS['cfg2']['F']).to(dev)
for donate in (True, False):
    mem, gnn, _ = bench.build_memory(dev, donate=donate)
    for _ in range(2):
        bench.rollout(mem, obs)
        gnn.zero_grad(set_to_none=True)
    if not donate:
        for _ in range(2):
            bench.rollout_api(mem, obs)
            gnn.zero_grad(set_to_none=True)
torch.cuda.synchronize()
print("done")
