#!/usr/bin/env python3
"""k_euclid_mfma2 alone on FULL graphs at cfg3's shape (what bench.py --config cfg3 reports as `roofline`), for
`rocprofv3 --kernel-trace --stats` (profiles/r03_euclid_full_kernel_stats.csv): the bench run's own trace mixes these
launches with the in-situ ones of a rollout that starts from empty graphs (rows >= cur skipped: shorter)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import bench  # noqa: E402

import torch  # noqa: E402

launch = bench.euclid_launcher(bench.CONFIGS["cfg3"])
for _ in range(200):
    launch()
torch.cuda.synchronize()
print("k_euclid_mfma2 on full graphs: 200 back-to-back launches done")
