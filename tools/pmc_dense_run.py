#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes of the dense-materialised regime (tools/collect_profiles.sh part `dense`):
DenseGCM + DenseEdge at cfg2's shapes (bench.py --config dense_edge), few launches (counter passes serialise dispatches):
  one donated rollout T = 128 fwd + bwd with the column-write cached step (k_step_colcache, k_bptt_rows<..,4>),
  16 steps past graph_size on the same chain (the ring form of the same kernel),
  one donated rollout T = 128 fwd + bwd with the form switched off (k_step_rows in the dense regime: every row <= cur
  live, the cur^2 F aggregation on the 16x16x4 fp32 MFMA; k_bptt_rows<..,4> over its records), 8 steps past graph_size."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-conv-memory_amd")]
import torch  # noqa: E402
import bench  # noqa: E402
from gcm.gcm import DenseGCM  # noqa: E402

DenseGCM.did_warn = True
dev = torch.device("cuda", 0)
c = bench.CONFIGS["dense_edge"]
for on, extra in ((True, 16), (False, 8)):
    mem, gnn, sel = bench.build_memory(dev, donate=True, selector="dense", cfg=c)
    mem.rows_col_cache = on
    bench.rollout(mem, bench.make_obs(c, 0, dev))
    gnn.zero_grad(set_to_none=True)
    bench.rollout(mem, bench.make_obs(dict(c, T=c["N"] + extra), 0, dev))
    torch.cuda.synchronize()
print("done")
