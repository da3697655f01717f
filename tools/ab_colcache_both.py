#!/usr/bin/env python3
"""A/B of the column-write cached step against the general live-row kernel for SPARSE column-writing chains: cfg2's shapes with
TemporalBackedge([1,2,4], direction="both") and DenseEdge, donated state, graph-replayed loop + backward, T = 128 / 256.
Dev tool."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-conv-memory_amd")]
import torch  # noqa: E402
import bench  # noqa: E402
from gcm.gcm import DenseGCM  # noqa: E402
from gcm.edge_selectors.temporal import TemporalBackedge  # noqa: E402
from gcm.edge_selectors.dense import DenseEdge  # noqa: E402

DenseGCM.did_warn = True
dev = torch.device("cuda", 0)
c = bench.CONFIGS["cfg2"]
B, N, F, H = c["B"], c["N"], c["F"], c["H"]
for name, mk in (("both[1,2,4]", lambda: TemporalBackedge([1, 2, 4], direction="both")),
                 ("backward[1,2,4]", lambda: TemporalBackedge([1, 2, 4], direction="backward")),
                 ("dense", lambda: DenseEdge())):
    for T in (128, 256):
        obs = torch.rand(T, B, F, device=dev)
        for on in (True, False):
            torch.manual_seed(0)
            gnn = bench.dense_gnn(F, H, dev)
            mem = DenseGCM(gnn, edge_selectors=mk(), graph_size=N, donate_state=True)
            mem.rows_col_cache = on
            g = bench.capture(lambda: bench.rollout(mem, obs), lambda: gnn.zero_grad(set_to_none=True))
            ms = bench.event_time(g.replay, 10, warm=3)
            print(f"{name:16s} T={T} col_cache={on!s:5s}: {ms * 1e3:8.1f} us per rollout, {B * T / ms / 1e3:6.2f} M belief-states/s, "
                  f"col steps {mem.rows_col_steps_taken()}", flush=True)
            del g
