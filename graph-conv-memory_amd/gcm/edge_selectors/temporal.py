"""TemporalBackedge (reference: src/gcm/edge_selectors/temporal.py:17-94).

adj[b, i, j] = 1 means node i aggregates from node j (temporal.py:7-15)."""
from typing import List

import torch

from .. import _ops


class TemporalBackedge(torch.nn.Module):
    """Add temporal directional back edges, e.g. node_t <- node_{t-hop}."""

    def __init__(self, hops: List[int] = [1], direction="forward", learned=False,
                 learning_window=10, deterministic=False, num_samples=3):
        super().__init__()
        assert direction in ["forward", "backward", "both"]
        if learned:
            # temporal.py:51-70 - per-graph Python loop over gumbel windows; the secondary,
            # slow variant is not part of the accelerated path
            raise NotImplementedError("TemporalBackedge(learned=True) is not implemented")
        self.hops = list(hops)
        self.direction = direction
        self.learned = False

    def native_desc(self):
        """Descriptor for the fused / rollout paths (struct gcm_selector_desc)."""
        from .. import _hip
        if len(self.hops) > 16:       # the descriptor holds 16 hops: the layered path takes over
            return None
        d = _hip.SelectorDesc(kind=_hip.SEL_TEMPORAL, n_hops=len(self.hops),
                              direction=_hip.DIR[self.direction])
        for i, h in enumerate(self.hops):
            d.hops[i] = h
        return d

    def forward(self, nodes, adj_mats, edge_weights, num_nodes, B):
        """temporal.py:72-88: for every hop and every graph with num_nodes >= hop set
        adj[b, n, n-hop] (forward/both) and/or adj[b, n-hop, n] (backward/both)."""
        if adj_mats.requires_grad:
            mask = _ops.edge_temporal_(torch.zeros_like(adj_mats), num_nodes, self.hops,
                                       self.direction)
            return torch.where(mask > 0, mask, adj_mats), edge_weights
        _ops.edge_temporal_(adj_mats, num_nodes, self.hops, self.direction)
        return adj_mats, edge_weights
