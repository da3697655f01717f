"""Sparse TemporalEdge (reference: src/gcm/sparse_edge_selectors/temporal.py:11-63)."""
from typing import List

import torch

from .. import _ops


class TemporalEdge(torch.nn.Module):
    """For every new node t in [T_b, T_b + tau_b) and every hop h add the edge
    (batch b, sink t, source t - h) when the source exists (t - h >= 0) and t > 0.

    Returns a torch.sparse_coo tensor with indices (batch, sink, source) and unit values,
    size (B, 1e5, 1e5) like the reference.  The edges come out of one closed-form kernel in
    coalesced order (no per-graph Python loop, no sort)."""

    def __init__(self, hops: List[int] = [1]):
        super().__init__()
        self.hops = torch.tensor(hops)
        # descending + unique => ascending sources inside each sink (coalesced COO order);
        # a repeated hop only makes a duplicate edge that coalesce()/normalisation removes
        self._hops_desc = sorted({int(h) for h in hops}, reverse=True)

    def forward(self, nodes, T, taus, B):
        idx = _ops.sparse_temporal_edges(T, taus, self._hops_desc)
        vals = torch.ones(idx.shape[1], device=idx.device)
        return torch.sparse_coo_tensor(idx, vals, size=(B, int(1e5), int(1e5)),
                                       is_coalesced=True)
