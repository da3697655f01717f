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

    new_sinks_only = True   # every edge ends in a NEW node: SparseGCM merges without a sort

    def plan(self, T, taus):
        """per-graph edge offsets [B+1] (device; no sync): SparseGCM reads the total back together
        with its own sizes and hands both to forward()"""
        return _ops.sparse_temporal_count(T, taus, self._hops_desc)

    def forward(self, nodes, T, taus, B, plan=None):
        edge_off, E = plan if plan is not None else (None, None)
        idx = _ops.sparse_temporal_edges(T, taus, self._hops_desc, edge_off, E)
        vals = torch.ones(idx.shape[1], device=idx.device)
        out = torch.sparse_coo_tensor(idx, vals, size=(B, int(1e5), int(1e5)), is_coalesced=True)
        out.gcm_bptr = edge_off      # rows of each graph (None: computed on demand)
        return out
