"""DenseEdge (reference: src/gcm/edge_selectors/dense.py:4-23)."""
import torch

from .. import _ops


class DenseEdge(torch.nn.Module):
    """Connect the new node to every earlier node in both directions, plus a self edge."""

    def native_desc(self):
        from .. import _hip
        return _hip.SelectorDesc(kind=_hip.SEL_DENSE)

    def forward(self, nodes, adj_mats, edge_weights, num_nodes, B):
        if adj_mats.requires_grad:
            mask = _ops.edge_dense_(torch.zeros_like(adj_mats), num_nodes)
            return torch.where(mask > 0, mask, adj_mats), edge_weights
        _ops.edge_dense_(adj_mats, num_nodes)
        return adj_mats, edge_weights
