"""Helpers of the hot path (reference: src/gcm/util.py:9-26)."""
import torch


class STEFunction(torch.autograd.Function):
    """util.py:9-18 - forward (x > 0) as float, backward passes the gradient through."""

    @staticmethod
    def forward(ctx, input):
        return (input > 0).float()

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output


class StraightThroughEstimator(torch.nn.Module):
    """util.py:21-26."""

    def forward(self, x):
        return STEFunction.apply(x)


class Spardmax(torch.nn.Module):
    """util.py:29-42 - unusable at the reference HEAD too (its `sparsemax`
    import is commented out, util.py:5 -> NameError at util.py:36)."""

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError(
            "Spardmax needs the `sparsemax` package, which the reference itself no longer "
            "imports (util.py:5); deterministic=True selectors are out of scope")


def pack_hidden(hidden, B, max_edges: int, edge_fill: int = -1, weight_fill: float = 1.0):
    """util.py:323-351 - sparse hidden (nodes, coo adj, T) -> the fixed-size form RLlib carries
    between calls: (nodes, dense_edges [B,2,max_edges] i64, dense_weights [B,1,max_edges], T).
    One kernel over the COO entries instead of a Python loop over B."""
    from . import _ops
    nodes, adj, T = hidden
    adj = adj.coalesce()
    dense_edges, dense_weights = _ops.pack_hidden(adj.indices(), adj.values(), B, max_edges,
                                                  edge_fill, weight_fill)
    return nodes, dense_edges, dense_weights, T


def unpack_hidden(hidden, B):
    """util.py:353-382 - the inverse: entries with dense_edges[b, 0, k] >= 0 become COO entries
    (b, dense_edges[b,0,k], dense_edges[b,1,k]) with value dense_weights[b,0,k]."""
    nodes, edges, weights, T = hidden
    batch_idx, edge_idx = (edges[:, 0] >= 0).nonzero().T.unbind()
    adj_idx = torch.stack([batch_idx, edges[batch_idx, 0, edge_idx], edges[batch_idx, 1, edge_idx]])
    adj = torch.sparse_coo_tensor(indices=adj_idx, values=weights[batch_idx, 0, edge_idx],
                                  size=(B, nodes.shape[1], nodes.shape[1]))
    return nodes, adj, T
