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


def get_causal_edges(T, taus, window=None):
    """util.py:242-282 - every (batch, sink, source) pair with sink a new node (T <= sink <
    T + tau), source < sink and, with a window, source >= max(0, T - window); coalesced order.
    Closed form on the device instead of a Python loop over tril_indices."""
    from . import _ops
    return _ops.CausalEdges(T, taus, window).indices


def sparse_gumbel_softmax(logits, dim, tau=1, hard=False, noise=None):
    """util.py:89-130 for the layout this path uses: logits a coalesced torch.sparse_coo
    [B,N,N] with indices (batch, sink, source), softmax over dim=2 inside every (batch, sink)
    row.  hard=True (scatter_max over rows) is not implemented."""
    from . import _ops
    if hard or dim not in (2, -1):
        raise NotImplementedError("only the soft, dim=2 form used by LearnedEdge is implemented")
    logits = logits.coalesce()
    idx, vals = logits.indices(), logits.values()
    key = idx[0] * logits.shape[1] + idx[1]
    first = torch.ones_like(key, dtype=torch.bool)
    first[1:] = key[1:] != key[:-1]
    seg_ptr = torch.cat([first.nonzero().flatten(), torch.tensor([key.numel()], device=key.device)])

    class _Rows:
        pass
    rows = _Rows()
    rows.seg_ptr, rows.S, rows.E = seg_ptr.contiguous(), seg_ptr.numel() - 1, key.numel()
    if noise is None:
        noise = -torch.empty_like(vals).exponential_().log()
    tau_t = tau if torch.is_tensor(tau) else torch.tensor([float(tau)], device=vals.device)
    soft = _ops.segment_softmax(vals, tau_t, noise, rows)
    return torch.sparse_coo_tensor(idx, soft, size=logits.shape)
