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


def sparsemax(z, dim=-1):
    """Sparsemax along `dim` (Martins & Astudillo 2016): the Euclidean projection of z onto the
    simplex, tau from the sorted cumulative sums; -inf entries stay out of the support.  This is
    what the `sparsemax` package behind the reference's Spardmax computes (util.py:29-42; the
    reference's own import of it is commented out at util.py:5, so its Spardmax raises NameError)."""
    z = z.transpose(dim, -1)
    zs, _ = torch.sort(z, dim=-1, descending=True)
    k = torch.arange(1, z.shape[-1] + 1, device=z.device, dtype=z.dtype)
    finite = torch.isfinite(zs)
    csum = torch.where(finite, zs, torch.zeros_like(zs)).cumsum(-1)
    support = (1 + k * zs > csum) & finite
    ksup = support.sum(-1, keepdim=True).clamp(min=1)
    tau = (csum.gather(-1, ksup - 1) - 1) / ksup.to(z.dtype)
    out = torch.clamp(torch.where(torch.isfinite(z), z - tau, torch.zeros_like(z)), min=0)
    return out.transpose(dim, -1)


class Spardmax(torch.nn.Module):
    """util.py:29-42 - a hard version of sparsemax (straight through)."""

    def __init__(self, dim=-1, cutoff=0):
        super().__init__()
        self.dim, self.cutoff = dim, cutoff

    def forward(self, x):
        y_soft = sparsemax(x, self.dim)
        return (y_soft > self.cutoff).float() - y_soft.detach() + y_soft


class Hardmax(torch.nn.Module):
    """util.py:45-56 - thresholded softmax (straight through)."""

    def __init__(self, dim=-1, cutoff=0.2):
        super().__init__()
        self.dim, self.cutoff = dim, cutoff

    def forward(self, x):
        y_soft = torch.softmax(x, self.dim)
        return (y_soft > self.cutoff).float() - y_soft.detach() + y_soft


def diff_or(tensors):
    """util.py:456-465 - differentiable OR of {0,1} tensors: res + t - res*t, left to right."""
    res = torch.zeros_like(tensors[0])
    for t in tensors:
        res = res + t - res * t
    return res


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
    """util.py:89-130: gumbel softmax of a torch.sparse_coo tensor along `dim`, over the stored
    entries of every row (a row = one setting of all other indices).  hard=True keeps, per row,
    only the largest soft entry with its soft value (the reference's torch_scatter.scatter_max:
    first entry on a tie; the value stays differentiable).

    The layout of the accelerated path - [B,N,N] with indices (batch, sink, source), dim = last -
    is already row-contiguous after coalesce(); any other `dim` goes through one stable sort of the
    row keys.  `noise` (test hook): the standard gumbel draws, one per coalesced entry."""
    from . import _ops
    logits = logits.coalesce()
    idx, vals = logits.indices(), logits.values()
    nd = idx.shape[0]
    dim = dim % nd
    E = vals.numel()
    if noise is None:
        noise = -torch.empty_like(vals).exponential_().log()
    if E == 0:
        return torch.sparse_coo_tensor(idx, vals, size=logits.shape)
    key = torch.zeros(E, dtype=torch.long, device=idx.device)
    for d in range(nd):
        if d != dim:
            key = key * logits.shape[d] + idx[d]
    order = None
    if dim != nd - 1:                    # rows are not contiguous in coalesced order
        key, order = torch.sort(key, stable=True)
        vals, noise = vals[order], noise[order]
    first = torch.ones_like(key, dtype=torch.bool)
    first[1:] = key[1:] != key[:-1]
    seg_ptr = torch.cat([first.nonzero().flatten(), torch.tensor([E], device=key.device)])

    class _Rows:
        pass
    rows = _Rows()
    rows.seg_ptr, rows.S, rows.E = seg_ptr.contiguous(), seg_ptr.numel() - 1, E
    tau_t = tau if torch.is_tensor(tau) else torch.tensor([float(tau)], device=vals.device)
    soft = _ops.segment_softmax(vals, tau_t, noise, rows)
    if not hard:
        if order is not None:
            soft = torch.empty_like(soft).index_put((order,), soft)
        return torch.sparse_coo_tensor(idx, soft, size=logits.shape)
    # util.py:110-130: per row the (first) largest entry
    seg = torch.cumsum(first, 0) - 1
    top = torch.full((rows.S,), float("-inf"), device=vals.device).scatter_reduce_(
        0, seg, soft.detach(), "amax", include_self=True)
    pos = torch.arange(E, device=key.device)
    arg = torch.full((rows.S,), E, dtype=torch.long, device=key.device).scatter_reduce_(
        0, seg, torch.where(soft.detach() == top[seg], pos, E), "amin", include_self=True)
    src = arg if order is None else order[arg]
    return torch.sparse_coo_tensor(idx[:, src], soft[arg], size=logits.shape)
