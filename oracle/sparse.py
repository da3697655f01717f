"""CPU restatement of the SparseGCM step and the sparse TemporalEdge selector.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows src/gcm/sparse_gcm.py:55-212, src/gcm/sparse_edge_selectors/temporal.py:11-63
and the helpers src/gcm/util.py:176-240,287-304,426-452.  The hidden adjacency
is a torch.sparse_coo tensor with indices (batch, sink, source), as in the
reference.
"""
import torch

from . import pyg


def initial_hidden(x, graph_size):
    """sparse_gcm.py:55-70."""
    B, _, F = x.shape
    nodes = torch.zeros(B, graph_size, F)
    adj = torch.zeros((B, graph_size, graph_size), layout=torch.sparse_coo)
    return nodes, adj, torch.zeros(B, dtype=torch.long)


def _ragged_arange(start, count):
    """(batch ids, start[b] + 0..count[b]-1) for every b - util.py:176-231
    build these with a Python list comprehension over B."""
    b_ids = torch.repeat_interleave(torch.arange(count.numel()), count)
    offs = torch.cumsum(count, 0) - count
    within = torch.arange(int(count.sum())) - offs[b_ids]
    return b_ids, start[b_ids] + within


def batch_offsets(lengths):
    """util.py:234-240."""
    ends = lengths.cumsum(0)
    return ends - lengths, ends


class TemporalEdge:
    """sparse_edge_selectors/temporal.py:18-63 - for every new node t in
    [T_b, T_b + tau_b) and every hop h: edge (b, sink=t, source=t-h) when
    source >= 0 and sink > 0."""

    def __init__(self, hops=(1,)):
        self.hops = torch.tensor(list(hops))

    def __call__(self, nodes, T, taus, B):
        b_ids, sinks = _ragged_arange(T, taus)
        H = self.hops.numel()
        sink = sinks.repeat_interleave(H)
        src = sink - self.hops.repeat(sinks.numel())
        bat = b_ids.repeat_interleave(H)
        ok = (src >= 0) & (sink > 0)
        idx = torch.stack([bat[ok], sink[ok], src[ok]])
        return torch.sparse_coo_tensor(idx, torch.ones(idx.shape[1]), size=(B, int(1e5), int(1e5)))


def sparse_step(x, taus, hidden, gnn, graph_size=128, edge_selectors=None,
                preprocessor=None, aux_edge_selectors=None, positional_encoder=None,
                max_hops=None):
    """sparse_gcm.py:72-212 - one SparseGCM.forward.
    x [B, t, F] zero padded, taus [B] -> (mx_dense [B, t, H], hidden')."""
    if hidden is None:
        hidden = initial_hidden(x, graph_size)
    nodes, adj, T = hidden
    adj = adj.coalesce()
    B, N = x.shape[0], nodes.shape[1]

    new_b, new_t = _ragged_arange(T, taus)                       # util.py:191-208
    pad_b, pad_t = _ragged_arange(torch.zeros_like(T), taus)     # util.py:176-188
    if int(new_t.max()) >= N:                                    # sparse_gcm.py:120-121
        raise Exception("Overflow")
    nodes = nodes.clone().index_put((new_b, new_t), x[pad_b, pad_t])
    dirty = nodes.clone()

    def merge(adj, sel, seen):
        add = sel(seen, T, taus, B).coalesce()
        idx = torch.cat([adj.indices(), add.indices()], dim=-1)
        val = torch.cat([adj.values(), add.values()], dim=-1)
        return torch.sparse_coo_tensor(idx, val, size=adj.shape).coalesce()

    if edge_selectors is not None:                               # :130-139
        adj = merge(adj, edge_selectors, dirty)
    if preprocessor is not None:
        dirty = preprocessor(dirty)
    if positional_encoder is not None:
        dirty = positional_encoder(dirty, T + taus)
    if aux_edge_selectors is not None:
        adj = merge(adj, aux_edge_selectors, dirty)
    v = adj.values()
    adj = torch.sparse_coo_tensor(adj.indices(), v / v.detach(), size=adj.shape)  # :160-164

    starts, _ = batch_offsets(T + taus)
    live_b, live_t = _ragged_arange(torch.zeros_like(T), T + taus)   # util.py:211-231
    flat_nodes = dirty[live_b, live_t]                               # util.py:426-452
    _, out_idx = _ragged_arange(starts + T, taus)
    ac = adj.coalesce()                                              # util.py:287-304
    flat_edges = ac.indices()[1:] + starts[ac.indices()[0]]
    weights = ac.values()
    edges = torch.flip(flat_edges, (0,))                             # (sink,source)->(source,sink)
    assert torch.all(edges[0] < edges[1]), "Causality violated"
    if edges.numel() > 0:
        edges, weights = pyg.coalesce(edges, weights, num_nodes=int((T + taus).sum()),
                                      reduce="mean")
    if max_hops is None:
        mx = gnn(flat_nodes, edges, weights)[out_idx]
    else:
        sub, sub_edges, node_map, emask = pyg.k_hop_subgraph(
            out_idx, max_hops, edges, relabel_nodes=True, num_nodes=int((T + taus).sum()))
        mx = gnn(flat_nodes[sub], sub_edges, weights[emask])[node_map]
    assert torch.all(torch.isfinite(mx)), "Got NaN in returned memory, try using tanh activation"
    mx_dense = torch.zeros((*x.shape[:-1], mx.shape[-1]), dtype=mx.dtype)   # (dtype-generic: float64 runs bound the tests)
    mx_dense = mx_dense.index_put((pad_b, pad_t), mx)
    return mx_dense, (nodes, adj, T + taus)


def canonical_gnn(F, H, act=torch.nn.Tanh, layers=2):
    """ray_sparse_gcm.py:37-39 / tests/test_sparse_gcm.py:310-323 - GraphConv + act."""
    mods, cin = [], F
    for _ in range(layers):
        mods.append((pyg.GraphConv(cin, H), "x, edges, weights -> x"))
        if act is not None:
            mods.append(act())
        cin = H
    return pyg.Sequential("x, edges, weights", mods)


def pack_hidden(hidden, B, max_edges, edge_fill=-1, weight_fill=1.0):
    """util.py:323-351."""
    nodes, adj, T = hidden
    adj = adj.coalesce()
    idx, val = adj.indices(), adj.values()
    dense_edges = torch.full((B, 2, max_edges), edge_fill, dtype=torch.long)
    dense_weights = torch.full((B, 1, max_edges), weight_fill, dtype=torch.float)
    for b in range(B):
        sel = torch.nonzero(idx[0] == b).reshape(-1)
        assert sel.numel() < max_edges, f"Cannot pack {sel.numel()} edges into {max_edges}"
        dense_edges[b, :, : sel.numel()] = idx[1:, sel]
        dense_weights[b, 0, : sel.numel()] = val[sel]
    return nodes, dense_edges, dense_weights, T


def unpack_hidden(hidden, B):
    """util.py:353-382."""
    nodes, edges, weights, T = hidden
    b_idx, e_idx = (edges[:, 0] >= 0).nonzero().T.unbind()
    idx = torch.stack([b_idx, edges[b_idx, 0, e_idx], edges[b_idx, 1, e_idx]])
    adj = torch.sparse_coo_tensor(idx, weights[b_idx, 0, e_idx], size=(B, nodes.shape[1], nodes.shape[1]))
    return nodes, adj, T


# --------------------------------------------------------------------------
# sparse LearnedEdge (sparse_edge_selectors/learned.py:90-160) - SURVEY 8(f) rank 3
# --------------------------------------------------------------------------
def get_causal_edges(T, taus, window=None):
    """util.py:242-282."""
    out = []
    for b in range(T.numel()):
        t, tau = int(T[b]), int(taus[b])
        edge = torch.tril_indices(t + tau, t + tau, offset=-1)
        if window is not None:
            edge = edge[:, edge[1] >= max(0, t - window)]
        edge = edge[:, edge[0] >= t]
        out.append(torch.cat((torch.full((1, edge.shape[1]), b, dtype=torch.long), edge), dim=0))
    return torch.cat(out, dim=-1)


def sparse_gumbel_softmax(logits, dim, tau=1.0, noise=None, hard=False):
    """util.py:89-130.  hard=True: util.py:110-130 calls torch_scatter.scatter_max (third-party,
    absent from /root/reference and this image; its documented semantics: per group the maximum and
    the index of the FIRST entry attaining it) - restated as a loop over rows; pinned by the
    reference's own known-answer test (tests/test_sparse_gcm.py:795-823)."""
    logits = logits.coalesce()
    g = noise if noise is not None else -torch.empty_like(logits.values()).exponential_().log()
    z = torch.sparse_coo_tensor(logits.indices(), (logits.values() + g) / tau, size=logits.shape)
    soft = torch.sparse.softmax(z, dim=dim).coalesce()
    if not hard:
        return soft
    idx, vals = soft.indices(), soft.values()
    dim = dim % idx.shape[0]
    rows = {}
    for e in range(vals.numel()):
        k = tuple(int(idx[d, e]) for d in range(idx.shape[0]) if d != dim)
        if k not in rows or float(vals[e].detach()) > float(vals[rows[k]].detach()):
            rows[k] = e
    keep = torch.tensor([rows[k] for k in sorted(rows)], dtype=torch.long)
    return torch.sparse_coo_tensor(idx[:, keep], vals[keep], size=logits.shape)


class LearnedEdge:
    """sparse_edge_selectors/learned.py:90-160 (non-deterministic branch); `noise_fn(n)` supplies
    the gumbel draws."""

    def __init__(self, edge_network, num_edge_samples=5, window=None, tau=1.0, noise_fn=None):
        self.net, self.k, self.window, self.tau = edge_network, num_edge_samples, window, tau
        self.noise_fn = noise_fn

    def __call__(self, nodes, T, taus, B):
        N = nodes.shape[1]
        if int((T + taus).max()) <= 1:
            return torch.sparse_coo_tensor(torch.zeros(3, 0, dtype=torch.long), torch.zeros(0),
                                           size=(B, N, N))
        idx = get_causal_edges(T, taus, self.window)
        b, sink, src = idx.unbind()
        logits = self.net(torch.cat((nodes[b, sink], nodes[b, src]), dim=-1)).squeeze()
        gs_in = torch.sparse_coo_tensor(idx, logits, size=(B, N, N))
        noise = self.noise_fn(logits.numel()) if self.noise_fn else None
        soft = sparse_gumbel_softmax(gs_in, 2, self.tau, noise)
        keep = soft.values() > 1 / (1 + self.k)
        v = soft.values()[keep]
        return torch.sparse_coo_tensor(soft.indices()[:, keep], v / v.detach(), size=(B, N, N))
