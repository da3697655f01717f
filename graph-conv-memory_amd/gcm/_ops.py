"""torch.autograd wrappers around the C-ABI kernels (include/gcm_hip.h).

Every function here launches HIP kernels on torch's current stream; none of
them synchronises with the host.
"""
import ctypes

import torch

from . import _ext, _hip

_f32 = torch.float32


class KernelTimer:
    """Optional per-launch timing with HIP events on the launch stream (bench.py uses it
    for the roofline figures).  Disabled (None) in normal operation: zero overhead."""

    def __init__(self):
        self.spans = {}

    def launch(self, name, fn, *args):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        self.spans.setdefault(name, []).append((a, b))
        return rc

    def summary(self):
        """name -> (launches, mean ms).  Call after a device synchronize."""
        out = {}
        for name, spans in self.spans.items():
            ms = [a.elapsed_time(b) for a, b in spans]
            out[name] = (len(ms), sum(ms) / len(ms))
        return out


TIMER = None


def _call(name, *args):
    fn = getattr(_hip.lib(), name)
    rc = TIMER.launch(name, fn, *args) if TIMER is not None else fn(*args)
    _hip.check(rc, name)


def _empty_like(t):
    return torch.empty_like(t, memory_format=torch.contiguous_format)


# ---------------------------------------------------------------------------
# DenseGCM state: insert + overflow wrap (gcm.py:262-278, 323-355)
# ---------------------------------------------------------------------------
class _StateAdvance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nodes, adj, weights, num_nodes, x, flags):
        B, N, F = nodes.shape
        has_w = weights.numel() != 0
        nodes, adj, x = nodes.contiguous(), adj.contiguous(), x.contiguous()
        weights = weights.contiguous()
        _hip.on_device(nodes, adj, weights, num_nodes, x, flags)
        nodes_out, adj_out = _empty_like(nodes), _empty_like(adj)
        weights_out = _empty_like(weights)
        cur = torch.empty_like(num_nodes)
        nn_out = torch.empty_like(num_nodes)
        _call(
            "gcm_state_advance_fwd", _hip.ptr(nodes), _hip.ptr(adj), _hip.ptr(weights) if has_w else None,
            _hip.ptr(num_nodes), _hip.ptr(x), _hip.ptr(nodes_out), _hip.ptr(adj_out),
            _hip.ptr(weights_out) if has_w else None, _hip.ptr(cur), _hip.ptr(nn_out),
            _hip.ptr(flags), B, N, F, _hip.stream())
        ctx.save_for_backward(num_nodes)
        ctx.shape = (B, N, F)
        ctx.has_w = has_w
        ctx.mark_non_differentiable(cur, nn_out)
        if not adj.requires_grad:
            ctx.mark_non_differentiable(adj_out)
        if not weights.requires_grad:
            ctx.mark_non_differentiable(weights_out)
        return nodes_out, adj_out, weights_out, cur, nn_out

    @staticmethod
    def backward(ctx, g_nodes, g_adj, g_weights, _g_cur, _g_nn):
        (num_nodes,) = ctx.saved_tensors
        B, N, F = ctx.shape
        lib = _hip.lib()
        need_nodes, need_adj, need_w, _, need_x, _ = ctx.needs_input_grad
        dev = num_nodes.device
        if g_nodes is None:
            g_nodes = torch.zeros(B, N, F, device=dev)
        g_nodes = g_nodes.contiguous()
        g_nodes_in = torch.empty_like(g_nodes)
        g_x = torch.empty(B, F, device=dev)
        planes = []
        if need_adj and g_adj is not None:
            planes.append(("adj", g_adj.contiguous()))
        if need_w and ctx.has_w and g_weights is not None:
            planes.append(("w", g_weights.contiguous()))
        outs = {}
        first = planes[0] if planes else None
        g_plane_in = torch.empty_like(first[1]) if first else None
        rc = lib.gcm_state_advance_bwd(
            _hip.ptr(g_nodes), _hip.ptr(first[1]) if first else None, _hip.ptr(num_nodes),
            _hip.ptr(g_nodes_in), _hip.ptr(g_plane_in), _hip.ptr(g_x), B, N, F, _hip.stream())
        _hip.check(rc, "gcm_state_advance_bwd")
        if first:
            outs[first[0]] = g_plane_in
        for name, g in planes[1:]:
            scratch_n, scratch_x = torch.empty_like(g_nodes), torch.empty_like(g_x)
            gp = torch.empty_like(g)
            rc = lib.gcm_state_advance_bwd(
                _hip.ptr(g_nodes), _hip.ptr(g), _hip.ptr(num_nodes), _hip.ptr(scratch_n),
                _hip.ptr(gp), _hip.ptr(scratch_x), B, N, F, _hip.stream())
            _hip.check(rc, "gcm_state_advance_bwd")
            outs[name] = gp
        return (g_nodes_in if need_nodes else None, outs.get("adj"), outs.get("w"), None,
                g_x if need_x else None, None)


def state_advance(nodes, adj, weights, num_nodes, x, flags):
    return _StateAdvance.apply(nodes, adj, weights, num_nodes, x, flags)


# ---------------------------------------------------------------------------
# belief row gather + finite flag (gcm.py:309-318)
# ---------------------------------------------------------------------------
class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, cur, flags):
        feats = feats.contiguous()
        _hip.on_device(feats, cur, flags)
        B, N, H = feats.shape
        out = torch.empty(B, H, device=feats.device, dtype=_f32)
        rc = _hip.lib().gcm_gather_rows_fwd(_hip.ptr(feats), _hip.ptr(cur), _hip.ptr(out),
                                            _hip.ptr(flags), B, N, H, _hip.stream())
        _hip.check(rc, "gcm_gather_rows_fwd")
        ctx.save_for_backward(cur)
        ctx.shape = (B, N, H)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (cur,) = ctx.saved_tensors
        B, N, H = ctx.shape
        g_out = g_out.contiguous()
        g_feats = torch.empty(B, N, H, device=g_out.device, dtype=_f32)
        rc = _hip.lib().gcm_gather_rows_bwd(_hip.ptr(g_out), _hip.ptr(cur), _hip.ptr(g_feats),
                                            B, N, H, _hip.stream())
        _hip.check(rc, "gcm_gather_rows_bwd")
        return g_feats, None, None


def gather_rows(feats, cur, flags):
    return _GatherRows.apply(feats, cur, flags)


# ---------------------------------------------------------------------------
# DenseGraphConv (PyG; README.md:56-62)
# ---------------------------------------------------------------------------
class _DenseGraphConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, adj, w_rel, b_rel, w_root, act):
        x, adj = x.contiguous(), adj.contiguous()
        w_rel, w_root = w_rel.contiguous(), w_root.contiguous()
        b_rel = None if b_rel is None else b_rel.contiguous()
        _hip.on_device(x, adj, w_rel, b_rel, w_root)
        B, N, Fi = x.shape
        Fo = w_rel.shape[0]
        assert adj.shape == (B, N, N), "adj must be [B, N, N]"
        assert w_rel.shape == (Fo, Fi) and w_root.shape == (Fo, Fi)
        out = torch.empty(B, N, Fo, device=x.device, dtype=_f32)
        need_bwd = any(ctx.needs_input_grad)
        agg = torch.empty(B, N, Fi, device=x.device, dtype=_f32) if need_bwd else None
        _call(
            "gcm_dense_graphconv_fwd", _hip.ptr(x), _hip.ptr(adj), _hip.ptr(w_rel), _hip.ptr(b_rel), _hip.ptr(w_root),
            _hip.ptr(out), _hip.ptr(agg), B, N, Fi, Fo, act, _hip.stream())
        ctx.save_for_backward(x, adj, w_rel, w_root, out, agg)
        ctx.act = act
        ctx.has_bias = b_rel is not None
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, adj, w_rel, w_root, out, agg = ctx.saved_tensors
        B, N, Fi = x.shape
        Fo = w_rel.shape[0]
        need_x, need_adj, need_wrel, need_b, need_wroot, _ = ctx.needs_input_grad
        need_b = need_b and ctx.has_bias
        g_out = g_out.contiguous()
        dev = x.device
        lib = _hip.lib()
        g_x = torch.empty_like(x) if need_x else None
        g_adj = torch.empty_like(adj) if need_adj else None
        g_wrel = torch.empty_like(w_rel) if need_wrel else None
        g_wroot = torch.empty_like(w_root) if need_wroot else None
        g_b = torch.empty(Fo, device=dev, dtype=_f32) if need_b else None
        ws_bytes = lib.gcm_dense_graphconv_bwd_workspace_bytes(B, N, Fi, Fo)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _call(
            "gcm_dense_graphconv_bwd", _hip.ptr(g_out), _hip.ptr(out), _hip.ptr(x), _hip.ptr(adj), _hip.ptr(agg),
            _hip.ptr(w_rel), _hip.ptr(w_root), _hip.ptr(g_x), _hip.ptr(g_adj), _hip.ptr(g_wrel),
            _hip.ptr(g_b), _hip.ptr(g_wroot), _hip.ptr(ws), ws_bytes, B, N, Fi, Fo, ctx.act,
            _hip.stream())
        return g_x, g_adj, g_wrel, g_b, g_wroot, None


def dense_graphconv(x, adj, w_rel, b_rel, w_root, act=_hip.ACT_NONE):
    return _DenseGraphConv.apply(x, adj, w_rel, b_rel, w_root, act)


# ---------------------------------------------------------------------------
# index-writing / distance selectors: in place on a fresh adjacency buffer
# ---------------------------------------------------------------------------
def edge_temporal_(adj, cur, hops, direction):
    _hip.on_device(adj, cur)
    B, N, _ = adj.shape
    arr = (ctypes.c_int32 * len(hops))(*hops)
    rc = _hip.lib().gcm_edge_temporal(_hip.ptr(adj), _hip.ptr(cur), ctypes.addressof(arr),
                                      len(hops), _hip.DIR[direction], B, N, _hip.stream())
    _hip.check(rc, "gcm_edge_temporal")
    return adj


class _TemporalWindow(torch.autograd.Function):
    """TemporalBackedge(learned=True) (temporal.py:51-70): adj[b, n_b, :n_b] += the OR of the sampled
    one-hots, in place on `adj` like the reference's slice assignment."""

    @staticmethod
    def forward(ctx, adj, window, cur, noise, S, deterministic, flags):
        window = window.contiguous()
        _hip.on_device(adj, window, cur, noise, flags)
        B, N, _ = adj.shape
        W = window.numel()
        Wn = min(W, N)
        soft = torch.empty(1 if deterministic else S, B, Wn, device=adj.device, dtype=_f32)
        scratch = torch.empty(B, Wn, device=adj.device, dtype=_f32)
        _call("gcm_temporal_window_fwd", _hip.ptr(window), _hip.ptr(noise), _hip.ptr(cur), _hip.ptr(adj),
              _hip.ptr(soft), _hip.ptr(scratch), B, N, W, S, int(deterministic), _hip.ptr(flags), _hip.stream())
        ctx.mark_dirty(adj)
        ctx.save_for_backward(soft, cur)
        ctx.cfg = (W, S, int(deterministic))
        return adj

    @staticmethod
    def backward(ctx, g_adj):
        soft, cur = ctx.saved_tensors
        W, S, det = ctx.cfg
        _, B, Wn = soft.shape
        g_adj = g_adj.contiguous()
        part = torch.empty(B, Wn, device=g_adj.device, dtype=_f32)
        _call("gcm_temporal_window_bwd", _hip.ptr(g_adj), _hip.ptr(soft), _hip.ptr(cur), _hip.ptr(part),
              B, g_adj.shape[-1], W, S, det, _hip.stream())
        g_window = part.sum(0)
        if Wn < W:
            g_window = torch.cat([g_window, g_window.new_zeros(W - Wn)])
        return g_adj, g_window, None, None, None, None, None


def temporal_window_(adj, window, cur, noise, num_samples, deterministic, flags):
    """noise: [S, B, min(W, N)] standard gumbel draws (None when deterministic)"""
    if noise is not None:
        noise = noise.contiguous()
    return _TemporalWindow.apply(adj, window, cur, noise, num_samples, deterministic, flags)


def edge_dense_(adj, cur):
    _hip.on_device(adj, cur)
    B, N, _ = adj.shape
    rc = _hip.lib().gcm_edge_dense(_hip.ptr(adj), _hip.ptr(cur), B, N, _hip.stream())
    _hip.check(rc, "gcm_edge_dense")
    return adj


def edge_distance_(nodes, adj, cur, mode, max_distance, dist_param=None, a=(0, 0), b=(0, 0),
                   bidirectional=False, want_dist=False, cur_rows=None):
    """cur_rows [n, F]: the current nodes of every rank (all-gathered) when a batch-sharded
    EuclideanEdge takes its mean over the global batch; None: the local graphs' own."""
    nodes = nodes.contiguous()
    _hip.on_device(nodes, adj, cur, dist_param, cur_rows)
    B, N, F = nodes.shape
    lib = _hip.lib()
    n_cur = 0 if cur_rows is None else cur_rows.shape[0]
    ws_bytes = lib.gcm_edge_distance_workspace_bytes(mode, max(B, n_cur), N, F)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=nodes.device)
    dist = torch.empty(B, N, device=nodes.device, dtype=_f32) if want_dist else None
    rc = lib.gcm_edge_distance_ex(_hip.ptr(nodes), _hip.ptr(adj), _hip.ptr(cur), mode,
                                  float(max_distance), _hip.ptr(dist_param), a[0], a[1], b[0], b[1],
                                  int(bidirectional), _hip.ptr(dist), _hip.ptr(cur_rows), n_cur, _hip.ptr(ws),
                                  ws_bytes, B, N, F, _hip.stream())
    _hip.check(rc, "gcm_edge_distance")
    return adj, dist


# ===========================================================================
# sparse path (sparse_gcm.py:72-212)
# ===========================================================================
_i64 = torch.int64


def _hops_arg(hops):
    return (ctypes.c_int32 * len(hops))(*hops)


def sparse_plan(T, taus):
    """-> node_off [B+1], new_off [B+1], totals [4] (all device int64); no host sync."""
    _hip.on_device(T, taus)
    B = T.numel()
    node_off = torch.empty(B + 1, dtype=_i64, device=T.device)
    new_off = torch.empty(B + 1, dtype=_i64, device=T.device)
    totals = torch.empty(4, dtype=_i64, device=T.device)
    _call("gcm_sparse_plan", _hip.ptr(T), _hip.ptr(taus), _hip.ptr(node_off), _hip.ptr(new_off),
          _hip.ptr(totals), B, _hip.stream())
    return node_off, new_off, totals


class _SparseInsert(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nodes, x, T, taus, flags):
        nodes, x = nodes.contiguous(), x.contiguous()
        _hip.on_device(nodes, x, T, taus, flags)
        B, N, F = nodes.shape
        t_pad = x.shape[1]
        out = torch.empty_like(nodes)
        _call("gcm_sparse_insert_fwd", _hip.ptr(nodes), _hip.ptr(x), _hip.ptr(T), _hip.ptr(taus),
              _hip.ptr(out), _hip.ptr(flags), B, N, F, t_pad, _hip.stream())
        ctx.save_for_backward(T, taus)
        ctx.dims = (B, N, F, t_pad)
        return out

    @staticmethod
    def backward(ctx, g_out):
        T, taus = ctx.saved_tensors
        B, N, F, t_pad = ctx.dims
        g_out = g_out.contiguous()
        g_nodes = torch.empty_like(g_out)
        g_x = torch.empty(B, t_pad, F, device=g_out.device, dtype=_f32)
        _call("gcm_sparse_insert_bwd", _hip.ptr(g_out), _hip.ptr(T), _hip.ptr(taus),
              _hip.ptr(g_nodes), _hip.ptr(g_x), B, N, F, t_pad, _hip.stream())
        return g_nodes, g_x, None, None, None


def sparse_insert(nodes, x, T, taus, flags):
    return _SparseInsert.apply(nodes, x, T, taus, flags)


def sparse_temporal_count(T, taus, hops_desc):
    """edge_off [B+1] of the TemporalEdge selector (closed form per graph); no host sync."""
    _hip.on_device(T, taus)
    B = T.numel()
    edge_off = torch.empty(B + 1, dtype=_i64, device=T.device)
    arr = _hops_arg(hops_desc)
    _call("gcm_sparse_temporal_count", _hip.ptr(T), _hip.ptr(taus), ctypes.addressof(arr),
          len(hops_desc), _hip.ptr(edge_off), B, _hip.stream())
    return edge_off


def coo_merge_segments(old_idx, old_val, new_idx, new_val, new_bptr, B, flags):
    """sparse_gcm.py:132-139 without a sort, for new entries that sort behind the stored ones of
    their graph (gcm_coo_merge_segments).  -> (idx [3, Ea+Eb], values)."""
    old_idx, new_idx = old_idx.contiguous(), new_idx.contiguous()
    Ea, Eb = old_idx.shape[1], new_idx.shape[1]
    dev = new_idx.device
    if Eb == 0:
        return old_idx, old_val
    if Ea == 0:
        return new_idx, new_val
    old_bptr = ptr_from_sorted(old_idx[0], B) if Ea else torch.zeros(B + 1, dtype=_i64, device=dev)
    if new_bptr is None:
        new_bptr = ptr_from_sorted(new_idx[0], B)
    out_idx = torch.empty(3, Ea + Eb, dtype=_i64, device=dev)
    grad = old_val.requires_grad or new_val.requires_grad
    out_val = None if grad else torch.empty(Ea + Eb, dtype=_f32, device=dev)
    perm = torch.empty(Ea + Eb, dtype=_i64, device=dev) if grad else None
    _call("gcm_coo_merge_segments", _hip.ptr(old_idx) if Ea else None, _hip.ptr(new_idx) if Eb else None,
          None if grad else _hip.ptr(old_val.contiguous()), None if grad else _hip.ptr(new_val.contiguous()),
          _hip.ptr(old_bptr), _hip.ptr(new_bptr), _hip.ptr(out_idx), _hip.ptr(out_val), _hip.ptr(perm),
          _hip.ptr(flags), Ea, Eb, B, _hip.stream())
    if grad:
        out_val = torch.cat([old_val, new_val])[perm]
    return out_idx, out_val


def sparse_temporal_edges(T, taus, hops_desc, edge_off=None, E=None):
    """COO indices [3, E] (batch, sink, source) of the TemporalEdge selector, already in
    coalesced order.  One host sync (E) unless the caller already knows E."""
    _hip.on_device(T, taus)
    B = T.numel()
    arr = _hops_arg(hops_desc)
    if edge_off is None:
        edge_off = sparse_temporal_count(T, taus, hops_desc)
    if E is None:
        E = int(edge_off[B].item())
    idx = torch.empty(3, E, dtype=_i64, device=T.device)
    _call("gcm_sparse_temporal_fill", _hip.ptr(T), _hip.ptr(taus), ctypes.addressof(arr),
          len(hops_desc), _hip.ptr(edge_off), _hip.ptr(idx), E, B, _hip.stream())
    return idx


class _SparseFlatten(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nodes, T, taus, node_off, M):
        nodes = nodes.contiguous()
        _hip.on_device(nodes, T, taus, node_off)
        B, N, F = nodes.shape
        flat = torch.empty(M, F, device=nodes.device, dtype=_f32)
        _call("gcm_sparse_flatten_fwd", _hip.ptr(nodes), _hip.ptr(T), _hip.ptr(taus),
              _hip.ptr(node_off), _hip.ptr(flat), B, N, F, M, _hip.stream())
        ctx.save_for_backward(T, taus, node_off)
        ctx.dims = (B, N, F, M)
        return flat

    @staticmethod
    def backward(ctx, g_flat):
        T, taus, node_off = ctx.saved_tensors
        B, N, F, M = ctx.dims
        g_flat = g_flat.contiguous()
        g_nodes = torch.empty(B, N, F, device=g_flat.device, dtype=_f32)
        _call("gcm_sparse_flatten_bwd", _hip.ptr(g_flat), _hip.ptr(T), _hip.ptr(taus),
              _hip.ptr(node_off), _hip.ptr(g_nodes), B, N, F, M, _hip.stream())
        return g_nodes, None, None, None, None


def sparse_flatten(nodes, T, taus, node_off, M):
    return _SparseFlatten.apply(nodes, T, taus, node_off, M)


def ptr_from_sorted(keys, M):
    keys = keys.contiguous()
    ptr = torch.empty(M + 1, dtype=_i64, device=keys.device)
    _call("gcm_ptr_from_sorted", _hip.ptr(keys), _hip.ptr(ptr), keys.numel(), M, _hip.stream())
    return ptr


class GraphIndex:
    """Device-resident index of one flat edge list: CSR by destination (forward gather) and,
    built lazily, CSC by source (backward gather).  Attached to the edge_index tensor handed
    to the GNN (attribute `gcm_graph`) so GraphConv layers share it."""

    def __init__(self, edge_index, row_ptr, M, csr_perm=None, mask=None, batches=None):
        self.edge_index = edge_index            # [2, E] (source, sink); CSR order unless csr_perm
        self.csr_perm = csr_perm                # positions of the CSR entries in edge_index
        self.row_ptr, self.M, self.mask = row_ptr, M, mask
        # (node_off [B+1], B, max nodes per graph) when the list is grouped by graph and no edge
        # leaves its graph (SparseGCM's flat list): the CSC view is then built without a sort
        self.batches = batches
        src = edge_index[0] if csr_perm is None else edge_index[0][csr_perm]
        self.col = src.contiguous()
        self._csc = None

    @property
    def E(self):
        return self.col.numel()

    def dst_csr(self):
        d = self.edge_index[1] if self.csr_perm is None else self.edge_index[1][self.csr_perm]
        return d.contiguous()

    def csc(self):
        """(col_ptr [M+1], rows [E], perm [E]): entry k of the CSC is CSR entry perm[k]."""
        if self._csc is None:
            bt = self.batches
            if bt is not None and self.csr_perm is None and bt[2] <= 8192:
                E, dev = self.E, self.col.device
                col_ptr = torch.empty(self.M + 1, dtype=_i64, device=dev)
                rows = torch.empty(E, dtype=_i64, device=dev)
                perm = torch.empty(E, dtype=_i64, device=dev)
                _call("gcm_csc_from_csr_batched", _hip.ptr(self.row_ptr), _hip.ptr(self.col),
                      _hip.ptr(self.dst_csr()), _hip.ptr(bt[0]), _hip.ptr(col_ptr), _hip.ptr(rows),
                      _hip.ptr(perm), bt[1], self.M, E, bt[2], _hip.stream())
                self._csc = (col_ptr, rows, perm)
            else:
                src_sorted, perm = torch.sort(self.col, stable=True)
                rows = self.dst_csr()[perm].contiguous()
                self._csc = (ptr_from_sorted(src_sorted, self.M), rows, perm.contiguous())
        return self._csc

    @staticmethod
    def from_edge_index(edge_index, M):
        """Generic entry: any [2, E] (source, sink) list -> CSR by destination."""
        dst_sorted, perm = torch.sort(edge_index[1], stable=True)
        return GraphIndex(edge_index, ptr_from_sorted(dst_sorted, M), M, csr_perm=perm)


def sparse_edges_to_csr(coo, node_off, M, B, flags, n_cap=None):
    """coo [3,E] sorted (batch, sink, source) -> (edge_index [2,E] (source, sink), GraphIndex).
    n_cap: the graph size (bound on the nodes of one graph), when known."""
    coo = coo.contiguous()
    _hip.on_device(coo, node_off, flags)
    E = coo.shape[1]
    edge_index = torch.empty(2, E, dtype=_i64, device=coo.device)
    row_ptr = torch.empty(M + 1, dtype=_i64, device=coo.device)
    _call("gcm_sparse_edges_to_csr", _hip.ptr(coo), _hip.ptr(node_off), _hip.ptr(edge_index),
          _hip.ptr(row_ptr), _hip.ptr(flags), E, M, B, _hip.stream())
    return edge_index, GraphIndex(edge_index, row_ptr, M,
                                  batches=None if n_cap is None else (node_off, B, int(n_cap)))


def khop_mask(graph, node_off, T, taus, hops, B, t_pad):
    M = graph.M
    mask = torch.empty(M, dtype=torch.uint8, device=graph.row_ptr.device)
    scratch = torch.empty(2 * M, dtype=torch.uint8, device=mask.device)
    _call("gcm_khop_mask", _hip.ptr(graph.row_ptr), _hip.ptr(graph.col), _hip.ptr(node_off),
          _hip.ptr(T), _hip.ptr(taus), hops, _hip.ptr(mask), _hip.ptr(scratch), M, B, t_pad,
          _hip.stream())
    return mask


class _SparseExtract(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, T, taus, node_off, B, t_pad, flags):
        feats = feats.contiguous()
        M, H = feats.shape
        out = torch.empty(B, t_pad, H, device=feats.device, dtype=_f32)
        _call("gcm_sparse_extract_fwd", _hip.ptr(feats), _hip.ptr(T), _hip.ptr(taus),
              _hip.ptr(node_off), _hip.ptr(out), _hip.ptr(flags), B, t_pad, H, M, _hip.stream())
        ctx.save_for_backward(T, taus, node_off)
        ctx.dims = (B, t_pad, H, M)
        return out

    @staticmethod
    def backward(ctx, g_out):
        T, taus, node_off = ctx.saved_tensors
        B, t_pad, H, M = ctx.dims
        g_out = g_out.contiguous()
        g_feats = torch.empty(M, H, device=g_out.device, dtype=_f32)
        _call("gcm_sparse_extract_bwd", _hip.ptr(g_out), _hip.ptr(T), _hip.ptr(taus),
              _hip.ptr(node_off), _hip.ptr(g_feats), B, t_pad, H, M, _hip.stream())
        return g_feats, None, None, None, None, None, None


def sparse_extract(feats, T, taus, node_off, B, t_pad, flags):
    return _SparseExtract.apply(feats, T, taus, node_off, B, t_pad, flags)


class _CsrGraphConv(torch.autograd.Function):
    """x [M,Fi]; w_edge [E] in CSR order or None."""

    @staticmethod
    def forward(ctx, x, w_edge, w_rel, b_rel, w_root, graph, act):
        x = x.contiguous()
        w_rel, w_root = w_rel.contiguous(), w_root.contiguous()
        b_rel = None if b_rel is None else b_rel.contiguous()
        w_edge = None if w_edge is None else w_edge.contiguous()
        _hip.on_device(x, w_edge, w_rel, b_rel, w_root)
        M, Fi = x.shape
        Fo = w_rel.shape[0]
        assert M == graph.M
        out = torch.empty(M, Fo, device=x.device, dtype=_f32)
        need_bwd = any(ctx.needs_input_grad)
        agg = torch.empty(M, Fi, device=x.device, dtype=_f32) if need_bwd else None
        _call("gcm_csr_graphconv_fwd", _hip.ptr(x), _hip.ptr(graph.row_ptr), _hip.ptr(graph.col),
              _hip.ptr(w_edge), _hip.ptr(graph.mask), _hip.ptr(w_rel), _hip.ptr(b_rel),
              _hip.ptr(w_root), _hip.ptr(out), _hip.ptr(agg), M, Fi, Fo, act, _hip.stream())
        ctx.save_for_backward(x, w_edge, w_rel, w_root, out, agg)
        ctx.graph, ctx.act, ctx.has_bias = graph, act, b_rel is not None
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, w_edge, w_rel, w_root, out, agg = ctx.saved_tensors
        graph = ctx.graph
        M, Fi = x.shape
        Fo = w_rel.shape[0]
        need_x, need_we, need_wrel, need_b, need_wroot, _, _ = ctx.needs_input_grad
        need_b = need_b and ctx.has_bias
        need_we = need_we and w_edge is not None
        g_out = g_out.contiguous()
        dev = x.device
        lib = _hip.lib()
        E = graph.E
        col_ptr = rows = perm = None
        if (need_x or need_we) and E > 0:
            col_ptr, rows, perm = graph.csc()
        g_x = torch.empty_like(x) if need_x else None
        g_we = torch.zeros_like(w_edge) if need_we else None
        g_wrel = torch.empty_like(w_rel) if need_wrel else None
        g_wroot = torch.empty_like(w_root) if need_wroot else None
        g_b = torch.empty(Fo, device=dev, dtype=_f32) if need_b else None
        ws_bytes = lib.gcm_csr_graphconv_bwd_workspace_bytes(M, Fi, Fo)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _call("gcm_csr_graphconv_bwd", _hip.ptr(g_out), _hip.ptr(out), _hip.ptr(x), _hip.ptr(agg),
              _hip.ptr(graph.row_ptr), _hip.ptr(graph.col), _hip.ptr(col_ptr), _hip.ptr(rows),
              _hip.ptr(perm), _hip.ptr(w_edge), _hip.ptr(graph.mask), _hip.ptr(w_rel),
              _hip.ptr(w_root), _hip.ptr(g_x), _hip.ptr(g_we), _hip.ptr(g_wrel), _hip.ptr(g_b),
              _hip.ptr(g_wroot), _hip.ptr(ws), ws_bytes, M, E, Fi, Fo, ctx.act, _hip.stream())
        return g_x, g_we, g_wrel, g_b, g_wroot, None, None


def csr_graphconv(x, w_edge, w_rel, b_rel, w_root, graph, act=_hip.ACT_NONE):
    return _CsrGraphConv.apply(x, w_edge, w_rel, b_rel, w_root, graph, act)


# ===========================================================================
# LearnedEdge (edge_selectors/learned.py:53-125)
# ===========================================================================
class _LearnedPairs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nodes, cur):
        nodes = nodes.contiguous()
        _hip.on_device(nodes, cur)
        B, N, F = nodes.shape
        pairs = torch.empty(B, N, 2 * F, device=nodes.device, dtype=_f32)
        _call("gcm_learned_pairs_fwd", _hip.ptr(nodes), _hip.ptr(cur), _hip.ptr(pairs), B, N, F,
              _hip.stream())
        ctx.save_for_backward(cur)
        ctx.dims = (B, N, F)
        return pairs

    @staticmethod
    def backward(ctx, g_pairs):
        (cur,) = ctx.saved_tensors
        B, N, F = ctx.dims
        g_pairs = g_pairs.contiguous()
        g_nodes = torch.empty(B, N, F, device=g_pairs.device, dtype=_f32)
        _call("gcm_learned_pairs_bwd", _hip.ptr(g_pairs), _hip.ptr(cur), _hip.ptr(g_nodes), B, N, F,
              _hip.stream())
        return g_nodes, None


def learned_pairs(nodes, cur):
    return _LearnedPairs.apply(nodes, cur)


class _LearnedSelect(torch.autograd.Function):
    """new_adj = adj with row cur rewritten (in place on `adj`, which the caller owns)."""

    @staticmethod
    def forward(ctx, adj, logits, noise, cur, cutoff):
        logits, noise = logits.contiguous(), noise.contiguous()
        _hip.on_device(adj, logits, noise, cur)
        B, N, _ = adj.shape
        soft = torch.empty(B, N, device=adj.device, dtype=_f32)
        _call("gcm_learned_select_fwd", _hip.ptr(logits), _hip.ptr(noise), _hip.ptr(cur),
              float(cutoff), _hip.ptr(adj), _hip.ptr(soft), B, N, _hip.stream())
        ctx.mark_dirty(adj)
        ctx.save_for_backward(soft, cur)
        return adj

    @staticmethod
    def backward(ctx, g_adj):
        soft, cur = ctx.saved_tensors
        B, N = soft.shape
        g_adj = g_adj.contiguous()
        g_logits = torch.empty(B, N, device=g_adj.device, dtype=_f32)
        _call("gcm_learned_select_bwd", _hip.ptr(g_adj), _hip.ptr(soft), _hip.ptr(cur),
              _hip.ptr(g_logits), B, N, _hip.stream())
        # both straight-through estimators are identities, so the incoming adjacency receives
        # g_adj unchanged - also at the rewritten entries (learned.py:108-110 adds adj inside
        # the STE)
        return g_adj, g_logits, None, None, None


def learned_select_(adj, logits, noise, cur, cutoff):
    return _LearnedSelect.apply(adj, logits, noise, cur, cutoff)


# ===========================================================================
# fused canonical step / time-batched rollout (csrc/fused.hip, csrc/rollout.hip)
# ===========================================================================
def gnn2_supported(N, F, H1, H2):
    return bool(_hip.lib().gcm_dense_gnn2_row_supported(N, F, H1, H2))


class StepConfig:
    """Everything static about a fused DenseGCM configuration (built once per module/shape)."""

    def __init__(self, descs, acts, has_bias, N, F, H1, H2, device):
        lib = _hip.lib()
        self.descs = descs
        self.arr = (_hip.SelectorDesc * max(1, len(descs)))(*descs)
        self.arr_ptr = ctypes.addressof(self.arr)
        self.n_desc = len(descs)
        self.acts, self.has_bias = acts, has_bias
        self.N, self.F, self.H1, self.H2 = N, F, H1, H2
        self.P = lib.gcm_dense_gnn2_param_count(F, H1, H2)
        self.ws_bytes = 0
        self.ws = None
        self.device = device
        self._layouts = {}
        self.fn_fwd = lib.gcm_dense_step_fwd
        self.fn_bwd = lib.gcm_dense_step_bwd
        self.has_distance = any(d.kind == _hip.SEL_DISTANCE for d in descs)
        self._cpp, self._cpp_handle, self._cpp_call = None, None, None
        # the live-row step (rows_step.hip): index-writing selectors (<= 16 hops in all) and at most
        # one distance selector, which then runs ahead of the step on the incoming state
        n_dist = sum(1 for d in descs if d.kind == _hip.SEL_DISTANCE)
        self.rows_ok = bool(
            all(d.kind in (_hip.SEL_TEMPORAL, _hip.SEL_DENSE, _hip.SEL_DISTANCE) for d in descs)
            and n_dist <= 1 and not any(d.kind == _hip.SEL_DISTANCE and d.bidirectional for d in descs)
            and sum(d.n_hops for d in descs if d.kind == _hip.SEL_TEMPORAL) <= 16
            and lib.gcm_dense_rows_supported(N, F, H1, H2)
            and _ext.module() is not None and hasattr(_ext.module(), "RowsFast"))
        self._rows_fast = None
        # ... and its form that also differentiates w.r.t. the observations / the incoming nodes
        self.dx_ok = bool(self.rows_ok and lib.gcm_dense_rows_dx_supported(N, F, H1, H2))

    # (Linear preprocessor | None, PositionalEncoding in "add" mode | None) folded into the live-row
    # step, or None (gcm.py:_fold_config)
    fold = None
    # Distance selectors of a batch-sharded run (EuclideanEdge(shard_group=...)): their current rows are
    # all-gathered ahead of every step (DenseGCM._gather_sharded)
    sharded = ()

    # -- DenseGCM + LearnedEdge (csrc/learned_step.hip) ----------------------------------------
    learned_sel = None          # the LearnedEdge module when this config is the fused learned step

    def set_learned(self, sel, mods):
        """sel: LearnedEdge with the default edge network `mods` (default_edge_network)."""
        lib = _hip.lib()
        self.learned_sel, self.mlp_mods = sel, mods
        self.Pm = lib.gcm_learned_mlp_param_count(self.F)
        self.P_total = self.P + self.Pm
        self.eps = (float(mods[2].eps), float(mods[5].eps))
        self.cutoff = 1.0 / (1 + sel.num_edge_samples)
        self._zero_chain, self._zero_params = {}, {}
        self.rows_ok = False

    def learned_cpp_handle(self):
        """address of the C++ twin of the learned-step config (0 when the extension lacks it)"""
        h = getattr(self, "_learned_cpp_h", None)
        if h is None:
            ext = _ext.module()
            if ext is None or not hasattr(ext, "LearnedCfg") or TIMER is not None:
                h = 0
            else:
                self._learned_cpp = ext.LearnedCfg(self.N, self.F, self.H1, self.H2, self.acts[0], self.acts[1],
                                                   self.has_bias, self.eps[0], self.eps[1], self.cutoff)
                h = self._learned_cpp.handle()
            self._learned_cpp_h = h
        return h

    def zero_chain(self, B, dev):
        """the proxy whose GRADIENT is the adjacency-gradient chain buffer: a [B,N,N] view of one
        zero (no memory)"""
        z = self._zero_chain.get(dev)
        if z is None:
            z = torch.zeros(1, 1, 1, device=dev)
            self._zero_chain[dev] = z
        return z.expand(B, self.N, self.N)       # a fresh view object per call (it gets a grad_fn)

    def zero_params(self, dev):
        z = self._zero_params.get(dev)
        if z is None:
            z = torch.zeros(self.P_total, device=dev)
            self._zero_params[dev] = z
        return z

    _learned_fast = None

    def learned_fast(self, module):
        """The host path of a continuing LearnedEdge chain (C++: LearnedFast in csrc/torch_ext/step_ext.cpp), one per
        configuration - RowsFast's twin: parameters read through the modules' own `_parameters` dicts, the hook dicts
        torch.nn.Module.__call__ would consult, the selector's `__dict__` (an injected `noise_fn` declines)."""
        f = self._learned_fast
        if f is None:
            rel0, root0, rel1, root1 = self.lins
            specs = [(rel0._parameters, "weight"), (root0._parameters, "weight"), (rel0._parameters, "bias"),
                     (rel1._parameters, "weight"), (root1._parameters, "weight"), (rel1._parameters, "bias")]
            l0, _, n0, l1, _, n1, l2 = self.mlp_mods
            for m in (l0, n0, l1, n1, l2):
                specs += [(m._parameters, "weight"), (m._parameters, "bias")]
            from torch.nn.modules import module as M
            hooks = [module._forward_hooks, module._forward_pre_hooks, module._backward_hooks,
                     module._backward_pre_hooks, M._global_backward_pre_hooks, M._global_backward_hooks,
                     M._global_forward_pre_hooks, M._global_forward_hooks,
                     M._global_forward_hooks_always_called, M._global_forward_hooks_with_kwargs]
            f = self._learned_fast = _ext.module().LearnedFast(specs, hooks, self.learned_sel.__dict__)
        return f

    def rows_fast(self, module):
        """The host path of the live-row step (C++: RowsFast in csrc/torch_ext/step_ext.cpp), one per
        configuration: it validates a continuing chain by itself - hidden state returned by the
        previous call, same parameter objects at the same versions (read through the modules' own
        `_parameters` dicts) - and runs the step without the interpreter."""
        f = self._rows_fast
        if f is None:
            rel0, root0, rel1, root1 = self.lins
            specs = [(rel0._parameters, "weight"), (root0._parameters, "weight"), (rel0._parameters, "bias"),
                     (rel1._parameters, "weight"), (root1._parameters, "weight"), (rel1._parameters, "bias")]
            if self.fold is not None and self.fold[0] is not None:
                specs += [(self.fold[0]._parameters, "weight"), (self.fold[0]._parameters, "bias")]
            # the dicts torch.nn.Module.__call__ would consult: with any of them non-empty the unchecked
            # entry declines and the call goes through nn.Module.__call__
            from torch.nn.modules import module as M
            hooks = [module._forward_hooks, module._forward_pre_hooks, module._backward_hooks,
                     module._backward_pre_hooks, M._global_backward_pre_hooks, M._global_backward_hooks,
                     M._global_forward_pre_hooks, M._global_forward_hooks,
                     M._global_forward_hooks_always_called, M._global_forward_hooks_with_kwargs]
            f = self._rows_fast = _ext.module().RowsFast(specs, hooks)
        return f

    def cpp_handle(self):
        """address of the C++ twin of this config (0 when the torch extension is not built)"""
        h = self._cpp_handle
        if h is None:
            ext = _ext.module()
            if ext is None:
                h = 0
            else:
                self._cpp = ext.StepCfg(self.arr_ptr, self.n_desc, self.acts[0], self.acts[1],
                                        self.has_bias, self.N, self.F, self.H1, self.H2)
                h = self._cpp.handle()
            self._cpp_handle = h
        return h

    def cpp_call(self):
        """(ext.fused_step, config handle, device index) or None when the extension is not built"""
        c = self._cpp_call
        if c is None:
            h = self.cpp_handle()
            c = (_ext.module().fused_step, h, self.device.index) if h else False
            self._cpp_call = c
        return c or None

    def refresh_pointers(self):
        """re-read the device pointers baked into the selector descriptors (a re-assigned
        Distance.dist_param, the gathered current rows of a sharded EuclideanEdge) into this config and
        its C++ twin"""
        for i, (d, src) in enumerate(zip(self.descs, getattr(self, "desc_sources", ()))):
            if d.kind == _hip.SEL_DISTANCE and src is not None:
                p, rows, n_rows = src()
                d.dist_param = p
                self.arr[i].dist_param = p
                d.cur_rows, d.n_cur_rows = rows, n_rows
                self.arr[i].cur_rows, self.arr[i].n_cur_rows = rows, n_rows
        if self._cpp is not None:
            self._cpp.update_descs(self.arr_ptr, self.n_desc)

    def workspace(self, B):
        if not self.has_distance:
            return None, 0
        need = 0
        for d in self.descs:
            if d.kind == _hip.SEL_DISTANCE:
                need = max(need, _hip.lib().gcm_edge_distance_workspace_bytes(d.mode, max(B, d.n_cur_rows),
                                                                               self.N, self.F))
        if need > self.ws_bytes:
            self.ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self.ws_bytes = need
        return (self.ws.data_ptr() if self.ws is not None else None), self.ws_bytes

    def layout(self, B, need_bwd):
        """(total floats, float offsets of nodes|adj|mx|h1|agg1|agg2, bwd layout) for a batch size"""
        key = (B, need_bwd)
        lay = self._layouts.get(key)
        if lay is None:
            N, F, H1, H2, P = self.N, self.F, self.H1, self.H2, self.P
            n_nodes, n_adj, n_mx = _pad64(B * N * F), _pad64(B * N * N), _pad64(B * H2)
            n_h1, n_agg2 = _pad64(B * N * H1), _pad64(B * H1)
            o_adj = n_nodes
            o_mx = o_adj + n_adj
            o_h1 = o_mx + n_mx
            o_agg1 = o_h1 + n_h1
            o_agg2 = o_agg1 + n_nodes
            total = (o_agg2 + n_agg2) if need_bwd else o_h1
            n_obs, n_p = _pad64(B * F), _pad64(P)
            bwd = (n_nodes + n_obs + n_p + B * P, n_nodes, n_nodes + n_obs, n_nodes + n_obs + n_p)
            lay = (total, o_adj, o_mx, o_h1, o_agg1, o_agg2, bwd)
            self._layouts[key] = lay
        return lay

    def unpack_ptrs(self, packed):
        """device pointers (w_rel1, b1, w_root1, w_rel2, b2, w_root2) into the packed vector"""
        F, H1, H2 = self.F, self.H1, self.H2
        base = packed.data_ptr()
        w_rel1 = base
        w_root1 = w_rel1 + 4 * H1 * F
        b1 = w_root1 + 4 * H1 * F
        w_rel2 = b1 + 4 * H1
        w_root2 = w_rel2 + 4 * H2 * H1
        b2 = w_root2 + 4 * H2 * H1
        return (w_rel1, b1 if self.has_bias & 1 else None, w_root1,
                w_rel2, b2 if self.has_bias & 2 else None, w_root2)


def _pad64(n):
    return (n + 63) & ~63


class _FusedStep(torch.autograd.Function):
    """One DenseGCM step as ONE autograd node and ONE C call per direction
    (gcm_dense_step_fwd / gcm_dense_step_bwd): (obs, nodes_in, packed params) ->
    (mx, nodes_out, adj_out, cur, count_out).  Outputs and the activations saved for backward
    live in one allocation.  (This is the per-step hot loop of the host: no helper layers.)"""

    @staticmethod
    def forward(ctx, obs, nodes_in, packed, adj_in, count_in, flags, cfg):
        if not (obs.is_contiguous() and nodes_in.is_contiguous() and adj_in.is_contiguous()):
            obs, nodes_in, adj_in = obs.contiguous(), nodes_in.contiguous(), adj_in.contiguous()
        B = obs.shape[0]
        N, F, H2 = cfg.N, cfg.F, cfg.H2
        need = ctx.needs_input_grad
        need_bwd = need[0] or need[1] or need[2]
        total, o_adj, o_mx, o_h1, o_agg1, o_agg2, _ = cfg.layout(B, need_bwd)
        buf = torch.empty(total, device=obs.device, dtype=_f32)
        ibuf = torch.empty(2, B, device=obs.device, dtype=torch.int64)
        base = buf.data_ptr()
        ib = ibuf.data_ptr()
        if need_bwd:
            p_h1, p_agg1, p_agg2 = base + 4 * o_h1, base + 4 * o_agg1, base + 4 * o_agg2
        else:
            p_h1 = p_agg1 = p_agg2 = None
        ws_ptr, ws_bytes = cfg.workspace(B)
        args = (obs.data_ptr(), nodes_in.data_ptr(), adj_in.data_ptr(), count_in.data_ptr(), base,
                base + 4 * o_adj, ib, ib + 8 * B, cfg.arr_ptr, cfg.n_desc, packed.data_ptr(),
                cfg.has_bias, cfg.acts[0], cfg.acts[1], base + 4 * o_mx, p_h1, p_agg1, p_agg2,
                flags.data_ptr(), ws_ptr, ws_bytes, B, N, F, cfg.H1, H2,
                torch.cuda.current_stream().cuda_stream)
        rc = TIMER.launch("gcm_dense_step_fwd", cfg.fn_fwd, *args) if TIMER is not None \
            else cfg.fn_fwd(*args)
        if rc:
            _hip.check(rc, "gcm_dense_step_fwd")
        nodes_out = buf[:B * N * F].view(B, N, F)
        adj_out = buf[o_adj:o_adj + B * N * N].view(B, N, N)
        mx = buf[o_mx:o_mx + B * H2].view(B, H2)
        cur, count_out = ibuf[0], ibuf[1]
        ctx.save_for_backward(buf, ibuf, count_in, packed)
        ctx.cfg, ctx.B = cfg, B
        ctx.mark_non_differentiable(adj_out, cur, count_out)
        ctx.set_materialize_grads(False)   # backward handles None: no zero-fill launches
        return mx, nodes_out, adj_out, cur, count_out

    @staticmethod
    def backward(ctx, g_mx, g_nodes_out, _ga, _gc, _gn):
        buf, ibuf, count_in, packed = ctx.saved_tensors
        cfg, B = ctx.cfg, ctx.B
        N, F, H2, P = cfg.N, cfg.F, cfg.H2, cfg.P
        _, o_adj, o_mx, o_h1, o_agg1, o_agg2, (tot_b, o_obs, o_par, o_ws) = cfg.layout(B, True)
        if g_mx is None:
            g_mx = torch.zeros(B, H2, device=buf.device)
        elif not g_mx.is_contiguous():
            g_mx = g_mx.contiguous()
        if g_nodes_out is None:
            g_no = None
        else:
            g_no = (g_nodes_out if g_nodes_out.is_contiguous() else g_nodes_out.contiguous()).data_ptr()
        out = torch.empty(tot_b, device=buf.device, dtype=_f32)
        ob = out.data_ptr()
        base = buf.data_ptr()
        args = (g_mx.data_ptr(), g_no, base, base + 4 * o_adj, ibuf.data_ptr(), count_in.data_ptr(),
                packed.data_ptr(), cfg.has_bias, cfg.acts[0], cfg.acts[1], base + 4 * o_mx,
                base + 4 * o_h1, base + 4 * o_agg1, base + 4 * o_agg2, ob, ob + 4 * o_obs,
                ob + 4 * o_par, ob + 4 * o_ws, 4 * B * P, B, N, F, cfg.H1, H2,
                torch.cuda.current_stream().cuda_stream)
        rc = TIMER.launch("gcm_dense_step_bwd", cfg.fn_bwd, *args) if TIMER is not None \
            else cfg.fn_bwd(*args)
        if rc:
            _hip.check(rc, "gcm_dense_step_bwd")
        need = ctx.needs_input_grad
        g_nodes_in = out[:B * N * F].view(B, N, F) if need[1] else None
        g_obs = out[o_obs:o_obs + B * F].view(B, F) if need[0] else None
        g_params = out[o_par:o_par + P] if need[2] else None
        return g_obs, g_nodes_in, g_params, None, None, None, None


class SlabHolder:
    """The parameter-gradient slab arrays [B, P] of one packed parameter vector (one per batch size
    seen): every step's backward kernel accumulates into them, _ParamGate sums them once."""

    def __init__(self, P, device):
        self.P, self.device = P, device
        self.slabs = {}

    def get(self, B):
        t = self.slabs.get(B)
        if t is None:
            t = torch.zeros(B, self.P, device=self.device, dtype=_f32)
            self.slabs[B] = t
        return t


class _ParamGate(torch.autograd.Function):
    """Identity on the packed parameter vector, placed between it and the step nodes: its backward
    runs after every step node of the pass (they are its consumers) and produces the parameter
    gradient of ALL of them at once -
      * fused-step nodes (gcm_dense_step_bwd_slabs) accumulated per-graph slabs into `holder`:
        one slab sum instead of T slab sums and T engine-side adds;
    (The live-row steps do not pass through the gate: all of a chain's steps share one autograd node
    - RowsChainNode in csrc/torch_ext/step_ext.cpp - that consumes the packed vector directly.)"""

    @staticmethod
    def forward(ctx, packed, holder):
        ctx.holder = holder
        ctx.set_materialize_grads(False)
        return packed.view_as(packed)

    @staticmethod
    def backward(ctx, g):
        holder = ctx.holder
        total = g
        for B, slabs in holder.slabs.items():
            out = torch.empty(holder.P, device=slabs.device, dtype=_f32)
            _call("gcm_sum_slabs_acc", _hip.ptr(slabs), B, holder.P, _hip.ptr(total), _hip.ptr(out),
                  _hip.stream())
            slabs.zero_()
            total = out
        return total, None


def param_gate(packed, holder):
    return _ParamGate.apply(packed, holder)


def fused_step(obs, nodes_in, packed, adj_in, count_in, flags, cfg, slab_acc=None, is_head=True):
    """The per-step node: the C++ autograd node when gcm/_lib/ext is built (same C-ABI calls,
    no interpreter on the path), else the Python Function above.  Kernel timing (TIMER) goes
    through the Python one, whose launches it can bracket.
    slab_acc [B, P]: where the C++ node's backward accumulates the parameter-gradient slabs (summed
    once by _ParamGate); is_head: first step of a chain of hidden states.
    -> (mx, nodes_out, adj_out, cur, count_out)"""
    if TIMER is None:
        handle = cfg.cpp_handle()
        if handle:
            return _ext.module().fused_step(obs, nodes_in, packed, adj_in, count_in, flags, handle,
                                            torch._C._cuda_getCurrentRawStream(obs.device.index),
                                            slab_acc, is_head)
    return _FusedStep.apply(obs, nodes_in, packed, adj_in, count_in, flags, cfg)


# Time-parallel BPTT keeps T*B*(N*F + F + P) floats of scratch; above this many bytes the
# rollout backward runs step by step instead (2*B*N*F + B*P floats).  0 forces the sequential one.
ROLLOUT_BWD_BATCHED_MAX_BYTES = 16 << 30


class _FusedRollout(torch.autograd.Function):
    """T DenseGCM steps as ONE autograd node (gcm_dense_rollout_fwd/bwd)."""

    @staticmethod
    def forward(ctx, obs, nodes0, packed, adj0, num_nodes0, flags, cfg):
        obs = obs.contiguous()
        _hip.on_device(obs, nodes0, adj0, num_nodes0, flags)
        T, B, F = obs.shape
        N, H1, H2 = cfg.N, cfg.H1, cfg.H2
        dev = obs.device
        need_bwd = any(ctx.needs_input_grad)
        w = cfg.unpack_ptrs(packed)
        if not need_bwd:
            # inference: no history.  The persistent kernel keeps the state in LDS and writes the
            # final hidden state once (two-slot arrays); shapes / selectors it does not cover take
            # the general path below.
            nodes2 = torch.empty(2, B, N, F, device=dev, dtype=_f32)
            adj2 = torch.empty(2, B, N, N, device=dev, dtype=_f32)
            count2 = torch.empty(2, B, device=dev, dtype=torch.int64)
            nodes2[0].copy_(nodes0)
            adj2[0].copy_(adj0)
            count2[0].copy_(num_nodes0)
            mx_all = torch.empty(T, B, H2, device=dev, dtype=_f32)
            fn = _hip.lib().gcm_dense_rollout_persistent_fwd
            rc = fn(_hip.ptr(obs), _hip.ptr(nodes2), _hip.ptr(adj2), _hip.ptr(count2), None,
                    cfg.arr_ptr, cfg.n_desc, w[0], w[1], w[2], cfg.acts[0], w[3], w[4], w[5],
                    cfg.acts[1], _hip.ptr(mx_all), None, None, None, _hip.ptr(flags), 0, T, B, N, F,
                    H1, H2, _hip.stream())
            if rc == 0:
                adj_T, count_T = adj2[1], count2[1]
                ctx.mark_non_differentiable(adj_T, count_T)
                return mx_all, nodes2[1], adj_T, count_T
            if rc != _hip.GCM_EUNSUPPORTED:
                _hip.check(rc, "gcm_dense_rollout_persistent_fwd")
        nodes_all = torch.empty(T + 1, B, N, F, device=dev, dtype=_f32)
        adj_all = torch.empty(T + 1, B, N, N, device=dev, dtype=_f32)
        count_all = torch.empty(T + 1, B, device=dev, dtype=torch.int64)
        nodes_all[0].copy_(nodes0)
        adj_all[0].copy_(adj0)
        count_all[0].copy_(num_nodes0)
        cur_all = torch.empty(T, B, device=dev, dtype=torch.int64)
        mx_all = torch.empty(T, B, H2, device=dev, dtype=_f32)
        h1_all = torch.empty(T, B, N, H1, device=dev, dtype=_f32) if need_bwd else None
        agg1_all = torch.empty(T, B, N, F, device=dev, dtype=_f32) if need_bwd else None
        agg2_all = torch.empty(T, B, H1, device=dev, dtype=_f32) if need_bwd else None
        ws_ptr, ws_bytes = cfg.workspace(B)
        _call("gcm_dense_rollout_fwd", _hip.ptr(obs), _hip.ptr(nodes_all), _hip.ptr(adj_all),
              _hip.ptr(count_all), _hip.ptr(cur_all), cfg.arr_ptr, cfg.n_desc,
              w[0], w[1], w[2], cfg.acts[0], w[3], w[4], w[5], cfg.acts[1], _hip.ptr(mx_all),
              _hip.ptr(h1_all), _hip.ptr(agg1_all), _hip.ptr(agg2_all), _hip.ptr(flags), ws_ptr,
              ws_bytes, T, B, N, F, H1, H2, _hip.stream())
        ctx.save_for_backward(nodes_all, adj_all, count_all, cur_all, mx_all, h1_all, agg1_all,
                              agg2_all, packed)
        ctx.cfg, ctx.dims = cfg, (T, B)
        # clones: a view would pin the whole [T+1, ...] history for as long as the hidden lives
        nodes_T, adj_T, count_T = nodes_all[T].clone(), adj_all[T].clone(), count_all[T].clone()
        ctx.mark_non_differentiable(adj_T, count_T)
        return mx_all, nodes_T, adj_T, count_T

    @staticmethod
    def backward(ctx, g_mx_all, g_nodes_T, _g_adj, _g_count):
        (nodes_all, adj_all, count_all, cur_all, mx_all, h1_all, agg1_all, agg2_all,
         packed) = ctx.saved_tensors
        cfg = ctx.cfg
        T, B = ctx.dims
        N, F, H1, H2, P = cfg.N, cfg.F, cfg.H1, cfg.H2, cfg.P
        dev = nodes_all.device
        lib = _hip.lib()
        if g_mx_all is None:
            g_mx_all = torch.zeros(T, B, H2, device=dev)
        need = ctx.needs_input_grad
        if (not need[0] and not need[1] and g_mx_all.dtype == _f32
                and lib.gcm_dense_rows_supported(N, F, H1, H2)):
            # only the parameters need a gradient: every graph-step is independent - one launch over
            # the live rows of all T*B of them (csrc/rows_bptt.hip), no reverse scan, no Q array; an
            # expanded gradient (mean()) is read through its strides
            flat = torch.empty(P, device=dev, dtype=_f32)
            ws_bytes = lib.gcm_dense_rollout_bwd_params_workspace_bytes(T, B, F, H1, H2)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            st_t, st_b, st_h = g_mx_all.stride()
            _call("gcm_dense_rollout_bwd_params", _hip.ptr(g_mx_all), st_t, st_b, st_h, _hip.ptr(nodes_all),
                  _hip.ptr(adj_all), _hip.ptr(cur_all), packed.data_ptr(), cfg.acts[0], cfg.acts[1],
                  _hip.ptr(mx_all), _hip.ptr(h1_all), _hip.ptr(agg1_all), _hip.ptr(agg2_all), _hip.ptr(flat),
                  _hip.ptr(ws), ws_bytes, T, B, N, F, H1, H2, _hip.stream())
            return None, None, flat if need[2] else None, None, None, None, None
        g_mx_all = g_mx_all.contiguous()
        g_nodes_T = None if g_nodes_T is None else g_nodes_T.contiguous()
        g_obs = torch.empty(T, B, F, device=dev, dtype=_f32)
        g_nodes0 = torch.empty(B, N, F, device=dev, dtype=_f32)
        flat = torch.empty(P, device=dev, dtype=_f32)
        ws_bytes = lib.gcm_dense_rollout_bwd_batched_workspace_bytes(T, B, N, F, H1, H2)
        if ws_bytes > ROLLOUT_BWD_BATCHED_MAX_BYTES:
            ws_bytes = lib.gcm_dense_rollout_bwd_workspace_bytes(B, N, F, H1, H2)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        w = cfg.unpack_ptrs(packed)
        _call("gcm_dense_rollout_bwd", _hip.ptr(g_mx_all), _hip.ptr(g_nodes_T), _hip.ptr(nodes_all),
              _hip.ptr(adj_all), _hip.ptr(count_all), _hip.ptr(cur_all), w[0], w[1], w[2],
              cfg.acts[0], w[3], w[4], w[5], cfg.acts[1], _hip.ptr(mx_all), _hip.ptr(h1_all),
              _hip.ptr(agg1_all), _hip.ptr(agg2_all), _hip.ptr(g_obs), _hip.ptr(g_nodes0),
              _hip.ptr(flat), _hip.ptr(ws), ws_bytes, T, B, N, F, H1, H2, _hip.stream())
        need = ctx.needs_input_grad
        return (g_obs if need[0] else None, g_nodes0 if need[1] else None,
                flat if need[2] else None, None, None, None, None)


class _LearnedStep(torch.autograd.Function):
    """One DenseGCM + LearnedEdge step (default edge network, observations without gradient) as ONE
    autograd node: state advance, fused edge network + gumbel selection, dense 2-layer GNN forward;
    ONE kernel backward (gcm_learned_step_bwd).  The adjacency's gradient chain through time travels
    as the gradient of the `dchain` proxy (ONE [B,N,N] buffer, updated sparsely: csrc/learned_step.hip), the
    parameter gradients accumulate into the gate's slab array.
    inputs: packed = GNN vector | edge-network vector (gated), dchain_in; the rest is data."""

    @staticmethod
    def forward(ctx, packed, dchain_in, obs, nodes_in, adj_in, count_in, noise, noise_is_exp, flags, cfg,
                slab_acc, is_head):
        lib = _hip.lib()
        B, N, F, H1, H2, P = obs.shape[0], cfg.N, cfg.F, cfg.H1, cfg.H2, cfg.P
        dev = obs.device
        st = _hip.stream()
        obs, nodes_in, adj_in = obs.contiguous(), nodes_in.contiguous(), adj_in.contiguous()
        noise = noise.contiguous()
        _hip.on_device(obs, nodes_in, adj_in, count_in, noise, packed, flags)
        need_bwd = ctx.needs_input_grad[0]
        nodes_out, adj_out = torch.empty_like(nodes_in), torch.empty_like(adj_in)
        ibuf = torch.empty(2, B, dtype=torch.int64, device=dev)
        cur, count_out = ibuf[0], ibuf[1]
        p = _hip.ptr
        _call("gcm_state_advance_fwd", p(nodes_in), p(adj_in), None, p(count_in), p(obs), p(nodes_out),
              p(adj_out), None, p(cur), p(count_out), p(flags), B, N, F, st)
        soft = torch.empty(B, N, device=dev, dtype=_f32)
        base = packed.data_ptr()
        _call("gcm_learned_select_fused", p(nodes_out), p(adj_out), p(cur), p(noise), int(noise_is_exp),
              base + 4 * P, cfg.eps[0], cfg.eps[1], cfg.cutoff, p(soft), B, N, F, st)
        mx = torch.empty(B, H2, device=dev, dtype=_f32)
        h1 = torch.empty(B, N, H1, device=dev, dtype=_f32) if need_bwd else None
        agg1 = torch.empty(B, N, F, device=dev, dtype=_f32) if need_bwd else None
        agg2 = torch.empty(B, H1, device=dev, dtype=_f32) if need_bwd else None
        w = cfg.unpack_ptrs(packed)
        _call("gcm_dense_gnn2_row_fwd", p(nodes_out), p(adj_out), p(cur), w[0], w[1], w[2], cfg.acts[0],
              w[3], w[4], w[5], cfg.acts[1], p(mx), p(h1), p(agg1), p(agg2), p(flags), B, N, F, H1, H2, st)
        if need_bwd:
            ctx.save_for_backward(packed, nodes_out, adj_out, ibuf, count_in, mx, h1, agg1, agg2, soft)
            ctx.cfg, ctx.slab_acc, ctx.is_head = cfg, slab_acc, is_head
        dchain_out = cfg.zero_chain(B, dev)
        ctx.mark_non_differentiable(nodes_out, adj_out, cur, count_out)
        ctx.set_materialize_grads(False)
        return mx, dchain_out, nodes_out, adj_out, cur, count_out

    @staticmethod
    def backward(ctx, g_mx, g_chain, _gn, _ga, _gc, _gk):
        packed, nodes_out, adj_out, ibuf, count_in, mx, h1, agg1, agg2, soft = ctx.saved_tensors
        cfg = ctx.cfg
        B, N, F, H1, H2, P = mx.shape[0], cfg.N, cfg.F, cfg.H1, cfg.H2, cfg.P
        dev = mx.device
        if g_mx is None and g_chain is None:
            return (None,) * 12
        if g_mx is None:
            g_mx = torch.zeros(B, H2, device=dev, dtype=_f32)
        g_mx = g_mx.contiguous()
        # the chain buffer: handed down from the step after this one (mutated in place), or new
        D = g_chain if (g_chain is not None and g_chain.is_contiguous()) else \
            (torch.zeros(B, N, N, device=dev, dtype=_f32) if g_chain is None else g_chain.contiguous())
        p = _hip.ptr
        base = packed.data_ptr()
        _call("gcm_learned_step_bwd", p(g_mx), p(nodes_out), p(adj_out), ibuf.data_ptr(), p(count_in), base,
              cfg.acts[0], cfg.acts[1], p(mx), p(h1), p(agg1), p(agg2), p(soft), base + 4 * P, cfg.eps[0],
              cfg.eps[1], p(D), p(ctx.slab_acc), 1, B, N, F, H1, H2, _hip.stream())
        g_packed = cfg.zero_params(dev) if ctx.is_head else None     # a defined gradient: the gate runs
        return (g_packed, D if ctx.needs_input_grad[1] else None) + (None,) * 10


def learned_step(packed, dchain_in, obs, nodes_in, adj_in, count_in, noise, noise_is_exp, flags, cfg,
                 slab_acc, is_head):
    """The per-step node of DenseGCM + LearnedEdge: the C++ autograd node when gcm/_lib/ext is built (same
    C-ABI calls, no interpreter on the path), else the Python Function above."""
    h = cfg.learned_cpp_handle()
    if h:
        return _ext.module().learned_step(packed, dchain_in, obs, nodes_in, adj_in, count_in, noise,
                                          int(noise_is_exp), flags, h,
                                          torch._C._cuda_getCurrentRawStream(obs.device.index), slab_acc, is_head)
    return _LearnedStep.apply(packed, dchain_in, obs, nodes_in, adj_in, count_in, noise, noise_is_exp, flags,
                              cfg, slab_acc, is_head)


def fused_rollout(obs, nodes0, packed, adj0, num_nodes0, flags, cfg):
    return _FusedRollout.apply(obs, nodes0, packed, adj0, num_nodes0, flags, cfg)


def rows_linear(x2, weight, bias=None, transpose=False, ln=None, out=None, ldy=0):
    """csrc/rows_linear.hip on rows x2 [M, *]: x2 W^T + bias (transpose: x2 W).  ln = (gamma, beta, eps):
    -> (pre-activation, LayerNorm(relu(pre-activation))).  out/ldy: write into a wider matrix."""
    x2 = x2.contiguous()
    weight = weight.contiguous()
    _hip.on_device(x2, weight, bias)
    M = x2.shape[0]
    O, I = weight.shape
    n_out = I if transpose else O
    y = out if out is not None else torch.empty(M, n_out, device=x2.device, dtype=_f32)
    h = torch.empty(M, n_out, device=x2.device, dtype=_f32) if ln is not None else None
    _call("gcm_rows_linear", _hip.ptr(x2), _hip.ptr(weight), _hip.ptr(bias), _hip.ptr(y), M, I, O,
          int(transpose), ldy, _hip.ptr(ln[0]) if ln else None, _hip.ptr(ln[1]) if ln else None,
          float(ln[2]) if ln else 0.0, _hip.ptr(h), _hip.stream())
    return (y, h) if ln is not None else y


class _SkinnyLinear(torch.autograd.Function):
    """y = x W^T + b for many rows and narrow layers (<= 64 features: the LearnedEdge edge network, the
    PositionalEncoding re-projection): forward and dX are gcm_rows_linear (a row stream through the matrix
    cores), the weight gradient - a [O x M] x [M x I] product with M = B*N rows, which a library GEMM runs
    on one or two workgroups - is gcm_skinny_wgrad (rows split over the grid)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        y = rows_linear(x.reshape(-1, x.shape[-1]), weight, bias)
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad
        gx = None
        if need_x:
            gx = rows_linear(g.reshape(-1, weight.shape[0]), weight, transpose=True).view(x.shape)
        gw = gb = None
        if need_w or need_b:
            O, I = weight.shape
            g2 = g.reshape(-1, O).contiguous()
            x2 = x.reshape(-1, I).contiguous()
            M = g2.shape[0]
            lib = _hip.lib()
            ws_bytes = lib.gcm_skinny_wgrad_workspace_bytes(M, O, I)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=g.device)
            out = torch.empty(O * I + O, device=g.device, dtype=_f32)
            _call("gcm_skinny_wgrad", _hip.ptr(g2), _hip.ptr(x2), _hip.ptr(out), _hip.ptr(ws), ws_bytes,
                  M, O, I, _hip.stream())
            gw = out[:O * I].view(O, I) if need_w else None
            gb = out[O * I:] if (need_b and ctx.has_bias) else None
        return gx, gw, gb


def skinny_linear(x, weight, bias):
    return _SkinnyLinear.apply(x, weight, bias)


class _ReluLayerNorm(torch.autograd.Function):
    """y = LayerNorm(relu(x)) over the last dim (gcm_relu_layernorm_fwd/bwd); x is saved, the row
    statistics are recomputed in the backward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        _hip.on_device(x, gamma, beta)
        F = x.shape[-1]
        M = x.numel() // F
        y = torch.empty_like(x)
        _call("gcm_relu_layernorm_fwd", _hip.ptr(x), _hip.ptr(gamma), _hip.ptr(beta), _hip.ptr(y), M, F,
              eps, _hip.stream())
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma = ctx.saved_tensors
        F = x.shape[-1]
        M = x.numel() // F
        g = g.contiguous()
        dx = torch.empty_like(x)
        dgb = torch.empty(2 * F, device=x.device, dtype=_f32)
        lib = _hip.lib()
        ws_bytes = lib.gcm_relu_layernorm_bwd_workspace_bytes(M, F)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        _call("gcm_relu_layernorm_bwd", _hip.ptr(g), _hip.ptr(x), _hip.ptr(gamma), _hip.ptr(dx), _hip.ptr(dgb),
              _hip.ptr(ws), ws_bytes, M, F, ctx.eps, _hip.stream())
        return dx, dgb[:F], dgb[F:], None


def relu_layernorm(x, gamma, beta, eps):
    return _ReluLayerNorm.apply(x, gamma, beta, eps)


def default_edge_network(net):
    """The modules of `net` when it is the reference's default edge network (learned.py:38-51:
    Linear - ReLU - LayerNorm - Linear - ReLU - LayerNorm - Linear, features <= 64, affine norms,
    no hooks), else None: that architecture runs through skinny_linear / relu_layernorm, anything
    else is called as the torch module it is."""
    nn = torch.nn
    kinds = (nn.Linear, nn.ReLU, nn.LayerNorm, nn.Linear, nn.ReLU, nn.LayerNorm, nn.Linear)
    if not isinstance(net, nn.Sequential) or len(net) != 7:
        return None
    mods = list(net)
    if not all(isinstance(m, k) for m, k in zip(mods, kinds)):
        return None
    for m in mods + [net]:
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks:
            return None
    for ln, lin in ((mods[2], mods[0]), (mods[5], mods[3])):
        if (not ln.elementwise_affine or ln.bias is None or tuple(ln.normalized_shape) != (lin.out_features,)
                or lin.out_features > 64):
            return None
    if max(mods[0].in_features, mods[3].in_features, mods[6].in_features) > 64 or mods[6].out_features > 64:
        return None
    return mods


def _mlp_fwd(x2, w0, b0, g0, be0, w1, b1, g1, be1, w2, b2, eps0, eps1):
    """Linear - ReLU - LayerNorm - Linear - ReLU - LayerNorm - Linear on rows x2 [M, I];
    -> (out [M, O], tensors the backward needs)."""
    # three launches of gcm_rows_linear, the first two with the ReLU + LayerNorm epilogue; the
    # pre-activations are saved, the row statistics recomputed in the backward
    p0, h0 = rows_linear(x2, w0, b0, ln=(g0, be0, eps0))
    p1, h1 = rows_linear(h0, w1, b1, ln=(g1, be1, eps1))
    return rows_linear(h1, w2, b2), (x2, p0, h0, p1, h1, w0, g0, w1, g1, w2)


def _mlp_bwd(g2, saved, eps, has_bias, need_x):
    """adjoint of _mlp_fwd for g2 [M, O] -> (g_x2 | None, [dw0, db0, dg0, dbe0, dw1, db1, dg1, dbe1, dw2, db2])"""
    x2, p0, h0, p1, h1, w0, g0, w1, g1, w2 = saved
    lib = _hip.lib()
    st = _hip.stream()
    dev = x2.device
    M = x2.shape[0]

    def wgrad(gy, xin, O, I):
        ws_bytes = lib.gcm_skinny_wgrad_workspace_bytes(M, O, I)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        out = torch.empty(O * I + O, device=dev, dtype=_f32)
        _call("gcm_skinny_wgrad", _hip.ptr(gy), _hip.ptr(xin), _hip.ptr(out), _hip.ptr(ws), ws_bytes,
              M, O, I, st)
        return out[:O * I].view(O, I), out[O * I:]

    def ln_bwd(gy, pre, gamma, e):
        F = pre.shape[1]
        dx = torch.empty_like(pre)
        dgb = torch.empty(2 * F, device=dev, dtype=_f32)
        ws_bytes = lib.gcm_relu_layernorm_bwd_workspace_bytes(M, F)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        _call("gcm_relu_layernorm_bwd", _hip.ptr(gy), _hip.ptr(pre), _hip.ptr(gamma), _hip.ptr(dx),
              _hip.ptr(dgb), _hip.ptr(ws), ws_bytes, M, F, e, st)
        return dx, dgb[:F], dgb[F:]

    dw2, db2 = wgrad(g2, h1, w2.shape[0], w2.shape[1])
    gp1, dg1, dbe1 = ln_bwd(rows_linear(g2, w2, transpose=True), p1, g1, eps[1])
    dw1, db1 = wgrad(gp1, h0, w1.shape[0], w1.shape[1])
    gp0, dg0, dbe0 = ln_bwd(rows_linear(gp1, w1, transpose=True), p0, g0, eps[0])
    dw0, db0 = wgrad(gp0, x2, w0.shape[0], w0.shape[1])
    gx = rows_linear(gp0, w0, transpose=True) if need_x else None
    hb = has_bias
    return gx, [dw0, db0 if hb[0] else None, dg0, dbe0, dw1, db1 if hb[1] else None, dg1, dbe1, dw2,
                db2 if hb[2] else None]


class _EdgeMLP(torch.autograd.Function):
    """The default edge network (learned.py:38-51) as ONE autograd node: Linear - ReLU - LayerNorm -
    Linear - ReLU - LayerNorm - Linear on M candidate rows.  Same kernels and library GEMMs as the
    module-by-module path (skinny_linear / relu_layernorm), chained by hand in both directions:
    five Python-level autograd nodes per step become one."""

    @staticmethod
    def forward(ctx, x, w0, b0, g0, be0, w1, b1, g1, be1, w2, b2, eps0, eps1):
        x2 = x.reshape(-1, x.shape[-1])
        out, saved = _mlp_fwd(x2, w0, b0, g0, be0, w1, b1, g1, be1, w2, b2, eps0, eps1)
        ctx.save_for_backward(*saved)
        ctx.eps = (eps0, eps1)
        ctx.has_bias = (b0 is not None, b1 is not None, b2 is not None)
        ctx.xshape = x.shape
        return out.view(*x.shape[:-1], out.shape[-1])

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        g2 = g.reshape(saved[0].shape[0], -1).contiguous()
        gx, pg = _mlp_bwd(g2, saved, ctx.eps, ctx.has_bias, ctx.needs_input_grad[0])
        return (gx.view(ctx.xshape) if gx is not None else None, *pg, None, None)


class _LearnedEdgeDefault(torch.autograd.Function):
    """Dense LearnedEdge with the default edge network as ONE autograd node (learned.py:53-113):
    candidate pairs, edge network, gumbel noise, gumbel-softmax + STE + adjacency-row write (in
    place on `adj`, which the caller owns).  `noise` None = draw it here with the device RNG the
    way torch.nn.functional.gumbel_softmax does."""

    @staticmethod
    def forward(ctx, nodes, adj, cur, noise, cutoff, w0, b0, g0, be0, w1, b1, g1, be1, w2, b2, eps0, eps1):
        nodes = nodes.contiguous()
        _hip.on_device(nodes, adj, cur)
        B, N, F = nodes.shape
        st = _hip.stream()
        pairs = torch.empty(B * N, 2 * F, device=nodes.device, dtype=_f32)
        _call("gcm_learned_pairs_fwd", _hip.ptr(nodes), _hip.ptr(cur), _hip.ptr(pairs), B, N, F, st)
        logits, saved = _mlp_fwd(pairs, w0, b0, g0, be0, w1, b1, g1, be1, w2, b2, eps0, eps1)
        logits = logits.view(B, N)
        if noise is None:
            noise = -torch.empty_like(logits).exponential_().log()
        noise = noise.contiguous()
        soft = torch.empty(B, N, device=adj.device, dtype=_f32)
        _call("gcm_learned_select_fwd", _hip.ptr(logits), _hip.ptr(noise), _hip.ptr(cur), float(cutoff),
              _hip.ptr(adj), _hip.ptr(soft), B, N, st)
        ctx.mark_dirty(adj)
        ctx.save_for_backward(soft, cur, *saved)
        ctx.eps = (eps0, eps1)
        ctx.has_bias = (b0 is not None, b1 is not None, b2 is not None)
        ctx.dims = (B, N, F)
        return adj

    @staticmethod
    def backward(ctx, g_adj):
        soft, cur, *saved = ctx.saved_tensors
        B, N, F = ctx.dims
        st = _hip.stream()
        g_adj = g_adj.contiguous()
        g_logits = torch.empty(B * N, 1, device=g_adj.device, dtype=_f32)
        _call("gcm_learned_select_bwd", _hip.ptr(g_adj), _hip.ptr(soft), _hip.ptr(cur), _hip.ptr(g_logits),
              B, N, st)
        g_pairs, pg = _mlp_bwd(g_logits, saved, ctx.eps, ctx.has_bias, ctx.needs_input_grad[0])
        g_nodes = None
        if g_pairs is not None:
            g_nodes = torch.empty(B, N, F, device=g_adj.device, dtype=_f32)
            _call("gcm_learned_pairs_bwd", _hip.ptr(g_pairs), _hip.ptr(cur), _hip.ptr(g_nodes), B, N, F, st)
        # both straight-through estimators are identities: the incoming adjacency receives g_adj
        # unchanged, also at the rewritten entries (learned.py:108-110 adds adj inside the STE)
        return (g_nodes, g_adj, None, None, None, *pg, None, None)


def learned_edge_default(net, nodes, adj, cur, noise, cutoff):
    """The fused dense LearnedEdge when `net` is the default edge network on device tensors, else None."""
    mods = default_edge_network(net)
    if mods is None or not nodes.is_cuda or nodes.dtype != _f32:
        return None
    l0, _, n0, l1, _, n1, l2 = mods
    if l2.out_features != 1 or l0.in_features != 2 * nodes.shape[2]:
        return None
    return _LearnedEdgeDefault.apply(nodes, adj, cur, noise, cutoff, l0.weight, l0.bias, n0.weight, n0.bias,
                                     l1.weight, l1.bias, n1.weight, n1.bias, l2.weight, l2.bias, n0.eps, n1.eps)


def edge_network_forward(net, x):
    """net(x) for the default architecture on device rows: one autograd node over the row-split
    kernels; anything else is called as the torch module it is."""
    mods = default_edge_network(net)
    if mods is None or not x.is_cuda or x.dtype != _f32 or x.numel() == 0:
        return net(x)
    l0, _, n0, l1, _, n1, l2 = mods
    return _EdgeMLP.apply(x, l0.weight, l0.bias, n0.weight, n0.bias, l1.weight, l1.bias, n1.weight, n1.bias,
                          l2.weight, l2.bias, n0.eps, n1.eps)


# ===========================================================================
# SURVEY 8(f) "next" rows: positional encoding, packed sparse hidden state
# ===========================================================================
class _PosEncAdd(torch.autograd.Function):
    """x[b, n] += pe[n] for n <= num_nodes[b], in place (gcm.py:120-131); backward = identity."""

    @staticmethod
    def forward(ctx, x, pe, num_nodes):
        _hip.on_device(x, pe, num_nodes)
        B, N, F = x.shape
        _call("gcm_posenc_add", _hip.ptr(x), _hip.ptr(pe), _hip.ptr(num_nodes), B, N, F, pe.shape[1],
              pe.shape[0], _hip.stream())
        ctx.mark_dirty(x)
        return x

    @staticmethod
    def backward(ctx, g):
        return g, None, None


class _PosEncCat(torch.autograd.Function):
    """PositionalEncoding(mode="cat") (gcm.py:133-140): rows i <= num_nodes[b] become
    [pe[i, :cat_dim] | reproject(x[b, i])], the rows beyond stay x - gcm_rows_linear writes the
    re-projection straight into the output's columns cat_dim.., one element-wise pass finishes it."""

    @staticmethod
    def forward(ctx, x, weight, bias, pe, num_nodes, cat_dim):
        x = x.contiguous()
        _hip.on_device(x, weight, bias, pe, num_nodes)
        B, N, F = x.shape
        out = torch.empty_like(x)
        rows_linear(x.view(B * N, F), weight, bias, out=out.view(B * N, F)[:, cat_dim:], ldy=F)
        _call("gcm_posenc_cat_finish", _hip.ptr(x), _hip.ptr(pe), pe.stride(0), _hip.ptr(num_nodes), _hip.ptr(out),
              B, N, F, cat_dim, _hip.stream())
        ctx.save_for_backward(x, weight, num_nodes)
        ctx.cat_dim = cat_dim
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, num_nodes = ctx.saved_tensors
        B, N, F = x.shape
        cat = ctx.cat_dim
        g = g.contiguous()
        g_x = torch.empty_like(x)
        g_proj = torch.empty(B * N, F - cat, device=x.device, dtype=_f32)
        _call("gcm_posenc_cat_bwd", _hip.ptr(g), _hip.ptr(num_nodes), _hip.ptr(g_x), _hip.ptr(g_proj), B, N, F, cat,
              _hip.stream())
        if ctx.needs_input_grad[0]:
            g_x = g_x + rows_linear(g_proj, weight, transpose=True).view(B, N, F)
        else:
            g_x = None
        gw = gb = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            O, I = weight.shape
            lib = _hip.lib()
            ws_bytes = lib.gcm_skinny_wgrad_workspace_bytes(B * N, O, I)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=g.device)
            out = torch.empty(O * I + O, device=g.device, dtype=_f32)
            _call("gcm_skinny_wgrad", _hip.ptr(g_proj), _hip.ptr(x), _hip.ptr(out), _hip.ptr(ws), ws_bytes,
                  B * N, O, I, _hip.stream())
            gw = out[:O * I].view(O, I)
            gb = out[O * I:] if ctx.has_bias else None
        return g_x, gw, gb, None, None, None


def posenc_cat(x, weight, bias, pe, num_nodes, cat_dim):
    return _PosEncCat.apply(x, weight, bias, pe, num_nodes, cat_dim)


def posenc_add_(x, pe, num_nodes):
    return _PosEncAdd.apply(x, pe, num_nodes)


def pack_hidden(coo, values, B, max_edges, edge_fill, weight_fill):
    """-> dense_edges [B,2,max_edges] i64, dense_weights [B,1,max_edges] f32 (util.py:323-351).
    One host readback (the per-graph counts, for the reference's assertion)."""
    coo, values = coo.contiguous(), values.contiguous()
    _hip.on_device(coo, values)
    E = coo.shape[1]
    dev = coo.device
    dense_edges = torch.full((B, 2, max_edges), edge_fill, dtype=_i64, device=dev)
    dense_weights = torch.full((B, 1, max_edges), weight_fill, dtype=_f32, device=dev)
    batch_ptr = ptr_from_sorted(coo[0], B)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    _call("gcm_pack_hidden", _hip.ptr(coo), _hip.ptr(values), _hip.ptr(batch_ptr),
          _hip.ptr(dense_edges), _hip.ptr(dense_weights), _hip.ptr(flags), E, B, max_edges,
          _hip.stream())
    counts = (batch_ptr[1:] - batch_ptr[:-1]).tolist()
    for n in counts:
        assert n < max_edges, f"Cannot pack {n} edges into {max_edges}, increase max edges"
    return dense_edges, dense_weights


# ===========================================================================
# sparse LearnedEdge (sparse_edge_selectors/learned.py:90-160)
# ===========================================================================
class CausalEdges:
    """Closed-form causal candidate edges of a batch (util.get_causal_edges util.py:242-282):
    indices [3, E] (batch, sink, source) in coalesced order + the sink-row segments."""

    def __init__(self, T, taus, window):
        _hip.on_device(T, taus)
        B = T.numel()
        dev = T.device
        self.T, self.taus, self.B = T, taus, B
        self.window = -1 if window is None else int(window)
        offs = torch.empty(2, B + 1, dtype=_i64, device=dev)
        self.edge_off, self.seg_off = offs[0], offs[1]
        _call("gcm_causal_count", _hip.ptr(T), _hip.ptr(taus), self.window, _hip.ptr(self.edge_off),
              _hip.ptr(self.seg_off), B, _hip.stream())
        self.E, self.S = (int(v) for v in offs[:, B].tolist())           # one readback
        self.indices = torch.empty(3, self.E, dtype=_i64, device=dev)
        self.seg_ptr = torch.empty(self.S + 1, dtype=_i64, device=dev)
        _call("gcm_causal_fill", _hip.ptr(T), _hip.ptr(taus), self.window, _hip.ptr(self.edge_off),
              _hip.ptr(self.seg_off), _hip.ptr(self.indices), _hip.ptr(self.seg_ptr), self.E,
              self.S, B, _hip.stream())


class _CausalPairs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nodes, edges):
        nodes = nodes.contiguous()
        B, N, F = nodes.shape
        pairs = torch.empty(edges.E, 2 * F, device=nodes.device, dtype=_f32)
        _call("gcm_causal_pairs_fwd", _hip.ptr(nodes), _hip.ptr(edges.indices), _hip.ptr(pairs),
              edges.E, B, N, F, _hip.stream())
        ctx.edges, ctx.dims = edges, (B, N, F)
        return pairs

    @staticmethod
    def backward(ctx, g_pairs):
        e = ctx.edges
        B, N, F = ctx.dims
        g_pairs = g_pairs.contiguous()
        g_nodes = torch.empty(B, N, F, device=g_pairs.device, dtype=_f32)
        _call("gcm_causal_pairs_bwd", _hip.ptr(g_pairs), _hip.ptr(e.T), _hip.ptr(e.taus), e.window,
              _hip.ptr(e.edge_off), _hip.ptr(g_nodes), e.E, B, N, F, _hip.stream())
        return g_nodes, None


def causal_pairs(nodes, edges):
    return _CausalPairs.apply(nodes, edges)


class _SegmentSoftmax(torch.autograd.Function):
    """soft = softmax over each sink row of (logits + noise) / tau  (util.py:89-113, hard=False)."""

    @staticmethod
    def forward(ctx, logits, tau, noise, edges):
        logits, noise = logits.contiguous(), noise.contiguous()
        tau_d = tau.detach().to(device=logits.device, dtype=_f32).contiguous()
        soft = torch.empty_like(logits)
        _call("gcm_segment_softmax_fwd", _hip.ptr(logits), _hip.ptr(noise), _hip.ptr(tau_d),
              _hip.ptr(edges.seg_ptr), _hip.ptr(soft), edges.S, edges.E, _hip.stream())
        ctx.save_for_backward(soft, logits, noise, tau_d)
        ctx.edges, ctx.tau_shape = edges, tau.shape
        return soft

    @staticmethod
    def backward(ctx, g_soft):
        soft, logits, noise, tau_d = ctx.saved_tensors
        e = ctx.edges
        g_soft = g_soft.contiguous()
        g_logits = torch.empty_like(logits)
        g_tau_rows = torch.zeros(max(e.S, 1), device=logits.device, dtype=_f32)
        _call("gcm_segment_softmax_bwd", _hip.ptr(g_soft), _hip.ptr(soft), _hip.ptr(logits),
              _hip.ptr(noise), _hip.ptr(tau_d), _hip.ptr(e.seg_ptr), _hip.ptr(g_logits),
              _hip.ptr(g_tau_rows), e.S, e.E, _hip.stream())
        g_tau = g_tau_rows.sum().reshape(ctx.tau_shape) if ctx.needs_input_grad[1] else None
        return g_logits, g_tau, None, None


def segment_softmax(logits, tau, noise, edges):
    return _SegmentSoftmax.apply(logits, tau, noise, edges)
