"""torch.autograd wrappers around the C-ABI kernels (include/gcm_hip.h).

Every function here launches HIP kernels on torch's current stream; none of
them synchronises with the host.
"""
import ctypes

import torch

from . import _hip

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


def edge_dense_(adj, cur):
    _hip.on_device(adj, cur)
    B, N, _ = adj.shape
    rc = _hip.lib().gcm_edge_dense(_hip.ptr(adj), _hip.ptr(cur), B, N, _hip.stream())
    _hip.check(rc, "gcm_edge_dense")
    return adj


def edge_distance_(nodes, adj, cur, mode, max_distance, dist_param=None, a=(0, 0), b=(0, 0),
                   bidirectional=False, want_dist=False):
    nodes = nodes.contiguous()
    _hip.on_device(nodes, adj, cur, dist_param)
    B, N, F = nodes.shape
    lib = _hip.lib()
    ws_bytes = lib.gcm_edge_distance_workspace_bytes(mode, B, N, F)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=nodes.device)
    dist = torch.empty(B, N, device=nodes.device, dtype=_f32) if want_dist else None
    rc = lib.gcm_edge_distance(_hip.ptr(nodes), _hip.ptr(adj), _hip.ptr(cur), mode,
                               float(max_distance), _hip.ptr(dist_param), a[0], a[1], b[0], b[1],
                               int(bidirectional), _hip.ptr(dist), _hip.ptr(ws), ws_bytes, B, N, F,
                               _hip.stream())
    _hip.check(rc, "gcm_edge_distance")
    return adj, dist
