"""SparseGCM - graph memory over a sparse (COO) adjacency (reference: src/gcm/sparse_gcm.py).

    out, (nodes, adj, T) = SparseGCM(gnn, edge_selectors=...)(x[B,t,F], taus[B], hidden)

Hidden state as in the reference: nodes f32[B,N,F], adj = torch.sparse_coo [B,N,N] with
indices (batch, sink, source), T i64[B].  A whole episode can go through one call.

The per-graph Python loops of the reference (util.py:176-240, 426-452) are replaced by
closed-form kernels driven by one small device-side plan; the GNN input is the same flat
node matrix + (source, sink) edge list, with a ready CSR attached so that gcm.nn.GraphConv
layers run their gather-reduce + linears as one kernel each.  One host readback per call
(the flat sizes, needed to allocate exact-size tensors like the reference returns).
"""
from typing import Tuple, Union

import torch

from . import _hip, _ops
from . import nn as _nn


class SparseGCM(torch.nn.Module):
    """Graph Associative Memory using sparse-graph representations"""

    did_warn = False

    def __init__(
        self,
        gnn: torch.nn.Module,
        preprocessor: torch.nn.Module = None,
        edge_selectors: torch.nn.Module = None,
        aux_edge_selectors: torch.nn.Module = None,
        graph_size: int = 128,
        max_hops: Union[int, None] = None,
        positional_encoder: torch.nn.Module = None,
        finite_check: str = "sync",
    ):
        super().__init__()
        assert finite_check in ("sync", "off")
        self.preprocessor = preprocessor
        self.gnn = gnn
        self.graph_size = graph_size
        self.edge_selectors = edge_selectors
        self.aux_edge_selectors = aux_edge_selectors
        self.positional_encoder = positional_encoder
        self.max_hops = max_hops
        self.finite_check = finite_check
        self._flags = {}

    def get_initial_hidden_state(self, x):
        """sparse_gcm.py:55-70."""
        assert x.dim() == 3
        B, _, feats = x.shape
        nodes = torch.zeros(B, self.graph_size, feats, device=x.device)
        adj = torch.zeros((B, self.graph_size, self.graph_size), device=x.device,
                          layout=torch.sparse_coo)
        T = torch.zeros(B, dtype=torch.long, device=x.device)
        return nodes, adj, T

    def _flag_word(self, device):
        f = self._flags.get(device)
        if f is None:
            f = torch.zeros(1, dtype=torch.int32, device=device)
            self._flags[device] = f
        return f

    def _merge(self, adj, new_adj, selector, B, flags, first=True):
        """sparse_gcm.py:132-139: concatenate the COO lists and coalesce.  Selectors whose edges
        all end in new nodes (every shipped one) merge as a segmented concatenation - no sort;
        a violated order is flagged on the device and surfaces at the call's flag check.
        `first`: the merge into the STORED state.  Only there does "stored entries, then the new
        ones" hold per graph: after it the list already has entries ending in the new nodes, so
        the aux selector's edges (sparse_gcm.py:146-152) interleave with them and may duplicate
        them - that merge is the reference's cat + coalesce (duplicates summed)."""
        if (first and getattr(selector, "new_sinks_only", False) and new_adj.is_coalesced()
                and adj.is_coalesced()):
            idx, val = _ops.coo_merge_segments(adj.indices(), adj.values(), new_adj.indices(),
                                               new_adj.values(), getattr(new_adj, "gcm_bptr", None), B, flags)
            return torch.sparse_coo_tensor(idx, val, size=adj.shape, is_coalesced=True)
        new_adj = new_adj.coalesce()
        if adj._nnz() == 0 and new_adj.is_coalesced():
            return torch.sparse_coo_tensor(new_adj.indices(), new_adj.values(), size=adj.shape,
                                           is_coalesced=True)
        idx = torch.cat([adj.indices(), new_adj.indices()], dim=-1)
        val = torch.cat([adj.values(), new_adj.values()], dim=-1)
        return torch.sparse_coo_tensor(idx, val, size=adj.shape).coalesce()

    def _native_gnn(self):
        """True when every graph layer of the GNN is one of ours (then the k-hop restriction
        runs as a row/edge mask inside the kernels instead of a relabelled subgraph)."""
        if not isinstance(self.gnn, _nn.Sequential):
            return False
        has = False
        for mod, ins, _ in self.gnn.stages():
            if isinstance(mod, _nn.GraphConv):
                has = True
            elif len(ins) != 1:
                return False
        return has

    def forward(self, x, taus, hidden):
        """x [B, t, feat] zero padded in t; taus [B] valid lengths; hidden (nodes, adj, T) or
        None.  Returns (mx [B, t, H] zero padded, (nodes, adj, T + taus))."""
        if hidden is None:
            hidden = self.get_initial_hidden_state(x)
        nodes, adj, T = hidden
        assert x.dim() == 3 and x.dtype == torch.float32
        assert taus.dtype == torch.long and T.dtype == torch.long
        adj = adj.coalesce()
        N = nodes.shape[1]
        B, t_pad, _ = x.shape
        flags = self._flag_word(x.device)

        node_off, _new_off, totals = _ops.sparse_plan(T, taus)
        sel, sel_plan = self.edge_selectors, None
        if hasattr(sel, "plan"):        # a selector that sizes itself on the device: one readback for both
            edge_off = sel.plan(T, taus)
            sizes = torch.cat([totals, edge_off[B:]]).tolist()
            sel_plan = (edge_off, int(sizes[4]))
        else:
            sizes = totals.tolist()
        M, _n_new, max_total, _max_tau = (int(v) for v in sizes[:4])         # the one readback
        if max_total > N:                                   # sparse_gcm.py:120-121
            raise Exception("Overflow")

        nodes = _ops.sparse_insert(nodes, x, T, taus, flags)
        user_code = self.preprocessor is not None or self.positional_encoder is not None
        dirty_nodes = nodes.clone() if user_code else nodes

        if self.edge_selectors:
            new_adj = sel(dirty_nodes, T, taus, B, plan=sel_plan) if sel_plan is not None \
                else sel(dirty_nodes, T, taus, B)
            adj = self._merge(adj, new_adj, sel, B, flags)
        if self.preprocessor:
            dirty_nodes = self.preprocessor(dirty_nodes)
        if self.positional_encoder:
            dirty_nodes = self.positional_encoder(dirty_nodes, T + taus)
        if self.aux_edge_selectors:
            adj = self._merge(adj, self.aux_edge_selectors(dirty_nodes, T, taus, B), self.aux_edge_selectors,
                              B, flags, first=False)

        # sparse_gcm.py:160-164: all weights become 1 while keeping the path to the logits
        v = adj.values()
        unit = not v.requires_grad
        v = v / v.detach() if v.requires_grad else torch.ones_like(v)
        if unit:
            v.gcm_unit_weights = True      # lets gcm.nn.GraphConv skip the multiplication by 1
        adj = torch.sparse_coo_tensor(adj.indices(), v, size=adj.shape, is_coalesced=True)

        flat_nodes = _ops.sparse_flatten(dirty_nodes, T, taus, node_off, M)
        edges, graph = _ops.sparse_edges_to_csr(adj.indices(), node_off, M, B, flags, n_cap=N)
        weights = v
        # (torch_geometric.utils.coalesce(reduce="mean") at sparse_gcm.py:172-175 only
        #  reorders here: the COO list is already duplicate free)
        if self.max_hops is None:
            edges.gcm_graph = graph
            node_feats = self.gnn(flat_nodes, edges, weights)
        elif self._native_gnn():
            mask = _ops.khop_mask(graph, node_off, T, taus, self.max_hops, B, t_pad)
            sub = _ops.GraphIndex(edges, graph.row_ptr, M, mask=mask, batches=graph.batches)
            edges.gcm_graph = sub
            node_feats = self.gnn(flat_nodes, edges, weights)
        else:
            node_feats = self._khop_generic(flat_nodes, edges, weights, graph, node_off, T, taus,
                                            B, t_pad, M)
        mx_dense = _ops.sparse_extract(node_feats, T, taus, node_off, B, t_pad, flags)

        if self.finite_check == "sync":
            bits = int(flags.item())
            if bits:
                flags.zero_()
            assert not bits & _hip.FLAG_MERGE_ORDER, \
                "the stored adjacency has entries at or behind the new nodes (not a state this module produced)"
            assert not bits & _hip.FLAG_ACAUSAL, "Causality violated"
            assert not bits & _hip.FLAG_NONFINITE, \
                "Got NaN in returned memory, try using tanh activation"
        return mx_dense, (nodes, adj, T + taus)

    def _khop_generic(self, flat_nodes, edges, weights, graph, node_off, T, taus, B, t_pad, M):
        """sparse_gcm.py:182-199 for a GNN that is not built from gcm.nn.GraphConv: hand it the
        relabelled k-hop subgraph exactly like torch_geometric.utils.k_hop_subgraph would."""
        mask = _ops.khop_mask(graph, node_off, T, taus, self.max_hops, B, t_pad).bool()
        subset = mask.nonzero().flatten()
        relabel = torch.full((M,), -1, dtype=torch.long, device=mask.device)
        relabel[subset] = torch.arange(subset.numel(), device=mask.device)
        keep = mask[edges[0]] & mask[edges[1]]
        sub_edges = relabel[edges[:, keep]]
        out_sub = self.gnn(flat_nodes[subset], sub_edges, weights[keep])
        full = torch.zeros(M, out_sub.shape[-1], device=out_sub.device)
        return full.index_put((subset,), out_sub)
