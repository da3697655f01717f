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
        self._fast_plan = None    # (structure analysed once): the canonical configuration's C++ host path
        self.fast_host = True     # False: the layered Python path always (A/B tests)
        # One node per call (x [B, 1, F]) in a chain from empty graphs: the new node's belief from the chain's
        # h1 / agg1 / x caches in ONE launch, one time-parallel backward launch per chain (step_ext.cpp:
        # SparseChain).  False: every call runs both GraphConv layers over every stored node.
        self.stepwise_cache = True
        self._chain = None

    def __getstate__(self):
        """copy.deepcopy / pickle of a module that has already run: the analysed structure, the flag words and
        the stepwise chain are runtime caches (rebuilt on the first call)."""
        d = dict(self.__dict__)
        d.update(_fast_plan=None, _flags={}, _chain=None)
        return d

    def get_initial_hidden_state(self, x):
        """sparse_gcm.py:55-70."""
        assert x.dim() == 3
        B, _, feats = x.shape
        nodes = torch.zeros(B, self.graph_size, feats, device=x.device)
        adj = torch.zeros((B, self.graph_size, self.graph_size), device=x.device,
                          layout=torch.sparse_coo)
        T = torch.zeros(B, dtype=torch.long, device=x.device)
        T._gcm_fresh = T._version      # (all zero by construction: lets a stepwise chain start on the caches,
        nodes._gcm_fresh = nodes._version   # and the insert kernel skip reading the zero node matrix)
        return nodes, adj, T

    def _empty_adj(self, x):
        return torch.zeros((x.shape[0], self.graph_size, self.graph_size), device=x.device, layout=torch.sparse_coo)

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

    def _canonical(self):
        """(hops, conv1, act1, conv2, act2) when this module is the canonical configuration - TemporalEdge
        selector, two GraphConv layers (each optionally followed by Tanh / ReLU), nothing else - whose whole
        forward runs as ONE call into the C++ host path (csrc/torch_ext/step_ext.cpp: sparse_temporal_step):
        plan, the one readback, insert, edges, merge, flatten, CSR, both layers, extract, one autograd node.
        None: the layered path below (any other selector / GNN / option)."""
        if self._fast_plan is None:
            from . import _ext
            from .sparse_edge_selectors.temporal import TemporalEdge
            plan = False
            ext = _ext.module()
            g, sel = self.gnn, self.edge_selectors
            if (ext is not None and hasattr(ext, "sparse_temporal_step") and type(sel) is TemporalEdge
                    and self.aux_edge_selectors is None and self.preprocessor is None
                    and self.positional_encoder is None and self.max_hops is None
                    and isinstance(g, _nn.Sequential) and len(g.arg_names) == 3):
                xn = g.arg_names[0]
                convs, acts = [], []
                ok = True
                for mod, ins, outs in g.stages():
                    if isinstance(mod, _nn.GraphConv) and ins == g.arg_names and outs == [xn]:
                        convs.append(mod)
                        acts.append(_hip.ACT_NONE)
                    elif (type(mod) in _nn._FUSABLE and convs and ins == [xn] and outs == [xn]
                          and acts[-1] == _hip.ACT_NONE):
                        acts[-1] = _nn._FUSABLE[type(mod)]
                    else:
                        ok = False
                        break
                if ok and len(convs) == 2 and convs[0].out_channels == convs[1].in_channels:
                    # the modules the one C++ call stands in for: a hook on any of them must still fire, so the
                    # call then takes the layered path (their hook dicts are consulted at every call)
                    mods = [g, sel, convs[0], convs[1], convs[0].lin_rel, convs[0].lin_root, convs[1].lin_rel,
                            convs[1].lin_root] + [m for m, _, _ in g.stages()]
                    hooks = [d for m in mods for d in (m._forward_hooks, m._forward_pre_hooks, m._backward_hooks,
                                                       m._backward_pre_hooks)]
                    plan = (ext.sparse_temporal_step, sel._hops_desc, convs[0], acts[0], convs[1], acts[1], hooks)
            self.__dict__["_fast_plan"] = plan
        plan = self._fast_plan
        if plan and any(plan[6]):      # (a hook was registered since)
            return None
        return plan or None

    # what _canonical() looked at: re-assigning any of it re-runs the analysis
    _PLAN_ATTRS = frozenset(("gnn", "edge_selectors", "aux_edge_selectors", "preprocessor", "positional_encoder",
                             "max_hops"))

    def __setattr__(self, name, value):
        if name in SparseGCM._PLAN_ATTRS and "_fast_plan" in self.__dict__:
            self.__dict__["_fast_plan"] = None
            self.__dict__["_chain"] = None
        super().__setattr__(name, value)

    def rollout(self, x, hidden=None, taus=None):
        """The time-batched entry (the shape RLlib's wrapper holds: x [B, T, feat], batch first):
        T memory steps of every graph in ONE call - which is what forward() already is for SparseGCM (the reference's
        own tests pin one call == T single-node calls, tests/test_sparse_gcm.py:395-540): forward(x, taus = T for every
        graph, hidden).  A stepwise caller (x [B, 1, F] per call, ray_sparse_gcm.py:198-213) pays ~50 us of host and
        launch cost per call; whole episodes through here run at the one-shot rate (bench.py --config cfg4)."""
        if taus is None:
            taus = torch.full((x.shape[0],), x.shape[1], dtype=torch.long, device=x.device)
        return self(x, taus, hidden)

    def forward(self, x, taus, hidden):
        """x [B, t, feat] zero padded in t; taus [B] valid lengths; hidden (nodes, adj, T) or
        None.  Returns (mx [B, t, H] zero padded, (nodes, adj, T + taus))."""
        fast = self._canonical() if self.fast_host else None
        lazy = False
        if hidden is None:
            if (fast is not None and x.is_cuda and x.dim() == 3 and fast[2].in_channels == x.shape[-1]
                    and x.device.index == torch.cuda.current_device()):
                # empty graphs on the one-call host path: the all-zero node matrix [B, N, F] (33 MB at cfg4) is
                # neither filled nor read - the insert kernel knows it is zero
                lazy = True
                hidden = (None, self._empty_adj(x), torch.zeros(x.shape[0], dtype=torch.long, device=x.device))
            else:
                hidden = self.get_initial_hidden_state(x)
        nodes, adj, T = hidden
        fresh = lazy or (getattr(T, "_gcm_fresh", None) == T._version
                         and getattr(nodes, "_gcm_fresh", None) == nodes._version)
        assert x.dim() == 3 and x.dtype == torch.float32
        assert taus.dtype == torch.long and T.dtype == torch.long
        adj = adj.coalesce()
        N = self.graph_size if lazy else nodes.shape[1]
        B, t_pad, _ = x.shape
        flags = self._flag_word(x.device)

        if (fast is not None and x.is_cuda and not adj.values().requires_grad
                and fast[2].in_channels == x.shape[-1] and x.device.index == torch.cuda.current_device()):
            fn, hops, c1, a1, c2, a2 = fast[:6]
            chain = None
            if self.stepwise_cache and t_pad == 1:
                chain = self._chain
                if chain is None:
                    from . import _ext
                    chain = self._chain = _ext.module().SparseChain()
            # (current parameter tensors through the modules' own dicts: nn.Module.__getattr__ chains cost
            #  ~0.5 us each and this runs every call)
            try:
                r1, t1, r2, t2 = (c1._modules["lin_rel"]._parameters, c1._modules["lin_root"]._parameters,
                                  c2._modules["lin_rel"]._parameters, c2._modules["lin_root"]._parameters)
                ws = (r1["weight"], r1["bias"], t1["weight"], r2["weight"], r2["bias"], t2["weight"])
            except KeyError:       # (parametrized / re-registered weights: the attribute protocol)
                ws = (c1.lin_rel.weight, c1.lin_rel.bias, c1.lin_root.weight,
                      c2.lin_rel.weight, c2.lin_rel.bias, c2.lin_root.weight)
            r = fn(x, taus, nodes, N, adj.indices(), T, hops, ws[0], ws[1], ws[2], a1, ws[3], ws[4], ws[5], a2,
                   flags, chain, fresh, self.finite_check == "sync")
            if r == 1:                                         # sparse_gcm.py:120-121
                raise Exception("Overflow")
            mx_dense, nodes_out, idx, vals, T_out, bits = r
            vals.gcm_unit_weights = True
            adj_out = torch.sparse_coo_tensor(idx, vals, size=adj.shape, is_coalesced=True)
            self._check_flags(flags, bits)      # (bits >= 0: the flag word came back with the call's sizes)
            return mx_dense, (nodes_out, adj_out, T_out)

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

        self._check_flags(flags)
        return mx_dense, (nodes, adj, T + taus)

    def _check_flags(self, flags, bits=-1):
        if self.finite_check == "sync":
            fast = self._fast_plan
            if bits >= 0:
                pass
            elif fast and flags.is_cuda and flags.device.index == torch.cuda.current_device():
                from . import _ext
                bits = _ext.module().read_flag_word(flags)
            else:
                bits = int(flags.item())
            if bits:
                flags.zero_()
            assert not bits & _hip.FLAG_MERGE_ORDER, \
                "the stored adjacency has entries at or behind the new nodes (not a state this module produced)"
            assert not bits & _hip.FLAG_ACAUSAL, "Causality violated"
            assert not bits & _hip.FLAG_NONFINITE, \
                "Got NaN in returned memory, try using tanh activation"

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
