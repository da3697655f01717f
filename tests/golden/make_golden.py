#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/*.npz by running the
REFERENCE itself (imported from /root/reference/src) in the build container.

Run:  python tests/golden/make_golden.py            (needs /root/reference)

The reference's Python never travels to the GPU box; only the .npz vectors
(inputs + expected outputs) produced here are committed.  This script is kept
for reproducibility and refuses to run where /root/reference is absent.

Import-time names the reference needs but this image lacks are
satisfied with EMPTY placeholder modules - the dense path never calls into
them.  The only third-party arithmetic on the path (torch_geometric's
DenseGraphConv / GraphConv / coalesce / k_hop_subgraph, not vendored by the
reference and not installable here) is supplied by the plain-torch restatement
in oracle/pyg.py; that boundary is "parity unpinned" (see oracle/__init__.py).
"""
import json
import os
import sys
import types
from typing import Any

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"


def install_placeholders():
    sys.path.insert(0, ROOT)
    from oracle import pyg

    tg = types.ModuleType("torch_geometric")
    tg.nn = types.ModuleType("torch_geometric.nn")
    tg.utils = types.ModuleType("torch_geometric.utils")
    tg.data = types.ModuleType("torch_geometric.data")
    tg.data.Data = type("Data", (), {})
    tg.data.Batch = type("Batch", (), {})
    tg.utils.coalesce = pyg.coalesce
    tg.utils.k_hop_subgraph = pyg.k_hop_subgraph
    ts = types.ModuleType("torch_scatter")
    ts.scatter_max = ts.scatter = None
    tt = types.ModuleType("torchtyping")

    class TensorType:
        def __class_getitem__(cls, item):
            return Any

    tt.TensorType = TensorType
    tt.patch_typeguard = lambda: None
    tgd = types.ModuleType("typeguard")
    tgd.typechecked = lambda f: f
    for name, mod in [("torch_geometric", tg), ("torch_geometric.nn", tg.nn),
                      ("torch_geometric.utils", tg.utils), ("torch_geometric.data", tg.data),
                      ("torch_scatter", ts), ("torchtyping", tt), ("typeguard", tgd)]:
        sys.modules[name] = mod
    sys.path.insert(0, REF)


ONLY = sys.argv[1] if len(sys.argv) > 1 else ""   # substring filter: regenerate matching fixtures only


def save(name, meta, **arrays):
    if ONLY and ONLY not in name:
        return
    out = {"meta": np.array(json.dumps(meta))}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KB")


def params_of(module, prefix="param:"):
    return {prefix + k: v.clone() for k, v in module.state_dict().items()}


def staggered_state(B, N, F, starts, gen):
    nodes = torch.zeros(B, N, F)
    adj = torch.zeros(B, N, N)
    for b, n in enumerate(starts):
        nodes[b, :n] = torch.rand(n, F, generator=gen)
        for i in range(1, n):
            adj[b, i, i - 1] = 1
    return nodes, adj, torch.zeros(0), torch.tensor(starts, dtype=torch.long)


def run_dense(name, gcm_mod, obs, h0, gnn, sel_module=None, extra=None, meta=None):
    """Drive reference DenseGCM for T steps, record outputs + grads."""
    T = obs.shape[0]
    obs = obs.clone().requires_grad_(True)
    hidden = None if h0 is None else tuple(t.clone() for t in h0)
    mxs, adj_sums = [], []
    for t in range(T):
        mx, hidden = gcm_mod(obs[t], hidden)
        mxs.append(mx)
        adj_sums.append(hidden[1].detach().sum(dim=(1, 2)))
    mxs = torch.stack(mxs)
    mxs.mean().backward()
    arrays = dict(obs=obs.detach(), mx=mxs, hT_nodes=hidden[0], hT_adj=hidden[1],
                  hT_num_nodes=hidden[3], adj_sums=torch.stack(adj_sums), grad_obs=obs.grad)
    if h0 is not None:
        arrays.update(h0_nodes=h0[0], h0_adj=h0[1], h0_weights=h0[2], h0_num_nodes=h0[3])
    arrays.update(params_of(gnn))
    for k, p in gnn.named_parameters():
        arrays["grad:" + k] = p.grad.clone()
    if sel_module is not None:
        arrays.update(params_of(sel_module, "sel_param:"))
        for k, p in sel_module.named_parameters():
            if p.grad is not None:
                arrays["sel_grad:" + k] = p.grad.clone()
    arrays.update(extra() if callable(extra) else (extra or {}))
    save(name, meta or {}, **arrays)


def main():
    assert os.path.isdir(REF), "golden vectors can only be generated where /root/reference exists"
    install_placeholders()
    from oracle import dense as od, pyg, sparse as osp
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.temporal import TemporalBackedge
    from gcm.edge_selectors.distance import EuclideanEdge, CosineEdge, SpatialEdge
    from gcm.edge_selectors.dense import DenseEdge
    from gcm.edge_selectors.learned import LearnedEdge
    from gcm.sparse_gcm import SparseGCM
    from gcm.sparse_edge_selectors.temporal import TemporalEdge
    import gcm.util

    DenseGCM.did_warn = True  # silence the one-time print

    # ---- G1 / G2: temporal back edges, staggered starts, crosses overflow ----
    for name, hops, direction in [("g1_temporal_h1", [1], "forward"),
                                  ("g2_temporal_h124_both", [1, 2, 4], "both")]:
        torch.manual_seed(0)
        gen = torch.Generator().manual_seed(1)
        B, N, F, H, T = 4, 32, 8, 32, 40
        gnn = od.canonical_gnn(F, H)
        m = DenseGCM(gnn, edge_selectors=TemporalBackedge(hops, direction=direction), graph_size=N)
        h0 = staggered_state(B, N, F, [0, 3, 31, 32], gen)
        obs = torch.rand(T, B, F, generator=gen)
        run_dense(name, m, obs, h0, gnn,
                  meta=dict(B=B, N=N, F=F, H=H, T=T, selector="temporal", hops=hops, direction=direction))

    # ---- G1b: cfg1 exactly (fresh state, T=32) -------------------------------
    torch.manual_seed(0)
    B, N, F, H, T = 4, 32, 8, 32, 32
    gnn = od.canonical_gnn(F, H)
    m = DenseGCM(gnn, edge_selectors=TemporalBackedge([1]), graph_size=N)
    run_dense("g1b_cfg1", m, torch.rand(T, B, F), None, gnn,
              meta=dict(B=B, N=N, F=F, H=H, T=T, selector="temporal", hops=[1], direction="forward"))

    # ---- G3 / G4: distance selectors on clustered data -----------------------
    def clustered(T, B, F, gen, n_c=4, spread=0.05, scale=4.0):
        centres = scale * torch.randn(n_c, F, generator=gen)
        k = torch.arange(T) % n_c
        return centres[k][:, None, :] + spread * torch.randn(T, B, F, generator=gen)

    B, N, F, H = 4, 16, 8, 16
    specs = [
        ("g3_euclid", lambda: EuclideanEdge(2.0), dict(selector="euclid", max_distance=2.0), 14, False),
        # every graph sits in a different cluster at time t: the cross-batch mean decides differently
        # from a per-graph distance; the threshold is put in the widest gap of the observed distances
        ("g3_euclid_mixed", lambda: EuclideanEdge(None), dict(selector="euclid"), 20, True),
        ("g4_spatial", lambda: SpatialEdge(1.0, slice(0, 3)), dict(selector="spatial", max_distance=1.0, a=[0, 3], b=[0, 3]), 20, False),
        ("g4_spatial_ab", lambda: SpatialEdge(1.0, slice(0, 3), slice(4, 7)), dict(selector="spatial", max_distance=1.0, a=[0, 3], b=[4, 7]), 14, False),
        ("g4_cosine", lambda: CosineEdge(0.5), dict(selector="cosine", max_distance=0.5), 14, True),
        ("g3_euclid_learned", lambda: EuclideanEdge(2.0, learned=True), dict(selector="euclid", max_distance=2.0, learned=True), 14, False),
    ]

    def live_distances(m, sel, obs):
        """All (d - nothing) values the selector compares against its threshold, live entries only."""
        hidden, vals = None, []
        T = obs.shape[0]
        for t in range(T):
            _, hidden = m(obs[t], hidden)
            n_b = hidden[3] - 1
            scale = sel.dist_param.detach() if sel.learned else 1.0
            d = sel.dist_fn(hidden[0][torch.arange(B), n_b] / scale, hidden[0] / scale).detach()
            live = torch.arange(N)[None, :] < n_b[:, None]
            vals.append(d[live])
        return torch.cat(vals)

    for name, mk, meta, T, mixed in specs:
        torch.manual_seed(0)
        gen = torch.Generator().manual_seed(3)
        gnn = od.canonical_gnn(F, H)
        sel = mk()
        obs = clustered(T, B, F, gen)
        if mixed:
            centres = 4.0 * torch.randn(4, F, generator=gen)
            k = (torch.arange(T)[:, None] + torch.arange(B)[None, :]) % 4
            obs = centres[k] + 0.05 * torch.randn(T, B, F, generator=gen)
        if "ab" in name:  # make slice b of past nodes comparable with slice a of the current one
            obs[:, :, 4:7] = obs[:, :, 0:3].roll(1, 0)
        if sel.max_distance is None:
            sel.max_distance = 1e9
            m = DenseGCM(gnn, edge_selectors=sel, graph_size=N)
            v = live_distances(m, sel, obs).sort().values
            lo, hi = int(0.25 * v.numel()), int(0.75 * v.numel())
            gaps = v[lo + 1: hi] - v[lo: hi - 1]
            g = int(gaps.argmax())
            sel.max_distance = float((v[lo + g] + v[lo + g + 1]) / 2)
            meta = dict(meta, max_distance=sel.max_distance)
        m = DenseGCM(gnn, edge_selectors=sel, graph_size=N)
        # threshold margin (keep |d - thr| >= 1e-3 so cdist rounding cannot flip an edge)
        margin = float((live_distances(m, sel, obs) - sel.max_distance).abs().min())
        assert margin >= 1e-3, (name, margin)
        meta = dict(meta, B=B, N=N, F=F, H=H, T=T, margin=margin)
        run_dense(name, m, obs, None, gnn, sel_module=sel, meta=meta)

    # ---- G13: tile-exact shapes (N, F, H multiples of 32): the shapes the live-tile / persistent
    # kernels are specialised for.  Staggered starts, runs through the overflow. ----------------
    torch.manual_seed(13)
    gen = torch.Generator().manual_seed(14)
    B, N, F, H, T = 3, 64, 32, 32, 80
    gnn = od.canonical_gnn(F, H)
    m = DenseGCM(gnn, edge_selectors=TemporalBackedge([1, 2, 4]), graph_size=N)
    h0 = staggered_state(B, N, F, [0, 40, 64], gen)
    run_dense("g13_exact_temporal", m, torch.rand(T, B, F, generator=gen), h0, gnn,
              meta=dict(B=B, N=N, F=F, H=H, T=T, selector="temporal", hops=[1, 2, 4], direction="forward"))
    torch.manual_seed(15)
    gen = torch.Generator().manual_seed(16)
    B, N, F, H, T = 2, 32, 32, 64, 40
    gnn = od.canonical_gnn(F, H)
    m = DenseGCM(gnn, edge_selectors=DenseEdge(), graph_size=N)
    run_dense("g13_exact_dense", m, torch.rand(T, B, F, generator=gen), None, gnn,
              meta=dict(B=B, N=N, F=F, H=H, T=T, selector="dense"))

    # ---- G5: DenseEdge --------------------------------------------------------
    torch.manual_seed(0)
    B, N, F, H, T = 3, 8, 5, 7, 11   # crosses overflow at t=8
    gnn = od.canonical_gnn(F, H)
    m = DenseGCM(gnn, edge_selectors=DenseEdge(), graph_size=N)
    run_dense("g5_dense_edge", m, torch.rand(T, B, F), None, gnn,
              meta=dict(B=B, N=N, F=F, H=H, T=T, selector="dense"))

    # ---- G6: LearnedEdge with captured gumbel noise --------------------------
    torch.manual_seed(0)
    B, N, F, H, T = 4, 12, 8, 8, 10
    gnn = od.canonical_gnn(F, H)
    sel = LearnedEdge(F, num_edge_samples=3)
    noises = []
    real_gs = torch.nn.functional.gumbel_softmax

    def recording_gumbel_softmax(logits, tau=1, hard=False, eps=1e-10, dim=-1):
        state = torch.get_rng_state()
        g = -torch.empty_like(logits).exponential_().log()
        out = ((logits + g) / tau).softmax(dim)
        torch.set_rng_state(state)
        ref = real_gs(logits, tau=tau, hard=hard, dim=dim)   # the real thing, same draws
        if hard:     # one-hot of the same soft sample (straight-through value)
            one_hot = torch.zeros_like(out).scatter_(dim, out.argmax(dim, keepdim=True), 1.0)
            assert torch.equal(ref.detach(), (one_hot - out.detach()) + out.detach())
        else:
            assert torch.equal(ref, out)
        noises.append(g.detach().clone())
        return ref

    torch.nn.functional.gumbel_softmax = recording_gumbel_softmax
    m = DenseGCM(gnn, edge_selectors=sel, graph_size=N)
    obs = torch.randn(T, B, F)

    def noise_arrays():
        # the selector returns early while max(num_nodes) < 1, so step t >= 1 owns draw t-1
        out = {}
        for i, g in enumerate(noises):
            pad = torch.zeros(g.shape[0], N)
            pad[:, : g.shape[1]] = g
            out[f"noise_{i + 1}"] = pad
        return out

    try:
        run_dense("g6_learned", m, obs, None, gnn, sel_module=sel,
                  meta=dict(B=B, N=N, F=F, H=H, T=T, selector="learned", num_edge_samples=3),
                  extra=noise_arrays)
    finally:
        torch.nn.functional.gumbel_softmax = real_gs

    # ---- G14: TemporalBackedge(learned=True): gumbel windows with captured noise --------------
    torch.manual_seed(14)
    gen = torch.Generator().manual_seed(15)
    B, N, F, H, T, W, S = 3, 12, 6, 8, 6, 10, 3
    starts = [0, 2, 4]
    gnn = od.canonical_gnn(F, H)
    sel = TemporalBackedge(learned=True, learning_window=W, num_samples=S)
    with torch.no_grad():
        sel.window.copy_(torch.randn(W, generator=gen))
    noises.clear()
    torch.nn.functional.gumbel_softmax = recording_gumbel_softmax
    m = DenseGCM(gnn, edge_selectors=sel, graph_size=N)
    h0 = staggered_state(B, N, F, starts, gen)

    def window_noise():
        # calls come graph by graph (empty graphs skipped), sample by sample; one [n_b] draw each
        out, it = torch.zeros(T, B, S, W), iter(noises)
        for t in range(T):
            for b in range(B):
                n = starts[b] + t
                for i in range(S if n > 0 else 0):
                    g = next(it)
                    assert g.shape == (n,)
                    out[t, b, i, :n] = g
        assert next(it, None) is None
        return {"noise": out}

    try:
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):      # util.diff_or prints on every call
            run_dense("g14_temporal_learned", m, torch.rand(T, B, F, generator=gen), h0, gnn, sel_module=sel,
                      meta=dict(B=B, N=N, F=F, H=H, T=T, selector="temporal_learned", learning_window=W,
                                num_samples=S, starts=starts),
                      extra=window_noise)
    finally:
        torch.nn.functional.gumbel_softmax = real_gs

    # ---- G7: wrap_overflow exact case (tests/test_gcm.py:105-152) -------------
    B, N, F = 2, 7, 5
    nodes = torch.arange(B * N * F, dtype=torch.float).reshape(B, N, F)
    adj = torch.zeros(B, N, N)
    weights = torch.ones(B, N, N)
    adj[:, 0, :] = 1
    adj[:, :, 0] = 1
    weights[:, 0, :] = 5
    weights[:, :, 0] = 5
    nodes[:, 0] = 0
    torch.manual_seed(0)
    g = pyg.Sequential("x, adj, weights, B, N", [(pyg.DenseGraphConv(F, F), "x, adj -> x"), torch.nn.ReLU()])
    for nm, w in [("g7_wrap_weights", weights), ("g7_wrap_noweights", torch.ones(0))]:
        s = DenseGCM(g)
        num_nodes = torch.tensor([1, 7])
        obs = torch.ones(B, F) * 5
        mx, (n2, a2, w2, nn2) = s(obs, (nodes.clone(), adj.clone(), w.clone(), num_nodes))
        save(nm, dict(B=B, N=N, F=F), obs=obs, h0_nodes=nodes, h0_adj=adj, h0_weights=w,
             h0_num_nodes=torch.tensor([1, 7]), mx=mx, hT_nodes=n2, hT_adj=a2, hT_weights=w2,
             hT_num_nodes=nn2, caller_num_nodes_after=num_nodes, **params_of(g))

    # ---- G10: PositionalEncoding in the step ----------------
    from gcm.gcm import PositionalEncoding
    for mode in ("add", "cat"):
        torch.manual_seed(0)
        B, N, F, H, T = 3, 10, 6, 5, 13
        gnn = od.canonical_gnn(F, H)
        pe = PositionalEncoding(max_len=N, mode=mode, cat_dim=2)
        m = DenseGCM(gnn, edge_selectors=TemporalBackedge([1]), aux_edge_selectors=TemporalBackedge([2]),
                     positional_encoder=pe, graph_size=N)
        obs = torch.rand(T, B, F)
        torch.manual_seed(7)            # the lazily created reproject layer (mode="cat")
        run_dense(f"g10_posenc_{mode}", m, obs, None, gnn, sel_module=pe if mode == "cat" else None,
                  meta=dict(B=B, N=N, F=F, H=H, T=T, mode=mode, cat_dim=2))
    pe = PositionalEncoding(max_len=7, mode="add")
    enc0 = pe(torch.zeros(2, 7, 5), torch.tensor([0, 7]))
    enc1 = pe(torch.zeros(2, 7, 5), torch.tensor([1, 8]))
    save("g10_posenc_table", dict(N=7, F=5), enc0=enc0, enc1=enc1, pe=pe.pe)

    # ---- G15: what the live-row step folds in at shapes its kernels take:
    # a Linear preprocessor (ray_gcm.py:117,131-135), index-writing aux selectors, PositionalEncoding
    # ("add" reaches the GNN through the reference's in-place write, gcm.py:131,294-301; "cat" is seen
    # by the aux selectors only).  Every run crosses the overflow. ---------------------------------
    specs15 = [
        # name, B, N, F_obs, F_gnn, H, T, selector, aux, pe mode, preprocessor bias
        ("g15_fold_pre", 3, 16, 8, 12, 8, 20, lambda: TemporalBackedge([1, 2]), None, None, True),
        ("g15_fold_pre_nobias", 2, 16, 8, 8, 12, 18, lambda: DenseEdge(), None, None, False),
        ("g15_fold_pre_aux_cat", 3, 16, 8, 12, 8, 20, lambda: TemporalBackedge([1]),
         lambda: TemporalBackedge([3], direction="both"), "cat", True),
        ("g15_fold_pe_add", 3, 16, 8, 8, 8, 20, lambda: TemporalBackedge([1]), lambda: TemporalBackedge([2]), "add", None),
        ("g15_fold_pre_exact", 2, 32, 32, 32, 32, 40, lambda: TemporalBackedge([1, 2, 4]), None, None, True),
        ("g15_fold_pe_add_exact", 2, 32, 32, 32, 32, 40, lambda: TemporalBackedge([1, 2, 4]), lambda: DenseEdge(), "add", None),
    ]
    for name, B, N, F, Fg, H, T, mk_sel, mk_aux, mode, pre_bias in specs15:
        torch.manual_seed(15)
        gen = torch.Generator().manual_seed(16)
        gnn = od.canonical_gnn(Fg, H)
        pre = torch.nn.Linear(F, Fg, bias=pre_bias) if pre_bias is not None else None
        pe = PositionalEncoding(max_len=N, mode=mode, cat_dim=2) if mode else None
        holder = torch.nn.ModuleDict({k: v for k, v in (("pre", pre), ("pe", pe)) if v is not None})
        m = DenseGCM(gnn, preprocessor=pre, edge_selectors=mk_sel(), aux_edge_selectors=mk_aux() if mk_aux else None,
                     positional_encoder=pe, graph_size=N)
        obs = torch.rand(T, B, F, generator=gen)
        torch.manual_seed(17)           # the lazily created reproject layer (mode="cat")
        run_dense(name, m, obs, None, gnn, sel_module=holder,
                  meta=dict(B=B, N=N, F=F, Fg=Fg, H=H, T=T, mode=mode, cat_dim=2, pre_bias=pre_bias))

    # ---- G11: pack_hidden / unpack_hidden --------------------
    gen = torch.Generator().manual_seed(11)
    B, N, F, max_edges = 4, 9, 3, 12
    counts = [5, 0, 11, 3]
    idx = []
    for b, c in enumerate(counts):
        pairs = torch.randperm(N * N, generator=gen)[:c].sort().values
        idx.append(torch.stack([torch.full((c,), b), pairs // N, pairs % N]))
    idx = torch.cat(idx, dim=1)
    vals = torch.rand(idx.shape[1], generator=gen)
    adj = torch.sparse_coo_tensor(idx, vals, size=(B, N, N)).coalesce()
    nodes, Tt = torch.rand(B, N, F, generator=gen), torch.tensor([3, 0, 9, 2])
    pn, pe_, pw, pT = gcm.util.pack_hidden((nodes, adj, Tt), B, max_edges)
    un, uadj, uT = gcm.util.unpack_hidden((pn, pe_, pw, pT), B)
    save("g11_pack", dict(B=B, N=N, F=F, max_edges=max_edges), coo=adj.indices(), values=adj.values(),
         nodes=nodes, T=Tt, dense_edges=pe_, dense_weights=pw, un_idx=uadj.coalesce().indices(),
         un_val=uadj.coalesce().values())

    # ---- G12: SparseGCM + sparse LearnedEdge with recorded gumbel noise ----
    from gcm.sparse_edge_selectors.learned import LearnedEdge as SparseLearnedEdge
    real_sgs = gcm.util.sparse_gumbel_softmax
    for name, window in [("g12_sparse_learned", None), ("g12_sparse_learned_win3", 3)]:
        torch.manual_seed(0)
        B, N, F, H = 3, 12, 4, 5
        gnn = osp.canonical_gnn(F, H, act=torch.nn.Tanh)
        sel = SparseLearnedEdge(F, num_edge_samples=3, window=window, store_grads=False)
        noises = []

        def recording(logits, dim, tau=1, hard=False):
            state = torch.get_rng_state()
            out = real_sgs(logits, dim=dim, tau=tau, hard=hard)
            after = torch.get_rng_state()
            torch.set_rng_state(state)
            noises.append(-torch.empty_like(logits.coalesce().values()).exponential_().log())
            torch.set_rng_state(after)
            return out

        gcm.util.sparse_gumbel_softmax = recording
        try:
            m = SparseGCM(gnn, edge_selectors=sel, graph_size=N)
            obs = torch.randn(B, 9, F).requires_grad_(True)
            plan = [torch.tensor([3, 1, 2]), torch.tensor([2, 2, 2]), torch.tensor([4, 1, 3])]
            hidden, outs, pos = None, [], torch.zeros(B, dtype=torch.long)
            for taus in plan:
                t = int(taus.max())
                x = torch.zeros(B, t, F)
                for b in range(B):
                    x[b, : taus[b]] = obs[b, pos[b]: pos[b] + taus[b]]
                out, hidden = m(x, taus, hidden)
                outs.append(out)
                pos = pos + taus
            loss = sum(o.sum() for o in outs) / sum(o.numel() for o in outs)
            loss.backward()
        finally:
            gcm.util.sparse_gumbel_softmax = real_sgs
        arrays = dict(obs=obs.detach(), grad_obs=obs.grad, taus=torch.stack(plan), hT_nodes=hidden[0],
                      hT_adj_indices=hidden[1].coalesce().indices(),
                      hT_adj_values=hidden[1].coalesce().values().detach(), hT_T=hidden[2])
        for i, o in enumerate(outs):
            arrays[f"out{i}"] = o
        for i, g in enumerate(noises):
            arrays[f"noise_{i}"] = g
        arrays.update(params_of(gnn))
        arrays.update(params_of(sel, "sel_param:"))
        for k, p in gnn.named_parameters():
            arrays["grad:" + k] = p.grad.clone()
        for k, p in sel.named_parameters():
            if p.grad is not None:
                arrays["sel_grad:" + k] = p.grad.clone()
        save(name, dict(B=B, N=N, F=F, H=H, window=window, num_edge_samples=3), **arrays)
    # candidate enumeration alone
    Tq, tq = torch.tensor([0, 3, 7, 1, 0]), torch.tensor([4, 2, 0, 1, 1])
    save("g12_causal_edges", dict(), T=Tq, taus=tq, all=gcm.util.get_causal_edges(Tq, tq),
         win2=gcm.util.get_causal_edges(Tq, tq, window=2), win0=gcm.util.get_causal_edges(Tq, tq, window=0))

    # ---- G8: SparseGCM + TemporalEdge ----------------------------------------
    def run_sparse(name, B, N, F, H, hops, obs, tau_plan, max_hops=None, act=None, aux_hops=None):
        torch.manual_seed(0)
        gnn = osp.canonical_gnn(F, H, act=act)
        m = SparseGCM(gnn, edge_selectors=TemporalEdge(hops), graph_size=N, max_hops=max_hops,
                      aux_edge_selectors=TemporalEdge(aux_hops) if aux_hops else None)
        obs = obs.clone().requires_grad_(True)
        hidden, outs, pos = None, [], torch.zeros(B, dtype=torch.long)
        for taus in tau_plan:
            t = int(taus.max())
            x = torch.zeros(B, t, F)
            for b in range(B):
                x[b, : taus[b]] = obs[b, pos[b]: pos[b] + taus[b]]
            out, hidden = m(x, taus, hidden)
            outs.append(out)
            pos = pos + taus
        loss = sum(o.sum() for o in outs) / sum(o.numel() for o in outs)
        loss.backward()
        arrays = dict(obs=obs.detach(), grad_obs=obs.grad, taus=torch.stack(tau_plan),
                      hT_nodes=hidden[0], hT_adj_indices=hidden[1].coalesce().indices(),
                      hT_adj_values=hidden[1].coalesce().values().detach(), hT_T=hidden[2])
        for i, o in enumerate(outs):
            arrays[f"out{i}"] = o
        arrays.update(params_of(gnn))
        for k, p in gnn.named_parameters():
            arrays["grad:" + k] = p.grad.clone()
        save(name, dict(B=B, N=N, F=F, H=H, hops=hops, max_hops=max_hops, act=bool(act),
                        aux_hops=aux_hops), **arrays)

    B, N, F, ts = 3, 8, 3, 8
    ar = torch.arange(B * ts * F, dtype=torch.float32).reshape(B, ts, F)
    run_sparse("g8_sparse_oneshot", B, N, F, F, [1, 2], ar, [torch.full((B,), ts)])
    run_sparse("g8_sparse_oneshot_2hop", B, N, F, F, [1, 2], ar, [torch.full((B,), ts)], max_hops=2)
    run_sparse("g8_sparse_stepwise", B, N, F, F, [1, 2], ar, [torch.ones(B, dtype=torch.long)] * ts)
    gen = torch.Generator().manual_seed(5)
    B, N, F, H = 4, 32, 8, 16
    robs = torch.rand(B, 30, F, generator=gen)
    plan = [torch.tensor([3, 1, 4, 2]), torch.tensor([5, 5, 1, 3]), torch.tensor([10, 2, 7, 9]),
            torch.tensor([1, 1, 1, 1])]
    run_sparse("g8_sparse_ragged", B, N, F, H, [1, 3], robs, plan, act=torch.nn.Tanh)
    run_sparse("g8_sparse_ragged_2hop", B, N, F, H, [1, 3], robs, plan, max_hops=2, act=torch.nn.Tanh)
    # main + aux selectors (sparse_gcm.py:146-152): the aux edges end in the same new nodes as the
    # main selector's, interleave with them and (second case) duplicate some of them
    run_sparse("g16_sparse_aux", B, N, F, H, [1], robs, plan, act=torch.nn.Tanh, aux_hops=[2])
    run_sparse("g16_sparse_aux_overlap", B, N, F, H, [1, 3], robs, plan, act=torch.nn.Tanh, aux_hops=[3, 2, 1])


if __name__ == "__main__":
    main()
