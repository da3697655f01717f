"""Parity of the HIP SparseGCM path against the CPU oracle and the reference's golden
vectors.  Needs an MI355X."""
import pytest
import torch

from _golden import Fixture
from oracle import dense as od, pyg, sparse as osp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev_sparse_gnn(ref, F, H, act, layers=2):
    from gcm import nn as G
    mods, cin = [], F
    for _ in range(layers):
        mods.append((G.GraphConv(cin, H), "x, edges, weights -> x"))
        if act is not None:
            mods.append(act())
        cin = H
    g = G.Sequential("x, edges, weights", mods)
    g.load_state_dict(ref.state_dict())
    return g.to(DEV)


@pytest.mark.parametrize("M,E,Fi,Fo,act,weighted", [
    (5, 0, 3, 3, None, False), (40, 90, 8, 16, "tanh", True), (300, 1500, 32, 32, "tanh", False),
    (1000, 4000, 33, 70, "relu", True), (129, 700, 128, 128, "tanh", True),
    # Fi = Fo = 32: the third-generation kernels (one wave per 32 rows; forward, and the backward with the
    # transpose aggregation folded in) - ragged last tile, heavy fan-in / fan-out, no edges at all
    (5000, 30000, 32, 32, "tanh", True), (1000, 2500, 32, 32, "relu", False), (65, 4000, 32, 32, None, True),
    (4096, 0, 32, 32, "tanh", False), (3333, 9000, 32, 64, "tanh", True),
])
def test_csr_graphconv_kernel(M, E, Fi, Fo, act, weighted):
    """Generic edge_index (unsorted, duplicates allowed) through gcm.nn.GraphConv."""
    from gcm import nn as G
    torch.manual_seed(M + E)
    ref = pyg.GraphConv(Fi, Fo)
    dev = G.GraphConv(Fi, Fo)
    dev.load_state_dict(ref.state_dict())
    dev = dev.to(DEV)
    x = torch.randn(M, Fi)
    ei = torch.randint(0, M, (2, E))
    w = torch.rand(E) if weighted else None
    acts = {"tanh": torch.tanh, "relu": torch.relu, None: lambda t: t}
    code = {"tanh": 1, "relu": 2, None: 0}[act]
    xc = x.clone().requires_grad_(True)
    wc = w.clone().requires_grad_(True) if weighted else None
    xd = x.to(DEV).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True) if weighted else None
    yc = acts[act](ref(xc, ei, wc))
    yd = dev(xd, ei.to(DEV), wd, _act=code)
    torch.testing.assert_close(yd.cpu(), yc, rtol=1e-5, atol=1e-5)
    g = torch.randn_like(yc)
    yc.backward(g)
    yd.backward(g.to(DEV))
    torch.testing.assert_close(xd.grad.cpu(), xc.grad, rtol=1e-5, atol=1e-5 * float(xc.grad.abs().max()) + 1e-9)
    if weighted and E:
        torch.testing.assert_close(wd.grad.cpu(), wc.grad, rtol=1e-5, atol=1e-5 * float(wc.grad.abs().max()) + 1e-9)
    for (k, pc), (_, pd) in zip(ref.named_parameters(), dev.named_parameters()):
        scale = float(pc.grad.abs().max()) + 1e-6
        torch.testing.assert_close(pd.grad.cpu(), pc.grad, rtol=1e-5, atol=1e-5 * scale + 1e-5, msg=k)


def test_csr_graphconv_constant_edge_weights():
    """Edge weights that carry no gradient: the fused backward (transpose aggregation inside the row
    kernel) reads them through the CSC permutation."""
    from gcm import nn as G
    torch.manual_seed(11)
    M, E, F = 2000, 7000, 32
    ref = pyg.GraphConv(F, F)
    dev = G.GraphConv(F, F)
    dev.load_state_dict(ref.state_dict())
    dev = dev.to(DEV)
    x, ei, w = torch.randn(M, F), torch.randint(0, M, (2, E)), torch.rand(E)
    xc, xd = x.clone().requires_grad_(True), x.to(DEV).requires_grad_(True)
    yc = torch.tanh(ref(xc, ei, w))
    yd = dev(xd, ei.to(DEV), w.to(DEV), _act=1)
    torch.testing.assert_close(yd.cpu(), yc, rtol=1e-5, atol=1e-5)
    g = torch.randn_like(yc)
    yc.backward(g)
    yd.backward(g.to(DEV))
    torch.testing.assert_close(xd.grad.cpu(), xc.grad, rtol=1e-5, atol=1e-5 * float(xc.grad.abs().max()) + 1e-9)
    for (k, pc), (_, pd) in zip(ref.named_parameters(), dev.named_parameters()):
        scale = float(pc.grad.abs().max()) + 1e-6
        torch.testing.assert_close(pd.grad.cpu(), pc.grad, rtol=1e-5, atol=1e-5 * scale + 1e-5, msg=k)


def test_temporal_edge_matches_oracle():
    from gcm.sparse_edge_selectors.temporal import TemporalEdge
    T = torch.tensor([0, 3, 0, 10, 1, 0])
    taus = torch.tensor([5, 2, 0, 7, 1, 1])
    for hops in ([1], [1, 2], [4], [3, 1], [2, 7, 1]):
        want = osp.TemporalEdge(hops)(None, T, taus, 6).coalesce()
        got = TemporalEdge(hops)(None, T.to(DEV), taus.to(DEV), 6)
        assert got.is_coalesced()
        assert torch.equal(got.indices().cpu(), want.indices()), hops
        assert torch.equal(got.values().cpu(), want.values())
        assert tuple(got.shape) == tuple(want.shape)


@pytest.mark.parametrize("B,N,p", [(5, 16, 0.5), (3, 200, 0.05), (7, 64, 1.0), (4, 33, 0.0), (2, 700, 0.02)])
def test_csc_view_without_sort(B, N, p):
    """The backward's CSC view (one wave per graph, LDS counters) against a stable sort by source."""
    from gcm import _ops
    torch.manual_seed(B * N)
    counts = torch.randint(0, N + 1, (B,))
    counts[0] = N
    coo = []
    for b in range(B):
        n = int(counts[b])
        m = torch.tril(torch.rand(n, n) < p, diagonal=-1)
        snk, src = m.nonzero(as_tuple=True)
        coo.append(torch.stack([torch.full_like(snk, b), snk, src]))
    coo = torch.cat(coo, dim=1).to(DEV)
    node_off = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)]).to(DEV)
    M = int(node_off[-1])
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    edges, graph = _ops.sparse_edges_to_csr(coo, node_off, M, B, flags, n_cap=N)
    col_ptr, rows, perm = graph.csc()
    plain = _ops.GraphIndex(edges, graph.row_ptr, M)            # the sort-based construction
    want_ptr, want_rows, want_perm = plain.csc()
    assert torch.equal(col_ptr, want_ptr)
    assert torch.equal(rows, want_rows) and torch.equal(perm, want_perm)
    assert int(flags.item()) == 0


def test_khop_mask_matches_k_hop_subgraph():
    from gcm import _ops
    torch.manual_seed(1)
    B = 4
    T = torch.tensor([6, 0, 9, 3])
    taus = torch.tensor([2, 4, 1, 3])
    tot = T + taus
    off = torch.cat([torch.zeros(1, dtype=torch.long), tot.cumsum(0)])
    M = int(off[-1])
    coo = osp.TemporalEdge([1, 3])(None, torch.zeros(B, dtype=torch.long), tot, B).coalesce().indices()
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    edges, graph = _ops.sparse_edges_to_csr(coo.to(DEV), off.to(DEV), M, B, flags)
    out_idx = torch.cat([torch.arange(off[b] + T[b], off[b] + tot[b]) for b in range(B)])
    ref_edges = torch.stack([coo[2] + off[coo[0]], coo[1] + off[coo[0]]])
    assert torch.equal(edges.cpu(), ref_edges)
    for hops in (0, 1, 2, 3):
        mask = _ops.khop_mask(graph, off.to(DEV), T.to(DEV), taus.to(DEV), hops, B, 4).cpu().bool()
        subset, _, _, _ = pyg.k_hop_subgraph(out_idx, hops, ref_edges, relabel_nodes=True, num_nodes=M)
        want = torch.zeros(M, dtype=torch.bool)
        want[subset] = True
        assert torch.equal(mask, want), hops
    assert int(flags.item()) == 0


SPARSE = ["g8_sparse_oneshot", "g8_sparse_oneshot_2hop", "g8_sparse_stepwise",
          "g8_sparse_ragged", "g8_sparse_ragged_2hop",
          # main + aux selectors (the aux merge must coalesce - its edges interleave with / duplicate
          # the main selector's)
          "g16_sparse_aux", "g16_sparse_aux_overlap"]


@pytest.mark.parametrize("name", SPARSE)
def test_sparse_rollout_matches_reference(name):
    from gcm.sparse_gcm import SparseGCM
    from gcm.sparse_edge_selectors.temporal import TemporalEdge
    fx = Fixture(name)
    m = fx.meta
    act = torch.nn.Tanh if m["act"] else None
    ref = osp.canonical_gnn(m["F"], m["H"], act=act)
    ref.load_state_dict(fx.group("param:"))
    g = dev_sparse_gnn(ref, m["F"], m["H"], act)
    mem = SparseGCM(g, edge_selectors=TemporalEdge(m["hops"]), graph_size=m["N"], max_hops=m["max_hops"],
                    aux_edge_selectors=TemporalEdge(m["aux_hops"]) if m.get("aux_hops") else None)
    obs = fx["obs"].to(DEV).requires_grad_(True)
    B = m["B"]
    hidden, outs, pos = None, [], torch.zeros(B, dtype=torch.long)
    for taus in fx["taus"]:
        t = int(taus.max())
        rows = [torch.cat([obs[b, pos[b]: pos[b] + taus[b]],
                           torch.zeros(t - int(taus[b]), m["F"], device=DEV)]) for b in range(B)]
        x = torch.stack(rows)
        out, hidden = mem(x, taus.to(DEV), hidden)
        outs.append(out)
        pos = pos + taus
    loss = sum(o.sum() for o in outs) / sum(o.numel() for o in outs)
    loss.backward()
    for i, o in enumerate(outs):
        torch.testing.assert_close(o.cpu(), fx[f"out{i}"], rtol=1e-5, atol=1e-5)
    assert torch.equal(hidden[0].cpu(), fx["hT_nodes"])
    assert torch.equal(hidden[1].coalesce().indices().cpu(), fx["hT_adj_indices"])   # bit exact
    assert torch.equal(hidden[1].coalesce().values().cpu(), fx["hT_adj_values"])
    assert torch.equal(hidden[2].cpu(), fx["hT_T"])
    gs = float(fx["grad_obs"].abs().max())
    torch.testing.assert_close(obs.grad.cpu(), fx["grad_obs"], rtol=1e-5, atol=1e-5 * gs)
    for k, p in g.named_parameters():
        want = fx["grad:" + k]
        torch.testing.assert_close(p.grad.cpu(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()) + 1e-7, msg=k)


def test_dense_equals_sparse():
    """tests/test_sparse_gcm.py:395-429 - same weights, TemporalBackedge([1,2]) vs
    TemporalEdge([1,2]): equal node matrices, equal edge sets, equal outputs."""
    from gcm.gcm import DenseGCM
    from gcm.sparse_gcm import SparseGCM
    from gcm import nn as G
    from gcm.edge_selectors.temporal import TemporalBackedge
    from gcm.sparse_edge_selectors.temporal import TemporalEdge
    F, B, ts = 3, 3, 8
    torch.manual_seed(0)
    dense_g = G.Sequential("x, adj, weights, B, N", [(G.DenseGraphConv(F, F), "x, adj -> x"),
                                                      (G.DenseGraphConv(F, F), "x, adj -> x")])
    sparse_g = G.Sequential("x, edges, weights", [(G.GraphConv(F, F), "x, edges, weights -> x"),
                                                   (G.GraphConv(F, F), "x, edges, weights -> x")])
    sparse_g.load_state_dict(dense_g.state_dict())
    dense = DenseGCM(dense_g.to(DEV), edge_selectors=TemporalBackedge([1, 2]), graph_size=8)
    obs = torch.arange(B * ts * F, dtype=torch.float32).reshape(B, ts, F).to(DEV)
    dh, douts = None, []
    for i in range(ts):
        o, dh = dense(obs[:, i], dh)
        douts.append(o)
    douts = torch.stack(douts, dim=1)
    taus = torch.full((B,), ts, dtype=torch.long, device=DEV)
    for max_hops in (None, 2):
        sparse = SparseGCM(sparse_g.to(DEV), edge_selectors=TemporalEdge([1, 2]), graph_size=8,
                           max_hops=max_hops)
        souts, sh = sparse(obs, taus, None)
        assert torch.equal(dh[0], sh[0])
        assert torch.equal(dh[1].nonzero().T, sh[1].coalesce().indices())
        torch.testing.assert_close(souts, douts, rtol=1e-6, atol=1e-3)   # values reach ~1e4 here


def test_sparse_overflow_raises():
    """sparse_gcm.py:120-121."""
    from gcm.sparse_gcm import SparseGCM
    from gcm import nn as G
    g = G.Sequential("x, edges, weights", [(G.GraphConv(2, 2), "x, edges, weights -> x")]).to(DEV)
    mem = SparseGCM(g, graph_size=4)
    with pytest.raises(Exception, match="Overflow"):
        mem(torch.zeros(2, 5, 2, device=DEV), torch.tensor([5, 1], device=DEV), None)


def test_user_gnn_gets_plain_tensors():
    """Plugin API #2 (sparse): an arbitrary GNN sees (flat_nodes [M,F], edge_index [2,E]
    (source, sink), weights [E]) and max_hops hands it the relabelled subgraph."""
    from gcm.sparse_gcm import SparseGCM
    from gcm.sparse_edge_selectors.temporal import TemporalEdge
    torch.manual_seed(0)
    ref = pyg.GraphConv(4, 4)

    class UserGNN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.inner = pyg.GraphConv(4, 4)
            self.inner.load_state_dict(ref.state_dict())
        def forward(self, x, edge_index, w):
            assert x.dim() == 2 and edge_index.shape[0] == 2 and w.shape[0] == edge_index.shape[1]
            assert bool((edge_index[0] < edge_index[1]).all())
            return self.inner(x, edge_index, w)

    x = torch.rand(3, 6, 4)
    taus = torch.tensor([6, 4, 5])
    want, _ = osp.sparse_step(x, taus, None, lambda a, b, c: ref(a, b, c), graph_size=8,
                              edge_selectors=osp.TemporalEdge([1, 2]))
    for max_hops in (None, 1):
        mem = SparseGCM(UserGNN().to(DEV), edge_selectors=TemporalEdge([1, 2]), graph_size=8,
                        max_hops=max_hops)
        got, _ = mem(x.to(DEV), taus.to(DEV), None)
        torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-6)


def test_sparse_backward_without_input_grad():
    """Regression: parameters are the only tensors that need gradients (obs without grad)."""
    from gcm.sparse_gcm import SparseGCM
    from gcm.sparse_edge_selectors.temporal import TemporalEdge
    from gcm import nn as G
    torch.manual_seed(0)
    g = G.Sequential("x, edges, weights", [(G.GraphConv(4, 4), "x, edges, weights -> x"), torch.nn.Tanh(),
                                           (G.GraphConv(4, 4), "x, edges, weights -> x")]).to(DEV)
    mem = SparseGCM(g, edge_selectors=TemporalEdge([1]), graph_size=8)
    out, _ = mem(torch.rand(3, 8, 4, device=DEV), torch.full((3,), 8, device=DEV), None)
    out.mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in g.parameters())


# --------------------------------------------------------------------------
# packed sparse hidden state (util.py:323-382)
# --------------------------------------------------------------------------
def test_pack_unpack_hidden_matches_reference():
    from gcm import util
    fx = Fixture("g11_pack")
    m = fx.meta
    adj = torch.sparse_coo_tensor(fx["coo"], fx["values"], size=(m["B"], m["N"], m["N"])).to(DEV)
    hidden = (fx["nodes"].to(DEV), adj, fx["T"].to(DEV))
    n, e, w, T = util.pack_hidden(hidden, m["B"], m["max_edges"])
    assert torch.equal(e.cpu(), fx["dense_edges"]) and torch.equal(w.cpu(), fx["dense_weights"])
    assert torch.equal(n.cpu(), fx["nodes"]) and torch.equal(T.cpu(), fx["T"])
    _, uadj, _ = util.unpack_hidden((n, e, w, T), m["B"])
    assert torch.equal(uadj.coalesce().indices().cpu(), fx["un_idx"])
    assert torch.equal(uadj.coalesce().values().cpu(), fx["un_val"])
    with pytest.raises(AssertionError, match="Cannot pack"):
        util.pack_hidden(hidden, m["B"], 11)            # graph 2 holds 11 edges: needs max_edges > 11


def test_pack_unpack_round_trips():
    """tests/test_sparse_gcm.py:82-133 - empty and small packed states survive unpack -> pack."""
    from gcm import util
    B, N, F, ME = 3, 5, 4, 10
    nodes = torch.zeros(B, N, F, device=DEV)
    for fill in (False, True):
        edge = torch.full((B, 2, ME), -1, dtype=torch.long, device=DEV)
        weight = torch.ones(B, 1, ME, device=DEV)
        if fill:
            edge[0, :, 0] = torch.tensor([0, 1]); edge[1, :, 0] = torch.tensor([0, 1])
            edge[1, :, 1] = torch.tensor([1, 2])
            weight[0, 0, 0], weight[1, 0, 0], weight[1, 0, 1] = 0.5, 0.33, 0.25
        T = torch.tensor([2, 3, 0], device=DEV)
        packed = (nodes, edge, weight, T)
        again = util.pack_hidden(util.unpack_hidden(packed, B), B, ME)
        for a, b in zip(packed, again):
            assert torch.equal(a, b)


# --------------------------------------------------------------------------
# sparse LearnedEdge (sparse_edge_selectors/learned.py, util.py:89-113,242-282)
# --------------------------------------------------------------------------
def test_causal_edges_match_reference():
    from gcm import util
    fx = Fixture("g12_causal_edges")
    T, taus = fx["T"].to(DEV), fx["taus"].to(DEV)
    assert torch.equal(util.get_causal_edges(T, taus).cpu(), fx["all"])
    assert torch.equal(util.get_causal_edges(T, taus, window=2).cpu(), fx["win2"])
    assert torch.equal(util.get_causal_edges(T, taus, window=0).cpu(), fx["win0"])


def test_sparse_gumbel_softmax_matches_torch_sparse_softmax():
    from gcm import util
    torch.manual_seed(0)
    idx = osp.get_causal_edges(torch.tensor([2, 0, 5]), torch.tensor([3, 4, 1]))
    vals = torch.randn(idx.shape[1])
    noise = torch.randn(idx.shape[1])
    logits = torch.sparse_coo_tensor(idx, vals, size=(3, 8, 8))
    want = osp.sparse_gumbel_softmax(logits, 2, tau=0.7, noise=noise)
    got = util.sparse_gumbel_softmax(logits.to(DEV), 2, tau=0.7, noise=noise.to(DEV)).coalesce()
    assert torch.equal(got.indices().cpu(), want.indices())
    torch.testing.assert_close(got.values().cpu(), want.values(), rtol=1e-5, atol=1e-7)


def _known_answer_logits():
    """The reference's own vector for hard=True (tests/test_sparse_gcm.py:795-823)."""
    idx = torch.tensor([[0, 0, 0, 0, 0, 0, 1, 1], [0, 0, 0, 0, 1, 1, 1, 1],
                        [0, 1, 2, 2, 0, 5, 4, 4], [0, 0, 1, 0, 0, 3, 0, 3]])
    values = torch.ones(8) * 1e15
    values[3] = 0
    values[-1] = 0
    want = torch.tensor([[0, 0, 0, 0, 0, 1], [0, 0, 0, 1, 1, 1], [0, 1, 2, 0, 5, 4], [0, 0, 1, 0, 3, 0]])
    return torch.sparse_coo_tensor(idx, values, size=(2, 2, 100, 100)), want


def test_sparse_gumbel_softmax_hard_known_answer():
    from gcm import util
    a, want = _known_answer_logits()
    res = util.sparse_gumbel_softmax(a.to(DEV), 3, hard=True).coalesce()
    assert torch.equal(res.indices().cpu(), want)
    assert torch.equal(res.values().cpu(), torch.ones(6))


@pytest.mark.parametrize("dim", [2, 1])
@pytest.mark.parametrize("hard", [False, True])
def test_sparse_gumbel_softmax_dims_and_hard_vs_oracle(dim, hard):
    """Rows along the last dim (contiguous in COO order) and along a middle dim (sorted by row key);
    hard: same surviving entries and values as the oracle's scatter_max loop, gradient included."""
    from gcm import util
    torch.manual_seed(3)
    dense = torch.rand(4, 9, 9) < 0.35
    idx = dense.nonzero().t().contiguous()
    vals = torch.randn(idx.shape[1])
    noise = -torch.empty(idx.shape[1]).exponential_().log()
    vc = vals.clone().requires_grad_(True)
    vd = vals.to(DEV).requires_grad_(True)
    want = osp.sparse_gumbel_softmax(torch.sparse_coo_tensor(idx, vc, size=(4, 9, 9)), dim, tau=0.8,
                                     noise=noise, hard=hard).coalesce()
    got = util.sparse_gumbel_softmax(torch.sparse_coo_tensor(idx.to(DEV), vd, size=(4, 9, 9)), dim, tau=0.8,
                                     noise=noise.to(DEV), hard=hard).coalesce()
    assert torch.equal(got.indices().cpu(), want.indices())
    torch.testing.assert_close(got.values().cpu(), want.values(), rtol=1e-5, atol=1e-7)
    w = torch.randn(want.values().numel())
    (want.values() * w).sum().backward()
    (got.values() * w.to(DEV)).sum().backward()
    torch.testing.assert_close(vd.grad.cpu(), vc.grad, rtol=1e-5, atol=1e-5 * float(vc.grad.abs().max()) + 1e-9)


@pytest.mark.parametrize("name", ["g12_sparse_learned", "g12_sparse_learned_win3"])
def test_sparse_learned_edge_matches_reference(name):
    from gcm.sparse_gcm import SparseGCM
    from gcm.sparse_edge_selectors.learned import LearnedEdge
    fx = Fixture(name)
    m = fx.meta
    ref = osp.canonical_gnn(m["F"], m["H"], act=torch.nn.Tanh)
    ref.load_state_dict(fx.group("param:"))
    g = dev_sparse_gnn(ref, m["F"], m["H"], torch.nn.Tanh)
    sel = LearnedEdge(m["F"], num_edge_samples=m["num_edge_samples"], window=m["window"], store_grads=False)
    sel.load_state_dict(fx.group("sel_param:"))
    sel = sel.to(DEV)
    call = {"i": 0}

    def noise(logits):
        gz = fx[f"noise_{call['i']}"].to(DEV)
        call["i"] += 1
        assert gz.numel() == logits.numel()
        return gz

    sel.noise_fn = noise
    mem = SparseGCM(g, edge_selectors=sel, graph_size=m["N"])
    obs = fx["obs"].to(DEV).requires_grad_(True)
    B = m["B"]
    hidden, outs, pos = None, [], torch.zeros(B, dtype=torch.long)
    for taus in fx["taus"]:
        t = int(taus.max())
        rows = [torch.cat([obs[b, pos[b]: pos[b] + taus[b]],
                           torch.zeros(t - int(taus[b]), m["F"], device=DEV)]) for b in range(B)]
        out, hidden = mem(torch.stack(rows), taus.to(DEV), hidden)
        outs.append(out)
        pos = pos + taus
    (sum(o.sum() for o in outs) / sum(o.numel() for o in outs)).backward()
    for i, o in enumerate(outs):
        torch.testing.assert_close(o.cpu(), fx[f"out{i}"], rtol=1e-5, atol=1e-5)
    assert torch.equal(hidden[1].coalesce().indices().cpu(), fx["hT_adj_indices"])   # sampled edges: bit exact
    assert torch.equal(hidden[0].cpu(), fx["hT_nodes"]) and torch.equal(hidden[2].cpu(), fx["hT_T"])
    gs = float(fx["grad_obs"].abs().max())
    torch.testing.assert_close(obs.grad.cpu(), fx["grad_obs"], rtol=1e-5, atol=1e-5 * gs)
    for k, p in g.named_parameters():
        want = fx["grad:" + k]
        torch.testing.assert_close(p.grad.cpu(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()) + 1e-7, msg=k)
    for k, p in sel.named_parameters():
        want = fx["sel_grad:" + k]
        torch.testing.assert_close(p.grad.cpu(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()) + 1e-8, msg=k)
    assert {"edges_per_node", "edge_density", "logits_mean", "logits_var", "temperature"} <= set(sel.stats)


def test_sparse_learned_edge_default_noise_runs():
    """tests/test_sparse_gcm.py:822-852 style smoke: device-RNG gumbel draws, grads reach the MLP."""
    from gcm.sparse_gcm import SparseGCM
    from gcm.sparse_edge_selectors.learned import LearnedEdge
    from gcm import nn as G
    torch.manual_seed(0)
    B, N, F = 8, 64, 16
    g = G.Sequential("x, edges, weights", [(G.GraphConv(F, F), "x, edges, weights -> x"), torch.nn.Tanh(),
                                           (G.GraphConv(F, F), "x, edges, weights -> x"), torch.nn.Tanh()]).to(DEV)
    sel = LearnedEdge(F, num_edge_samples=4, window=16).to(DEV)
    mem = SparseGCM(g, edge_selectors=sel, graph_size=N)
    hidden, outs = None, []
    for _ in range(3):
        out, hidden = mem(torch.randn(B, 10, F, device=DEV), torch.full((B,), 10, device=DEV), hidden)
        outs.append(out)
    torch.cat(outs, 1).mean().backward()
    idx = hidden[1].coalesce().indices()
    assert bool((idx[2] < idx[1]).all())                      # causal
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in sel.edge_network.parameters())
    assert any(k.startswith("gnorm_") for k in sel.stats)     # grad hooks (store_grads)


def test_sparse_learned_edge_smoke_size_matches_oracle():
    """The reference's smoke size (tests/test_sparse_gcm.py:822-852: B=8, N=256, F=32) with injected
    gumbel draws against the oracle: outputs, sampled edges (bit exact) and every gradient.  At this
    size the edge network runs on ~20 k candidate rows per call - gcm_rows_linear / gcm_skinny_wgrad /
    gcm_relu_layernorm_bwd, no library GEMM."""
    from gcm.sparse_gcm import SparseGCM
    from gcm.sparse_edge_selectors.learned import LearnedEdge
    torch.manual_seed(5)
    B, N, F, tau_len, calls, window = 8, 256, 32, 24, 3, 40
    ref = osp.canonical_gnn(F, F, act=torch.nn.Tanh)
    g = dev_sparse_gnn(ref, F, F, torch.nn.Tanh)
    sel = LearnedEdge(F, num_edge_samples=3, window=window, store_grads=False)
    net = od.build_edge_network(F)
    net.load_state_dict({k[len("edge_network."):]: v for k, v in sel.state_dict().items()
                         if k.startswith("edge_network.")})
    sel = sel.to(DEV)
    draws = []

    def dev_noise(logits):
        gz = -torch.empty(logits.numel()).exponential_().log()
        draws.append(gz)
        return gz.to(DEV)

    sel.noise_fn = dev_noise
    mem = SparseGCM(g, edge_selectors=sel, graph_size=N)
    obs = torch.randn(calls, B, tau_len, F)
    od_ = obs.to(DEV).requires_grad_(True)
    taus = torch.full((B,), tau_len, dtype=torch.long)
    hidden, outs = None, []
    for c in range(calls):
        out, hidden = mem(od_[c], taus.to(DEV), hidden)
        outs.append(out)
    torch.stack(outs).mean().backward()

    it = iter(draws)
    osel = osp.LearnedEdge(net, 3, window=window, tau=sel.state_dict()["tau_param"].cpu(),
                           noise_fn=lambda n: next(it))
    oo = obs.clone().requires_grad_(True)
    ohid, oouts = None, []
    for c in range(calls):
        out, ohid = osp.sparse_step(oo[c], taus, ohid, ref, graph_size=N, edge_selectors=osel)
        oouts.append(out)
    torch.stack(oouts).mean().backward()
    assert torch.equal(hidden[1].coalesce().indices().cpu(), ohid[1].coalesce().indices())
    for a, b in zip(outs, oouts):
        torch.testing.assert_close(a.cpu(), b, rtol=1e-5, atol=1e-5)
    gs = float(oo.grad.abs().max())
    torch.testing.assert_close(od_.grad.cpu(), oo.grad, rtol=1e-5, atol=1e-5 * gs)
    for (k, p), (_, q) in zip(g.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(p.grad.cpu(), q.grad, rtol=1e-5, atol=1e-5 * float(q.grad.abs().max()) + 1e-7, msg=k)
    for (k, p), (_, q) in zip(sel.edge_network.named_parameters(), net.named_parameters()):
        torch.testing.assert_close(p.grad.cpu(), q.grad, rtol=1e-5, atol=1e-5 * float(q.grad.abs().max()) + 1e-8, msg=k)


# --------------------------------------------------------------------------
# SparseGCM one node per call on the chain's caches (step_ext.cpp: SparseChain; gcm_sparse_step_cached)
# --------------------------------------------------------------------------
def _stepwise_case(hops, B, N, F, H, T, act, p_skip, seed):
    gen = torch.Generator().manual_seed(seed)
    obs = torch.rand(T, B, 1, F, generator=gen)
    taus = (torch.rand(T, B, generator=gen) >= p_skip).long()      # some graphs get no node in some calls
    weight = torch.rand(T, B, 1, H, generator=gen) - 0.3
    ref = osp.canonical_gnn(F, H, act=act)
    return obs, taus, weight, ref


def _oracle_stepwise(ref, obs, taus, weight, hops, N, dtype):
    import copy
    r = copy.deepcopy(ref).to(dtype)
    n0, a0, T0 = osp.initial_hidden(obs[0], N)
    hid, loss, outs = (n0.to(dtype), a0.to(dtype), T0), 0.0, []
    for t in range(obs.shape[0]):
        o, hid = osp.sparse_step(obs[t].to(dtype), taus[t], hid, lambda a, b, c: r(a, b, c), graph_size=N,
                                 edge_selectors=osp.TemporalEdge(hops))
        outs.append(o)
        loss = loss + (o * weight[t].to(dtype)).sum()
    loss.backward()
    return torch.stack(outs).detach(), hid, {k: p.grad for k, p in r.named_parameters()}


@pytest.mark.parametrize("hops,B,N,F,H,T,act,p_skip", [
    ([1], 6, 16, 32, 32, 16, torch.nn.Tanh, 0.0),
    ([1, 2, 4], 5, 32, 32, 32, 30, torch.nn.Tanh, 0.25),
    ([3, 1], 4, 12, 64, 32, 12, torch.nn.ReLU, 0.1),
    ([2], 3, 700, 32, 64, 40, None, 0.3),            # (graph_size beyond the dense kernels' 128)
])
def test_sparse_stepwise_cached_chain_vs_oracle(hops, B, N, F, H, T, act, p_skip):
    """x [B, 1, F] per call from hidden = None: every call on the caches (one launch + the state advance), the
    backward as one launch at the chain's gate.  Against the oracle called the same way: state bit exact, beliefs
    and parameter gradients inside the float64 bound (3x the reference formulation's own fp32 error); and against
    this module's general path (both GraphConv layers over every stored node)."""
    from gcm.sparse_gcm import SparseGCM
    from gcm.sparse_edge_selectors.temporal import TemporalEdge
    obs, taus, weight, ref = _stepwise_case(hops, B, N, F, H, T, act, p_skip, seed=len(hops) * 100 + B)
    out32, hid32, g32 = _oracle_stepwise(ref, obs, taus, weight, hops, N, torch.float32)
    out64, _, g64 = _oracle_stepwise(ref, obs, taus, weight, hops, N, torch.float64)
    g = dev_sparse_gnn(ref, F, H, act)
    results = {}
    for cache in (True, False):
        mem = SparseGCM(g, edge_selectors=TemporalEdge(hops), graph_size=N)
        mem.stepwise_cache = cache
        g.zero_grad(set_to_none=True)
        hid, outs = None, []
        for t in range(T):
            o, hid = mem(obs[t].to(DEV), taus[t].to(DEV), hid)
            outs.append(o)
        if cache:
            assert mem._chain.steps() == T and mem._chain.live()
        else:
            assert mem._chain is None
        (torch.stack(outs) * weight.to(DEV)).sum().backward()
        results[cache] = (torch.stack(outs).detach().cpu(), hid, {k: p.grad.cpu() for k, p in g.named_parameters()})
    got, hid, grads = results[True]
    atol_o = max(2e-6, 3.0 * float((out32.double() - out64).abs().max()))
    assert float((got.double() - out64).abs().max()) <= atol_o
    assert torch.equal(hid[0].cpu(), hid32[0])
    assert torch.equal(hid[1].coalesce().indices().cpu(), hid32[1].coalesce().indices())
    assert torch.equal(hid[1].coalesce().values().cpu(), hid32[1].coalesce().values())
    assert torch.equal(hid[2].cpu(), hid32[2])
    for k, want in g64.items():
        err_ref = float((g32[k].double() - want).abs().max())
        atol = max(3.0 * err_ref, 5e-7 * float(want.abs().max()))
        assert float((grads[k].double() - want).abs().max()) <= atol, k
    # the general path of this module: same beliefs to 1e-5, same state bit for bit
    base, hid_b, _ = results[False]
    torch.testing.assert_close(got, base, rtol=1e-5, atol=1e-5)
    assert torch.equal(hid[0], hid_b[0]) and torch.equal(hid[2], hid_b[2])
    assert torch.equal(hid[1].coalesce().indices(), hid_b[1].coalesce().indices())


def test_sparse_stepwise_cached_chain_ends_and_restarts():
    """What ends a chain on the caches: a fork (the previous state handed in again), a parameter written in place,
    a call with more than one node; the calls behind it run the general path and stay correct; hidden = None or
    get_initial_hidden_state() starts a new one.  Two backward passes over one chain (retain_graph)."""
    from gcm.sparse_gcm import SparseGCM
    from gcm.sparse_edge_selectors.temporal import TemporalEdge
    hops, B, N, F, H, T = [1, 2], 4, 16, 32, 32, 10
    obs, taus, weight, ref = _stepwise_case(hops, B, N, F, H, T, torch.nn.Tanh, 0.0, seed=7)
    out32, _, g32 = _oracle_stepwise(ref, obs, taus, weight, hops, N, torch.float32)
    g = dev_sparse_gnn(ref, F, H, torch.nn.Tanh)
    mem = SparseGCM(g, edge_selectors=TemporalEdge(hops), graph_size=N)
    one = torch.ones(B, dtype=torch.long, device=DEV)

    def run(hid, t0, t1, outs):
        for t in range(t0, t1):
            o, hid = mem(obs[t].to(DEV), one, hid)
            outs.append(o)
        return hid

    # fork: step 4 evaluated twice from the same state; the second evaluation and everything behind it is general
    outs = []
    hid = run(mem.get_initial_hidden_state(obs[0].to(DEV)), 0, 4, outs)
    assert mem._chain.steps() == 4
    o_a, hid_a = mem(obs[4].to(DEV), one, hid)
    assert mem._chain.steps() == 5
    o_b, hid_b = mem(obs[4].to(DEV), one, hid)
    assert not mem._chain.live()
    torch.testing.assert_close(o_a, o_b, rtol=1e-5, atol=1e-6)
    outs.append(o_b)
    hid = run(hid_b, 5, T, outs)
    assert not mem._chain.live()
    loss = (torch.stack(outs) * weight.to(DEV)).sum()
    loss.backward(retain_graph=True)
    first = {k: p.grad.clone() for k, p in g.named_parameters()}
    torch.testing.assert_close(torch.stack(outs).detach().cpu(), out32, rtol=1e-5, atol=1e-5)
    for k, p in g.named_parameters():
        torch.testing.assert_close(p.grad.cpu(), g32[k], rtol=1e-5, atol=1e-5 * float(g32[k].abs().max()), msg=k)
    loss.backward()                                     # a second pass over the same records: gradients add up
    for k, p in g.named_parameters():
        torch.testing.assert_close(p.grad, 2 * first[k], rtol=1e-5, atol=1e-6, msg=k)
    g.zero_grad(set_to_none=True)

    # a parameter written in place mid-chain: the cached rows of h1 are stale - the chain ends there
    outs = []
    hid = run(None, 0, 3, outs)
    assert mem._chain.live() and mem._chain.steps() == 3
    with torch.no_grad():
        next(g.parameters()).mul_(1.0)
    hid = run(hid, 3, 5, outs)
    assert not mem._chain.live()
    torch.testing.assert_close(torch.stack(outs).detach().cpu(), out32[:5], rtol=1e-5, atol=1e-5)
    # several nodes in one call: general path; a new chain from None afterwards
    x2 = torch.cat([obs[5], obs[6]], 1).to(DEV)
    o2, hid = mem(x2, 2 * one, hid)
    torch.testing.assert_close(o2.detach().cpu(), torch.cat([out32[5], out32[6]], 1), rtol=1e-5, atol=1e-5)
    run(None, 0, 2, [])
    assert mem._chain.live() and mem._chain.steps() == 2
    # no_grad calls ride on the caches too (no records)
    with torch.no_grad():
        outs = []
        run(None, 0, T, outs)
    assert mem._chain.steps() == T
    torch.testing.assert_close(torch.stack(outs).cpu(), out32, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("hops,B,N", [([1], 5, 40), ([4, 2, 1], 7, 33), ([9, 5, 3], 4, 16), ([2], 3, 2),
                                      ([16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1], 3, 50)])
def test_temporal_structure_closed_form_equals_the_general_kernels(hops, B, N):
    """Whole episodes from empty graphs: gcm_sparse_temporal_structure writes the COO entries, the flat list, the CSR
    pointers and the CSC view in one launch, in closed form - against the oracle's TemporalEdge (temporal.py:18-63) and
    the kernels it stands in for (k_temporal_fill, k_edges_flat, k_ptr_from_sorted, k_csc_batched), ragged taus
    including empty and one-node graphs.  Bit exact (indices)."""
    import ctypes
    from gcm import _hip, _ops
    lib, p, st = _hip.lib(), _hip.ptr, _hip.stream()
    torch.manual_seed(N + len(hops))
    taus = torch.randint(0, N + 1, (B,))
    taus[0], taus[1 % B] = N, 0
    if B > 2:
        taus[2] = 1
    T0 = torch.zeros(B, dtype=torch.long)
    want_coo = osp.TemporalEdge(hops)(None, T0, taus, B).coalesce().indices()
    E = want_coo.shape[1]
    off = torch.cat([torch.zeros(1, dtype=torch.long), taus.cumsum(0)])
    M = int(off[-1])
    d = lambda t: t.to(DEV)
    taus_d, off_d, T0_d = d(taus), d(off), d(T0)      # (kept alive: the calls below take raw pointers)
    edge_off = _ops.sparse_temporal_count(T0_d, taus_d, hops)
    assert int(edge_off[-1]) == E
    coo = torch.full((3, E), -1, dtype=torch.long, device=DEV)
    vals = torch.zeros(E, device=DEV)
    edge_index = torch.full((2, E), -1, dtype=torch.long, device=DEV)
    row_ptr = torch.full((M + 1,), -1, dtype=torch.long, device=DEV)
    col_ptr = torch.full((M + 1,), -1, dtype=torch.long, device=DEV)
    rows = torch.full((E,), -1, dtype=torch.long, device=DEV)
    perm = torch.full((E,), -1, dtype=torch.long, device=DEV)
    h = (ctypes.c_int32 * len(hops))(*hops)
    rc = lib.gcm_sparse_temporal_structure(p(taus_d), ctypes.addressof(h), len(hops), p(off_d), p(edge_off), p(coo),
                                           p(vals), p(edge_index), p(row_ptr), p(col_ptr), p(rows), p(perm), E, M, B, st)
    assert rc == 0
    assert torch.equal(coo.cpu(), want_coo) and bool((vals == 1).all())
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    want_d = d(want_coo)
    edges, graph = _ops.sparse_edges_to_csr(want_d, off_d, M, B, flags, n_cap=N)
    assert torch.equal(edge_index, edges) and torch.equal(row_ptr, graph.row_ptr)
    w_ptr, w_rows, w_perm = graph.csc()
    assert torch.equal(col_ptr, w_ptr) and torch.equal(rows, w_rows) and torch.equal(perm, w_perm)
    # hops the closed form does not take: the general kernels
    bad = (ctypes.c_int32 * 2)(1, 2)
    assert lib.gcm_sparse_temporal_structure(p(taus_d), ctypes.addressof(bad), 2, p(off_d), p(edge_off), p(coo),
                                             p(vals), p(edge_index), p(row_ptr), p(col_ptr), p(rows), p(perm), E, M, B,
                                             st) == _hip.GCM_EUNSUPPORTED


def test_one_shot_from_none_returns_aliases_that_are_watched():
    """hidden = None, whole episodes: the returned rows are the last layer's output itself (no extract copy) - an
    in-place write by the caller before backward() is reported like any saved tensor's; the returned node matrix is
    a fresh tensor (not the caller's x)."""
    from gcm.sparse_gcm import SparseGCM
    from gcm import nn as G
    from gcm.sparse_edge_selectors.temporal import TemporalEdge
    B, N, F, H = 6, 64, 32, 32
    torch.manual_seed(0)
    g = G.Sequential("x, edges, weights", [(G.GraphConv(F, H), "x, edges, weights -> x"), torch.nn.Tanh(),
                                           (G.GraphConv(H, H), "x, edges, weights -> x"), torch.nn.Tanh()]).to(DEV)
    mem = SparseGCM(g, edge_selectors=TemporalEdge([1, 2]), graph_size=N)
    x = torch.rand(B, N, F, device=DEV)
    taus = torch.full((B,), N, dtype=torch.long, device=DEV)
    out, hid = mem(x, taus, None)
    assert hid[0].data_ptr() != x.data_ptr() and torch.equal(hid[0], x)
    out.mul_(2.0)
    with pytest.raises(RuntimeError, match="inplace"):
        out.sum().backward()
    # and the NaN check still fires on this path (sparse_gcm.py:201-203)
    xb = x.clone()
    xb[3, 10, 5] = float("nan")
    with pytest.raises(AssertionError):
        mem(xb, taus, None)


def test_one_shot_sizes_are_reused_for_the_same_taus_tensor():
    """Whole-episode calls from hidden = None with the SAME taus tensor (object and version counter) reuse the flat sizes
    the first call read back - no host round trip in front of the call's launches; ragged taus; a taus tensor edited in
    place (its version counter moves) is read back again and gives the new sizes; one edited behind the counter's back
    (.data) is caught by the device's own figures at the call's closing flag read."""
    from gcm.sparse_gcm import SparseGCM
    from gcm import nn as G, _ext
    from gcm.sparse_edge_selectors.temporal import TemporalEdge
    ext = _ext.module()
    B, N, F, H = 5, 32, 32, 32
    torch.manual_seed(1)
    ref = osp.canonical_gnn(F, H)
    g = G.Sequential("x, edges, weights", [(G.GraphConv(F, H), "x, edges, weights -> x"), torch.nn.Tanh(),
                                           (G.GraphConv(H, H), "x, edges, weights -> x"), torch.nn.Tanh()])
    g.load_state_dict(ref.state_dict())
    g = g.to(DEV)
    mem = SparseGCM(g, edge_selectors=TemporalEdge([1, 3]), graph_size=N)
    x = torch.rand(B, N, F)
    taus = torch.tensor([32, 7, 0, 19, 32])

    def check(taus_cpu, taus_dev):
        out, hid = mem(x.to(DEV), taus_dev, None)
        out_o, hid_o = osp.sparse_step(x, taus_cpu, None, ref, graph_size=N, edge_selectors=osp.TemporalEdge([1, 3]))
        torch.testing.assert_close(out.cpu(), out_o.detach(), rtol=1e-5, atol=2e-6)
        assert torch.equal(hid[1].coalesce().indices().cpu(), hid_o[1].coalesce().indices())
        assert torch.equal(hid[2].cpu(), hid_o[2])

    td = taus.to(DEV)
    h0 = ext.sparse_sizes_memo_hits()
    check(taus, td)
    assert ext.sparse_sizes_memo_hits() == h0            # first sight of this tensor: read back
    check(taus, td)
    check(taus, td)
    assert ext.sparse_sizes_memo_hits() == h0 + 2        # ... then reused
    td[1] = 12                                           # in place: the version counter moves
    taus2 = taus.clone()
    taus2[1] = 12
    check(taus2, td)
    assert ext.sparse_sizes_memo_hits() == h0 + 2
    check(taus2, td)
    assert ext.sparse_sizes_memo_hits() == h0 + 3
    td.data[3] = 2                                       # behind the counter's back: caught at the closing flag read
    with pytest.raises(RuntimeError, match="version counter"):
        mem(x.to(DEV), td, None)
    taus3 = taus2.clone()
    taus3[3] = 2
    check(taus3, td)                                     # (the memo was dropped: read back, correct again)


def test_sparse_paths_do_not_depend_on_uninitialised_memory():
    """SparseGCM's whole-episode call and its stepwise cached chain allocate their flat buffers, index structures,
    caches and records without a zero fill where a kernel writes them whole.  With torch filling every uninitialised
    allocation with NaN (deterministic mode's fill_uninitialized_memory) they must still match the reference vectors /
    the oracle."""
    prev_det = torch.are_deterministic_algorithms_enabled()
    prev_fill = torch.utils.deterministic.fill_uninitialized_memory
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
    try:
        test_sparse_rollout_matches_reference(SPARSE[0])
        import os
        if os.environ.get("GCM_NO_TORCH_EXT") != "1":     # (the cached chain lives in the C++ host path)
            test_sparse_stepwise_cached_chain_vs_oracle([1, 2, 4], 6, 24, 32, 32, 20, torch.nn.Tanh, 0.0)
    finally:
        torch.utils.deterministic.fill_uninitialized_memory = prev_fill
        torch.use_deterministic_algorithms(prev_det)
