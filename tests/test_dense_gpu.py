"""Parity of the HIP DenseGCM path (through the C ABI) against the CPU oracle and
the golden vectors captured from the reference.  Needs an MI355X."""
import pytest
import torch

from _golden import Fixture, oracle_selector
from oracle import dense as od, pyg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL, ATOL = 1e-5, 1e-6   # north_star: 1e-5 rtol fp32 on outputs; adjacency bit exact


def dev_gnn_from(ref_gnn, spec):
    """Product GNN with the oracle GNN's weights. spec: [(Fi, Fo, act_cls|None), ...]"""
    from gcm import nn as G

    mods = []
    for fi, fo, act in spec:
        mods.append((G.DenseGraphConv(fi, fo), "x, adj -> x"))
        if act is not None:
            mods.append(act())
    g = G.Sequential("x, adj, weights, B, N", mods)
    g.load_state_dict(ref_gnn.state_dict())
    return g.to(DEV)


def product_selector(meta, sel_params=None):
    from gcm.edge_selectors.temporal import TemporalBackedge
    from gcm.edge_selectors.dense import DenseEdge
    from gcm.edge_selectors.distance import EuclideanEdge, CosineEdge, SpatialEdge

    kind = meta["selector"]
    if kind == "temporal":
        return TemporalBackedge(meta["hops"], direction=meta["direction"])
    if kind == "dense":
        return DenseEdge()
    learned = bool(meta.get("learned"))
    if kind == "euclid":
        s = EuclideanEdge(meta["max_distance"], learned=learned)
    elif kind == "cosine":
        s = CosineEdge(meta["max_distance"], learned=learned)
    else:
        s = SpatialEdge(meta["max_distance"], slice(*meta["a"]), slice(*meta["b"]), learned=learned)
    if learned:
        s.load_state_dict(sel_params)
    return s.to(DEV)


# --------------------------------------------------------------------------
# DenseGraphConv kernel: forward + every gradient, odd shapes included
# --------------------------------------------------------------------------
@pytest.mark.parametrize("B,N,Fi,Fo,act", [
    (2, 7, 5, 5, "relu"), (3, 8, 3, 3, None), (5, 10, 11, 11, "relu"), (4, 32, 8, 32, "tanh"),
    (2, 33, 20, 40, "tanh"), (3, 100, 64, 32, "tanh"), (2, 128, 32, 32, "tanh"),
    (1, 130, 33, 65, None), (2, 64, 128, 128, "tanh"), (1, 257, 96, 16, "relu"),
])
def test_graphconv_kernel(B, N, Fi, Fo, act):
    from gcm import nn as G
    torch.manual_seed(B * 1000 + N)
    ref = pyg.DenseGraphConv(Fi, Fo)
    dev = G.DenseGraphConv(Fi, Fo)
    dev.load_state_dict(ref.state_dict())
    dev = dev.to(DEV)
    x = torch.randn(B, N, Fi)
    adj = (torch.rand(B, N, N) < 0.2).float() * torch.rand(B, N, N).round(decimals=1)
    acts = {"tanh": torch.tanh, "relu": torch.relu, None: lambda t: t}
    code = {"tanh": 1, "relu": 2, None: 0}[act]
    xc, ac = x.clone().requires_grad_(True), adj.clone().requires_grad_(True)
    xd, ad = x.to(DEV).requires_grad_(True), adj.to(DEV).requires_grad_(True)
    yc = acts[act](ref(xc, ac))
    yd = dev(xd, ad, _act=code)
    torch.testing.assert_close(yd.cpu(), yc, rtol=RTOL, atol=1e-5)
    g = torch.randn_like(yc)
    yc.backward(g)
    yd.backward(g.to(DEV))
    torch.testing.assert_close(xd.grad.cpu(), xc.grad, rtol=1e-5, atol=1e-5 * float(xc.grad.abs().max()) + 1e-9)
    torch.testing.assert_close(ad.grad.cpu(), ac.grad, rtol=1e-5, atol=1e-5 * float(ac.grad.abs().max()) + 1e-9)
    for (k, pc), (_, pd) in zip(ref.named_parameters(), dev.named_parameters()):
        scale = float(pc.grad.abs().max()) + 1e-6
        torch.testing.assert_close(pd.grad.cpu(), pc.grad, rtol=1e-5, atol=1e-5 * scale + 1e-5, msg=k)


def test_graphconv_identity_known_answer():
    """tests/test_gcm.py:282-323 - identity lin_root, identity lin_rel, no edges => out == obs."""
    from gcm.gcm import DenseGCM
    from gcm import nn as G
    feats, B, N = 11, 5, 10
    g = G.Sequential("x, adj, weights, B, N", [
        (G.DenseGraphConv(feats, feats), "x, adj -> x"), torch.nn.ReLU(),
        (G.DenseGraphConv(feats, feats), "x, adj -> x"), torch.nn.ReLU()])
    for m in g.modules():
        if isinstance(m, G.DenseGraphConv):
            m.lin_root.weight = torch.nn.Parameter(torch.eye(feats))
            m.lin_rel.weight = torch.nn.Parameter(torch.eye(feats))
            m.lin_rel.bias = torch.nn.Parameter(torch.zeros(feats))
    s = DenseGCM(g.to(DEV))
    hidden = (torch.zeros(B, N, feats, device=DEV), torch.zeros(B, N, N, device=DEV),
              torch.ones(B, N, N, device=DEV), torch.zeros(B, dtype=torch.long, device=DEV))
    for k in (1, 2, 3):
        obs = k * torch.ones(B, feats, device=DEV)
        out, hidden = s(obs, hidden)
        assert torch.equal(out, obs)
    want = torch.zeros(B, N, feats)
    want[:, 0], want[:, 1], want[:, 2] = 1, 2, 3
    assert torch.equal(hidden[0].cpu(), want)


# --------------------------------------------------------------------------
# state advance / wrap overflow (tests/test_gcm.py:105-184)
# --------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["g7_wrap_weights", "g7_wrap_noweights"])
def test_wrap_overflow_golden(name):
    from gcm.gcm import DenseGCM
    fx = Fixture(name)
    ref = pyg.Sequential("x, adj, weights, B, N",
                         [(pyg.DenseGraphConv(5, 5), "x, adj -> x"), torch.nn.ReLU()])
    ref.load_state_dict(fx.group("param:"))
    g = dev_gnn_from(ref, [(5, 5, torch.nn.ReLU)])
    for mutate in (False, True):
        s = DenseGCM(g, mutate_num_nodes_on_overflow=mutate, finite_check="sync")
        h0 = tuple(t.to(DEV) for t in fx.h0())
        mx, (n2, a2, w2, nn2) = s(fx["obs"].to(DEV), h0)
        assert torch.equal(n2.cpu(), fx["hT_nodes"]) and torch.equal(a2.cpu(), fx["hT_adj"])
        assert torch.equal(w2.cpu(), fx["hT_weights"]) and torch.equal(nn2.cpu(), fx["hT_num_nodes"])
        torch.testing.assert_close(mx.cpu(), fx["mx"], rtol=RTOL, atol=ATOL)
        # inputs are never modified ... except the reference's num_nodes quirk when asked for
        assert torch.equal(h0[0].cpu(), fx["h0_nodes"]) and torch.equal(h0[1].cpu(), fx["h0_adj"])
        want_nn = fx["caller_num_nodes_after"] if mutate else fx["h0_num_nodes"]
        assert torch.equal(h0[3].cpu(), want_nn)


def test_state_advance_random_vs_oracle():
    from gcm import _ops
    torch.manual_seed(3)
    B, N, F = 9, 13, 6
    nodes, adj, w = torch.randn(B, N, F), torch.rand(B, N, N), torch.rand(B, N, N)
    nn_ = torch.tensor([0, 1, 5, 12, 13, 13, 7, 13, 2])
    x = torch.randn(B, F)
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    for weights in (w, torch.zeros(0)):
        n2, a2, w2, cur, nxt = _ops.state_advance(nodes.to(DEV), adj.to(DEV), weights.to(DEV),
                                                  nn_.to(DEV), x.to(DEV), flags)
        rn, ra, rw, rnn = od.wrap_overflow(nodes, adj, weights, nn_)
        rn = rn.index_put((torch.arange(B), rnn), x)
        assert torch.equal(n2.cpu(), rn) and torch.equal(a2.cpu(), ra) and torch.equal(w2.cpu(), rw)
        assert torch.equal(cur.cpu(), rnn) and torch.equal(nxt.cpu(), rnn + 1)
    assert int(flags.item()) & 1


# --------------------------------------------------------------------------
# full rollouts against the reference's golden vectors
# --------------------------------------------------------------------------
DENSE = ["g1_temporal_h1", "g2_temporal_h124_both", "g1b_cfg1", "g3_euclid", "g3_euclid_mixed",
         "g3_euclid_learned", "g4_spatial", "g4_spatial_ab", "g4_cosine", "g5_dense_edge",
         "g13_exact_temporal", "g13_exact_dense"]


@pytest.mark.parametrize("name", DENSE)
def test_dense_rollout_matches_reference(name):
    from gcm.gcm import DenseGCM
    fx = Fixture(name)
    m = fx.meta
    ref = od.canonical_gnn(m["F"], m["H"])
    ref.load_state_dict(fx.group("param:"))
    g = dev_gnn_from(ref, [(m["F"], m["H"], torch.nn.Tanh), (m["H"], m["H"], torch.nn.Tanh)])
    mem = DenseGCM(g, edge_selectors=product_selector(m, fx.group("sel_param:")), graph_size=m["N"])
    obs = fx["obs"].to(DEV).requires_grad_(True)
    h0 = fx.h0()
    hidden = None if h0 is None else tuple(t.to(DEV) for t in h0)
    mxs, sums = [], []
    for t in range(m["T"]):
        mx, hidden = mem(obs[t], hidden)
        mxs.append(mx)
        sums.append(hidden[1].sum(dim=(1, 2)))
    mxs = torch.stack(mxs)
    mxs.mean().backward()
    mem.check_flags()
    assert torch.equal(hidden[1].cpu(), fx["hT_adj"])            # adjacency: bit exact
    assert torch.equal(torch.stack(sums).cpu(), fx["adj_sums"])
    assert torch.equal(hidden[0].cpu(), fx["hT_nodes"])
    assert torch.equal(hidden[3].cpu(), fx["hT_num_nodes"])
    # DenseEdge rows add up to N terms per aggregate in a different order than the reference's GEMM
    atol = 5e-6 if m["selector"] == "dense" else ATOL
    torch.testing.assert_close(mxs.cpu(), fx["mx"], rtol=RTOL, atol=atol)
    _check_grads_fp64(fx, ref, obs, g)


def _check_grads_fp64(fx, ref, obs, g):
    """Observation and parameter gradients against the float64 evaluation of the same rollout:
    |ours - g64| <= max(3 x |reference fp32 - g64|, 5e-7 x scale)  (tests/_golden.py::fp64_grad_bound)."""
    from _golden import fp64_grad_bound
    m = fx.meta
    bounds = fp64_grad_bound(ref, fx, oracle_selector(m, fx.group("sel_param:")))
    got = dict(g.named_parameters())
    for k, (g64, atol) in bounds.items():
        mine = obs.grad if k == "obs" else got[k].grad
        err = float((mine.cpu().double() - g64).abs().max())
        assert err <= atol, (k, err, atol)


@pytest.mark.parametrize("name", ["g13_exact_temporal", "g13_exact_dense", "g2_temporal_h124_both", "g5_dense_edge"])
def test_rollout_entry_matches_reference(name):
    """The time-batched entry against the reference's own vectors: persistent forward kernel and
    time-parallel BPTT on the tile-exact fixtures (g13_*), launch-sequence rollout on the ragged ones."""
    from gcm.gcm import DenseGCM
    fx = Fixture(name)
    m = fx.meta
    ref = od.canonical_gnn(m["F"], m["H"])
    ref.load_state_dict(fx.group("param:"))
    g = dev_gnn_from(ref, [(m["F"], m["H"], torch.nn.Tanh), (m["H"], m["H"], torch.nn.Tanh)])
    mem = DenseGCM(g, edge_selectors=product_selector(m, fx.group("sel_param:")), graph_size=m["N"])
    obs = fx["obs"].to(DEV).requires_grad_(True)
    h0 = fx.h0()
    hidden = None if h0 is None else tuple(t.to(DEV) for t in h0)
    mxs, hidden = mem.rollout(obs, hidden)
    mxs.mean().backward()
    mem.check_flags()
    assert torch.equal(hidden[1].cpu(), fx["hT_adj"])            # adjacency: bit exact
    assert torch.equal(hidden[0].cpu(), fx["hT_nodes"])
    assert torch.equal(hidden[3].cpu(), fx["hT_num_nodes"])
    # DenseEdge rows add up to N terms per aggregate in a different order than the reference's GEMM
    atol = 5e-6 if m["selector"] == "dense" else ATOL
    torch.testing.assert_close(mxs.cpu(), fx["mx"], rtol=RTOL, atol=atol)
    _check_grads_fp64(fx, ref, obs, g)
    # inference mode (no history kept): same beliefs and final state
    with torch.no_grad():
        h0 = fx.h0()
        mx2, hid2 = mem.rollout(fx["obs"].to(DEV), None if h0 is None else tuple(t.to(DEV) for t in h0))
    torch.testing.assert_close(mx2.cpu(), fx["mx"], rtol=RTOL, atol=atol)
    assert torch.equal(hid2[1].cpu(), fx["hT_adj"]) and torch.equal(hid2[0].cpu(), fx["hT_nodes"])


def test_distance_matrix_matches_oracle():
    """The thresholded quantity itself (all three dist_fn variants) vs the oracle."""
    from gcm.edge_selectors.distance import EuclideanEdge, CosineEdge, SpatialEdge
    torch.manual_seed(0)
    B, N, F = 6, 40, 24
    nodes = torch.randn(B, N, F)
    nn_ = torch.tensor([0, 1, 13, 39, 20, 7])
    pairs = [(EuclideanEdge(1.0), od.EuclideanEdge(1.0)), (CosineEdge(0.1), od.CosineEdge(0.1)),
             (SpatialEdge(1.0, slice(2, 9), slice(11, 18)), od.SpatialEdge(1.0, slice(2, 9), slice(11, 18)))]
    for dev_sel, ref_sel in pairs:
        d = dev_sel.distances(nodes.to(DEV), nn_.to(DEV)).cpu()
        want = ref_sel.distances(nodes, nn_)
        # entry (b, cur_b) contains a zero self-distance: torch.cdist's matmul formulation
        # (|a|^2+|b|^2-2ab, used above 25 rows) cancels there with ~1e-3 absolute error, the
        # kernel's direct sum does not.  That entry is never thresholded (distance.py:31-33).
        self_entry = torch.zeros(B, N, dtype=torch.bool)
        self_entry[torch.arange(B), nn_] = True
        torch.testing.assert_close(d[~self_entry], want[~self_entry], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(d[self_entry], want[self_entry], rtol=1e-3, atol=2e-3)


@pytest.mark.parametrize("B,N,F,learned", [(40, 50, 24, False), (64, 128, 64, True), (33, 20, 100, False), (48, 40, 80, True),
                                            (300, 16, 128, False)])
def test_euclid_matrix_core_path(B, N, F, learned):
    """B >= 32 runs the MFMA formulation (|n|^2+|c|^2-2nc, like torch.cdist above 25 rows)."""
    from gcm.edge_selectors.distance import EuclideanEdge
    torch.manual_seed(B + N)
    nodes = torch.randn(B, N, F)
    nn_ = torch.randint(0, N, (B,))
    thr = float(torch.cdist(nodes[:, 0], nodes[:1]).mean()) * (0.5 if learned else 1.0)
    dev_sel = EuclideanEdge(thr, learned=learned).to(DEV)
    ref_sel = od.EuclideanEdge(thr, dist_param=torch.tensor([thr]) if learned else None)
    d = dev_sel.distances(nodes.to(DEV), nn_.to(DEV)).cpu()
    want = ref_sel.distances(nodes, nn_)
    torch.testing.assert_close(d, want, rtol=1e-5, atol=1e-5 * float(want.abs().max()) + 1e-9)
    # thresholded result, away from the threshold
    adj = torch.zeros(B, N, N, device=DEV)
    got, _ = dev_sel(nodes.to(DEV), adj, torch.zeros(0, device=DEV), nn_.to(DEV), B)
    ref, _ = ref_sel(nodes, torch.zeros(B, N, N), torch.zeros(0), nn_, B)
    live = torch.arange(N)[None, :] < nn_[:, None]
    safe = ((want - ref_sel.max_distance).abs() > 1e-3) & live
    rows = got.cpu()[torch.arange(B), nn_]
    ref_rows = ref[torch.arange(B), nn_]
    assert torch.equal(rows[safe], ref_rows[safe])
    assert float(got.cpu().sum()) >= float(rows[safe].sum())


# --------------------------------------------------------------------------
# known answers ported from the reference's unit tests
# --------------------------------------------------------------------------
def _identity_gcm(sel, feats=11, N=10):
    from gcm.gcm import DenseGCM
    from gcm import nn as G
    g = G.Sequential("x, adj, weights, B, N", [(G.DenseGraphConv(feats, feats), "x, adj -> x"),
                                                torch.nn.ReLU()])
    return DenseGCM(g.to(DEV), edge_selectors=sel, graph_size=N)


def test_temporal_known_answers():
    """tests/test_gcm.py:581-617."""
    from gcm.edge_selectors.temporal import TemporalBackedge
    B, feats, N = 5, 11, 10
    s = _identity_gcm(TemporalBackedge([1]))
    hidden = None
    for _ in range(2):
        _, hidden = s(torch.ones(B, feats, device=DEV), hidden)
    want = torch.zeros(B, N, N)
    want[:, 1, 0] = 1
    assert torch.equal(hidden[1].cpu(), want)
    s = _identity_gcm(TemporalBackedge([4]))
    hidden = None
    for _ in range(10):
        _, hidden = s(torch.ones(B, feats, device=DEV), hidden)
    want = torch.zeros(B, N, N)
    for i in range(4, 10):
        want[:, i, i - 4] = 1
    assert torch.equal(hidden[1].cpu(), want)


def test_distance_known_answers():
    """tests/test_gcm.py:708-729 - zero distance => edge [b,1,0]; distance sqrt(11) > 1 => none."""
    from gcm.edge_selectors.distance import EuclideanEdge
    B, feats, N = 5, 11, 10
    s = _identity_gcm(EuclideanEdge(max_distance=1))
    hidden = None
    for _ in range(2):
        _, hidden = s(torch.ones(B, feats, device=DEV), hidden)
    want = torch.zeros(B, N, N)
    want[:, 1, 0] = 1
    assert torch.equal(hidden[1].cpu(), want)
    s = _identity_gcm(EuclideanEdge(max_distance=1))
    _, hidden = s(torch.zeros(B, feats, device=DEV), None)
    _, hidden = s(torch.ones(B, feats, device=DEV), hidden)
    assert torch.equal(hidden[1].cpu(), torch.zeros(B, N, N))


def test_dense_edge_known_answer():
    """tests/test_gcm.py:784-801."""
    from gcm.edge_selectors.dense import DenseEdge
    B, feats, N = 5, 11, 10
    s = _identity_gcm(DenseEdge())
    hidden = None
    for _ in range(2):
        _, hidden = s(torch.ones(B, feats, device=DEV), hidden)
    want = torch.zeros(B, N, N)
    want[:, 0, 0] = want[:, 1, 1] = want[:, 0, 1] = want[:, 1, 0] = 1
    assert torch.equal(hidden[1].cpu(), want)


def test_chained_selectors():
    """tests/test_gcm.py:646-682 - two selectors chained through Sequential."""
    from gcm.gcm import DenseGCM
    from gcm import nn as G
    from gcm.edge_selectors.temporal import TemporalBackedge
    B, feats, N = 3, 4, 6
    sel = G.Sequential("x, adj, weights, num_nodes, B", [
        (TemporalBackedge([1]), "x, adj, weights, num_nodes, B -> adj, weights"),
        (TemporalBackedge([2]), "x, adj, weights, num_nodes, B -> adj, weights")])
    g = G.Sequential("x, adj, weights, B, N", [(G.DenseGraphConv(feats, feats), "x, adj -> x")])
    s = DenseGCM(g.to(DEV), edge_selectors=sel, graph_size=N)
    hidden = None
    for _ in range(4):
        _, hidden = s(torch.ones(B, feats, device=DEV), hidden)
    want = torch.zeros(B, N, N)
    for i in range(1, 4):
        want[:, i, i - 1] = 1
    for i in range(2, 4):
        want[:, i, i - 2] = 1
    assert torch.equal(hidden[1].cpu(), want)


def test_nonfinite_is_reported():
    """gcm.py:316-318 - same AssertionError text, raised at the step in sync mode."""
    from gcm.gcm import DenseGCM
    from gcm import nn as G
    g = G.Sequential("x, adj, weights, B, N", [(G.DenseGraphConv(4, 4), "x, adj -> x")])
    s = DenseGCM(g.to(DEV), graph_size=4, finite_check="sync")
    with pytest.raises(AssertionError, match="Got NaN in returned memory"):
        s(torch.full((2, 4), float("nan"), device=DEV), None)
    s = DenseGCM(g.to(DEV), graph_size=4, finite_check="deferred")
    s(torch.full((2, 4), float("inf"), device=DEV), None)
    with pytest.raises(AssertionError, match="Got NaN in returned memory"):
        s.check_flags()


# --------------------------------------------------------------------------
# LearnedEdge against the reference run with recorded gumbel noise
# --------------------------------------------------------------------------
def test_learned_edge_matches_reference():
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.learned import LearnedEdge
    fx = Fixture("g6_learned")
    m = fx.meta
    ref = od.canonical_gnn(m["F"], m["H"])
    ref.load_state_dict(fx.group("param:"))
    g = dev_gnn_from(ref, [(m["F"], m["H"], torch.nn.Tanh), (m["H"], m["H"], torch.nn.Tanh)])
    sel = LearnedEdge(m["F"], num_edge_samples=m["num_edge_samples"])
    sel.load_state_dict(fx.group("sel_param:"))
    sel = sel.to(DEV)
    step = {"t": 0}

    def noise(logits):   # the reference drew nothing at t = 0 (learned.py:120-121)
        key = f"noise_{step['t']}"
        return fx[key].to(DEV) if key in fx else torch.zeros_like(logits)

    sel.noise_fn = noise
    mem = DenseGCM(g, edge_selectors=sel, graph_size=m["N"])
    obs = fx["obs"].to(DEV).requires_grad_(True)
    hidden, mxs = None, []
    for t in range(m["T"]):
        step["t"] = t
        mx, hidden = mem(obs[t], hidden)
        mxs.append(mx)
    mxs = torch.stack(mxs)
    mxs.mean().backward()
    mem.check_flags()
    assert hidden[1].requires_grad                                   # adj carries grad (test_gcm.py:851-862)
    assert torch.equal(hidden[1].detach().cpu(), fx["hT_adj"])       # sampled edges: bit exact
    torch.testing.assert_close(mxs.cpu(), fx["mx"], rtol=RTOL, atol=ATOL)
    gs = float(fx["grad_obs"].abs().max())
    torch.testing.assert_close(obs.grad.cpu(), fx["grad_obs"], rtol=1e-5, atol=1e-5 * gs)
    for k, p in g.named_parameters():
        want = fx["grad:" + k]
        torch.testing.assert_close(p.grad.cpu(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()), msg=k)
    for k, p in sel.named_parameters():
        want = fx["sel_grad:" + k]
        torch.testing.assert_close(p.grad.cpu(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()) + 1e-8, msg=k)


def test_temporal_backedge_learned_matches_reference():
    """TemporalBackedge(learned=True) (temporal.py:51-70) against the reference run with
    recorded gumbel draws: sampled edges bit exact, window-logit gradient included."""
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.temporal import TemporalBackedge
    fx = Fixture("g14_temporal_learned")
    m = fx.meta
    ref = od.canonical_gnn(m["F"], m["H"])
    ref.load_state_dict(fx.group("param:"))
    g = dev_gnn_from(ref, [(m["F"], m["H"], torch.nn.Tanh), (m["H"], m["H"], torch.nn.Tanh)])
    sel = TemporalBackedge(learned=True, learning_window=m["learning_window"], num_samples=m["num_samples"])
    sel.load_state_dict(fx.group("sel_param:"))
    sel = sel.to(DEV)
    step = {"t": 0}
    sel.noise_fn = lambda shape, dev: fx["noise"][step["t"]].permute(1, 0, 2).contiguous().to(dev)
    mem = DenseGCM(g, edge_selectors=sel, graph_size=m["N"])
    obs = fx["obs"].to(DEV).requires_grad_(True)
    hidden, mxs = tuple(t.to(DEV) for t in fx.h0()), []
    for t in range(m["T"]):
        step["t"] = t
        mx, hidden = mem(obs[t], hidden)
        mxs.append(mx)
    mxs = torch.stack(mxs)
    mxs.mean().backward()
    mem.check_flags()
    assert torch.equal(hidden[1].detach().cpu(), fx["hT_adj"])
    assert torch.equal(hidden[0].cpu(), fx["hT_nodes"]) and torch.equal(hidden[3].cpu(), fx["hT_num_nodes"])
    torch.testing.assert_close(mxs.cpu(), fx["mx"], rtol=RTOL, atol=ATOL)
    gs = float(fx["grad_obs"].abs().max())
    torch.testing.assert_close(obs.grad.cpu(), fx["grad_obs"], rtol=1e-5, atol=1e-5 * gs)
    want = fx["sel_grad:window"]
    torch.testing.assert_close(sel.window.grad.cpu(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))
    for k, p in g.named_parameters():
        want = fx["grad:" + k]
        torch.testing.assert_close(p.grad.cpu(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()), msg=k)


def test_temporal_backedge_learned_variants():
    """Device-RNG draws, the deterministic (hard sparsemax) variant against the oracle restatement,
    and the reference's failure beyond the learning window."""
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.temporal import TemporalBackedge
    from gcm import nn as G
    torch.manual_seed(0)
    B, N, F, W = 4, 16, 8, 6

    def gnn():
        torch.manual_seed(1)
        return G.Sequential("x, adj, weights, B, N", [(G.DenseGraphConv(F, F), "x, adj -> x"), torch.nn.Tanh()]).to(DEV)

    sel = TemporalBackedge(learned=True, learning_window=W, num_samples=2).to(DEV)
    mem = DenseGCM(gnn(), edge_selectors=sel, graph_size=N)
    hidden, outs = None, []
    for t in range(W + 1):
        mx, hidden = mem(torch.randn(B, F, device=DEV), hidden)
        outs.append(mx)
        adj = hidden[1].detach()
        assert bool(((adj == 0) | (adj == 1)).all())
        assert int(adj.sum()) <= 2 * B * t and (t == 0 or int(adj[:, t].sum()) >= B)
    torch.stack(outs).sum().backward()
    assert sel.window.grad is not None and bool(torch.isfinite(sel.window.grad).all())
    with pytest.raises(RuntimeError):                     # n_b = W + 1 nodes: window too short (the kernel
        mem(torch.randn(B, F, device=DEV), hidden)        # raises a flag, surfaced with the module's others)
        mem.check_flags()
    with pytest.raises(RuntimeError):                     # ... and on the spot for a selector used on its own
        sel._gcm_flags = None
        sel(hidden[0], hidden[1].detach().clone(), hidden[2], hidden[3], B)

    # deterministic: hard sparsemax over window[:n_b] vs the oracle's per-graph loop
    torch.manual_seed(2)
    win = torch.randn(W)
    obs = torch.randn(W, B, F)
    sel = TemporalBackedge(learned=True, learning_window=W, deterministic=True)
    with torch.no_grad():
        sel.window.copy_(win)
    mem = DenseGCM(gnn(), edge_selectors=sel.to(DEV), graph_size=N)
    osel = od.TemporalBackedge(learned=True, learning_window=W, deterministic=True)
    osel.window = win.clone().requires_grad_(True)
    torch.manual_seed(1)
    ognn = pyg.Sequential("x, adj, weights, B, N", [(pyg.DenseGraphConv(F, F), "x, adj -> x"), torch.nn.Tanh()])
    hidden, ohid, got, want = None, None, [], []
    for t in range(W):
        mx, hidden = mem(obs[t].to(DEV), hidden)
        omx, ohid = od.dense_step(obs[t], ohid, ognn, graph_size=N, edge_selectors=osel)
        got.append(mx); want.append(omx)
    assert torch.equal(hidden[1].detach().cpu(), ohid[1].detach())
    torch.testing.assert_close(torch.stack(got).cpu(), torch.stack(want), rtol=RTOL, atol=ATOL)
    torch.stack(got).sum().backward()
    torch.stack(want).sum().backward()
    scale = float(osel.window.grad.abs().max())
    torch.testing.assert_close(sel.window.grad.cpu(), osel.window.grad, rtol=1e-5, atol=1e-5 * scale + 1e-8)


def test_learned_edge_default_noise_runs():
    """Device-RNG gumbel draws: adjacency stays binary, new edges only in row cur, grads flow."""
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.learned import LearnedEdge
    from gcm import nn as G
    torch.manual_seed(0)
    B, N, F = 8, 16, 8
    g = G.Sequential("x, adj, weights, B, N", [(G.DenseGraphConv(F, F), "x, adj -> x"), torch.nn.Tanh()])
    sel = LearnedEdge(F, num_edge_samples=3)
    mem = DenseGCM(g.to(DEV), edge_selectors=sel.to(DEV), graph_size=N)
    hidden, outs = None, []
    for t in range(6):
        mx, hidden = mem(torch.randn(B, F, device=DEV), hidden)
        outs.append(mx)
    torch.stack(outs).sum().backward()
    adj = hidden[1].detach()
    assert set(adj.unique().tolist()) <= {0.0, 1.0}
    assert float(adj.triu().sum()) == 0.0            # only past -> current edges
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in sel.parameters())


# --------------------------------------------------------------------------
# fused step kernels and the time-batched rollout entry
# --------------------------------------------------------------------------
def _mk(F, H1, H2, act1, act2, sel, N, fused, seed=0):
    from gcm.gcm import DenseGCM
    from gcm import nn as G
    torch.manual_seed(seed)
    mods = [(G.DenseGraphConv(F, H1), "x, adj -> x")]
    if act1:
        mods.append(act1())
    mods.append((G.DenseGraphConv(H1, H2), "x, adj -> x"))
    if act2:
        mods.append(act2())
    g = G.Sequential("x, adj, weights, B, N", mods).to(DEV)
    return DenseGCM(g, edge_selectors=sel, graph_size=N, fused=fused), g


def _run(mem, g, obs, h0, rollout=False):
    obs = obs.clone().requires_grad_(True)
    g.zero_grad(set_to_none=True)
    if rollout:
        out, hid = mem.rollout(obs, h0)
    else:
        hid, outs = h0, []
        for t in range(obs.shape[0]):
            mx, hid = mem(obs[t], hid)
            outs.append(mx)
        out = torch.stack(outs)
    (out * torch.linspace(0.5, 1.5, out.numel(), device=DEV).view_as(out)).sum().backward()
    mem.check_flags()
    return out.detach(), hid, obs.grad, {k: p.grad.clone() for k, p in g.named_parameters()}


@pytest.mark.parametrize("B,N,F,H1,H2,T,sel_kind", [
    (3, 7, 5, 6, 4, 10, "temporal"), (4, 32, 8, 32, 32, 40, "temporal_both"),
    (2, 33, 20, 40, 8, 36, "dense"), (5, 128, 32, 32, 32, 6, "temporal"),
    (3, 64, 64, 32, 16, 70, "temporal"), (4, 16, 8, 16, 16, 20, "spatial"),
    (2, 100, 33, 64, 70, 5, "none"),
    # tile-exact shapes: rollout() runs as one persistent launch (state resident in LDS)
    (4, 32, 32, 32, 32, 40, "temporal_both"), (3, 64, 64, 64, 64, 70, "dense"),
    (2, 96, 32, 64, 32, 100, "temporal"), (3, 128, 64, 32, 64, 9, "none"),
])
def test_fused_matches_layered_and_rollout(B, N, F, H1, H2, T, sel_kind):
    from gcm.edge_selectors.temporal import TemporalBackedge
    from gcm.edge_selectors.dense import DenseEdge
    from gcm.edge_selectors.distance import SpatialEdge

    def sel():
        return {"temporal": lambda: TemporalBackedge([1, 2, 4]),
                "temporal_both": lambda: TemporalBackedge([1, 3], direction="both"),
                "dense": lambda: DenseEdge(), "spatial": lambda: SpatialEdge(0.8, slice(0, 3)),
                "none": lambda: None}[sel_kind]()

    torch.manual_seed(B * N + T)
    obs = torch.rand(T, B, F, device=DEV)
    starts = torch.randint(0, N + 1, (B,), device=DEV)
    nodes0 = torch.rand(B, N, F, device=DEV) * (torch.arange(N, device=DEV)[None, :, None] < starts[:, None, None])
    adj0 = ((torch.rand(B, N, N, device=DEV) < 0.1) & (torch.arange(N, device=DEV)[None, :, None] < starts[:, None, None])
            & (torch.arange(N, device=DEV)[None, None, :] < starts[:, None, None])).float()
    nodes0.requires_grad_(True)
    h0 = (nodes0, adj0, torch.zeros(0, device=DEV), starts)
    lay, g_lay = _mk(F, H1, H2, torch.nn.Tanh, torch.nn.ReLU, sel(), N, fused=False)
    fus, g_fus = _mk(F, H1, H2, torch.nn.Tanh, torch.nn.ReLU, sel(), N, fused=True)
    assert lay._structure() is None and fus._structure() is not None
    want = _run(lay, g_lay, obs, h0)
    want_g0 = nodes0.grad.clone(); nodes0.grad = None
    for rollout in (False, True):
        got = _run(fus, g_fus, obs, h0, rollout=rollout)
        got_g0 = nodes0.grad.clone(); nodes0.grad = None
        # two fp32 implementations with different summation orders (dense rows sum ~N terms)
        # (atol relative to the largest belief: ReLU outputs of dense 64-node rows reach ~10)
        torch.testing.assert_close(got[0], want[0], rtol=1e-5,
                                   atol=1e-5 * max(1.0, float(want[0].abs().max())))
        for a, b in zip(got[1], want[1]):
            assert torch.equal(a, b)                       # state: bit exact
        # dense rows sum ~N terms in a different order; the test GNN ends in a ReLU, whose mask can
        # flip for a pre-activation within that noise of zero
        g_atol = 2e-5 if sel_kind == "dense" else 5e-6
        torch.testing.assert_close(got[2], want[2], rtol=1e-5, atol=g_atol)
        torch.testing.assert_close(got_g0, want_g0, rtol=1e-5, atol=g_atol)
        for k in want[3]:
            scale = float(want[3][k].abs().max()) + 1e-12
            torch.testing.assert_close(got[3][k], want[3][k], rtol=1e-5, atol=1e-5 * scale, msg=k)


@pytest.mark.parametrize("N,F,H1,H2,T,sel_kind", [(32, 32, 32, 32, 50, "temporal"), (64, 64, 64, 64, 70, "dense"),
                                                  (128, 32, 32, 32, 140, "temporal"), (96, 32, 64, 32, 30, "none")])
def test_rollout_params_only_backward(N, F, H1, H2, T, sel_kind):
    """Observations without a gradient: the rollout's backward is the one live-row parameter kernel
    over the kept history (no dX chain).  Same parameter gradients as the full backward."""
    from gcm.edge_selectors.temporal import TemporalBackedge
    from gcm.edge_selectors.dense import DenseEdge
    sel = {"temporal": lambda: TemporalBackedge([1, 2, 4]), "dense": lambda: DenseEdge(), "none": lambda: None}[sel_kind]
    torch.manual_seed(N + T)
    B = 5
    obs = torch.rand(T, B, F, device=DEV)
    starts = torch.randint(0, N // 2, (B,), device=DEV)
    nodes0 = torch.rand(B, N, F, device=DEV) * (torch.arange(N, device=DEV)[None, :, None] < starts[:, None, None])
    adj0 = ((torch.rand(B, N, N, device=DEV) < 0.1) & (torch.arange(N, device=DEV)[None, :, None] < starts[:, None, None])
            & (torch.arange(N, device=DEV)[None, None, :] < starts[:, None, None])).float()
    h0 = (nodes0, adj0, torch.zeros(0, device=DEV), starts)
    mem, g = _mk(F, H1, H2, torch.nn.Tanh, torch.nn.Tanh, sel(), N, fused=True)
    want = _run(mem, g, obs, h0, rollout=True)
    g.zero_grad(set_to_none=True)
    out, hid = mem.rollout(obs, h0)
    (out * torch.linspace(0.5, 1.5, out.numel(), device=DEV).view_as(out)).sum().backward()
    mem.check_flags()
    torch.testing.assert_close(out.detach(), want[0], rtol=0, atol=0)
    for k, p in g.named_parameters():
        scale = float(want[3][k].abs().max()) + 1e-12
        torch.testing.assert_close(p.grad, want[3][k], rtol=1e-5, atol=1e-5 * scale, msg=k)


def test_rollout_bptt_schedules_agree(monkeypatch):
    """time-parallel BPTT (one batched adjoint launch + reverse scan) == step-by-step BPTT."""
    from gcm.edge_selectors.temporal import TemporalBackedge
    from gcm import _ops
    B, N, F, H, T = 3, 32, 32, 32, 50          # overflows from step 32 on
    obs = torch.rand(T, B, F, device=DEV)
    res = []
    for cap in (1 << 40, 0):
        monkeypatch.setattr(_ops, "ROLLOUT_BWD_BATCHED_MAX_BYTES", cap)
        mem, g = _mk(F, H, H, torch.nn.Tanh, torch.nn.Tanh, TemporalBackedge([1, 2, 4]), N, fused=True)
        res.append(_run(mem, g, obs, None, rollout=True))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=0, atol=0)
    torch.testing.assert_close(res[0][2], res[1][2], rtol=1e-5, atol=1e-7)
    for k in res[0][3]:
        scale = float(res[1][3][k].abs().max()) + 1e-12
        torch.testing.assert_close(res[0][3][k], res[1][3][k], rtol=1e-5, atol=1e-6 * scale, msg=k)


@pytest.mark.parametrize("N,F,H,T,sel_kind", [(32, 32, 32, 45, "temporal"), (64, 32, 64, 80, "dense"),
                                              (128, 32, 32, 140, "temporal")])
def test_rollout_inference_keeps_no_history(N, F, H, T, sel_kind):
    """no_grad rollout = persistent kernel without the per-step history; same beliefs and final
    hidden state as the step-by-step loop (state bit exact), overflow included."""
    from gcm.edge_selectors.temporal import TemporalBackedge
    from gcm.edge_selectors.dense import DenseEdge
    sel = TemporalBackedge([1, 2, 4]) if sel_kind == "temporal" else DenseEdge()
    mem, g = _mk(F, H, H, torch.nn.Tanh, torch.nn.Tanh, sel, N, fused=True)
    B = 3
    obs = torch.rand(T, B, F, device=DEV)
    with torch.no_grad():
        out, hid = mem.rollout(obs)
        h, outs = None, []
        for t in range(T):
            mx, h = mem(obs[t], h)
            outs.append(mx)
    mem.check_flags()
    # two fp32 summation orders; dense rows add up to N terms before the tanh - and past N steps the per-step loop's
    # column-write cached steps keep each row's aggregate by rank-1 corrections (+ the new node, - the dropped one:
    # csrc/rows_colcache.hip) where rollout() sums afresh: both sit inside the float64 bound of the oracle tests
    # (tests/test_rows_gpu.py::test_rows_colcache_steps_vs_oracle), 6e-5 apart from each other here
    torch.testing.assert_close(out, torch.stack(outs), rtol=1e-5, atol=1e-6 if sel_kind == "temporal" else 1.5e-4)
    for a, b in zip(hid, h):
        assert torch.equal(a, b)


def test_rollout_falls_back_for_non_native_trees():
    """rollout() on a tree the fused kernels do not cover = the plain loop."""
    from gcm.edge_selectors.learned import LearnedEdge
    from gcm import nn as G
    from gcm.gcm import DenseGCM
    torch.manual_seed(0)
    g = G.Sequential("x, adj, weights, B, N", [(G.DenseGraphConv(4, 4), "x, adj -> x"), torch.nn.Tanh()]).to(DEV)
    mem = DenseGCM(g, edge_selectors=LearnedEdge(4).to(DEV), graph_size=6)
    obs = torch.rand(5, 2, 4, device=DEV)
    out, hid = mem.rollout(obs)
    assert out.shape == (5, 2, 4) and hid[3].tolist() == [5, 5]


# --------------------------------------------------------------------------
# PositionalEncoding (gcm.py:92-143)
# --------------------------------------------------------------------------
def test_posenc_known_answers():
    """tests/test_gcm.py:39-86."""
    import math
    from gcm.gcm import PositionalEncoding
    fx = Fixture("g10_posenc_table")
    pe = PositionalEncoding(max_len=7, mode="add")
    nodes = torch.zeros(2, 7, 5, device=DEV)
    enc = pe(nodes.clone(), torch.tensor([0, 7], device=DEV))
    assert torch.equal(enc.cpu(), fx["enc0"])                    # same table, same rows touched
    assert torch.all(enc[0, 1, :] == 0) and not torch.all(enc[0, 0, :] == 0)
    enc = pe(nodes.clone(), torch.tensor([1, 8], device=DEV))
    assert torch.equal(enc.cpu(), fx["enc1"])
    want = torch.tensor([math.sin(1.0), math.cos(1.0), math.sin((1 / 10000) ** (2 / 6)),
                         math.cos((1 / 10000) ** (2 / 6)), math.sin((1 / 10000) ** (4 / 6))])
    assert float((enc[0, 1].cpu() - want).abs().sum()) < 0.01


@pytest.mark.parametrize("mode", ["add", "cat"])
def test_posenc_in_step_matches_reference(mode):
    from gcm.gcm import DenseGCM, PositionalEncoding
    from gcm.edge_selectors.temporal import TemporalBackedge
    fx = Fixture(f"g10_posenc_{mode}")
    m = fx.meta
    ref = od.canonical_gnn(m["F"], m["H"])
    ref.load_state_dict(fx.group("param:"))
    g = dev_gnn_from(ref, [(m["F"], m["H"], torch.nn.Tanh), (m["H"], m["H"], torch.nn.Tanh)])
    pe = PositionalEncoding(max_len=m["N"], mode=mode, cat_dim=m["cat_dim"])
    if mode == "cat":
        pe.run_once(torch.zeros(1, 1, m["F"], device=DEV))
        pe.load_state_dict(fx.group("sel_param:"))
    mem = DenseGCM(g, edge_selectors=TemporalBackedge([1]), aux_edge_selectors=TemporalBackedge([2]),
                   positional_encoder=pe, graph_size=m["N"])
    obs = fx["obs"].to(DEV).requires_grad_(True)
    hidden, mxs = None, []
    for t in range(m["T"]):
        mx, hidden = mem(obs[t], hidden)
        mxs.append(mx)
    mxs = torch.stack(mxs)
    mxs.mean().backward()
    mem.check_flags()
    assert torch.equal(hidden[0].cpu(), fx["hT_nodes"])          # returned nodes stay un-encoded
    assert torch.equal(hidden[1].cpu(), fx["hT_adj"])
    torch.testing.assert_close(mxs.cpu(), fx["mx"], rtol=RTOL, atol=ATOL)
    gs = float(fx["grad_obs"].abs().max())
    torch.testing.assert_close(obs.grad.cpu(), fx["grad_obs"], rtol=1e-5, atol=1e-5 * gs)
    for k, p in g.named_parameters():
        want = fx["grad:" + k]
        torch.testing.assert_close(p.grad.cpu(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()), msg=k)


@pytest.mark.parametrize("B,N,F,cat", [(5, 12, 10, 4), (3, 130, 64, 8), (2, 7, 3, 1)])
def test_posenc_cat_kernel_matches_oracle(B, N, F, cat):
    """PositionalEncoding(mode="cat") (gcm.py:133-140) as gcm_rows_linear + gcm_posenc_cat_finish against the
    oracle: output, and the gradients w.r.t. the nodes and the re-projection layer."""
    from gcm.gcm import PositionalEncoding
    torch.manual_seed(B * 100 + N)
    x = torch.randn(B, N, F)
    num_nodes = torch.randint(0, N, (B,))
    num_nodes[0] = 0
    num_nodes[-1] = N - 1
    weight = torch.randn(B, N, F)
    pe = PositionalEncoding(max_len=N + 3, mode="cat", cat_dim=cat)
    xd = x.to(DEV).requires_grad_(True)
    pe.run_once(xd)
    rp = torch.nn.Linear(F, F - cat)
    rp.load_state_dict({k: v.cpu() for k, v in pe.reproject.state_dict().items()})
    ope = od.PositionalEncoding(max_len=N + 3, mode="cat", cat_dim=cat, reproject=rp)
    xo = x.clone().requires_grad_(True)
    want = ope(xo, num_nodes)
    got = pe(xd, num_nodes.to(DEV))
    torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-6)
    live = torch.arange(N)[None, :] <= num_nodes[:, None]
    assert torch.equal(got.detach().cpu()[~live], x[~live])                      # rows beyond stay bit exact
    assert torch.equal(got.detach().cpu()[live][:, :cat], want.detach()[live][:, :cat])   # table columns too
    (got * weight.to(DEV)).sum().backward()
    (want * weight).sum().backward()
    torch.testing.assert_close(xd.grad.cpu(), xo.grad, rtol=1e-5, atol=1e-6)
    for (k, p), (_, q) in zip(pe.reproject.named_parameters(), rp.named_parameters()):
        scale = float(q.grad.abs().max())
        torch.testing.assert_close(p.grad.cpu(), q.grad, rtol=1e-5, atol=2e-6 * scale, msg=k)


@pytest.mark.parametrize("M,I,O", [(1, 5, 7), (300, 64, 64), (4097, 32, 1), (129, 1, 33), (2048, 48, 17)])
def test_rows_linear_matches_torch(M, I, O):
    """gcm_rows_linear (the edge network's Linear layers, learned.py:38-51): y = x W^T + b, the input
    gradient x W, and the ReLU + LayerNorm epilogue, against torch in float64."""
    from gcm import _ops
    torch.manual_seed(M + I + O)
    x, w, b = torch.randn(M, I), torch.randn(O, I) / I ** 0.5, torch.randn(O)
    g, gamma, beta = torch.randn(M, O), torch.rand(O) + 0.5, torch.randn(O)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    y = _ops.rows_linear(xd, wd, bd)
    want = torch.nn.functional.linear(x.double(), w.double(), b.double())
    torch.testing.assert_close(y.cpu().double(), want, rtol=1e-5, atol=1e-5)
    gx = _ops.rows_linear(g.to(DEV), wd, transpose=True)
    torch.testing.assert_close(gx.cpu().double(), g.double() @ w.double(), rtol=1e-5, atol=1e-5)
    p, h = _ops.rows_linear(xd, wd, bd, ln=(gamma.to(DEV), beta.to(DEV), 1e-5))
    assert torch.equal(p, y)
    if O > 1:
        want_h = torch.nn.functional.layer_norm(torch.relu(want), (O,), gamma.double(), beta.double(), 1e-5)
        torch.testing.assert_close(h.cpu().double(), want_h, rtol=1e-5, atol=1e-5 * float(want_h.abs().max()) + 1e-9)
    # into a wider matrix (the PositionalEncoding re-projection)
    wide = torch.full((M, O + 3), 7.0, device=DEV)
    _ops.rows_linear(xd, wd, bd, out=wide[:, 3:], ldy=O + 3)
    assert torch.equal(wide[:, 3:], y) and bool((wide[:, :3] == 7.0).all())
