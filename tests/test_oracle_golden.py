"""Pin the CPU oracle against golden vectors captured from the reference
(tests/golden/make_golden.py).  CPU only."""
import pytest
import torch

from _golden import Fixture, oracle_selector, FOLD_SPECS, fold_selector
from oracle import dense as od, pyg, sparse as osp

DENSE = ["g1_temporal_h1", "g2_temporal_h124_both", "g1b_cfg1", "g3_euclid", "g3_euclid_mixed",
         "g3_euclid_learned", "g4_spatial", "g4_spatial_ab", "g4_cosine", "g5_dense_edge",
         "g13_exact_temporal", "g13_exact_dense"]


def _run_dense(fx, gnn, sel):
    m = fx.meta
    obs = fx["obs"].clone().requires_grad_(True)
    hidden, mxs, sums = fx.h0(), [], []
    for t in range(m["T"]):
        mx, hidden = od.dense_step(obs[t], hidden, gnn, graph_size=m["N"], edge_selectors=sel)
        mxs.append(mx)
        sums.append(hidden[1].detach().sum(dim=(1, 2)))
    mxs = torch.stack(mxs)
    mxs.mean().backward()
    return obs, mxs, hidden, torch.stack(sums)


@pytest.mark.parametrize("name", DENSE)
def test_dense_oracle_matches_reference(name):
    fx = Fixture(name)
    m = fx.meta
    gnn = od.canonical_gnn(m["F"], m["H"])
    gnn.load_state_dict(fx.group("param:"))
    sel = oracle_selector(m, fx.group("sel_param:"))
    obs, mxs, hidden, sums = _run_dense(fx, gnn, sel)
    assert torch.equal(hidden[0], fx["hT_nodes"])
    assert torch.equal(hidden[1], fx["hT_adj"])          # adjacency: bit exact
    assert torch.equal(hidden[3], fx["hT_num_nodes"])
    assert torch.equal(sums, fx["adj_sums"])
    torch.testing.assert_close(mxs, fx["mx"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(obs.grad, fx["grad_obs"], rtol=1e-5, atol=1e-7)
    for k, p in gnn.named_parameters():
        torch.testing.assert_close(p.grad, fx["grad:" + k], rtol=1e-5, atol=1e-7)


def test_learned_edge_oracle_matches_reference():
    fx = Fixture("g6_learned")
    m = fx.meta
    gnn = od.canonical_gnn(m["F"], m["H"])
    gnn.load_state_dict(fx.group("param:"))
    net = od.build_edge_network(m["F"])
    net.load_state_dict({k[len("edge_network."):]: v for k, v in fx.group("sel_param:").items()})
    step = {"t": 0}

    def noise(shape):
        return fx[f"noise_{step['t']}"][:, : shape[1]]

    sel = od.LearnedEdge(net, m["num_edge_samples"], noise_fn=noise)
    obs = fx["obs"].clone().requires_grad_(True)
    hidden, mxs = None, []
    for t in range(m["T"]):
        step["t"] = t
        mx, hidden = od.dense_step(obs[t], hidden, gnn, graph_size=m["N"], edge_selectors=sel)
        mxs.append(mx)
    mxs = torch.stack(mxs)
    mxs.mean().backward()
    assert torch.equal(hidden[1].detach(), fx["hT_adj"])
    torch.testing.assert_close(mxs, fx["mx"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(obs.grad, fx["grad_obs"], rtol=1e-5, atol=1e-7)
    for k, p in net.named_parameters():
        torch.testing.assert_close(p.grad, fx["sel_grad:edge_network." + k], rtol=1e-4, atol=1e-7)


def test_temporal_learned_oracle_matches_reference():
    """TemporalBackedge(learned=True) (temporal.py:51-70) with the reference's recorded gumbel draws."""
    fx = Fixture("g14_temporal_learned")
    m = fx.meta
    gnn = od.canonical_gnn(m["F"], m["H"])
    gnn.load_state_dict(fx.group("param:"))
    calls = {"t": 0, "k": 0}

    def noise(n):                       # graphs ascending (empty ones skipped), samples ascending
        order = [(b, i) for b in range(m["B"]) if m["starts"][b] + calls["t"] > 0 for i in range(m["num_samples"])]
        b, i = order[calls["k"]]
        calls["k"] += 1
        return fx["noise"][calls["t"], b, i, :n]

    sel = od.TemporalBackedge(learned=True, learning_window=m["learning_window"],
                              num_samples=m["num_samples"], noise_fn=noise)
    sel.window = fx["sel_param:window"].clone().requires_grad_(True)
    obs = fx["obs"].clone().requires_grad_(True)
    hidden, mxs = fx.h0(), []
    for t in range(m["T"]):
        calls["t"], calls["k"] = t, 0
        mx, hidden = od.dense_step(obs[t], hidden, gnn, graph_size=m["N"], edge_selectors=sel)
        mxs.append(mx)
    mxs = torch.stack(mxs)
    mxs.mean().backward()
    assert torch.equal(hidden[1].detach(), fx["hT_adj"])
    torch.testing.assert_close(mxs, fx["mx"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(obs.grad, fx["grad_obs"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(sel.window.grad, fx["sel_grad:window"], rtol=1e-4, atol=1e-7)
    for k, p in gnn.named_parameters():
        torch.testing.assert_close(p.grad, fx["grad:" + k], rtol=1e-5, atol=1e-7)


def test_sparsemax_restatement_properties():
    """Spardmax's sparsemax (third-party package, absent: parity unpinned) - simplex projection
    properties and the closed form for two logits."""
    torch.manual_seed(0)
    for n in (1, 2, 5, 17):
        z = 2 * torch.randn(n)
        p = od.sparsemax(z)
        assert abs(float(p.sum()) - 1) < 1e-6 and bool((p >= 0).all())
        # support = the largest entries; inside it p - z is constant (= -tau)
        sup = p > 0
        assert float(z[sup].min()) >= float(z[~sup].max()) if (~sup).any() else True
        d = (p - z)[sup]
        assert float(d.max() - d.min()) < 1e-6
    torch.testing.assert_close(od.sparsemax(torch.tensor([0.3, 0.1])), torch.tensor([0.6, 0.4]))
    torch.testing.assert_close(od.sparsemax(torch.tensor([3.0, 0.1])), torch.tensor([1.0, 0.0]))


@pytest.mark.parametrize("name", ["g7_wrap_weights", "g7_wrap_noweights"])
def test_wrap_overflow_oracle(name):
    fx = Fixture(name)
    g = pyg.Sequential("x, adj, weights, B, N",
                       [(pyg.DenseGraphConv(5, 5), "x, adj -> x"), torch.nn.ReLU()])
    g.load_state_dict(fx.group("param:"))
    mx, (n2, a2, w2, nn2) = od.dense_step(fx["obs"], fx.h0(), g, graph_size=7)
    assert torch.equal(n2, fx["hT_nodes"]) and torch.equal(a2, fx["hT_adj"])
    assert torch.equal(w2, fx["hT_weights"]) and torch.equal(nn2, fx["hT_num_nodes"])
    torch.testing.assert_close(mx, fx["mx"], rtol=1e-6, atol=1e-6)
    # gcm.py:354 quirk - the reference decremented the CALLER's num_nodes in place
    assert fx["caller_num_nodes_after"].tolist() == [1, 6]


SPARSE = ["g8_sparse_oneshot", "g8_sparse_oneshot_2hop", "g8_sparse_stepwise",
          "g8_sparse_ragged", "g8_sparse_ragged_2hop", "g16_sparse_aux", "g16_sparse_aux_overlap"]


@pytest.mark.parametrize("name", SPARSE)
def test_sparse_oracle_matches_reference(name):
    fx = Fixture(name)
    m = fx.meta
    gnn = osp.canonical_gnn(m["F"], m["H"], act=torch.nn.Tanh if m["act"] else None)
    gnn.load_state_dict(fx.group("param:"))
    sel = osp.TemporalEdge(m["hops"])
    aux = osp.TemporalEdge(m["aux_hops"]) if m.get("aux_hops") else None
    obs = fx["obs"].clone().requires_grad_(True)
    B = m["B"]
    hidden, outs, pos = None, [], torch.zeros(B, dtype=torch.long)
    for taus in fx["taus"]:
        t = int(taus.max())
        x = torch.zeros(B, t, m["F"])
        for b in range(B):
            x[b, : taus[b]] = obs[b, pos[b]: pos[b] + taus[b]]
        out, hidden = osp.sparse_step(x, taus, hidden, gnn, graph_size=m["N"], edge_selectors=sel,
                                      max_hops=m["max_hops"], aux_edge_selectors=aux)
        outs.append(out)
        pos = pos + taus
    loss = sum(o.sum() for o in outs) / sum(o.numel() for o in outs)
    loss.backward()
    for i, o in enumerate(outs):
        torch.testing.assert_close(o, fx[f"out{i}"], rtol=1e-6, atol=1e-6)
    assert torch.equal(hidden[0], fx["hT_nodes"])
    assert torch.equal(hidden[1].coalesce().indices(), fx["hT_adj_indices"])   # bit exact
    assert torch.equal(hidden[2], fx["hT_T"])
    torch.testing.assert_close(obs.grad, fx["grad_obs"], rtol=1e-5, atol=1e-7)
    for k, p in gnn.named_parameters():
        torch.testing.assert_close(p.grad, fx["grad:" + k], rtol=1e-5, atol=1e-6)


# --------------------------------------------------------------------------
# SURVEY 8(f) "next" rows: PositionalEncoding in the step, pack/unpack_hidden
# --------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["add", "cat"])
def test_posenc_oracle_matches_reference(mode):
    fx = Fixture(f"g10_posenc_{mode}")
    m = fx.meta
    gnn = od.canonical_gnn(m["F"], m["H"])
    gnn.load_state_dict(fx.group("param:"))
    reproject = None
    if mode == "cat":
        reproject = torch.nn.Linear(m["F"], m["F"] - m["cat_dim"])
        reproject.load_state_dict({k[len("reproject."):]: v for k, v in fx.group("sel_param:").items()
                                   if k.startswith("reproject.")})
    pe = od.PositionalEncoding(max_len=m["N"], mode=mode, cat_dim=m["cat_dim"], reproject=reproject)
    obs = fx["obs"].clone().requires_grad_(True)
    hidden, mxs = None, []
    for t in range(m["T"]):
        mx, hidden = od.dense_step(obs[t], hidden, gnn, graph_size=m["N"],
                                   edge_selectors=od.TemporalBackedge([1]),
                                   aux_edge_selectors=od.TemporalBackedge([2]), positional_encoder=pe)
        mxs.append(mx)
    mxs = torch.stack(mxs)
    mxs.mean().backward()
    assert torch.equal(hidden[0], fx["hT_nodes"]) and torch.equal(hidden[1], fx["hT_adj"])
    torch.testing.assert_close(mxs, fx["mx"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(obs.grad, fx["grad_obs"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("name", sorted(FOLD_SPECS))
def test_fold_fixtures_oracle_matches_reference(name):
    """G15: Linear preprocessor / aux selectors / PositionalEncoding between selectors and GNN."""
    fx = Fixture(name)
    m = fx.meta
    gnn = od.canonical_gnn(m["Fg"], m["H"])
    gnn.load_state_dict(fx.group("param:"))
    sp = fx.group("sel_param:")
    pre = None
    if m["pre_bias"] is not None:
        pre = torch.nn.Linear(m["F"], m["Fg"], bias=m["pre_bias"])
        pre.load_state_dict({k[len("pre."):]: v for k, v in sp.items() if k.startswith("pre.")})
    pe = None
    if m["mode"]:
        reproject = None
        if m["mode"] == "cat":
            reproject = torch.nn.Linear(m["Fg"], m["Fg"] - m["cat_dim"])
            reproject.load_state_dict({k[len("pe.reproject."):]: v for k, v in sp.items()
                                       if k.startswith("pe.reproject.")})
        pe = od.PositionalEncoding(max_len=m["N"], mode=m["mode"], cat_dim=m["cat_dim"], reproject=reproject)
    sel, aux = (fold_selector(s, od.TemporalBackedge, od.DenseEdge) for s in FOLD_SPECS[name])
    obs = fx["obs"].clone().requires_grad_(True)
    hidden, mxs = None, []
    for t in range(m["T"]):
        mx, hidden = od.dense_step(obs[t], hidden, gnn, graph_size=m["N"], edge_selectors=sel,
                                   preprocessor=pre, aux_edge_selectors=aux, positional_encoder=pe)
        mxs.append(mx)
    mxs = torch.stack(mxs)
    mxs.mean().backward()
    assert torch.equal(hidden[0], fx["hT_nodes"]) and torch.equal(hidden[1], fx["hT_adj"])
    torch.testing.assert_close(mxs, fx["mx"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(obs.grad, fx["grad_obs"], rtol=1e-5, atol=1e-7)
    for k, p in gnn.named_parameters():
        torch.testing.assert_close(p.grad, fx["grad:" + k], rtol=1e-5, atol=1e-7)
    if pre is not None:
        for k, p in pre.named_parameters():
            torch.testing.assert_close(p.grad, fx["sel_grad:pre." + k], rtol=1e-5, atol=1e-7)


def test_posenc_table_oracle():
    fx = Fixture("g10_posenc_table")
    pe = od.PositionalEncoding(max_len=7, mode="add")
    assert torch.equal(pe(torch.zeros(2, 7, 5), torch.tensor([0, 7])), fx["enc0"])
    assert torch.equal(pe(torch.zeros(2, 7, 5), torch.tensor([1, 8])), fx["enc1"])


def test_pack_hidden_oracle():
    fx = Fixture("g11_pack")
    m = fx.meta
    adj = torch.sparse_coo_tensor(fx["coo"], fx["values"], size=(m["B"], m["N"], m["N"]))
    n, e, w, T = osp.pack_hidden((fx["nodes"], adj, fx["T"]), m["B"], m["max_edges"])
    assert torch.equal(e, fx["dense_edges"]) and torch.equal(w, fx["dense_weights"])
    _, uadj, _ = osp.unpack_hidden((n, e, w, T), m["B"])
    assert torch.equal(uadj.coalesce().indices(), fx["un_idx"])
    assert torch.equal(uadj.coalesce().values(), fx["un_val"])


# --------------------------------------------------------------------------
# sparse LearnedEdge
# --------------------------------------------------------------------------
def test_causal_edges_oracle():
    fx = Fixture("g12_causal_edges")
    assert torch.equal(osp.get_causal_edges(fx["T"], fx["taus"]), fx["all"])
    assert torch.equal(osp.get_causal_edges(fx["T"], fx["taus"], window=2), fx["win2"])
    assert torch.equal(osp.get_causal_edges(fx["T"], fx["taus"], window=0), fx["win0"])


@pytest.mark.parametrize("name", ["g12_sparse_learned", "g12_sparse_learned_win3"])
def test_sparse_learned_oracle_matches_reference(name):
    fx = Fixture(name)
    m = fx.meta
    gnn = osp.canonical_gnn(m["F"], m["H"], act=torch.nn.Tanh)
    gnn.load_state_dict(fx.group("param:"))
    net = od.build_edge_network(m["F"])
    sp = fx.group("sel_param:")
    net.load_state_dict({k[len("edge_network."):]: v for k, v in sp.items() if k.startswith("edge_network.")})
    call = {"i": 0}

    def noise(n):
        g = fx[f"noise_{call['i']}"]
        call["i"] += 1
        assert g.numel() == n
        return g

    sel = osp.LearnedEdge(net, m["num_edge_samples"], window=m["window"], tau=sp["tau_param"], noise_fn=noise)
    obs = fx["obs"].clone().requires_grad_(True)
    B = m["B"]
    hidden, outs, pos = None, [], torch.zeros(B, dtype=torch.long)
    for taus in fx["taus"]:
        t = int(taus.max())
        x = torch.zeros(B, t, m["F"])
        for b in range(B):
            x[b, : taus[b]] = obs[b, pos[b]: pos[b] + taus[b]]
        out, hidden = osp.sparse_step(x, taus, hidden, gnn, graph_size=m["N"], edge_selectors=sel)
        outs.append(out)
        pos = pos + taus
    (sum(o.sum() for o in outs) / sum(o.numel() for o in outs)).backward()
    for i, o in enumerate(outs):
        torch.testing.assert_close(o, fx[f"out{i}"], rtol=1e-6, atol=1e-6)
    assert torch.equal(hidden[1].coalesce().indices(), fx["hT_adj_indices"])
    torch.testing.assert_close(obs.grad, fx["grad_obs"], rtol=1e-4, atol=1e-7)
    for k, p in net.named_parameters():
        torch.testing.assert_close(p.grad, fx["sel_grad:edge_network." + k], rtol=1e-4, atol=1e-7)


def test_sparse_gumbel_softmax_hard_oracle_known_answer():
    """hard=True against the reference's own known-answer vector (tests/test_sparse_gcm.py:795-823)."""
    idx = torch.tensor([[0, 0, 0, 0, 0, 0, 1, 1], [0, 0, 0, 0, 1, 1, 1, 1],
                        [0, 1, 2, 2, 0, 5, 4, 4], [0, 0, 1, 0, 0, 3, 0, 3]])
    values = torch.ones(8) * 1e15
    values[3] = 0
    values[-1] = 0
    res = osp.sparse_gumbel_softmax(torch.sparse_coo_tensor(idx, values, size=(2, 2, 100, 100)), 3,
                                    hard=True).coalesce()
    want = torch.tensor([[0, 0, 0, 0, 0, 1], [0, 0, 0, 1, 1, 1], [0, 1, 2, 0, 5, 4], [0, 0, 1, 0, 3, 0]])
    assert torch.equal(res.indices(), want) and torch.equal(res.values(), torch.ones(6))
