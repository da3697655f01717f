"""Edge cases of the drop-in surface against the oracle: the weights plane, pooled GNNs,
a preprocessor, tiny / large / odd shapes, empty and ragged sparse calls.  Needs an MI355X."""
import pytest
import torch

from oracle import dense as od, pyg, sparse as osp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dense_pair(F, H, layers, act=torch.nn.Tanh):
    """(oracle gnn, product gnn) with the same weights."""
    from gcm import nn as G
    ref = od.canonical_gnn(F, H, act=act, layers=layers)
    mods, cin = [], F
    for _ in range(layers):
        mods += [(G.DenseGraphConv(cin, H), "x, adj -> x"), act()]
        cin = H
    g = G.Sequential("x, adj, weights, B, N", mods)
    g.load_state_dict(ref.state_dict())
    return ref, g.to(DEV)


def _compare_rollouts(ref_kw, dev_mem, ref_gnn, obs, h0=None, rtol=1e-5, atol=1e-5):
    obs_c = obs.clone().requires_grad_(True)
    obs_d = obs.to(DEV).requires_grad_(True)
    hid_c = h0
    hid_d = None if h0 is None else tuple(t.to(DEV) for t in h0)
    outs_c, outs_d = [], []
    for t in range(obs.shape[0]):
        mc, hid_c = od.dense_step(obs_c[t], hid_c, ref_gnn, **ref_kw)
        md, hid_d = dev_mem(obs_d[t], hid_d)
        outs_c.append(mc)
        outs_d.append(md)
    oc, odv = torch.stack(outs_c), torch.stack(outs_d)
    oc.mean().backward()
    odv.mean().backward()
    dev_mem.check_flags()
    torch.testing.assert_close(odv.cpu(), oc, rtol=rtol, atol=atol)
    for a, b in zip(hid_d, hid_c):
        assert torch.equal(a.cpu(), b)
    torch.testing.assert_close(obs_d.grad.cpu(), obs_c.grad, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("B,N,F,H,T", [(1, 1, 1, 1, 3), (1, 2, 3, 2, 5), (2, 129, 8, 8, 4), (3, 200, 70, 40, 3),
                                        (2, 16, 100, 128, 4)])
def test_shapes_outside_the_fused_kernels(B, N, F, H, T):
    """N = 1, N > 128, F/H > 64: tiled layered kernels."""
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.temporal import TemporalBackedge
    torch.manual_seed(N + F)
    ref, g = _dense_pair(F, H, 2)
    mem = DenseGCM(g, edge_selectors=TemporalBackedge([1, 2]), graph_size=N)
    _compare_rollouts(dict(graph_size=N, edge_selectors=od.TemporalBackedge([1, 2])), mem, ref,
                      torch.rand(T, B, F))


@pytest.mark.parametrize("layers", [1, 3])
def test_one_and_three_layer_gnns(layers):
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.dense import DenseEdge
    torch.manual_seed(layers)
    ref, g = _dense_pair(6, 9, layers)
    mem = DenseGCM(g, edge_selectors=DenseEdge(), graph_size=12)
    assert mem._structure() is None
    _compare_rollouts(dict(graph_size=12, edge_selectors=od.DenseEdge()), mem, ref, torch.rand(15, 3, 6))


def test_edge_weights_plane_rolls_with_the_state():
    """edge_weights=True: the weights plane is copied / rolled like adj (gcm.py:343-352)."""
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.temporal import TemporalBackedge
    torch.manual_seed(0)
    B, N, F = 3, 6, 4
    ref, g = _dense_pair(F, F, 2)
    mem = DenseGCM(g, edge_selectors=TemporalBackedge([1]), graph_size=N, edge_weights=True)
    nodes, adj = torch.rand(B, N, F), (torch.rand(B, N, N) < 0.3).float()
    w = torch.rand(B, N, N)
    h0 = (nodes, adj, w, torch.tensor([6, 2, 5]))
    _compare_rollouts(dict(graph_size=N, edge_selectors=od.TemporalBackedge([1]), edge_weights=True), mem,
                      ref, torch.rand(4, B, F), h0=h0)


def test_pooled_gnn_and_preprocessor():
    """pooled=True returns the GNN output as is; preprocessor runs on the dirty nodes only."""
    from gcm.gcm import DenseGCM
    from gcm import nn as G
    from gcm.edge_selectors.temporal import TemporalBackedge
    torch.manual_seed(0)
    B, N, F, H = 4, 7, 5, 6
    pre_c = torch.nn.Linear(F, F)
    pre_d = torch.nn.Linear(F, F)
    pre_d.load_state_dict(pre_c.state_dict())
    conv_c = pyg.DenseGraphConv(F, H)

    class PoolC(torch.nn.Module):
        def forward(self, x, adj, w, B, N):
            return torch.tanh(conv_c(x, adj)).mean(dim=1)

    conv_d = G.DenseGraphConv(F, H)
    conv_d.load_state_dict(conv_c.state_dict())
    conv_d = conv_d.to(DEV)

    class PoolD(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = conv_d
        def forward(self, x, adj, w, B, N):
            return torch.tanh(self.conv(x, adj)).mean(dim=1)

    mem = DenseGCM(PoolD(), preprocessor=pre_d.to(DEV), edge_selectors=TemporalBackedge([1]), graph_size=N,
                   pooled=True, finite_check="sync")
    _compare_rollouts(dict(graph_size=N, edge_selectors=od.TemporalBackedge([1]), preprocessor=pre_c,
                           pooled=True), mem, PoolC(), torch.rand(9, B, F))


def test_sparse_empty_and_ragged_calls():
    """taus with zeros, a call that adds nothing to some graphs, graphs of different lengths."""
    from gcm.sparse_gcm import SparseGCM
    from gcm.sparse_edge_selectors.temporal import TemporalEdge
    from gcm import nn as G
    torch.manual_seed(0)
    B, N, F, H = 4, 10, 3, 5
    ref = osp.canonical_gnn(F, H, act=torch.nn.Tanh)
    g = G.Sequential("x, edges, weights", [(G.GraphConv(F, H), "x, edges, weights -> x"), torch.nn.Tanh(),
                                           (G.GraphConv(H, H), "x, edges, weights -> x"), torch.nn.Tanh()])
    g.load_state_dict(ref.state_dict())
    mem = SparseGCM(g.to(DEV), edge_selectors=TemporalEdge([2, 1]), graph_size=N)
    plans = [torch.tensor([2, 0, 3, 1]), torch.tensor([0, 0, 1, 0]), torch.tensor([3, 4, 0, 2])]
    hc, hd = None, None
    for taus in plans:
        t = max(1, int(taus.max()))
        x = torch.rand(B, t, F)
        for b in range(B):
            x[b, taus[b]:] = 0
        oc, hc = osp.sparse_step(x, taus, hc, ref, graph_size=N, edge_selectors=osp.TemporalEdge([2, 1]))
        od_, hd = mem(x.to(DEV), taus.to(DEV), hd)
        torch.testing.assert_close(od_.cpu(), oc, rtol=1e-5, atol=1e-6)
        assert torch.equal(hd[0].cpu(), hc[0]) and torch.equal(hd[2].cpu(), hc[2])
        assert torch.equal(hd[1].coalesce().indices().cpu(), hc[1].coalesce().indices())


def test_inference_without_grad_and_state_reuse():
    """no_grad inference; an old hidden state stays valid after later steps (functional state)."""
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.temporal import TemporalBackedge
    torch.manual_seed(0)
    B, N, F = 5, 8, 8
    ref, g = _dense_pair(F, F, 2)
    mem = DenseGCM(g, edge_selectors=TemporalBackedge([1, 3]), graph_size=N)
    obs = torch.rand(12, B, F)
    with torch.no_grad():
        hid, keep = None, None
        for t in range(12):
            mx, hid = mem(obs[t].to(DEV), hid)
            if t == 4:
                keep = tuple(x.clone() for x in hid), hid
        # the state captured at t = 4 was not touched by the 7 later steps (overflow included)
        for snap, live in zip(*keep):
            assert torch.equal(snap, live)
        out_c, hid_c = od.dense_rollout(obs, None, ref, graph_size=N,
                                        edge_selectors=od.TemporalBackedge([1, 3]))
    torch.testing.assert_close(mx.cpu(), out_c[-1], rtol=1e-5, atol=1e-6)
    assert torch.equal(hid[1].cpu(), hid_c[1])
