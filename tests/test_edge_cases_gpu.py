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
    torch.testing.assert_close(obs_d.grad.cpu(), obs_c.grad, rtol=1e-5, atol=1e-5 * float(obs_c.grad.abs().max()) + 1e-9)


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


# --------------------------------------------------------------------------
# host path: C++ autograd node, parameter-gradient chain
# --------------------------------------------------------------------------
def _cfg2_like(N=32, F=32, H=32, seed=0):
    from gcm.gcm import DenseGCM
    from gcm import nn as G
    from gcm.edge_selectors.temporal import TemporalBackedge
    torch.manual_seed(seed)
    g = G.Sequential("x, adj, weights, B, N", [(G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
                                               (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(DEV)
    return DenseGCM(g, edge_selectors=TemporalBackedge([1, 2, 4]), graph_size=N), g


def _loop(mem, obs, hidden=None, detach_at=()):
    outs = []
    for t in range(obs.shape[0]):
        if t in detach_at:
            hidden = tuple(h.detach() for h in hidden)
        mx, hidden = mem(obs[t], hidden)
        outs.append(mx)
    return torch.stack(outs), hidden


def test_cpp_node_matches_python_node(monkeypatch):
    """The C++ autograd node and the Python Function make the same C-ABI calls: identical
    beliefs, state and gradients (bit for bit), overflow included."""
    from gcm import _ext, _ops
    if _ext.module() is None:
        pytest.skip("torch extension not built (python __graft_entry__.py builds it)")
    obs = torch.rand(40, 3, 32, device=DEV)
    res = []
    for use_cpp in (True, False):
        mem, g = _cfg2_like()
        mem.rows_dx = False          # the round-1 fused step (one kernel per step and direction), both hosts
        if not use_cpp:
            monkeypatch.setattr(_ops.StepConfig, "cpp_handle", lambda self: 0)
        o = obs.clone().requires_grad_(True)
        out, hid = _loop(mem, o)
        (out * torch.linspace(0.5, 1.5, out.numel(), device=DEV).view_as(out)).sum().backward()
        res.append((out.detach(), hid, o.grad, [p.grad.clone() for p in g.parameters()]))
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    assert torch.equal(res[0][2], res[1][2])
    for a, b in zip(res[0][3], res[1][3]):
        # C++ node: per-graph slabs accumulate over the steps, one sum at the gate; Python node:
        # one slab sum per step, the engine adds the T results - same terms, different order
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max()))


def test_parameter_gradient_chain_survives_detach_and_restarts():
    """Truncated BPTT (hidden detached mid-sequence), two sequences through one module and an
    optimizer step between sequences: parameter gradients equal the ones of the rollout entry /
    of per-segment sums."""
    mem, g = _cfg2_like(seed=3)
    obs = torch.rand(24, 2, 32, device=DEV)
    # (a) detach at t = 10: gradients = sum of the two segments' own BPTT
    out, _ = _loop(mem, obs, detach_at=(10,))
    out.sum().backward()
    got = [p.grad.clone() for p in g.parameters()]
    g.zero_grad(set_to_none=True)
    o1, h1 = mem.rollout(obs[:10])
    o2, _ = mem.rollout(obs[10:], tuple(h.detach() for h in h1))
    (o1.sum() + o2.sum()).backward()
    for a, p in zip(got, g.parameters()):
        torch.testing.assert_close(a, p.grad, rtol=1e-5, atol=1e-5 * float(p.grad.abs().max()) + 1e-9)
    # (b) two independent sequences, one backward
    g.zero_grad(set_to_none=True)
    oa, _ = _loop(mem, obs[:8])
    ob, _ = _loop(mem, obs[8:20])
    (oa.sum() + ob.sum()).backward()
    got = [p.grad.clone() for p in g.parameters()]
    g.zero_grad(set_to_none=True)
    ra, _ = mem.rollout(obs[:8])
    rb, _ = mem.rollout(obs[8:20])
    (ra.sum() + rb.sum()).backward()
    for a, p in zip(got, g.parameters()):
        torch.testing.assert_close(a, p.grad, rtol=1e-5, atol=1e-5 * float(p.grad.abs().max()) + 1e-9)
    # (c) a parameter update in the middle of a kept hidden state: the next step must see the new
    # parameters (the chain restarts from the re-packed vector)
    g.zero_grad(set_to_none=True)
    out1, hid = _loop(mem, obs[:5])
    with torch.no_grad():
        for p in g.parameters():
            p.mul_(0.5)
    out2, _ = _loop(mem, obs[5:9], hid)
    ref2, _ = mem.rollout(obs[5:9], tuple(h.detach() for h in hid))
    torch.testing.assert_close(out2, ref2, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("M,I,O,bias", [(32768, 64, 32, True), (5000, 32, 32, True), (2049, 32, 1, True),
                                        (4097, 33, 17, False), (100, 64, 32, True)])
def test_skinny_linear_matches_nn_linear(M, I, O, bias):
    """gcm.nn.SkinnyLinear == nn.Linear: forward and input gradient (gcm_rows_linear) and weight / bias
    gradients (rows split over the grid) to fp32 summation order - each no further from the float64
    result than 3x nn.Linear's own fp32 result is.  M = 100 takes nn.Linear's own path."""
    from gcm import nn as G
    torch.manual_seed(M)
    ref = torch.nn.Linear(I, O, bias=bias).to(DEV)
    lin = G.SkinnyLinear(I, O, bias=bias).to(DEV)
    lin.load_state_dict(ref.state_dict())
    x = torch.randn(M, I, device=DEV)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = lin(xa.view(M // 1, I)), ref(xb)
    gy = torch.randn(M, O, device=DEV)
    ya.backward(gy)
    yb.backward(gy)
    w64, b64 = ref.weight.detach().double(), (ref.bias.detach().double() if bias else None)
    y64 = torch.nn.functional.linear(x.double(), w64, b64)
    gx64 = gy.double() @ w64
    for got, lib, want in ((ya, yb, y64), (xa.grad, xb.grad, gx64)):
        err_lib = float((lib.detach().double() - want).abs().max())
        assert float((got.detach().double() - want).abs().max()) <= max(3 * err_lib, 2e-6 * float(want.abs().max()))
    scale = float(ref.weight.grad.abs().max())
    torch.testing.assert_close(lin.weight.grad, ref.weight.grad, rtol=1e-5, atol=1e-5 * scale)
    if bias:
        torch.testing.assert_close(lin.bias.grad, ref.bias.grad, rtol=1e-5, atol=1e-5 * float(ref.bias.grad.abs().max()))
    assert set(lin.state_dict()) == set(ref.state_dict())


def test_parameter_gate_under_partial_and_repeated_backward():
    """The slab array the step nodes accumulate into is flushed by the gate on every pass:
    autograd.grad for the observations only (parameters not requested) leaves nothing behind,
    retain_graph + a second backward gives the same parameter gradients again, and a step whose
    outputs do not reach the loss contributes nothing."""
    mem, g = _cfg2_like(seed=5)
    obs = torch.rand(12, 3, 32, device=DEV, requires_grad=True)
    params = list(g.parameters())
    out, _ = _loop(mem, obs)
    loss = (out[:9] ** 2).sum()                      # the last three steps do not reach the loss
    (g_obs,) = torch.autograd.grad(loss, [obs], retain_graph=True)
    assert float(g_obs[9:].abs().max()) == 0.0
    first = torch.autograd.grad(loss, params, retain_graph=True)
    second = torch.autograd.grad(loss, params)
    for a, b in zip(first, second):
        assert torch.equal(a, b)
    ref_out, _ = mem.rollout(obs[:9])
    ref = torch.autograd.grad((ref_out ** 2).sum(), params)
    for a, b in zip(first, ref):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max()))


@pytest.mark.parametrize("M,F", [(32768, 32), (5000, 17), (2500, 64), (1, 32), (300, 33)])
def test_relu_layernorm_matches_torch(M, F):
    """gcm_relu_layernorm_fwd/bwd == LayerNorm(relu(x)) of torch (outputs 1e-5, gradients 1e-4)."""
    from gcm import _ops
    torch.manual_seed(M + F)
    ln = torch.nn.LayerNorm(F).to(DEV)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.uniform_(-0.5, 0.5)
    x = torch.randn(M, F, device=DEV)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ga, ba = ln.weight.detach().clone().requires_grad_(True), ln.bias.detach().clone().requires_grad_(True)
    ya = _ops.relu_layernorm(xa, ga, ba, ln.eps)
    yb = ln(torch.relu(xb))
    torch.testing.assert_close(ya, yb, rtol=1e-5, atol=1e-5)
    gy = torch.randn(M, F, device=DEV)
    ya.backward(gy)
    yb.backward(gy)
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-5, atol=1e-5 * float(xb.grad.abs().max()) + 1e-9)
    torch.testing.assert_close(ga.grad, ln.weight.grad, rtol=1e-5, atol=1e-5 * float(ln.weight.grad.abs().max()) + 1e-6)
    torch.testing.assert_close(ba.grad, ln.bias.grad, rtol=1e-5, atol=1e-5 * float(ln.bias.grad.abs().max()) + 1e-6)


def test_default_edge_network_path_matches_module():
    """LearnedEdge's default edge network through skinny_linear / relu_layernorm == the nn.Sequential
    called as a module (outputs and every parameter gradient); a user-supplied network is left alone."""
    from gcm import _ops
    from gcm.edge_selectors.learned import LearnedEdge
    torch.manual_seed(0)
    sel = LearnedEdge(32).to(DEV)
    net = sel.edge_network
    assert _ops.default_edge_network(net) is not None
    x = torch.randn(64, 128, 64, device=DEV)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = _ops.edge_network_forward(net, xa)
    ya.square().sum().backward()
    got = [p.grad.clone() for p in net.parameters()]
    net.zero_grad(set_to_none=True)
    yb = net(xb)
    yb.square().sum().backward()
    torch.testing.assert_close(ya, yb, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-5, atol=1e-5 * float(xb.grad.abs().max()))
    for a, p in zip(got, net.parameters()):
        torch.testing.assert_close(a, p.grad, rtol=1e-5, atol=1e-5 * float(p.grad.abs().max()))
    custom = torch.nn.Sequential(torch.nn.Linear(64, 8), torch.nn.Tanh(), torch.nn.Linear(8, 1)).to(DEV)
    assert _ops.default_edge_network(custom) is None
    assert _ops.edge_network_forward(custom, x).shape == (64, 128, 1)


def test_learned_edge_single_node_path_equals_stepwise_path():
    """LearnedEdge with the default edge network and the device RNG runs as one autograd node;
    with the same RNG state it must equal the op-by-op path (which the golden vectors pin) bit for
    bit: adjacency, beliefs and every gradient."""
    from gcm.gcm import DenseGCM
    from gcm import nn as G
    from gcm.edge_selectors.learned import LearnedEdge
    B, N, F, H, T = 32, 64, 32, 32, 12          # B * N = 2048 rows: the fused selector applies

    def build():
        torch.manual_seed(11)
        g = G.Sequential("x, adj, weights, B, N", [(G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
                                                   (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(DEV)
        sel = LearnedEdge(F).to(DEV)
        return DenseGCM(g, edge_selectors=sel, graph_size=N), g, sel

    obs = torch.rand(T, B, F, device=DEV)
    res = []
    for stepwise in (False, True):
        mem, g, sel = build()
        if stepwise:   # an explicit noise function takes the op-by-op path; same draws from the same RNG state
            sel.noise_fn = lambda logits: -torch.empty_like(logits).exponential_().log()
        torch.manual_seed(99)
        o = obs.clone().requires_grad_(True)
        hid, outs = None, []
        for t in range(T):
            mx, hid = mem(o[t], hid)
            outs.append(mx)
        torch.stack(outs).square().sum().backward()
        res.append((torch.stack(outs).detach(), hid[1].detach(), o.grad,
                    [p.grad.clone() for p in list(g.parameters()) + list(sel.parameters())]))
    assert torch.equal(res[0][1], res[1][1]) and float(res[0][1].sum()) > 0
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][2], res[1][2])
    for a, b in zip(res[0][3], res[1][3]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("kind", ["euclid", "learned", "learned_short", "fold_pre", "temporal"])
@pytest.mark.parametrize("start", ["none", "state"])
def test_rollout_entry_equals_single_steps_every_selector(kind, start):
    """DenseGCM.rollout for the configurations that run it as the loop of per-step kernels on a
    state the call owns (advanced in place whatever donate_state says): == T single functional steps - state bit
    exact, beliefs / parameter gradients to summation order - from hidden = None and from a caller's state, which
    must come back untouched; batch_first=True ([B, T, F] in, [B, T, H] out: ray_gcm.py:186-209) is the same run."""
    from gcm import nn as G
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.temporal import TemporalBackedge
    from gcm.edge_selectors.distance import EuclideanEdge
    from gcm.edge_selectors.learned import LearnedEdge
    B, N, F, H, T = 40, 32, 32, 32, 45          # (T > N: the graphs overflow inside the rollout)
    if kind == "learned_short":                 # T <= N from hidden = None: the three-launch time-parallel forward
        kind, T = "learned", 29
    torch.manual_seed(3)
    centres = 3 * torch.randn(5, F)
    obs = (centres[torch.arange(T) % 5][:, None, :] + 0.05 * torch.randn(T, B, F)).to(DEV)
    noise = -torch.empty(T, B, N).exponential_().log().to(DEV)
    w = torch.rand(T, B, H, device=DEV)

    def make():
        torch.manual_seed(11)
        g = G.Sequential("x, adj, weights, B, N", [
            (G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
            (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(DEV)
        pre, step = None, {"t": 0}
        if kind == "euclid":
            sel = EuclideanEdge(3.0)
        elif kind == "learned":
            sel = LearnedEdge(F, num_edge_samples=4).to(DEV)
            sel.noise_fn = lambda like: noise[step["t"]]
        elif kind == "fold_pre":
            sel, pre = TemporalBackedge([1, 3]), torch.nn.Linear(F, F).to(DEV)
        else:
            sel = TemporalBackedge([1, 2, 4])
        mem = DenseGCM(g, preprocessor=pre, edge_selectors=sel, graph_size=N)
        return mem, g, sel, pre, step

    def state0():
        if start == "none":
            return None
        torch.manual_seed(5)
        c0 = torch.randint(0, N // 2, (B,))
        n0 = torch.rand(B, N, F) * (torch.arange(N)[None, :, None] < c0[:, None, None])
        a0 = torch.zeros(B, N, N)
        i = torch.arange(1, N)
        a0[:, i, i - 1] = 1.0
        a0 = a0 * (torch.arange(N)[None, :, None] < c0[:, None, None])
        return (n0.to(DEV), a0.to(DEV), torch.zeros(0, device=DEV), c0.to(DEV))

    # T single steps, functional state
    mem, g, sel, pre, step = make()
    hid, outs = state0(), []
    for t in range(T):
        step["t"] = t
        mx, hid = mem(obs[t], hid)
        outs.append(mx)
    want = torch.stack(outs)
    (want * w).sum().backward()
    mem.check_flags()
    mods = [g] + ([sel] if kind == "learned" else []) + ([pre] if pre is not None else [])
    want_g = [p.grad.clone() for m in mods for p in m.parameters()]
    for bf in (False, True):
        mem2, g2, sel2, pre2, step2 = make()
        if kind == "learned":       # rollout() makes its own steps: the draws by call count
            cnt = {"t": 0}

            def nf(like, cnt=cnt):
                cnt["t"] += 1
                return noise[cnt["t"] - 1]
            sel2.noise_fn = nf
        h0 = state0()
        keep = None if h0 is None else tuple(t.clone() for t in h0)
        o = obs.transpose(0, 1).contiguous() if bf else obs
        got, hid2 = mem2.rollout(o, h0, batch_first=bf)
        if bf:
            assert got.shape == (B, T, H)
            got = got.transpose(0, 1)
        (got * w).sum().backward()
        mem2.check_flags()
        assert not mem2.donate_state
        if keep is not None:        # the caller's state is untouched
            assert all(torch.equal(a, b_) for a, b_ in zip(h0, keep))
            assert hid2[0].data_ptr() != h0[0].data_ptr()
        assert torch.equal(hid2[0], hid[0]) and torch.equal(hid2[1], hid[1]) and torch.equal(hid2[3], hid[3])
        torch.testing.assert_close(got, want, rtol=1e-5, atol=2e-6)
        mods2 = [g2] + ([sel2] if kind == "learned" else []) + ([pre2] if pre2 is not None else [])
        got_g = [p.grad for m in mods2 for p in m.parameters()]
        scale = max(float(x.abs().max()) for x in want_g)
        for a, b_ in zip(got_g, want_g):
            torch.testing.assert_close(a, b_, rtol=1e-5, atol=2e-6 * scale)


def test_sparse_rollout_entry_is_one_call():
    """SparseGCM.rollout(x [B, T, F]) == T single-node calls (tests/test_sparse_gcm.py:469-540 of the reference pin
    exactly that for forward())."""
    from gcm.sparse_gcm import SparseGCM
    from gcm import nn as G
    from gcm.sparse_edge_selectors.temporal import TemporalEdge
    B, N, F, H, T = 5, 24, 32, 32, 24
    torch.manual_seed(0)
    g = G.Sequential("x, edges, weights", [(G.GraphConv(F, H), "x, edges, weights -> x"), torch.nn.Tanh(),
                                           (G.GraphConv(H, H), "x, edges, weights -> x"), torch.nn.Tanh()]).to(DEV)
    mem = SparseGCM(g, edge_selectors=TemporalEdge([1, 2]), graph_size=N)
    x = torch.rand(B, T, F, device=DEV)
    out, hid = mem.rollout(x)
    one = torch.ones(B, dtype=torch.long, device=DEV)
    h, outs = None, []
    for t in range(T):
        o, h = mem(x[:, t:t + 1].contiguous(), one, h)
        outs.append(o)
    torch.testing.assert_close(out, torch.cat(outs, 1), rtol=1e-5, atol=2e-6)
    assert torch.equal(hid[1].coalesce().indices(), h[1].coalesce().indices()) and torch.equal(hid[0], h[0])
    assert torch.equal(hid[2], h[2])
