"""The live-row DenseGCM step (csrc/rows_step.hip, rows_bptt.hip): one kernel per forward step on
the rows that reach the belief, donated or functional state, time-parallel parameter backward run by
the chain's one autograd node.  Parity against the reference's golden vectors and the CPU oracle.  Needs an MI355X."""
import ctypes
import os

import pytest
import torch

from _golden import Fixture, oracle_selector
from oracle import dense as od
from test_dense_gpu import dev_gnn_from, product_selector, DEV, RTOL, ATOL

pytestmark = pytest.mark.gpu


from _golden import fp64_bound as _fp64_bound, fp64_grad_bound as _fp64_grad_bound  # noqa: E402
from _golden import fp64_rollout_bounds as _fp64_rollout_bounds  # noqa: E402


def _rows_taken(mem):
    """True when the module's last step ran on the live-row kernels (csrc/rows_step.hip)."""
    return mem.rows_steps() > 0


GOLD = ["g1_temporal_h1", "g2_temporal_h124_both", "g1b_cfg1", "g5_dense_edge", "g13_exact_temporal",
        "g13_exact_dense",
        # distance selectors: they run ahead of the step kernel, on the incoming state
        "g3_euclid", "g3_euclid_mixed", "g3_euclid_learned", "g4_spatial", "g4_spatial_ab", "g4_cosine"]


@pytest.mark.parametrize("donate", [False, True])
@pytest.mark.parametrize("name", GOLD)
def test_rows_path_matches_reference(name, donate):
    """obs without gradient => the live-row kernels run (forward state bit exact, beliefs 1e-5,
    parameter gradients from the time-parallel backward) - against the reference's own vectors."""
    from gcm.gcm import DenseGCM
    fx = Fixture(name)
    m = fx.meta
    ref = od.canonical_gnn(m["F"], m["H"])
    ref.load_state_dict(fx.group("param:"))
    g = dev_gnn_from(ref, [(m["F"], m["H"], torch.nn.Tanh), (m["H"], m["H"], torch.nn.Tanh)])
    mem = DenseGCM(g, edge_selectors=product_selector(m, fx.group("sel_param:")), graph_size=m["N"],
                   donate_state=donate)
    obs = fx["obs"].to(DEV)
    h0 = fx.h0()
    hidden = None if h0 is None else tuple(t.to(DEV).clone() for t in h0)
    mxs, sums = [], []
    for t in range(m["T"]):
        mx, hidden = mem(obs[t], hidden)
        mxs.append(mx)
        sums.append(hidden[1].sum(dim=(1, 2)))
    if m["N"] % 4 == 0 and m["F"] % 4 == 0:      # else: the fused kernels (same results)
        assert _rows_taken(mem)
    mxs = torch.stack(mxs)
    mxs.mean().backward()
    mem.check_flags()
    assert torch.equal(hidden[1].cpu(), fx["hT_adj"])            # adjacency: bit exact
    assert torch.equal(torch.stack(sums).cpu(), fx["adj_sums"])
    assert torch.equal(hidden[0].cpu(), fx["hT_nodes"])
    assert torch.equal(hidden[3].cpu(), fx["hT_num_nodes"])
    if m["selector"] == "dense":       # up to N terms per aggregate: bound through float64
        out64, atol = _fp64_bound(ref, fx["obs"], fx.h0(), fx["mx"], graph_size=m["N"],
                                  edge_selectors=oracle_selector(m))
        assert float((mxs.detach().cpu().double() - out64).abs().max()) <= atol
    else:
        torch.testing.assert_close(mxs.cpu(), fx["mx"], rtol=RTOL, atol=ATOL)
    # parameter gradients: bounded through the float64 evaluation (≈ 1e-6 of the gradient scale)
    bounds = _fp64_grad_bound(ref, fx, oracle_selector(m, fx.group("sel_param:")))
    for k, p in g.named_parameters():
        g64, atol = bounds[k]
        err = float((p.grad.cpu().double() - g64).abs().max())
        assert err <= atol, (k, err, atol)


CASES = [
    # B, N, F, H1, H2, T, selector
    (3, 8, 4, 16, 16, 13, ("temporal", [1, 2], "forward")),
    (5, 12, 8, 8, 24, 30, ("temporal", [1, 3], "both")),
    (4, 32, 32, 32, 32, 40, ("temporal", [1, 2, 4], "backward")),
    (2, 128, 32, 32, 32, 140, ("temporal", [1, 2, 4], "forward")),
    (3, 20, 12, 40, 8, 26, ("dense", None, None)),
    (2, 64, 64, 64, 64, 70, ("dense", None, None)),
    (2, 36, 36, 20, 44, 40, ("temporal", [2, 5], "both")),
    (300, 16, 8, 16, 16, 20, ("temporal", [1], "forward")),
]


def _mk(B, N, F, H1, H2, sel, donate):
    from gcm import nn as G
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.temporal import TemporalBackedge
    from gcm.edge_selectors.dense import DenseEdge
    from oracle import pyg
    ref = pyg.Sequential("x, adj, weights, B, N", [
        (pyg.DenseGraphConv(F, H1), "x, adj -> x"), torch.nn.Tanh(),
        (pyg.DenseGraphConv(H1, H2), "x, adj -> x"), torch.nn.Tanh()])
    g = G.Sequential("x, adj, weights, B, N", [
        (G.DenseGraphConv(F, H1), "x, adj -> x"), torch.nn.Tanh(),
        (G.DenseGraphConv(H1, H2), "x, adj -> x"), torch.nn.Tanh()])
    g.load_state_dict(ref.state_dict())
    g = g.to(DEV)
    if sel[0] == "temporal":
        ps, osel = TemporalBackedge(sel[1], direction=sel[2]), od.TemporalBackedge(sel[1], sel[2])
    else:
        ps, osel = DenseEdge(), od.DenseEdge()
    mem = DenseGCM(g, edge_selectors=ps, graph_size=N, donate_state=donate)
    return ref, g, mem, osel


@pytest.mark.parametrize("donate", [False, True])
@pytest.mark.parametrize("case", CASES)
def test_rows_path_vs_oracle(case, donate):
    """Ragged and tile-exact shapes, staggered starts, overflow crossings, vs the CPU oracle."""
    B, N, F, H1, H2, T, sel = case
    torch.manual_seed(B * 31 + N)
    ref, g, mem, osel = _mk(B, N, F, H1, H2, sel, donate)
    obs = torch.rand(T, B, F)
    # staggered starts: graph b already holds count0[b] nodes (and the matching temporal adjacency
    # would be there too - the oracle and the product both start from this very state)
    count0 = torch.randint(0, N + 1, (B,))
    nodes0 = torch.rand(B, N, F) * (torch.arange(N)[None, :, None] < count0[:, None, None])
    adj0 = torch.zeros(B, N, N)
    for b in range(B):
        for i in range(1, int(count0[b])):
            adj0[b, i, i - 1] = 1.0
    h_o = (nodes0.clone(), adj0.clone(), torch.zeros(0), count0.clone())
    h_d = (nodes0.to(DEV), adj0.to(DEV), torch.zeros(0, device=DEV), count0.to(DEV))
    out_c, hid_c = od.dense_rollout(obs, h_o, ref, graph_size=N, edge_selectors=osel)
    wgt = torch.rand(T, B, H2)
    (out_c * wgt).sum().backward()
    obs_d = obs.to(DEV)
    outs, hidden = [], h_d
    for t in range(T):
        mx, hidden = mem(obs_d[t], hidden)
        outs.append(mx)
    out_d = torch.stack(outs)
    (out_d * wgt.to(DEV)).sum().backward()
    mem.check_flags()
    assert _rows_taken(mem)
    if donate:   # the state was advanced in place: same tensor objects all the way
        assert hidden[0] is h_d[0] and hidden[1] is h_d[1] and hidden[3] is h_d[3]
    else:
        assert torch.equal(h_d[0].cpu(), nodes0) and torch.equal(h_d[1].cpu(), adj0)
        assert torch.equal(h_d[3].cpu(), count0)
    assert torch.equal(hidden[1].cpu(), hid_c[1]) and torch.equal(hidden[0].cpu(), hid_c[0])
    assert torch.equal(hidden[3].cpu(), hid_c[3])
    if sel[0] == "dense":              # up to N terms per aggregate: bound through float64
        out64, atol = _fp64_bound(ref, obs, h_o, out_c, graph_size=N, edge_selectors=osel)
        assert float((out_d.detach().cpu().double() - out64).abs().max()) <= atol
    else:
        torch.testing.assert_close(out_d.cpu(), out_c, rtol=RTOL, atol=2e-6)
    for (k, pc), (_, pd) in zip(ref.named_parameters(), g.named_parameters()):
        scale = float(pc.grad.abs().max()) + 1e-12
        torch.testing.assert_close(pd.grad.cpu(), pc.grad, rtol=1e-5, atol=2e-5 * scale, msg=k)


def test_rows_foreign_adjacency():
    """A caller's own state may hold anything: entries in row cur and beyond (the reference keeps
    and uses them).  The live-row list is built from the real row, not from the selector."""
    B, N, F, H = 4, 16, 8, 16
    torch.manual_seed(3)
    ref, g, mem, osel = _mk(B, N, F, H, H, ("temporal", [1], "forward"), False)
    count0 = torch.tensor([3, 5, 0, 9])
    nodes0 = torch.rand(B, N, F)
    adj0 = (torch.rand(B, N, N) < 0.3).float() * torch.rand(B, N, N).round(decimals=2)
    obs = torch.rand(3, B, F)
    out_c, hid_c = od.dense_rollout(obs, (nodes0.clone(), adj0.clone(), torch.zeros(0), count0.clone()),
                                    ref, graph_size=N, edge_selectors=osel)
    hidden = (nodes0.to(DEV), adj0.to(DEV), torch.zeros(0, device=DEV), count0.to(DEV))
    outs = []
    with torch.no_grad():
        for t in range(3):
            mx, hidden = mem(obs[t].to(DEV), hidden)
            outs.append(mx)
    torch.testing.assert_close(torch.stack(outs).cpu(), out_c.detach(), rtol=RTOL, atol=5e-6)
    assert torch.equal(hidden[1].cpu(), hid_c[1]) and torch.equal(hidden[0].cpu(), hid_c[0])


def test_rows_step_c_abi_against_functional_kernel():
    """gcm_dense_rows_step_fwd (donated and functional) against gcm_dense_step_fwd, raw C ABI."""
    from gcm import _hip
    lib = _hip.lib()
    B, N, F, H = 7, 64, 32, 32
    torch.manual_seed(0)
    P = lib.gcm_dense_gnn2_param_count(F, H, H)
    params = (torch.randn(P) * 0.2).to(DEV)
    count = torch.tensor([0, 1, 5, 31, 63, 64, 40]).to(DEV)
    nodes = torch.rand(B, N, F) * (torch.arange(N)[None, :, None] < count.cpu()[:, None, None])
    nodes = nodes.to(DEV)
    adj = torch.zeros(B, N, N)
    for b in range(B):
        for i in range(2, int(count[b])):
            adj[b, i, i - 1] = adj[b, i, i - 2] = 1.0
            adj[b, i - 2, i] = 1.0
    adj = adj.to(DEV)
    obs = torch.rand(B, F, device=DEV)
    d = _hip.SelectorDesc(kind=_hip.SEL_TEMPORAL, n_hops=2, direction=_hip.DIR["both"])
    d.hops[0], d.hops[1] = 1, 2
    arr = (_hip.SelectorDesc * 1)(d)
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    st = _hip.stream()
    p = _hip.ptr
    # functional reference kernels
    n_ref, a_ref = torch.empty_like(nodes), torch.empty_like(adj)
    ib = torch.empty(2, B, dtype=torch.int64, device=DEV)
    mx_ref = torch.empty(B, H, device=DEV)
    rc = lib.gcm_dense_step_fwd(p(obs), p(nodes), p(adj), p(count), p(n_ref), p(a_ref), ib.data_ptr(),
                                ib.data_ptr() + 8 * B, ctypes.addressof(arr), 1, p(params), 3, 1, 1,
                                p(mx_ref), None, None, None, p(flags), None, 0, B, N, F, H, H, st)
    assert rc == 0
    lay = (ctypes.c_size_t * 6)()
    assert lib.gcm_dense_rows_layout(B, N, F, H, H, ctypes.addressof(lay)) == 0
    for donate in (False, True):
        n_in, a_in, c_in = nodes.clone(), adj.clone(), count.clone()
        n_out = n_in if donate else torch.empty_like(nodes)
        a_out = a_in if donate else torch.empty_like(adj)
        c_out = c_in if donate else torch.empty_like(count)
        saved = torch.zeros(lay[0], device=DEV)
        rc = lib.gcm_dense_rows_step_fwd(p(obs), p(n_in), p(a_in), p(c_in), p(n_out), p(a_out), p(c_out),
                                         None, ctypes.addressof(arr), 1, p(params), 3, 1, 1, p(saved),
                                         p(saved), p(flags), B, N, F, H, H, st)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(n_out, n_ref) and torch.equal(a_out, a_ref)
        assert torch.equal(c_out, ib[1])
        torch.testing.assert_close(saved[:B * H].view(B, H), mx_ref, rtol=RTOL, atol=ATOL)
        if not donate:
            assert torch.equal(n_in, nodes) and torch.equal(a_in, adj) and torch.equal(c_in, count)
        hdr = saved[lay[2]:lay[2] + 4 * B].view(torch.int32).view(B, 4).cpu()
        cur = (ib[0]).cpu()
        assert torch.equal(hdr[:, 2].long(), cur)
        want_L = torch.tensor([int((a_ref[b, int(cur[b])] != 0).sum()) + (0 if a_ref[b, int(cur[b]), int(cur[b])] != 0 else 1)
                               for b in range(B)])
        assert torch.equal(hdr[:, 0].long(), want_L)


def test_rows_mixed_with_obs_gradient_steps():
    """A chain whose observations need a gradient takes the fused kernels; switching between the
    two kinds of step inside one chain keeps results and gradients right."""
    B, N, F, H, T = 3, 32, 32, 32, 12
    torch.manual_seed(5)
    ref, g, mem, osel = _mk(B, N, F, H, H, ("temporal", [1, 2], "forward"), False)
    obs = torch.rand(T, B, F)
    need = [t % 3 == 1 for t in range(T)]
    oc = [obs[t].clone().requires_grad_(need[t]) for t in range(T)]
    odv = [obs[t].to(DEV).requires_grad_(need[t]) for t in range(T)]
    hid_c, outs_c = None, []
    for t in range(T):
        mx, hid_c = od.dense_step(oc[t], hid_c, ref, graph_size=N, edge_selectors=osel)
        outs_c.append(mx)
    torch.stack(outs_c).mean().backward()
    hid, outs = None, []
    for t in range(T):
        mx, hid = mem(odv[t], hid)
        outs.append(mx)
    torch.stack(outs).mean().backward()
    torch.testing.assert_close(torch.stack(outs).cpu(), torch.stack(outs_c).detach(), rtol=RTOL, atol=ATOL)
    for t in range(T):
        if need[t]:
            torch.testing.assert_close(odv[t].grad.cpu(), oc[t].grad, rtol=1e-5, atol=1e-5 * float(oc[t].grad.abs().max()) + 1e-9)
    for (k, pc), (_, pd) in zip(ref.named_parameters(), g.named_parameters()):
        torch.testing.assert_close(pd.grad.cpu(), pc.grad, rtol=1e-5, atol=1e-5 * float(pc.grad.abs().max()), msg=k)


def test_rows_two_backward_passes_and_partial_loss():
    """retain_graph, a loss on part of the steps, two sequences through one module."""
    B, N, F, H, T = 4, 16, 8, 16, 9
    torch.manual_seed(7)
    ref, g, mem, osel = _mk(B, N, F, H, H, ("temporal", [1], "both"), True)
    obs = torch.rand(2, T, B, F)
    def run_ref():
        tot = 0
        for s in range(2):
            out, _ = od.dense_rollout(obs[s], None, ref, graph_size=N, edge_selectors=osel)
            tot = tot + out[T // 2:].sum() * (s + 1)
        return tot
    run_ref().backward()
    tot = 0
    for s in range(2):
        hid, outs = None, []
        for t in range(T):
            mx, hid = mem(obs[s, t].to(DEV), hid)
            outs.append(mx)
        tot = tot + torch.stack(outs[T // 2:]).sum() * (s + 1)
    tot.backward(retain_graph=True)
    for (k, pc), (_, pd) in zip(ref.named_parameters(), g.named_parameters()):
        torch.testing.assert_close(pd.grad.cpu(), pc.grad, rtol=1e-5, atol=1e-5 * float(pc.grad.abs().max()), msg=k)
    g.zero_grad()
    tot.backward()
    for (k, pc), (_, pd) in zip(ref.named_parameters(), g.named_parameters()):
        torch.testing.assert_close(pd.grad.cpu(), pc.grad, rtol=1e-5, atol=1e-5 * float(pc.grad.abs().max()), msg=k)


def test_rows_graph_capture_replay():
    """The per-step loop + backward captured in a HIP graph (in process) and replayed: state bit
    exact against the reference's g13 vectors after every replay, parameter gradients equal."""
    from gcm.gcm import DenseGCM
    fx = Fixture("g13_exact_temporal")
    m = fx.meta
    ref = od.canonical_gnn(m["F"], m["H"])
    ref.load_state_dict(fx.group("param:"))
    g = dev_gnn_from(ref, [(m["F"], m["H"], torch.nn.Tanh), (m["H"], m["H"], torch.nn.Tanh)])
    mem = DenseGCM(g, edge_selectors=product_selector(m, fx.group("sel_param:")), graph_size=m["N"],
                   donate_state=True)
    obs = fx["obs"].to(DEV)
    T, B = m["T"], obs.shape[1]
    h0 = tuple(t.to(DEV) for t in fx.h0())

    def rollout():
        hidden = tuple(t.clone() for t in h0)      # (re-created inside the graph on every replay)
        outs = []
        for t in range(T):
            mx, hidden = mem(obs[t], hidden)
            outs.append(mx)
        out = torch.stack(outs)
        out.mean().backward()
        return out, hidden

    for _ in range(3):                       # warm-up on a side stream, as torch.cuda.graphs asks
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            g.zero_grad(set_to_none=True)
            rollout()
        torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g.zero_grad(set_to_none=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out, hidden = rollout()
    for rep in range(3):
        for p in g.parameters():
            p.grad.zero_()
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(hidden[1].cpu(), fx["hT_adj"])
        assert torch.equal(hidden[0].cpu(), fx["hT_nodes"])
        assert torch.equal(hidden[3].cpu(), fx["hT_num_nodes"])
        torch.testing.assert_close(out.cpu(), fx["mx"], rtol=RTOL, atol=ATOL)
        for k, p in g.named_parameters():
            want = fx["grad:" + k]
            torch.testing.assert_close(p.grad.cpu(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()), msg=k)
    mem.check_flags()


@pytest.mark.parametrize("kind,B,N,F,T", [("euclid", 40, 128, 64, 140), ("euclid", 33, 32, 20, 40),
                                          ("cosine", 6, 64, 32, 70), ("spatial", 5, 36, 12, 40)])
def test_rows_path_distance_selectors_vs_fused_path(kind, B, N, F, T):
    """Distance selectors ahead of the live-row step (matrix-core Euclidean kernel for B >= 32) ==
    the fused path that runs them on the advanced state: same adjacency bit for bit through the
    overflow, same beliefs and parameter gradients."""
    from gcm import nn as G
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.distance import EuclideanEdge, CosineEdge, SpatialEdge
    torch.manual_seed(B + N)
    H = 32
    centres = 3 * torch.randn(6, F)
    # all graphs visit the same cluster at time t: the cross-batch mean distance separates the clusters
    obs = (centres[torch.arange(T) % 6][:, None, :] + 0.05 * torch.randn(T, B, F)).to(DEV)

    def build(donate):
        torch.manual_seed(1)
        g = G.Sequential("x, adj, weights, B, N", [
            (G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
            (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(DEV)
        sel = {"euclid": lambda: EuclideanEdge(3.0), "cosine": lambda: CosineEdge(0.5),
               "spatial": lambda: SpatialEdge(1.0, slice(0, 3))}[kind]()
        return DenseGCM(g, edge_selectors=sel, graph_size=N, donate_state=donate), g

    res = []
    for mode in ("fused", "rows", "rows_donated"):
        mem, g = build(mode == "rows_donated")
        mem.rows_dx = mode != "fused"                        # the round-1 fused kernels for the obs-gradient steps
        x = obs.clone().requires_grad_(mode == "fused")     # a gradient w.r.t. obs => the fused path
        hidden, outs, sums = None, [], []
        for t in range(T):
            mx, hidden = mem(x[t], hidden)
            outs.append(mx)
            sums.append(hidden[1].sum(dim=(1, 2)).clone())
        assert _rows_taken(mem) == (mode != "fused")
        # a donated chain from empty graphs: its first N steps run on the chain's caches (the selector's decision row
        # handed to k_step_rows_cached_sel; widths that are not 32 / 64 - F = 20, 12 here - in the padded form), the
        # rest - the graphs roll - on the general kernel
        # (round 5: EuclideanEdge alone in its one-launch form stays on the chain past N steps - the steady-state step,
        #  gcm_edge_distance_step_ring, counted with the cached ones)
        if mode == "rows_donated":
            ring = kind == "euclid" and B >= 32 and F in (32, 64)
            assert mem.rows_cached_steps_taken() == (T if ring else min(T, N))
            assert mem.rows_rolled_steps_taken() == (max(0, T - N) if ring else 0)
        out = torch.stack(outs)
        (out * torch.linspace(0.5, 1.5, out.numel(), device=DEV).view_as(out)).sum().backward()
        mem.check_flags()
        res.append((out.detach(), hidden[1].clone(), torch.stack(sums), {k_: p.grad.clone() for k_, p in g.named_parameters()}))
    assert float(res[0][2].max()) > 0                        # the thresholds do connect nodes
    for r in res[1:]:
        assert torch.equal(r[1], res[0][1]) and torch.equal(r[2], res[0][2])
        # (two fp32 evaluations with different orders of the row sums - dozens of rows per aggregate here -: 1e-5,
        #  north_star's tolerance, on each)
        torch.testing.assert_close(r[0], res[0][0], rtol=1e-5, atol=1e-5)
        for k_ in r[3]:
            scale = float(res[0][3][k_].abs().max()) + 1e-12
            torch.testing.assert_close(r[3][k_], res[0][3][k_], rtol=1e-5, atol=2e-5 * scale, msg=k_)


@pytest.mark.parametrize("kind", ["fold_pre", "euclid", "learned"])
def test_graph_capture_replay_other_step_kinds(kind):
    """The folded step, a distance selector ahead of the step and the fused LearnedEdge step are free of
    host syncs too: loop + backward captured once and replayed == the same loop run eagerly (state bit
    exact; LearnedEdge with injected draws so that both runs sample the same edges)."""
    from gcm import nn as G
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.temporal import TemporalBackedge
    from gcm.edge_selectors.distance import EuclideanEdge
    from gcm.edge_selectors.learned import LearnedEdge
    torch.manual_seed(3)
    B, N, F, H, T = 40, 32, 32, 32, 40
    centres = 3 * torch.randn(5, F)
    obs = (centres[torch.arange(T) % 5][:, None, :] + 0.05 * torch.randn(T, B, F)).to(DEV)
    noise = -torch.empty(T, B, N, device=DEV).exponential_().log()
    step = {"t": 0}

    def build():
        torch.manual_seed(4)
        g = G.Sequential("x, adj, weights, B, N", [
            (G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
            (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(DEV)
        kw, mods = {}, [g]
        if kind == "fold_pre":
            kw = dict(preprocessor=torch.nn.Linear(F, F).to(DEV), edge_selectors=TemporalBackedge([1, 2]))
            mods.append(kw["preprocessor"])
        elif kind == "euclid":
            kw = dict(edge_selectors=EuclideanEdge(3.0))
        else:
            sel = LearnedEdge(F, num_edge_samples=3).to(DEV)
            sel.noise_fn = lambda logits: noise[step["t"]]
            kw = dict(edge_selectors=sel)
            mods.append(sel)
        return DenseGCM(g, graph_size=N, **kw), mods

    def rollout(mem):
        hidden, outs = None, []
        for t in range(T):
            step["t"] = t
            mx, hidden = mem(obs[t], hidden)
            outs.append(mx)
        out = torch.stack(outs)
        (out * torch.linspace(0.5, 1.5, out.numel(), device=DEV).view_as(out)).sum().backward()
        return out, hidden

    mem_e, mods_e = build()
    out_e, hid_e = rollout(mem_e)
    mem_e.check_flags()
    grads_e = [p.grad.clone() for m_ in mods_e for p in m_.parameters() if p.grad is not None]
    assert float(hid_e[1].sum()) > 0

    mem_g, mods_g = build()
    params = [p for m_ in mods_g for p in m_.parameters()]
    for _ in range(2):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for p in params:
                p.grad = None
            rollout(mem_g)
        torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    for p in params:
        p.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_g, hid_g = rollout(mem_g)
    for _ in range(2):
        for p in params:
            if p.grad is not None:
                p.grad.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(hid_g[1].detach(), hid_e[1].detach()) and torch.equal(hid_g[0], hid_e[0])
        torch.testing.assert_close(out_g.detach(), out_e.detach(), rtol=0, atol=0)
        grads_g = [p.grad for p in params if p.grad is not None]
        assert len(grads_g) == len(grads_e)
        for a, b_ in zip(grads_g, grads_e):
            torch.testing.assert_close(a, b_, rtol=1e-5, atol=1e-6 * float(b_.abs().max()) + 1e-9)
    mem_g.check_flags()


def test_rows_inplace_on_returned_belief_is_detected():
    """the belief tensor aliases the head of the record the time-parallel backward reads; a
    legal in-place op on it must make backward raise (as stock autograd does for a saved tensor), not
    produce silently wrong parameter gradients."""
    ref, g, mem, _ = _mk(4, 16, 8, 16, 16, ("temporal", [1, 2], "forward"), False)
    obs = torch.rand(6, 4, 8, device=DEV)
    hidden, outs = None, []
    for t in range(6):
        mx, hidden = mem(obs[t], hidden)
        outs.append(mx)
    assert _rows_taken(mem)
    with torch.no_grad():
        outs[3].mul_(2.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        torch.stack(outs).sum().backward()


def test_rows_module_with_hooks_and_deepcopy():
    """Hooks registered on the module (or globally) are honoured: the unchecked C++ entry declines and the
    call goes through torch.nn.Module.__call__.  A module that has already run can be deep-copied (its
    runtime caches are rebuilt) and the copy computes the same thing."""
    import copy
    ref, g, mem, _ = _mk(3, 16, 8, 16, 16, ("temporal", [1, 2], "forward"), True)
    obs = torch.rand(8, 3, 8, device=DEV)
    seen = []
    hidden = None
    for t in range(4):
        mx, hidden = mem(obs[t], hidden)
    h = mem.register_forward_hook(lambda m, a, out: seen.append(a[0].shape))
    for t in range(4, 8):
        mx, hidden = mem(obs[t], hidden)
    h.remove()
    assert len(seen) == 4
    mem2 = copy.deepcopy(mem)
    assert mem2._fast is None and mem2.rows_steps() == 0
    outs = []
    for m in (mem, mem2):
        hid, o = None, []
        with torch.no_grad():
            for t in range(8):
                mx, hid = m(obs[t], hid)
                o.append(mx)
        outs.append(torch.stack(o))
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("mode", [True, "steps"])
@pytest.mark.parametrize("sel,N,T", [(("temporal", [1, 2, 4], "forward"), 16, 40), (("dense",), 12, 20),
                                     (("temporal", [1, 3], "both"), 128, 24)])
def test_rows_obs_gradient_chain_vs_oracle(mode, sel, N, T):
    """Observations (and the initial node matrix) with gradient on the live-row kernels, both backward forms:
    the chain's one node with the time-parallel launch (rows_dx=True) and one node per step ("steps") - through
    the overflow, staggered starts, a loss on part of the steps, a gradient handed to a returned node matrix,
    and torch.autograd.grad w.r.t. the observations only."""
    B, F, H = 5, 8, 16
    torch.manual_seed(N * 7 + T)
    ref, g, mem, osel = _mk(B, N, F, H, H, sel, False)
    mem.rows_dx = mode
    obs = torch.rand(T, B, F)
    count0 = torch.randint(0, N + 1, (B,))
    nodes0 = torch.rand(B, N, F) * (torch.arange(N)[None, :, None] < count0[:, None, None])
    adj0 = torch.zeros(B, N, N)
    for b in range(B):
        for i in range(1, int(count0[b])):
            adj0[b, i, i - 1] = 1.0
    w = torch.rand(T, B, H)
    w[T // 3] = 0                                            # a step without a gradient
    wn = torch.rand(B, N, F)

    def loss_of(step, x, n0, dev):
        hid, outs, mid = (n0, adj0.to(dev), torch.zeros(0, device=dev), count0.to(dev)), [], None
        for t in range(T):
            mx, hid = step(x[t], hid)
            outs.append(mx)
            if t == T // 2:
                mid = hid[0]
        keep = [t for t in range(T) if t != T // 3]
        loss = sum((outs[t] * w[t].to(dev)).sum() for t in keep) + (mid * wn.to(dev)).sum()
        return loss, torch.stack(outs)

    xo, no = obs.clone().requires_grad_(True), nodes0.clone().requires_grad_(True)
    lo, out_o = loss_of(lambda x, h: od.dense_step(x, h, ref, graph_size=N, edge_selectors=osel), xo, no, "cpu")
    lo.backward()
    xd, nd = obs.to(DEV).requires_grad_(True), nodes0.to(DEV).requires_grad_(True)
    ld, out_d = loss_of(mem, xd, nd, DEV)
    assert _rows_taken(mem) and mem.rows_steps() == T
    (gx_only,) = torch.autograd.grad(ld, [xd], retain_graph=True)
    assert all(p.grad is None for p in g.parameters())
    ld.backward()
    mem.check_flags()
    # (the forward is pinned with the float64 bound in test_rows_path_vs_oracle; DenseEdge sums up to N terms)
    torch.testing.assert_close(out_d.detach().cpu(), out_o.detach(), rtol=1e-5, atol=5e-6)
    scale = float(xo.grad.abs().max())
    torch.testing.assert_close(xd.grad.cpu(), xo.grad, rtol=1e-5, atol=1e-5 * scale)
    torch.testing.assert_close(gx_only.cpu(), xo.grad, rtol=1e-5, atol=1e-5 * scale)
    torch.testing.assert_close(nd.grad.cpu(), no.grad, rtol=1e-5, atol=1e-5 * float(no.grad.abs().max()))
    for (k, pc), (_, pd) in zip(ref.named_parameters(), g.named_parameters()):
        torch.testing.assert_close(pd.grad.cpu(), pc.grad, rtol=1e-5, atol=1e-5 * float(pc.grad.abs().max()), msg=k)


def test_rows_obs_gradient_feedback_and_stacked_memories():
    """(a) A policy that feeds belief t-1 into observation t: a single chain node would sit on a cycle - the
    module notices (the producer's topological number is above the chain's) and continues with one node per
    step.  (b) Two memories stacked (the second reads the first's beliefs): both keep the one-node form."""
    B, N, F, H, T = 4, 16, 8, 8, 14
    torch.manual_seed(3)
    ref, g, mem, osel = _mk(B, N, F, H, H, ("temporal", [1, 2], "forward"), False)
    lin_o = torch.nn.Linear(H, F)
    lin_d = torch.nn.Linear(H, F).to(DEV)
    lin_d.load_state_dict(lin_o.state_dict())
    obs = torch.rand(T, B, F)

    def feedback(step, lin, x, dev):
        hid, outs, prev = None, [], torch.zeros(B, H, device=dev)
        for t in range(T):
            mx, hid = step(x[t] + torch.tanh(lin(prev)), hid)
            outs.append(mx)
            prev = mx
        return torch.stack(outs)

    xo = obs.clone().requires_grad_(True)
    out_o = feedback(lambda x, h: od.dense_step(x, h, ref, graph_size=N, edge_selectors=osel), lin_o, xo, "cpu")
    out_o.sum().backward()
    xd = obs.to(DEV).requires_grad_(True)
    out_d = feedback(mem, lin_d, xd, DEV)
    assert mem.rows_steps() == T
    out_d.sum().backward()
    torch.testing.assert_close(out_d.detach().cpu(), out_o.detach(), rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(xd.grad.cpu(), xo.grad, rtol=1e-5, atol=1e-5 * float(xo.grad.abs().max()))
    for (k, pc), (_, pd) in list(zip(ref.named_parameters(), g.named_parameters())) + \
            list(zip(lin_o.named_parameters(), lin_d.named_parameters())):
        torch.testing.assert_close(pd.grad.cpu(), pc.grad, rtol=1e-5, atol=1e-5 * float(pc.grad.abs().max()), msg=k)

    # (b) stacked
    torch.manual_seed(4)
    ref1, g1, mem1, osel1 = _mk(B, N, F, H, F, ("temporal", [1], "forward"), False)
    ref2, g2, mem2, osel2 = _mk(B, N, F, H, H, ("dense",), False)

    def stacked(s1, s2, x):
        h1 = h2 = None
        outs = []
        for t in range(T):
            m1, h1 = s1(x[t], h1)
            m2, h2 = s2(m1, h2)
            outs.append(m2)
        return torch.stack(outs)

    xo = obs.clone().requires_grad_(True)
    out_o = stacked(lambda x, h: od.dense_step(x, h, ref1, graph_size=N, edge_selectors=osel1),
                    lambda x, h: od.dense_step(x, h, ref2, graph_size=N, edge_selectors=osel2), xo)
    (out_o ** 2).sum().backward()
    xd = obs.to(DEV).requires_grad_(True)
    out_d = stacked(mem1, mem2, xd)
    (out_d ** 2).sum().backward()
    assert mem1.rows_steps() == T and mem2.rows_steps() == T
    torch.testing.assert_close(out_d.detach().cpu(), out_o.detach(), rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(xd.grad.cpu(), xo.grad, rtol=1e-5, atol=1e-5 * float(xo.grad.abs().max()))
    for (k, pc), (_, pd) in list(zip(ref1.named_parameters(), g1.named_parameters())) + \
            list(zip(ref2.named_parameters(), g2.named_parameters())):
        torch.testing.assert_close(pd.grad.cpu(), pc.grad, rtol=1e-5, atol=1e-5 * float(pc.grad.abs().max()), msg=k)


def test_rows_obs_gradient_with_donated_state():
    """donate_state=True and observations with gradient: the one-node form needs the records only, so the
    state is still advanced in place (the returned node matrix is then a plain tensor); same beliefs,
    observation and parameter gradients as the oracle, through the overflow."""
    B, N, F, H, T = 6, 16, 8, 16, 30
    torch.manual_seed(11)
    ref, g, mem, osel = _mk(B, N, F, H, H, ("temporal", [1, 2, 4], "forward"), True)
    obs = torch.rand(T, B, F)
    w = torch.rand(T, B, H)
    xo = obs.clone().requires_grad_(True)
    out_o, _ = od.dense_rollout(xo, None, ref, graph_size=N, edge_selectors=osel)
    (out_o * w).sum().backward()
    xd = obs.to(DEV).requires_grad_(True)
    hid, outs, ptrs = None, [], set()
    for t in range(T):
        mx, hid = mem(xd[t], hid)
        outs.append(mx)
        ptrs.add(hid[0].data_ptr())
        assert not hid[0].requires_grad
    assert len(ptrs) == 1 and mem.rows_steps() == T          # advanced in place, on the live-row kernels
    out_d = torch.stack(outs)
    (out_d * w.to(DEV)).sum().backward()
    mem.check_flags()
    torch.testing.assert_close(out_d.detach().cpu(), out_o.detach(), rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(xd.grad.cpu(), xo.grad, rtol=1e-5, atol=1e-5 * float(xo.grad.abs().max()))
    for (k, pc), (_, pd) in zip(ref.named_parameters(), g.named_parameters()):
        torch.testing.assert_close(pd.grad.cpu(), pc.grad, rtol=1e-5, atol=1e-5 * float(pc.grad.abs().max()), msg=k)


@pytest.mark.parametrize("sel,B,N,F,H,T", [(("temporal", [1, 2, 4], "forward"), 5, 16, 32, 32, 40),
                                           (("temporal", [2, 5], "forward"), 4, 12, 32, 64, 20),
                                           (("temporal", [0, 1, 3], "forward"), 3, 128, 32, 32, 300),   # (T > 2N)
                                           (("temporal", [1, 2, 4], "forward"), 7, 128, 32, 32, 270),
                                           (("temporal", [3, 7], "forward"), 4, 12, 32, 32, 40),      # (N <= 2 max hop: no steady-state form)
                                           (("temporal", [1], "forward"), 6, 32, 64, 32, 30),
                                           (("temporal", [1, 2, 3, 4, 5, 6], "forward"), 2, 20, 64, 64, 26),
                                           (("temporal", [1, 2], "forward"), 3, 16, 8, 16, 20),    # (widths below 32: the padded form)
                                           (("temporal", [1, 2], "forward"), 3, 16, 8, 16, 40),    # (... and T > N: the usual steps behind)
                                           (("temporal", [1, 3], "forward"), 4, 24, 20, 48, 30)])  # (F -> 32, H -> 64 padded)
def test_rows_cached_steps_vs_oracle(sel, B, N, F, H, T):
    """A donated rollout from hidden = None whose selectors only write row cur: its first N steps are cached steps
    (rows_cached.hip: row cur alone over the chain's h1 / agg1 / node caches), and the steps behind them - every
    graph full, every step drops its oldest node (gcm.py:323-355) - the steady-state cached steps (round 4:
    k_step_rows_cached_roll - the caches as rings, the band adjacency a fixed point of the roll and left untouched,
    the node matrix rolled in place) when N > 2 max(hop), else the usual live-row ones; one backward over both
    kinds of record.  Against the oracle (state bit exact, T up to 300 > 2N) and against the same rollout with the
    cached steps switched off."""
    res = []
    obs = None
    for cached in (True, False):
        torch.manual_seed(N + T)
        ref, g, mem, osel = _mk(B, N, F, H, H, sel, True)
        mem.rows_cached_steps = cached
        obs = torch.rand(T, B, F)
        w = torch.rand(T, B, H)
        hid, outs = None, []
        for t in range(T):
            mx, hid = mem(obs[t].to(DEV), hid)
            outs.append(mx)
        takes = F <= 64 and H <= 64                  # (any widths up to 64: padded to 32 / 64 inside)
        steady = N > 2 * max(sel[1]) and F in (32, 64) and H in (32, 64)   # (the steady-state form: see above)
        assert mem.rows_steps() == T
        assert mem.rows_cached_steps_taken() == ((T if steady else min(T, N)) if cached and takes else 0)
        assert mem.rows_rolled_steps_taken() == (max(0, T - N) if cached and takes and steady else 0)
        if cached:   # selectors that write older rows keep the usual step
            _, _, mem_b, _ = _mk(B, N, F, H, H, ("temporal", [1], "both"), True)
            mem_b(obs[0].to(DEV), None)
            assert mem_b.rows_cached_steps_taken() == 0
        out = torch.stack(outs)
        (out * w.to(DEV)).sum().backward()
        mem.check_flags()
        res.append((out.detach().cpu(), [t.cpu() for t in (hid[0], hid[1], hid[3])],
                    {k: p.grad.cpu().clone() for k, p in g.named_parameters()}, ref, osel, w))
    ref, osel, w = res[0][3], res[0][4], res[0][5]
    out32, hid32, bounds, (out64, out_atol) = _fp64_rollout_bounds(ref, obs, None, w, lambda: osel, N)
    for out, state, grads, *_ in res:
        assert torch.equal(state[0], hid32[0]) and torch.equal(state[1], hid32[1]) and torch.equal(state[2], hid32[3])
        assert float((out.double() - out64).abs().max()) <= out_atol
        for k, gd in grads.items():
            g64, atol = bounds[k]
            assert float((gd.double() - g64).abs().max()) <= atol, k


@pytest.mark.parametrize("sel,B,N,F,H1,H2,T", [(("dense", None, None), 5, 16, 32, 32, 32, 40),
                                               (("dense", None, None), 3, 128, 32, 32, 32, 140),
                                               (("temporal", [1, 2, 4], "both"), 4, 32, 32, 32, 32, 50),
                                               (("temporal", [1, 3], "backward"), 4, 24, 64, 32, 48, 30),
                                               (("temporal", [0, 2], "both"), 3, 20, 32, 64, 16, 26),
                                               (("dense", None, None), 2, 64, 64, 64, 64, 70),
                                               (("temporal", [1, 2, 4], "both"), 300, 128, 32, 32, 32, 20),
                                               (("dense", None, None), 3, 32, 32, 32, 32, 150),      # (> 4 N steps)
                                               (("temporal", [0, 3, 5], "both"), 4, 20, 32, 32, 32, 75),
                                               (("temporal", [2, 7], "backward"), 2, 8, 64, 64, 40, 30),
                                               (("dense", None, None), 3, 16, 8, 16, 16, 20)])   # (widths without the form)
def test_rows_colcache_steps_vs_oracle(sel, B, N, F, H1, H2, T):
    """A donated rollout from hidden = None whose selectors also write COLUMN cur of the adjacency - DenseEdge
    (dense.py:16-21), TemporalBackedge "backward" / "both" (temporal.py:82-87): its first N steps are column-write
    cached steps (rows_colcache.hip: rank-1 updates of the chain's layer-1 aggregate, one matrix-core product of the
    live rows), and so are the steps behind them (the caches as rings: a full graph drops its oldest node every step,
    the rows that held it as a source get the opposite rank-1 correction); one backward over the records.  Against the
    oracle (state bit exact, beliefs and gradients inside the float64 bound) and against the same rollout with the
    form switched off."""
    res = []
    obs = None
    takes = F in (32, 64) and H1 in (32, 64)
    for on in (True, False):
        torch.manual_seed(N + T)
        ref, g, mem, osel = _mk(B, N, F, H1, H2, sel, True)
        mem.rows_col_cache = on
        obs = torch.rand(T, B, F) - 0.5
        w = torch.rand(T, B, H2)
        hid, outs = None, []
        for t in range(T):
            mx, hid = mem(obs[t].to(DEV), hid)
            outs.append(mx)
        assert mem.rows_steps() == T
        assert mem.rows_col_steps_taken() == (T if on and takes else 0)      # (past N steps: the ring form)
        assert mem.rows_cached_steps_taken() == 0
        out = torch.stack(outs)
        (out * w.to(DEV)).sum().backward()
        mem.check_flags()
        res.append((out.detach().cpu(), [t.cpu() for t in (hid[0], hid[1], hid[3])],
                    {k: p.grad.cpu().clone() for k, p in g.named_parameters()}, ref, osel, w))
    ref, osel, w = res[0][3], res[0][4], res[0][5]
    if B > 16:      # (the oracle on a slice of the batch: graphs are independent)
        pick = [0, B // 2, B - 1]
        wz = torch.zeros_like(w)
        wz[:, pick] = w[:, pick]
        res2 = []
        for on in (True, False):
            torch.manual_seed(N + T)
            _, g, mem, _ = _mk(B, N, F, H1, H2, sel, True)
            mem.rows_col_cache = on
            hid, outs = None, []
            for t in range(T):
                mx, hid = mem(obs[t].to(DEV), hid)
                outs.append(mx)
            (torch.stack(outs) * wz.to(DEV)).sum().backward()
            res2.append((torch.stack(outs).detach().cpu()[:, pick], [t.cpu()[pick] for t in (hid[0], hid[1], hid[3])],
                         {k: p.grad.cpu().clone() for k, p in g.named_parameters()}))
        res, obs, w = res2, obs[:, pick], w[:, pick]
    out32, hid32, bounds, (out64, out_atol) = _fp64_rollout_bounds(ref, obs, None, w, lambda: osel, N)
    for out, state, grads, *_ in res:
        assert torch.equal(state[0], hid32[0]) and torch.equal(state[1], hid32[1]) and torch.equal(state[2], hid32[3])
        assert float((out.double() - out64).abs().max()) <= out_atol
        for k, gd in grads.items():
            g64, atol = bounds[k]
            assert float((gd.double() - g64).abs().max()) <= atol, k


@pytest.mark.parametrize("sel,B,N,F,H,T", [(("dense", None, None), 4, 16, 32, 32, 40),
                                           (("temporal", [1, 3], "both"), 3, 32, 64, 32, 50),
                                           (("dense", None, None), 260, 128, 32, 32, 12)])
def test_rows_colcache_functional_state(sel, B, N, F, H, T):
    """The reference's default - functional state: every step returns fresh nodes / adj / num_nodes and leaves its
    inputs untouched (gcm.py:262,278,286) - on the column-write cached steps: the new state comes from extra workgroups of
    the same launch.  Every intermediate state is kept and compared with the oracle's (bit exact), beliefs and
    gradients inside the float64 bound; a second rollout of the same module starts a new chain; a call that branches
    off an OLDER state leaves the cached run."""
    torch.manual_seed(N + T)
    ref, g, mem, osel = _mk(B, N, F, H, H, sel, False)
    obs = torch.rand(T, B, F) - 0.5
    w = torch.rand(T, B, H)
    for rep in range(2):
        g.zero_grad(set_to_none=True)
        hid, outs, states = None, [], []
        for t in range(T):
            mx, hid = mem(obs[t].to(DEV), hid)
            outs.append(mx)
            states.append(hid)
        assert mem.rows_col_steps_taken() == T
        out = torch.stack(outs)
        (out * w.to(DEV)).sum().backward()
        mem.check_flags()
    assert len({h[0].data_ptr() for h in states}) == T       # fresh tensors every step
    pick = list(range(B)) if B <= 16 else [0, B // 2, B - 1]
    if B > 16:
        g.zero_grad(set_to_none=True)
        wz = torch.zeros_like(w)
        wz[:, pick] = w[:, pick]
        hid, outs = None, []
        for t in range(T):
            mx, hid = mem(obs[t].to(DEV), hid)
            outs.append(mx)
        (torch.stack(outs) * wz.to(DEV)).sum().backward()
    out32, hid32, bounds, (out64, out_atol) = _fp64_rollout_bounds(ref, obs[:, pick], None, w[:, pick], lambda: osel, N)
    assert float((out.detach().cpu()[:, pick].double() - out64).abs().max()) <= out_atol
    for k, p in g.named_parameters():
        g64, atol = bounds[k]
        assert float((p.grad.cpu().double() - g64).abs().max()) <= atol, k
    # the state after every step against the oracle's, step by step
    h_o = None
    for t in range(T):
        _, h_o = od.dense_rollout(obs[t:t + 1, pick], h_o, ref, graph_size=N, edge_selectors=osel)
        h_o = tuple(x.detach() for x in h_o)
        n_d, a_d, _, c_d = states[t]
        assert torch.equal(n_d.cpu()[pick], h_o[0]) and torch.equal(a_d.cpu()[pick], h_o[1]), t
        assert torch.equal(c_d.cpu()[pick], h_o[3]), t
    # branching off an older state: the general kernel, the same values as the chain had at that point
    if T > 5:
        with torch.no_grad():
            mx_b, _ = mem(obs[4].to(DEV), states[3])
        assert mem.rows_col_steps_taken() == 0      # (a new chain, armed on a caller's state: no cached form)
        torch.testing.assert_close(mx_b.cpu(), out.detach().cpu()[4], rtol=1e-5, atol=5e-6)


@pytest.mark.parametrize("kind", ["dense", "both"])
def test_rows_colcache_c_abi_four_and_eight_waves(kind):
    """gcm_dense_rows_step_colcache through the raw C ABI at F = H = 32, where two kernels exist - eight waves per
    graph on 16-slot tiles (the default) and four waves on 32-slot tiles (GCM_STEP_FOUR_WAVES in has_bias: what the
    wider shapes run) - on the same chain from empty graphs, 2.2 N steps (the ring form behind step N), against the
    general live-row kernel gcm_dense_rows_step_fwd: states bit for bit, beliefs 1e-5, the records' header and live rows
    (as sets: the general kernel lists them in another order); and the functional entry against the donated one."""
    from gcm import _hip
    lib = _hip.lib()
    B, N, F, H, T = 5, 32, 32, 32, 70
    torch.manual_seed(3)
    P = lib.gcm_dense_gnn2_param_count(F, H, H)
    params = (torch.randn(P) * 0.2).to(DEV)
    obs = (torch.rand(T, B, F) - 0.5).to(DEV)
    if kind == "dense":
        d = _hip.SelectorDesc(kind=_hip.SEL_DENSE)
    else:
        d = _hip.SelectorDesc(kind=_hip.SEL_TEMPORAL, n_hops=3, direction=_hip.DIR["both"])
        d.hops[0], d.hops[1], d.hops[2] = 0, 1, 3
    arr = (_hip.SelectorDesc * 1)(d)
    flags = torch.zeros(1, dtype=torch.int32, device=DEV)
    st, p = _hip.stream(), _hip.ptr
    lay = (ctypes.c_size_t * 6)()
    assert lib.gcm_dense_rows_layout(B, N, F, H, H, ctypes.addressof(lay)) == 0
    FOUR = 256   # GCM_STEP_FOUR_WAVES

    def chain(form):
        nodes, adj = torch.zeros(B, N, F, device=DEV), torch.zeros(B, N, N, device=DEV)
        count = torch.zeros(B, dtype=torch.int64, device=DEV)
        cA, cR = torch.full((B, N, F), float("nan"), device=DEV), torch.full((B, N, H), float("nan"), device=DEV)
        out = []
        for t in range(T):
            saved = torch.zeros(lay[0], device=DEV)
            if form == "general":
                rc = lib.gcm_dense_rows_step_fwd(p(obs[t]), p(nodes), p(adj), p(count), p(nodes), p(adj), p(count), None,
                                                 ctypes.addressof(arr), 1, p(params), 3, 1, 1, p(saved), p(saved),
                                                 p(flags), B, N, F, H, H, st)
            elif form == "functional":
                n2, a2, c2 = torch.empty_like(nodes), torch.empty_like(adj), torch.empty_like(count)
                rc = lib.gcm_dense_rows_step_colcache_functional(p(obs[t]), p(nodes), p(adj), p(count), p(n2), p(a2), p(c2),
                                                                 ctypes.addressof(arr), 1, p(params), 3, 1, 1, p(cA), p(cR),
                                                                 p(saved), 1, t, p(flags), B, N, F, H, H, st)
                nodes, adj, count = n2, a2, c2
            else:
                rc = lib.gcm_dense_rows_step_colcache(p(obs[t]), p(nodes), p(adj), p(count), ctypes.addressof(arr), 1,
                                                      p(params), 3 | (FOUR if form == "four" else 0), 1, 1, p(cA), p(cR),
                                                      p(saved), 1, t, p(flags), B, N, F, H, H, st)
            assert rc == 0, (form, t, rc)
            out.append((saved, nodes.clone(), adj.clone(), count.clone()))
        torch.cuda.synchronize()
        return out

    ref = chain("general")
    for form in ("eight", "four", "functional"):
        got = chain(form)
        for t in range(T):
            (sv, n, a, c), (sv0, n0, a0, c0) = got[t], ref[t]
            assert torch.equal(n, n0) and torch.equal(a, a0) and torch.equal(c, c0), (form, t)
            # (beliefs: up to N-term sums in another order than the general kernel's - 1e-5 of the pre-activation's
            #  scale, an absolute tolerance on tanh's output; the oracle tests bound the same through float64)
            err = float((sv[:B * H] - sv0[:B * H]).abs().max())
            assert err <= (4e-5 if kind == "dense" else 5e-6), (form, t, err)
            hdr = sv[lay[2]:lay[2] + 4 * B].view(torch.int32).view(B, 4).cpu()
            hdr0 = sv0[lay[2]:lay[2] + 4 * B].view(torch.int32).view(B, 4).cpu()
            assert torch.equal(hdr[:, 0], hdr0[:, 0]) and torch.equal(hdr[:, 2], hdr0[:, 2]), (form, t)
            # the live rows' x sections as multisets of row sums (their order differs between the kernels)
            rw, L = int(lay[5]), int(hdr[0, 0])
            rows = sv[lay[4]:lay[4] + B * N * rw].view(B, N, rw)[:, :L, H + F:].sum(-1).sort(dim=1).values
            rows0 = sv0[lay[4]:lay[4] + B * N * rw].view(B, N, rw)[:, :L, H + F:].sum(-1).sort(dim=1).values
            torch.testing.assert_close(rows, rows0, rtol=0, atol=0)
    assert int(flags.item()) & ~1 == 0      # (GCM_FLAG_WRAPPED past N steps, nothing else)


def test_rows_colcache_leaves_an_edited_chain():
    """A caller that edits the donated state in place between two steps (zeroing the graphs of finished episodes) ends
    the column-write cached run: the next step is the general live-row kernel on the state as it is."""
    torch.manual_seed(5)
    B, N, F, H = 4, 32, 32, 32
    ref, g, mem, osel = _mk(B, N, F, H, H, ("dense", None, None), True)
    obs = torch.rand(12, B, F)
    hid = None
    with torch.no_grad():
        for t in range(6):
            _, hid = mem(obs[t].to(DEV), hid)
        assert mem.rows_col_steps_taken() == 6
        hid[0][1].zero_(); hid[1][1].zero_(); hid[3][1] = 0          # graph 1 starts a new episode
        outs = []
        for t in range(6, 12):
            mx, hid = mem(obs[t].to(DEV), hid)
            outs.append(mx)
        assert mem.rows_col_steps_taken() == 6 and mem.rows_steps() == 12
    mem.check_flags()
    h_o = None
    out_a, h_o = od.dense_rollout(obs[:6], None, ref, graph_size=N, edge_selectors=osel)
    h_o = tuple(t.clone() for t in h_o)
    h_o[0][1].zero_(); h_o[1][1].zero_(); h_o[3][1] = 0
    out_b, h_o = od.dense_rollout(obs[6:], h_o, ref, graph_size=N, edge_selectors=osel)
    torch.testing.assert_close(torch.stack(outs).cpu(), out_b.detach(), rtol=1e-5, atol=5e-6)
    assert torch.equal(hid[1].cpu(), h_o[1]) and torch.equal(hid[0].cpu(), h_o[0]) and torch.equal(hid[3].cpu(), h_o[3])


@pytest.mark.parametrize("hops,B,N,F,H,T", [([1, 2, 4], 6, 16, 32, 32, 30), ([0, 1, 3], 3, 128, 32, 32, 130),
                                            ([1, 2, 3, 4, 5, 6, 7, 8, 9, 10], 2, 24, 64, 32, 24)])
def test_rows_cached_steps_with_obs_gradient(hops, B, N, F, H, T):
    """Observations with gradient on a donated state from hidden = None: the cached steps' part of dL/dx comes from
    the caches and the hop table (gcm_dense_rows_bptt_dx_all_cached), the steps behind them - the graphs overflow -
    from their own records; beliefs, state, observation and parameter gradients against the oracle."""
    torch.manual_seed(N + T)
    ref, g, mem, osel = _mk(B, N, F, H, H, ("temporal", hops, "forward"), True)
    obs = torch.rand(T, B, F)
    w = torch.rand(T, B, H)
    w[T // 3] = 0
    xo = obs.clone().requires_grad_(True)
    out_o, hid_o = od.dense_rollout(xo, None, ref, graph_size=N, edge_selectors=osel)
    (out_o * w).sum().backward()
    xs = [obs[t].to(DEV).requires_grad_(True) for t in range(T)]       # per-step leaves (an encoder's outputs)
    hid, outs = None, []
    for t in range(T):
        mx, hid = mem(xs[t], hid)
        outs.append(mx)
    assert mem.rows_steps() == T and mem.rows_cached_steps_taken() == min(T, N)
    out_d = torch.stack(outs)
    (out_d * w.to(DEV)).sum().backward()
    mem.check_flags()
    torch.testing.assert_close(out_d.detach().cpu(), out_o.detach(), rtol=RTOL, atol=ATOL)
    assert torch.equal(hid[1].cpu(), hid_o[1]) and torch.equal(hid[0].cpu(), hid_o[0])
    gx = torch.stack([x.grad.cpu() for x in xs])
    torch.testing.assert_close(gx, xo.grad, rtol=1e-5, atol=1e-5 * float(xo.grad.abs().max()))
    for (k, pc), (_, pd) in zip(ref.named_parameters(), g.named_parameters()):
        torch.testing.assert_close(pd.grad.cpu(), pc.grad, rtol=1e-5, atol=1e-5 * float(pc.grad.abs().max()), msg=k)


def test_euclid_chain_one_launch_per_step_equals_two():
    """A donated chain from empty graphs whose only selector is EuclideanEdge: the distance kernel and the cached
    step as ONE launch (the step is the tail of k_euclid_mfma2's first wave) against the same chain as two launches
    (gcm_edge_distance_pre, then k_step_rows_cached_img with the decision row): the same state bit for bit, the same
    beliefs and parameter gradients to fp32 summation order."""
    from gcm import nn as G
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.distance import EuclideanEdge
    B, N, F, H, T = 40, 128, 64, 32, 100
    torch.manual_seed(5)
    centres = 3 * torch.randn(6, F)
    obs = (centres[torch.arange(T) % 6][:, None, :] + 0.05 * torch.randn(T, B, F)).to(DEV)
    w = torch.linspace(0.5, 1.5, T * B * H, device=DEV).view(T, B, H)
    res = []
    for fused in (True, False):
        torch.manual_seed(1)
        g = G.Sequential("x, adj, weights, B, N", [
            (G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
            (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(DEV)
        mem = DenseGCM(g, edge_selectors=EuclideanEdge(3.0), graph_size=N, donate_state=True)
        mem.rows_one_launch_distance = fused      # (a per-call flag of the C ABI: GCM_STEP_TWO_LAUNCH)
        hidden, outs = None, []
        for t in range(T):
            mx, hidden = mem(obs[t], hidden)
            outs.append(mx)
        assert mem.rows_cached_steps_taken() == T
        assert mem.rows_cached_launches_per_step(B) == (1 if fused else 2)
        out = torch.stack(outs)
        (out * w).sum().backward()
        mem.check_flags()
        res.append((out.detach(), [h.clone() for h in hidden], {k: p.grad.clone() for k, p in g.named_parameters()}))
    a, b = res
    assert float(a[1][1].sum()) > 0
    for x, y in zip(a[1], b[1]):
        assert torch.equal(x, y)
    torch.testing.assert_close(a[0], b[0], rtol=1e-5, atol=1e-5)
    for k in a[2]:
        scale = float(b[2][k].abs().max()) + 1e-12
        torch.testing.assert_close(a[2][k], b[2][k], rtol=1e-5, atol=2e-5 * scale, msg=k)


@pytest.mark.parametrize("sel,F", [(("temporal", [1, 2, 4], "forward"), 32), (("temporal", [1, 3], "both"), 32)])
def test_rows_donated_state_reset_in_place_mid_chain(sel, F):
    """A caller that owns a donated state may edit it between two steps - the usual RL idiom at an episode's end:
    `num_nodes[done] = 0; nodes[done] = 0; adj[done] = 0`.  A cached step reads its per-chain caches and the host's
    step count, not the state (rows_cached.hip): the host path watches the state's version counters and hands the
    rest of such a chain to the kernel that reads the state.  Against the oracle with the same edit."""
    B, N, H, T, t_reset, done = 6, 16, 32, 30, 9, [1, 4]
    torch.manual_seed(77)
    ref, g, mem, osel = _mk(B, N, F, H, H, sel, True)
    obs = torch.rand(T, B, F)
    w = torch.rand(T, B, H)
    # oracle: two segments around the edit
    out_a, hid_o = od.dense_rollout(obs[:t_reset], None, ref, graph_size=N, edge_selectors=osel)
    nodes_o, adj_o, w_o, count_o = (t.clone() for t in hid_o)
    count_o[done] = 0
    nodes_o[done] = 0
    adj_o[done] = 0
    out_b, hid_o = od.dense_rollout(obs[t_reset:], (nodes_o, adj_o, w_o, count_o), ref, graph_size=N,
                                    edge_selectors=osel)
    out_o = torch.cat([out_a, out_b])
    (out_o * w).sum().backward()
    hid, outs = None, []
    for t in range(T):
        if t == t_reset:
            hid[3][done] = 0
            hid[0][done] = 0
            hid[1][done] = 0
        mx, hid = mem(obs[t].to(DEV), hid)
        outs.append(mx)
    assert mem.rows_steps() == T
    if sel[2] == "forward":      # cached steps up to the edit, the state-reading kernel behind it
        assert mem.rows_cached_steps_taken() == t_reset
    out_d = torch.stack(outs)
    (out_d * w.to(DEV)).sum().backward()
    mem.check_flags()
    assert torch.equal(hid[0].cpu(), hid_o[0]) and torch.equal(hid[1].cpu(), hid_o[1])
    assert torch.equal(hid[3].cpu(), hid_o[3])
    torch.testing.assert_close(out_d.detach().cpu(), out_o.detach(), rtol=RTOL, atol=2e-6)
    for (k, pc), (_, pd) in zip(ref.named_parameters(), g.named_parameters()):
        scale = float(pc.grad.abs().max()) + 1e-12
        torch.testing.assert_close(pd.grad.cpu(), pc.grad, rtol=1e-5, atol=2e-5 * scale, msg=k)


@pytest.mark.parametrize("hops,B,N,F,H1,H2,T", [([1, 2, 4], 37, 16, 32, 32, 32, 40), ([0, 1, 3], 5, 64, 64, 64, 16, 131),
                                                ([2, 5], 64, 12, 32, 64, 64, 11), ([1], 33, 32, 64, 32, 8, 70),
                                                ([3, 7], 4, 12, 32, 32, 32, 40)])
def test_rollout_time_parallel_vs_oracle(hops, B, N, F, H1, H2, T):
    """rollout() from hidden = None with forward temporal hops (csrc/rollout_tp.hip) against the oracle's per-step loop:
    ragged batch sizes (partial 32-graph tiles), F / H1 in {32, 64}, narrow H2, a self loop (hop 0), T below and beyond
    the graph size; state bit exact, beliefs and parameter gradients inside the float64 bound.  N <= 2 max(hop) with
    T > N has no time-parallel form (a live row loses a source to the roll): the persistent kernel runs."""
    torch.manual_seed(N + T)
    ref, g, mem, osel = _mk(B, N, F, H1, H2, ("temporal", hops, "forward"), False)
    obs = torch.rand(T, B, F)
    w = torch.rand(T, B, H2)
    out, hid = mem.rollout(obs.to(DEV))
    cfg = mem._cfg_last[3] if mem._cfg_last else None      # (shapes the fused plan does not take: the layered loop)
    tp = cfg is not None and cfg.rows_ok and not (T > N and N <= 2 * max(hops))
    assert (out.grad_fn.name() == "GcmRowsRollout") == tp
    assert tp or hops == [3, 7]
    (out * w.to(DEV)).sum().backward()
    mem.check_flags()
    out32, hid32, bounds, (out64, out_atol) = _fp64_rollout_bounds(ref, obs, None, w, lambda: osel, N)
    assert torch.equal(hid[0].cpu(), hid32[0]) and torch.equal(hid[1].cpu(), hid32[1])
    assert torch.equal(hid[3].cpu(), hid32[3])
    assert float((out.detach().cpu().double() - out64).abs().max()) <= out_atol
    for k, p in g.named_parameters():
        g64, atol = bounds[k]
        assert float((p.grad.cpu().double() - g64).abs().max()) <= atol, k


@pytest.mark.parametrize("H2,hops", [(64, [1, 2, 4]), (32, [1, 2, 4, 8]), (16, [3]), (48, [1, 2, 3, 5, 9])])
def test_rows_cached_step_variants_of_the_exact_widths(H2, hops):
    """F = H1 = 32 with the output width around 32 and the hop count around four: the cached temporal step has a form
    per case - k split over the half-waves (H2 <= 32: lanes 32 - 63 hold no output) or the full-wave products
    (H2 > 32); the source rows straight from four hop slots or, with more hops, from the mask walk - and so has the
    steady-state step behind it.  T > N so both kernels run; against the oracle (state bit exact, beliefs and gradients
    inside the float64 bound)."""
    B, N, F, H1, T = 5, 40, 32, 32, 70
    torch.manual_seed(H2 + len(hops))
    ref, g, mem, osel = _mk(B, N, F, H1, H2, ("temporal", hops, "forward"), True)
    obs = torch.rand(T, B, F)
    w = torch.rand(T, B, H2)
    hid, outs = None, []
    for t in range(T):
        mx, hid = mem(obs[t].to(DEV), hid)
        outs.append(mx)
    steady = N > 2 * max(hops)
    assert mem.rows_cached_steps_taken() == (T if steady else N)
    out = torch.stack(outs)
    (out * w.to(DEV)).sum().backward()
    mem.check_flags()
    out32, hid32, bounds, (out64, out_atol) = _fp64_rollout_bounds(ref, obs, None, w, lambda: osel, N)
    assert torch.equal(hid[0].cpu(), hid32[0]) and torch.equal(hid[1].cpu(), hid32[1]) and torch.equal(hid[3].cpu(), hid32[3])
    assert float((out.detach().cpu().double() - out64).abs().max()) <= out_atol
    for k, p in g.named_parameters():
        g64, atol = bounds[k]
        assert float((p.grad.cpu().double() - g64).abs().max()) <= atol, k


@pytest.mark.parametrize("sel,F,H", [(("temporal", [1, 2, 4], "forward"), 32, 32), (("temporal", [2, 5], "forward"), 20, 48)])
def test_rows_chain_does_not_depend_on_uninitialised_memory(sel, F, H):
    """The cached-step chain allocates its caches and records without a zero fill.  With torch filling every
    uninitialised allocation with NaN (deterministic mode's fill_uninitialized_memory) a donated rollout past graph_size
    - cached steps, then the steady-state / live-row steps - and its backward must still match the oracle."""
    B, N, T = 4, 24, 40
    prev_det = torch.are_deterministic_algorithms_enabled()
    prev_fill = torch.utils.deterministic.fill_uninitialized_memory
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
    try:
        torch.manual_seed(5)
        ref, g, mem, osel = _mk(B, N, F, H, H, sel, True)
        obs = torch.rand(T, B, F)
        w = torch.rand(T, B, H)
        hid, outs = None, []
        for t in range(T):
            mx, hid = mem(obs[t].to(DEV), hid)
            outs.append(mx)
        out = torch.stack(outs)
        (out * w.to(DEV)).sum().backward()
        mem.check_flags()
    finally:
        torch.utils.deterministic.fill_uninitialized_memory = prev_fill
        torch.use_deterministic_algorithms(prev_det)
    out32, hid32, bounds, (out64, out_atol) = _fp64_rollout_bounds(ref, obs, None, w, lambda: osel, N)
    assert torch.equal(hid[0].cpu(), hid32[0]) and torch.equal(hid[1].cpu(), hid32[1])
    assert float((out.detach().cpu().double() - out64).abs().max()) <= out_atol
    for k, p in g.named_parameters():
        g64, atol = bounds[k]
        assert float((p.grad.cpu().double() - g64).abs().max()) <= atol, k


def test_rows_per_item_backward_behind_the_ab_switch_matches_the_oracle():
    """The per-item backward over cached records (k_bptt_rows<.., 3>) stays behind GCM_STEP_FOUR_WAVES in has_bias for the
    A/B of tools/ab_cfg2.sh - at F = 32 / 64, H1 = 32 the default run takes the per-graph kernel (k_bptt_cached_graph).
    The module reads GCM_BPTT_PER_ITEM once per process: the cached-step oracle comparisons of this file run once more in a
    child that sets it."""
    import subprocess
    import sys
    env = dict(os.environ, GCM_BPTT_PER_ITEM="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run(
        [sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.abspath(__file__),
         "-k", "test_rows_cached_steps_vs_oracle"],
        env=env, capture_output=True, timeout=900, cwd=root)
    tail = p.stdout.decode()[-1500:]
    assert p.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
