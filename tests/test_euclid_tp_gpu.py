"""The time-parallel EuclideanEdge rollout (csrc/euclid_tp.hip): DenseGCM.rollout(obs[T,B,F]) from empty graphs with
EuclideanEdge as the only selector (edge_selectors/distance.py:18-49, gcm.py:262-321 driven by ray_gcm.py:186-209) -
every step's decisions as one causal MFMA contraction per graph, the GNN of all steps in one launch per graph."""
import pytest
import torch

from oracle import dense as od
from test_dense_gpu import DEV
from _golden import fp64_rollout_bounds

pytestmark = pytest.mark.gpu


def _mk(F, H1, H2, N, maxd, learned=False, donate=False):
    from gcm import nn as G
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.distance import EuclideanEdge
    from oracle import pyg
    ref = pyg.Sequential("x, adj, weights, B, N", [
        (pyg.DenseGraphConv(F, H1), "x, adj -> x"), torch.nn.Tanh(),
        (pyg.DenseGraphConv(H1, H2), "x, adj -> x"), torch.nn.Tanh()])
    g = G.Sequential("x, adj, weights, B, N", [
        (G.DenseGraphConv(F, H1), "x, adj -> x"), torch.nn.Tanh(),
        (G.DenseGraphConv(H1, H2), "x, adj -> x"), torch.nn.Tanh()])
    g.load_state_dict(ref.state_dict())
    g = g.to(DEV)
    sel = EuclideanEdge(maxd, learned=learned).to(DEV)
    mem = DenseGCM(g, edge_selectors=sel, graph_size=N, donate_state=donate)
    return ref, g, mem


def _clustered(T, B, F, n_c=6, seed=0):
    gen = torch.Generator().manual_seed(seed)
    centres = 3 * torch.randn(n_c, F, generator=gen)
    return centres[torch.arange(T) % n_c][:, None, :] + 0.05 * torch.randn(T, B, F, generator=gen)


@pytest.mark.parametrize("B,N,F,H1,H2,T,learned", [
    (40, 128, 64, 32, 32, 100, False),      # one chunk of current rows, a partial column tile
    (33, 64, 32, 32, 16, 64, False),        # F = 32, narrow H2, T = N
    (64, 32, 64, 64, 64, 32, True),         # `learned`: the node matrix divided by dist_param (distance.py:21-22)
    (160, 128, 32, 32, 32, 37, False),      # two chunks, the second with one column tile
    (300, 16, 64, 32, 32, 16, False),       # three chunks, a small graph
    (256, 128, 64, 32, 32, 20, False),      # cfg3's batch
])
def test_euclid_rollout_time_parallel_vs_oracle(B, N, F, H1, H2, T, learned):
    """rollout() against the oracle's per-step loop (cdist + mean over the batch, threshold, two DenseGraphConv): the
    state bit exact (adjacency = every step's decisions), beliefs and parameter gradients inside the float64 bound."""
    torch.manual_seed(B + T)
    maxd = 3.0
    ref, g, mem = _mk(F, H1, H2, N, maxd, learned)
    obs = _clustered(T, B, F, seed=B)
    w = torch.rand(T, B, H2)
    out, hid = mem.rollout(obs.to(DEV))
    assert out.grad_fn.name() == "GcmRowsRollout"
    (out * w.to(DEV)).sum().backward()
    mem.check_flags()

    def osel():
        return od.EuclideanEdge(maxd, dist_param=torch.tensor([maxd]) if learned else None)
    out32, hid32, bounds, (out64, out_atol) = fp64_rollout_bounds(ref, obs, None, w, osel, N)
    assert float(hid32[1].sum()) > 0
    assert torch.equal(hid[1].cpu(), hid32[1]), "every step's decisions"
    assert torch.equal(hid[0].cpu(), hid32[0]) and torch.equal(hid[3].cpu(), hid32[3])
    assert float((out.detach().cpu().double() - out64).abs().max()) <= out_atol
    for k, p in g.named_parameters():
        g64, atol = bounds[k]
        assert float((p.grad.cpu().double() - g64).abs().max()) <= atol, k


@pytest.mark.parametrize("B,N,F,T", [(96, 128, 64, 90), (256, 64, 32, 64)])
def test_euclid_rollout_time_parallel_equals_single_steps_at_the_threshold(B, N, F, T):
    """Uniform random observations with the threshold at the median distance - thousands of candidates within 1e-4 of
    it: the time-parallel contraction performs the per-step kernel's arithmetic value for value, so the decisions
    (the adjacency) equal those of T single steps bit for bit; beliefs / gradients to summation order."""
    from gcm.edge_selectors.distance import EuclideanEdge
    torch.manual_seed(3)
    obs = torch.rand(T, B, F, device=DEV)
    # the mean distance between uniform points: threshold there
    probe = EuclideanEdge(1e9)
    nodes = obs[: min(T, N)].transpose(0, 1).contiguous()
    nodes = torch.cat([nodes, torch.zeros(B, N - nodes.shape[1], F, device=DEV)], 1) if nodes.shape[1] < N else nodes
    d = probe.distances(nodes, torch.full((B,), min(T, N) - 1, dtype=torch.long, device=DEV))
    maxd = float(d[:, : min(T, N) - 1].median())
    res = []
    for tp in (True, False):
        torch.manual_seed(7)                  # (the same parameters in both modules)
        ref, g, mem = _mk(F, 32, 32, N, maxd, donate=not tp)
        mem.rollout_time_parallel = tp
        if tp:
            out, hid = mem.rollout(obs)
            assert out.grad_fn.name() == "GcmRowsRollout"
        else:
            hid, outs = None, []
            for t in range(T):
                mx, hid = mem(obs[t], hid)
                outs.append(mx)
            out = torch.stack(outs)
        out.sum().backward()
        mem.check_flags()
        res.append((out.detach(), [h.clone() for h in hid], {k: p.grad.clone() for k, p in g.named_parameters()}))
    a, b = res
    frac = float(a[1][1].sum()) / (B * T * (T - 1) / 2)
    assert 0.2 < frac < 0.8, frac
    assert torch.equal(a[1][1], b[1][1]) and torch.equal(a[1][0], b[1][0]) and torch.equal(a[1][3], b[1][3])
    # (aggregates of up to ~T/2 rows of magnitude ~1: two fp32 summation orders differ by a few 1e-5 - DESIGN 4)
    torch.testing.assert_close(a[0], b[0], rtol=1e-5, atol=5e-5)
    for k in a[2]:
        scale = float(b[2][k].abs().max()) + 1e-12
        torch.testing.assert_close(a[2][k], b[2][k], rtol=1e-5, atol=5e-5 * scale, msg=k)


@pytest.mark.parametrize("B,N,F,T,learned", [(48, 32, 64, 85, False), (130, 24, 32, 60, True)])
def test_euclid_tp_decisions_beyond_graph_size(B, N, F, T, learned):
    """gcm_euclid_rollout_tp_decide through the C ABI for T > N (the ring: node n in slot n mod N, the slot of the node
    the overflow roll drops is dead) against the per-step path, step by step: row cur of the adjacency after step t
    holds that step's decisions in image coordinates (image column j = node t - cur_t + j)."""
    from gcm import _hip
    from gcm.edge_selectors.distance import EuclideanEdge
    from gcm.gcm import DenseGCM
    lib = _hip.lib()
    torch.manual_seed(11)
    obs = torch.rand(T, B, F, device=DEV)
    maxd = 0.41 * (F / 6.0) ** 0.5 * 2.45     # near the mean distance of uniform points in [0,1]^F
    sel = EuclideanEdge(maxd, learned=learned).to(DEV)    # learned: nodes / dist_param against threshold 1
    bits = torch.empty(T, B, 4, dtype=torch.int32, device=DEV)
    rc = lib.gcm_euclid_rollout_tp_decide(obs.data_ptr(), float(sel.max_distance),
                                          sel.dist_param.data_ptr() if learned else None, bits.data_ptr(), T, B, N, F,
                                          _hip.stream())
    assert rc == 0
    torch.cuda.synchronize()
    bits = bits.cpu().numpy().astype("uint32")
    DenseGCM.did_warn = True
    _, g, mem = _mk(F, 32, 32, N, maxd)
    mem.edge_selectors = sel
    hid = None
    n_set = 0
    with torch.no_grad():
        for t in range(T):
            _, hid = mem(obs[t], hid)
            cur = min(t, N - 1)
            row = hid[1][:, cur, :].cpu()
            want = torch.zeros(B, N)
            for j in range(cur):
                n = t - cur + j                 # the node image column j holds
                s = n % N
                want[:, j] = torch.from_numpy(((bits[t, :, s >> 5] >> (s & 31)) & 1).astype("float32"))
            assert torch.equal(row, want), t
            n_set += int(want.sum())
    assert n_set > B * T        # (the threshold sits inside the distribution)


@pytest.mark.parametrize("B,N,F,H,T", [(40, 32, 64, 32, 85), (33, 16, 32, 32, 50), (96, 64, 64, 32, 150)])
def test_euclid_chain_steady_state_one_launch_vs_oracle(B, N, F, H, T):
    """A donated EuclideanEdge chain from empty graphs carried past graph_size steps: from step N on every step drops
    every graph's oldest node (gcm.py:263-271, 323-355) and runs gcm_edge_distance_step_ring - distances, the roll of the
    state in place, the live rows re-evaluated from the chain's bit image of the adjacency, ONE launch.  Against the
    oracle's per-step loop: state bit exact, beliefs and parameter gradients inside the float64 bound."""
    from gcm.gcm import DenseGCM
    DenseGCM.did_warn = True
    torch.manual_seed(B + T)
    maxd = 3.0
    ref, g, mem = _mk(F, H, H, N, maxd, donate=True)
    obs = _clustered(T, B, F, n_c=5, seed=T)
    w = torch.rand(T, B, H)
    obs_d = obs.to(DEV)
    hid, outs = None, []
    for t in range(T):
        mx, hid = mem(obs_d[t], hid)
        outs.append(mx)
    assert mem.rows_cached_steps_taken() == T and mem.rows_rolled_steps_taken() == T - N
    out = torch.stack(outs)
    (out * w.to(DEV)).sum().backward()
    mem.check_flags()
    out32, hid32, bounds, (out64, out_atol) = fp64_rollout_bounds(ref, obs, None, w, lambda: od.EuclideanEdge(maxd), N)
    assert float(hid32[1].sum()) > B * N
    assert torch.equal(hid[1].cpu(), hid32[1]) and torch.equal(hid[0].cpu(), hid32[0]) and torch.equal(hid[3].cpu(), hid32[3])
    assert float((out.detach().cpu().double() - out64).abs().max()) <= out_atol
    for k, p in g.named_parameters():
        g64, atol = bounds[k]
        assert float((p.grad.cpu().double() - g64).abs().max()) <= atol, k


def test_euclid_chain_steady_state_full_size_equals_general_kernels():
    """cfg3's shape (B = 256, N = 128, F = 64) for 2 N + 8 steps, thresholds inside the distance distribution: the
    one-launch chain (cached steps, then the steady-state step) against the same chain on the general kernels (distance
    kernel + k_step_rows per step): the state after every 16th step bit for bit, beliefs / gradients to summation order."""
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.distance import EuclideanEdge
    DenseGCM.did_warn = True
    B, N, F, T = 256, 128, 64, 264
    torch.manual_seed(5)
    obs = torch.rand(T, B, F, device=DEV)
    probe = EuclideanEdge(1e9)
    nodes = obs[:N].transpose(0, 1).contiguous()
    d = probe.distances(nodes, torch.full((B,), N - 1, dtype=torch.long, device=DEV))
    maxd = float(d[:, : N - 1].quantile(0.15))
    res = []
    for one_launch in (True, False):
        torch.manual_seed(9)
        ref, g, mem = _mk(F, 32, 32, N, maxd, donate=True)
        mem.rows_cached_steps = one_launch
        hid, outs, snaps = None, [], []
        for t in range(T):
            mx, hid = mem(obs[t], hid)
            outs.append(mx)
            if t % 16 == 15 or t == T - 1:
                snaps.append((hid[0].clone(), hid[1].clone(), hid[3].clone()))
        assert mem.rows_rolled_steps_taken() == (T - N if one_launch else 0)
        out = torch.stack(outs)
        out.sum().backward()
        mem.check_flags()
        res.append((out.detach(), snaps, {k: p.grad.clone() for k, p in g.named_parameters()}))
    a, b = res
    for sa, sb in zip(a[1], b[1]):
        for x, y in zip(sa, sb):
            assert torch.equal(x, y)
    deg = float(a[1][-1][1].sum()) / (B * N)
    assert 5 < deg < 60, deg
    torch.testing.assert_close(a[0], b[0], rtol=1e-5, atol=5e-5)
    for k in a[2]:
        scale = float(b[2][k].abs().max()) + 1e-12
        torch.testing.assert_close(a[2][k], b[2][k], rtol=1e-5, atol=5e-5 * scale, msg=k)


def test_euclid_steady_chain_falls_back_when_the_caller_edits_the_state():
    """The steady-state step writes the rolled adjacency from the chain's own bit image - valid only while the donated
    state is the chain's.  After an in-place edit of the adjacency by the caller (torch's version counter shows it) the
    chain runs the general kernels on the edited matrix: against the oracle given the same edit."""
    from gcm.gcm import DenseGCM
    DenseGCM.did_warn = True
    B, N, F, H = 40, 32, 64, 32
    T1, T2 = N + 7, 6
    torch.manual_seed(3)
    maxd = 3.0
    ref, g, mem = _mk(F, H, H, N, maxd, donate=True)
    obs = _clustered(T1 + T2, B, F, n_c=5, seed=17)
    osel = od.EuclideanEdge(maxd)

    def edit(adj):
        adj[:, 11, 2] = 1.0
        adj[:, 25, 0] = 1.0
        adj[:, 25, 9] = 0.0

    hid_o, hid_p, outs_o, outs_p = None, None, [], []
    with torch.no_grad():
        for t in range(T1 + T2):
            if t == T1:
                a = hid_o[1].clone()
                edit(a)
                hid_o = (hid_o[0], a, hid_o[2], hid_o[3])
                edit(hid_p[1])
            mo, hid_o = od.dense_step(obs[t], hid_o, ref, graph_size=N, edge_selectors=osel)
            mp, hid_p = mem(obs[t].to(DEV), hid_p)
            outs_o.append(mo)
            outs_p.append(mp.cpu())
    mem.check_flags()
    assert mem.rows_rolled_steps_taken() == T1 - N
    assert torch.equal(hid_p[1].cpu(), hid_o[1]) and torch.equal(hid_p[0].cpu(), hid_o[0])
    torch.testing.assert_close(torch.stack(outs_p), torch.stack(outs_o), rtol=1e-5, atol=2e-6)


def test_euclid_paths_do_not_depend_on_uninitialised_memory():
    """The time-parallel rollout writes its caches, records and final state whole (no zero fill) and the per-step chain
    allocates its caches and records empty.  With torch filling every uninitialised allocation with NaN (deterministic
    mode's fill_uninitialized_memory) both must still match the oracle - T = 37 leaves the last 32-row blocks partly
    unwritten; the chain runs past graph_size into the steady-state step."""
    prev_det = torch.are_deterministic_algorithms_enabled()
    prev_fill = torch.utils.deterministic.fill_uninitialized_memory
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
    try:
        test_euclid_rollout_time_parallel_vs_oracle(160, 128, 32, 32, 32, 37, False)
        test_euclid_chain_steady_state_one_launch_vs_oracle(33, 16, 32, 32, 50)
    finally:
        torch.utils.deterministic.fill_uninitialized_memory = prev_fill
        torch.use_deterministic_algorithms(prev_det)
