"""DenseGCM + LearnedEdge on the fused kernels (csrc/learned_step.hip): edge network, gumbel
selection and the compact adjacency-gradient chain, against the reference's recorded-noise vectors
and the CPU oracle with injected noise - up to BASELINE cfg5's per-GPU size.  Needs an MI355X."""
import os

import pytest
import torch

from _golden import Fixture
from oracle import dense as od
from test_dense_gpu import dev_gnn_from, DEV, RTOL, ATOL

pytestmark = pytest.mark.gpu


def _taken(mem):
    cfg = mem._cfg_last[3] if mem._cfg_last else None
    return cfg is not None and cfg.learned_sel is not None


def test_learned_fused_matches_reference_vectors():
    """g6 (the reference run with its gumbel draws recorded), observations WITHOUT gradient: the
    fused path.  Sampled adjacency bit exact, beliefs 1e-5, GNN and edge-network gradients."""
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

    def noise(like):   # the reference drew nothing at t = 0 (learned.py:120-121)
        key = f"noise_{step['t']}"
        return fx[key].to(DEV) if key in fx else torch.zeros_like(like)

    sel.noise_fn = noise
    mem = DenseGCM(g, edge_selectors=sel, graph_size=m["N"])
    obs = fx["obs"].to(DEV)
    hidden, mxs = None, []
    for t in range(m["T"]):
        step["t"] = t
        mx, hidden = mem(obs[t], hidden)
        mxs.append(mx)
    assert _taken(mem)
    mxs = torch.stack(mxs)
    mxs.mean().backward()
    mem.check_flags()
    assert torch.equal(hidden[1].cpu(), fx["hT_adj"])       # sampled edges: bit exact
    assert torch.equal(hidden[0].cpu(), fx["hT_nodes"])
    torch.testing.assert_close(mxs.cpu(), fx["mx"], rtol=RTOL, atol=ATOL)
    for k, p in g.named_parameters():
        want = fx["grad:" + k]
        torch.testing.assert_close(p.grad.cpu(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()), msg=k)
    for k, p in sel.named_parameters():
        want = fx["sel_grad:" + k]
        torch.testing.assert_close(p.grad.cpu(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()) + 1e-8, msg=k)


def _pair(F, H, N, k, seed, donate=False):
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.learned import LearnedEdge
    torch.manual_seed(seed)
    ref = od.canonical_gnn(F, H)
    net = od.build_edge_network(F)
    with torch.no_grad():          # livelier than the default init: the sampled rows change over time
        for p in net.parameters():
            p.mul_(2.0)
    g = dev_gnn_from(ref, [(F, H, torch.nn.Tanh), (H, H, torch.nn.Tanh)])
    sel = LearnedEdge(F, num_edge_samples=k)
    sel.edge_network.load_state_dict(net.state_dict())
    sel = sel.to(DEV)
    return ref, net, g, sel, DenseGCM(g, edge_selectors=sel, graph_size=N, donate_state=donate)


def _oracle_learned(ref, net, obs, noise, wgt, k, N, hid0, dtype=torch.float32):
    """The oracle's per-step loop (learned.py:53-113 restated) with injected gumbel draws, loss = sum(out * wgt), in
    float32 or float64 (on deep copies) -> (out, final hidden, {name: gradient}; edge network under "net.")."""
    import copy
    ref_, net_ = copy.deepcopy(ref).to(dtype), copy.deepcopy(net).to(dtype)
    for m_ in (ref_, net_):
        m_.zero_grad(set_to_none=True)
    step = {"t": 0}
    osel = od.LearnedEdge(net_, num_edge_samples=k,
                          noise_fn=lambda shape: noise[step["t"]][:, : shape[1]].to(dtype))
    hid = None if hid0 is None else tuple(t.to(dtype) if t.is_floating_point() else t.clone() for t in hid0)
    outs = []
    for t in range(obs.shape[0]):
        step["t"] = t
        mx, hid = od.dense_step(obs[t].to(dtype), hid, ref_, graph_size=N, edge_selectors=osel)
        outs.append(mx)
    out = torch.stack(outs)
    (out * wgt.to(dtype)).sum().backward()
    grads = {kk: p.grad for kk, p in ref_.named_parameters()}
    grads.update({"net." + kk: p.grad for kk, p in net_.named_parameters()})
    return out.detach(), tuple(t.detach() for t in hid), grads


def _check_learned_grads(got, g32, g64, same_edges, rtol_sel=2e-3, must_bound=False):
    """Parameter gradients (GNN + edge network "net.*") against the oracle.  With the same sampled edges in the
    oracle's float32 and float64 runs: the float64 bound of tests/_golden.py - no further from the float64 gradient
    than 3x the oracle's own fp32 evaluation is, floor 2e-6 of the gradient scale (the last LayerNorm's bias and
    the output bias get sum_j g_logit[j] = 0 analytically - softmax gradients sum to zero -: the floor there comes
    from the edge network's common scale; 2e-6 = 17 ulp of a sum over T x B x N terms that the kernels and torch's
    CPU kernels add in different orders - one seed of the ragged shapes sat at 1.2e-6 with the oracle's own fp32
    run at 0.33e-6).  Otherwise (a softmax value within rounding of the threshold flipped an
    edge between the two precisions, so float64 describes another graph): rtol against the float32 oracle."""
    scale = max(float(v.abs().max()) for kk, v in g64.items() if kk.startswith("net."))
    assert scale > 0
    assert same_edges or not must_bound, "the float32 and float64 oracle runs sampled different edges: pick another seed"
    for kk, gd in got.items():
        sc = scale if kk.startswith("net.") else float(g64[kk].abs().max())
        if same_edges:
            err_ref = float((g32[kk].double() - g64[kk]).abs().max())
            err = float((gd.double() - g64[kk]).abs().max())
            assert err <= max(3.0 * err_ref, 2e-6 * sc), (kk, err, err_ref, sc)
        else:
            torch.testing.assert_close(gd, g32[kk], rtol=rtol_sel,
                                       atol=2e-5 * float(g32[kk].abs().max()) + 2e-6 * sc, msg=kk)


def _run_both(B, N, F, H, T, k, seed, count0, pick, rtol_sel=2e-3, donate=False, must_bound=False, rollout=False):
    """product on the whole batch, oracle on the graphs `pick`, same injected gumbel noise; the loss
    weights only the picked graphs.  count0 = None: from hidden = None (empty graphs: the cached steps)."""
    ref, net, g, sel, mem = _pair(F, H, N, k, seed, donate)
    gen = torch.Generator().manual_seed(seed + 1)
    obs = torch.rand(T, B, F, generator=gen)
    noise = -torch.empty(T, B, N).exponential_(generator=gen).log()
    h0 = hidden = None
    if count0 is not None:
        nodes0 = torch.rand(B, N, F, generator=gen) * (torch.arange(N)[None, :, None] < count0[:, None, None])
        adj0 = torch.zeros(B, N, N)
        i = torch.arange(1, N)
        adj0[:, i, i - 1] = 1.0
        adj0 = adj0 * (torch.arange(N)[None, :, None] < count0[:, None, None])
        h0 = (nodes0[pick].clone(), adj0[pick].clone(), torch.zeros(0), count0[pick].clone())
        hidden = (nodes0.to(DEV), adj0.to(DEV), torch.zeros(0, device=DEV), count0.to(DEV))
    wgt = torch.rand(T, len(pick), H, generator=gen)
    # oracle on the slice, in float32 and in float64 (same draws; the same edges unless a softmax value sits within
    # rounding of the threshold): the latter bounds the gradient comparison below
    out_c, hid, g32 = _oracle_learned(ref, net, obs[:, pick], noise[:, pick], wgt, k, N, h0)
    _, hid64, g64 = _oracle_learned(ref, net, obs[:, pick], noise[:, pick], wgt, k, N, h0, torch.float64)
    same_edges = torch.equal(hid64[1].float(), hid[1])
    # product on everything
    pstep = {"t": 0}
    sel.noise_fn = lambda like: noise[pstep["t"]].to(DEV)
    h_first = None if hidden is None else (hidden[0].data_ptr(), hidden[1].data_ptr())
    if rollout:      # the time-batched entry: from hidden = None and T <= N, three launches for the whole forward
        assert hidden is None
        cnt = {"t": 0}

        def by_count(like):
            cnt["t"] += 1
            return noise[cnt["t"] - 1].to(DEV)
        sel.noise_fn = by_count
        out_d, hidden = mem.rollout(obs.to(DEV))
        assert cnt["t"] == T
    else:
        outs = []
        for t in range(T):
            pstep["t"] = t
            mx, hidden = mem(obs[t].to(DEV), hidden)
            outs.append(mx)
        out_d = torch.stack(outs)
    assert _taken(mem)
    if donate and N % 4 == 0 and F % 4 == 0 and h_first:   # advanced in place: the caller's own tensors all the way
        assert hidden[0].data_ptr() == h_first[0] and hidden[1].data_ptr() == h_first[1]
    (out_d[:, pick] * wgt.to(DEV)).sum().backward()
    mem.check_flags()
    assert torch.equal(hidden[1][pick].cpu(), hid[1])          # sampled adjacency: bit exact
    assert torch.equal(hidden[0][pick].cpu(), hid[0]) and torch.equal(hidden[3][pick].cpu(), hid[3])
    torch.testing.assert_close(out_d[:, pick].detach().cpu(), out_c, rtol=RTOL, atol=2e-6)
    got = {kk: p.grad.cpu() for kk, p in g.named_parameters()}
    got.update({"net." + kk: p.grad.cpu() for kk, p in sel.edge_network.named_parameters()})
    _check_learned_grads(got, g32, g64, same_edges, rtol_sel, must_bound)
    return hidden, mem


@pytest.mark.parametrize("donate", [False, True])
@pytest.mark.parametrize("B,N,F,H,T,k", [(5, 16, 8, 16, 24, 3), (4, 32, 32, 32, 40, 5), (3, 12, 4, 8, 9, 2),
                                         (4, 72, 20, 24, 90, 4), (3, 50, 10, 12, 60, 3)])
def test_learned_fused_vs_oracle(B, N, F, H, T, k, donate):
    """ragged shapes, staggered starts, overflow crossings; functional and donated state (the state advanced
    in place, the step's record keeps the node matrix and row cur of the adjacency).  N = 72 / 50: a last 32-row
    block that is partly beyond the graph (the backward works per block, k_learned_bptt_mlp); F = 10: rows that are
    no whole number of 16-byte loads."""
    torch.manual_seed(B + N)
    count0 = torch.randint(0, N + 1, (B,))
    _run_both(B, N, F, H, T, k, seed=B * 7 + N, count0=count0, pick=list(range(B)), donate=donate)


def test_learned_fused_cfg5_size_slice():
    """BASELINE cfg5's per-GPU share: B = 256, N = 128, F = H = 32, graphs almost full (so every
    step scores ~100+ candidates per graph and the last steps overflow); oracle on a slice."""
    B, N, F, H, T = 256, 128, 32, 32, 6
    gen = torch.Generator().manual_seed(3)
    count0 = torch.randint(100, 127, (B,), generator=gen)
    hidden, _ = _run_both(B, N, F, H, T, 5, seed=11, count0=count0, pick=[0, 97, 255])
    assert int(hidden[3].max()) == N


def test_learned_fused_cfg5_size_long_rollout():
    """cfg5's per-GPU share for T = 44 steps from almost-full graphs: ~110+ candidates scored per
    graph and step, every graph overflows and then rolls at every step (the chain buffer rolls with the
    state); oracle on a slice of 3 graphs with the same injected gumbel draws."""
    B, N, F, H, T = 256, 128, 32, 32, 44
    gen = torch.Generator().manual_seed(5)
    count0 = torch.randint(108, 127, (B,), generator=gen)
    hidden, _ = _run_both(B, N, F, H, T, 5, seed=13, count0=count0, pick=[1, 130, 254])
    assert int(hidden[3].min()) == N          # every graph is in steady-state overflow by the end
    hidden, _ = _run_both(B, N, F, H, T, 5, seed=13, count0=count0, pick=[1, 130, 254], donate=True)
    assert int(hidden[3].min()) == N


def test_learned_rollout_entry_time_parallel_full_size():
    """DenseGCM.rollout with LearnedEdge at cfg5's per-GPU size (B = 256, N = 128, F = H = 32, T = 64 from
    hidden = None): the three-launch time-parallel forward (k_learned_roll_logits, k_learned_roll_pick, k_learned_roll_l2) and the chain's
    time-parallel backward over its records - against the oracle's per-step loop on a 3-graph slice with the same
    injected gumbel draws: sampled adjacency bit exact, beliefs 1e-5, every gradient inside the float64 bound."""
    B, N, F, H, T = 256, 128, 32, 32, 64
    hidden, mem = _run_both(B, N, F, H, T, 5, seed=17, count0=None, pick=[2, 101, 255], must_bound=True, rollout=True)
    assert int(hidden[3].min()) == T and int(hidden[3].max()) == T
    assert float(hidden[1].sum()) > B * T


@pytest.mark.parametrize("B,N,F,H,T,k", [(5, 72, 20, 24, 60, 4), (3, 50, 10, 12, 50, 3), (6, 128, 32, 16, 100, 5)])
def test_learned_rollout_entry_ragged_blocks(B, N, F, H, T, k):
    """The time-parallel rollout (forward per 32-row block with a candidate row: k_learned_roll_logits) at graph
    sizes that are no multiple of 32, narrow / odd feature widths and a hidden width different from the observation's:
    against the oracle's per-step loop, sampled adjacency bit exact."""
    hidden, mem = _run_both(B, N, F, H, T, k, seed=23 + N, count0=None, pick=list(range(B)), rollout=True)
    assert int(hidden[3].min()) == T


@pytest.mark.parametrize("donate", [True, False])
def test_learned_cfg5_timed_path_from_empty_graphs_full_size(donate):
    """The kernels bench.py --config cfg5 is timed on, at its per-GPU size, directly against the oracle (VERDICT r3
    #1b): B = 256, N = 128, F = H = 32, a chain from hidden = None - every step a cached step, ONE launch
    (k_learned_select<MODE, TAIL>: selection + the GNN on row cur over the chain's caches), the backward the two
    time-parallel passes (k_bptt_rows<.., 2>, k_learned_bptt_sel, k_learned_bptt_mlp) over the caches - T = 32 steps, injected gumbel
    draws, oracle on a 3-graph slice in float32 and float64.  Sampled adjacency bit exact, beliefs 1e-5, EVERY
    gradient (GNN and edge network) inside the float64 bound - no rtol."""
    B, N, F, H, T = 256, 128, 32, 32, 32
    hidden, mem = _run_both(B, N, F, H, T, 5, seed=17, count0=None, pick=[2, 101, 255], donate=donate, must_bound=True)
    assert mem._learned_chain[1].cached_steps() == T        # all of them on the cached step
    assert int(hidden[3].min()) == T and int(hidden[3].max()) == T
    assert float(hidden[1].sum()) > B * T                    # (edges were sampled at all)


def test_learned_noise_pool_statistics_and_equivalence():
    """The default noise path of the fused LearnedEdge step: Exp(1) draws from the device RNG, 16
    steps' worth per launch (`DenseGCM._noise_pool`), turned into gumbel noise inside the kernel
    (torch.nn.functional.gumbel_softmax draws -log(Exp(1)) the same way, learned.py:89).
    (1) the draws are Exp(1) (mean, variance, independence of consecutive pools and of a pool's
    slices); (2) a run on the pool equals a run that gets the very same draws injected per step
    through `noise_fn` (the parity tests' path)."""
    B, N, F, H, T = 64, 32, 32, 32, 40
    ref, net, g, sel, mem = _pair(F, H, N, 5, seed=21)
    mem.noise_pool_steps = 16          # (the module's default is 64: three pools in these 40 steps)
    obs = torch.rand(T, B, F, device=DEV)
    pools, used = [], []
    hidden, outs = None, []
    with torch.no_grad():
        for t in range(T):
            mx, hidden = mem(obs[t], hidden)
            pool = mem._noise_pool
            if not pools or pools[-1] is not pool[0]:
                pools.append(pool[0])
            used.append(pool[0][pool[1] - 1])
            outs.append(mx)
    assert _taken(mem)
    assert len(pools) == 3 and all(p.shape == (16, B, N) for p in pools)      # 40 steps = 16 + 16 + 8
    draws = torch.stack(pools).double()
    n = draws.numel()
    assert float(draws.min()) > 0
    assert abs(float(draws.mean()) - 1.0) < 5.0 / n ** 0.5                     # sd of the mean = 1/sqrt(n)
    assert abs(float(draws.var()) - 1.0) < 5.0 * (8.0 / n) ** 0.5              # var of the sample variance = 8/n
    assert abs(float((draws < 0.6931471805599453).double().mean()) - 0.5) < 5.0 * 0.5 / n ** 0.5   # median
    flat = draws.reshape(3 * 16, -1)
    c = torch.corrcoef(flat)                                                   # slices / pools are independent
    assert float((c - torch.eye(48, dtype=c.dtype, device=c.device)).abs().max()) < 6.0 / flat.shape[1] ** 0.5
    # (2) the same draws injected step by step
    ref2, net2, g2, sel2, mem2 = _pair(F, H, N, 5, seed=21)
    step = {"t": 0}
    sel2.noise_fn = lambda like: -used[step["t"]].log()
    hidden2, outs2 = None, []
    with torch.no_grad():
        for t in range(T):
            step["t"] = t
            mx, hidden2 = mem2(obs[t], hidden2)
            outs2.append(mx)
    same = (hidden[1] == hidden2[1]).flatten(1).all(dim=1)
    assert float(same.double().mean()) >= 0.98      # (-log(x) is evaluated by the kernel in one run, by torch in the other)
    assert float(hidden[1].sum()) > float(B * 8)    # and edges were sampled at all
    torch.testing.assert_close(torch.stack(outs)[:, same], torch.stack(outs2)[:, same], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("donate", [True, False])
@pytest.mark.parametrize("B,N,F,H,T,k", [(5, 16, 8, 16, 24, 3), (4, 32, 32, 32, 40, 5), (3, 12, 4, 8, 9, 2),
                                         (6, 128, 32, 32, 20, 5), (3, 128, 32, 32, 100, 5)])   # (T = 100: the per-graph pass B1 past 64 steps)
def test_learned_cached_steps_vs_oracle(B, N, F, H, T, k, donate):
    """A rollout from hidden = None (donated or functional state): its first N steps are cached steps (ONE launch each: the
    GNN behind the selection on the chain's h1 / agg1 / node caches, gcm_learned_step_cached), the steps behind
    them - the graphs overflow - the usual ones; the backward mixes both kinds.  Against the oracle with the same
    injected gumbel draws, and against the same rollout without cached steps."""
    res = []
    for cached in (True, False):
        ref, net, g, sel, mem = _pair(F, H, N, k, seed=7, donate=donate)
        mem.learned_cached_steps = cached
        gen = torch.Generator().manual_seed(11)
        obs = torch.rand(T, B, F, generator=gen)
        noise = -torch.empty(T, B, N).exponential_(generator=gen).log()
        wgt = torch.rand(T, B, H, generator=gen)
        pstep = {"t": 0}
        sel.noise_fn = lambda like: noise[pstep["t"]].to(DEV)
        hidden, outs = None, []
        for t in range(T):
            pstep["t"] = t
            mx, hidden = mem(obs[t].to(DEV), hidden)
            outs.append(mx)
        assert _taken(mem)
        n_c = mem._learned_chain[1].cached_steps()
        assert n_c == (min(T, N) if cached else 0), n_c
        out_d = torch.stack(outs)
        (out_d * wgt.to(DEV)).sum().backward()
        mem.check_flags()
        res.append((out_d.detach().cpu(), [t.detach().cpu() for t in (hidden[0], hidden[1], hidden[3])],
                    {k_: p.grad.cpu().clone() for k_, p in list(g.named_parameters()) +
                     [("net." + n_, p_) for n_, p_ in sel.edge_network.named_parameters()]}))
    # oracle, float32 and float64
    out_c, hid, g32 = _oracle_learned(ref, net, obs, noise, wgt, k, N, None)
    _, hid64, g64 = _oracle_learned(ref, net, obs, noise, wgt, k, N, None, torch.float64)
    same_edges = torch.equal(hid64[1].float(), hid[1])
    scale = max(float(v.abs().max()) for kk, v in g32.items() if kk.startswith("net."))
    for out_d, state, grads in res:
        assert torch.equal(state[1], hid[1]) and torch.equal(state[0], hid[0]) and torch.equal(state[2], hid[3])
        torch.testing.assert_close(out_d, out_c, rtol=RTOL, atol=2e-6)
        _check_learned_grads(grads, g32, g64, same_edges)
    for k_ in res[0][2]:       # cached against not cached: the same arithmetic up to summation order
        a, b = res[0][2][k_], res[1][2][k_]   # (gradients that are zero analytically: the floor from the common scale)
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5 * float(b.abs().max()) +
                                   (2e-6 * scale if k_.startswith("net.") else 0.0), msg=k_)


@pytest.mark.parametrize("donate", [True, False])
@pytest.mark.parametrize("B,N,F,H,T", [(7, 32, 32, 32, 34), (5, 24, 12, 16, 20)])   # (T + 7 short of a pool refill)
def test_learned_fast_host_path_equals_interpreter_path(B, N, F, H, T, donate):
    """A continuing LearnedEdge chain goes from DenseGCM.__call__ straight into the C++ host path (LearnedFast:
    the checks of forward() / _forward_learned / _packed_params done natively, the gumbel draws from the module's
    own pool): every step but the chain's first, the same kernels on the same draws - beliefs, state and every
    parameter gradient bit for bit those of the interpreter's path; it declines (and the interpreter's path runs)
    once a parameter is written to, a hook is registered or a noise function injected."""
    res = []
    for fast in (True, False):
        ref, net, g, sel, mem = _pair(F, H, N, 4, seed=5, donate=donate)
        mem.learned_fast_path = fast
        mem.noise_pool_steps = 16      # (a refill inside these T steps)
        torch.manual_seed(99)
        obs = torch.rand(T, B, F, device=DEV)
        hidden, outs = None, []
        for t in range(T):
            mx, hidden = mem(obs[t], hidden)
            outs.append(mx)
        assert _taken(mem)
        assert mem.learned_fast_steps() == (T - 1 - (T - 1) // 16 if fast else 0)   # (a pool refill per 16 steps goes the long way)
        out = torch.stack(outs)
        (out * torch.linspace(0.5, 1.5, out.numel(), device=DEV).view_as(out)).sum().backward()
        mem.check_flags()
        grads = [p.grad.clone() for p in list(g.parameters()) + list(sel.edge_network.parameters())]
        res.append((out.detach().clone(), [hidden[0].clone(), hidden[1].clone(), hidden[3].clone()], grads))
        if fast:       # declines: a written parameter, a forward hook, an injected noise function
            n0 = mem.learned_fast_steps()
            mx, hidden = mem(obs[0], hidden)
            assert mem.learned_fast_steps() == n0          # (the chain ran backward: a new chain starts the long way)
            mx, hidden = mem(obs[1], hidden)
            assert mem.learned_fast_steps() == n0 + 1
            with torch.no_grad():
                next(g.parameters()).mul_(1.0)
            mx, hidden = mem(obs[2], hidden)
            assert mem.learned_fast_steps() == n0 + 1
            mx, hidden = mem(obs[3], hidden)
            assert mem.learned_fast_steps() == n0 + 2
            hk = mem.register_forward_hook(lambda m, i, o: None)
            mx, hidden = mem(obs[4], hidden)
            assert mem.learned_fast_steps() == n0 + 2
            hk.remove()
            mx, hidden = mem(obs[5], hidden)
            sel.noise_fn = lambda like: torch.zeros_like(like)
            n1 = mem.learned_fast_steps()
            mx, hidden = mem(obs[6], hidden)
            assert mem.learned_fast_steps() == n1
            sel.noise_fn = None
            mem.check_flags()
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    for a, b in zip(res[0][2], res[1][2]):
        assert torch.equal(a, b)
    # the C++ host path keeps no reference cycle through the module alive (it holds its config weakly)
    import gc
    import weakref
    mem_f = _pair(F, H, N, 4, seed=5, donate=donate)[4]
    hidden = None
    for t in range(3):
        mx, hidden = mem_f(obs[t], hidden)
    assert mem_f.learned_fast_steps() == 2
    wr = weakref.ref(mem_f)
    del mem_f, mx, hidden
    gc.collect()
    assert wr() is None


def test_learned_rollout_call_boundary_truncate():
    """what DenseGCM.rollout does with the adjacency's gradient chain at the call boundary, pinned.  Two
    consecutive chunks of one sequence, the loss on the SECOND chunk only: with truncate=False the edge network still
    receives the gradient that reaches the first chunk's selections through the hidden state (exactly what 2 T
    forward() calls give: learned.py:89-113, the returned adjacency carries its gumbel-softmax history); with
    truncate=True (default; RLlib-style detached states, ray_gcm.py:186-209) the first chunk's selections get none."""
    from gcm import nn as G
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.learned import LearnedEdge
    B, N, F, H, T = 6, 32, 32, 32, 10
    torch.manual_seed(3)
    obs = torch.rand(2 * T, B, F, device=DEV)
    noise = torch.rand(2 * T, B, N, device=DEV).clamp_(1e-6, 1 - 1e-6)
    noise = -torch.log(-torch.log(noise))

    def run(mode):
        torch.manual_seed(11)
        g = G.Sequential("x, adj, weights, B, N", [
            (G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
            (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(DEV)
        sel = LearnedEdge(F).to(DEV)
        step = {"t": 0}

        def nf(like):
            t = step["t"]
            step["t"] += 1
            return noise[t]
        sel.noise_fn = nf
        mem = DenseGCM(g, edge_selectors=sel, graph_size=N)
        if mode == "steps":
            hid, outs = None, []
            for t in range(2 * T):
                mx, hid = mem(obs[t], hid)
                outs.append(mx)
            out2 = torch.stack(outs[T:])
        else:
            _, hid = mem.rollout(obs[:T], None, truncate=mode == "truncate")
            out2, hid = mem.rollout(obs[T:], hid, truncate=mode == "truncate")
        out2.sum().backward()
        mem.check_flags()
        return out2.detach(), hid[1].detach(), {k: p.grad.clone() for k, p in sel.named_parameters()}

    ref = run("steps")
    keep = run("keep")
    cut = run("truncate")
    assert torch.equal(keep[1], ref[1]) and torch.equal(cut[1], ref[1])          # the same sampled edges
    torch.testing.assert_close(keep[0], ref[0], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(cut[0], ref[0], rtol=1e-5, atol=1e-6)
    differs = False
    for k in ref[2]:
        scale = float(ref[2][k].abs().max()) + 1e-12
        torch.testing.assert_close(keep[2][k], ref[2][k], rtol=1e-4, atol=2e-5 * scale, msg=k)
        differs = differs or float((cut[2][k] - ref[2][k]).abs().max()) > 1e-3 * scale
    assert differs, "truncate=True must cut the gradient that crosses the call boundary"


@pytest.mark.parametrize("B,N,F,H,T,k", [(6, 32, 32, 32, 80, 5), (5, 16, 8, 16, 40, 3), (4, 72, 20, 24, 100, 4),
                                         (3, 128, 32, 32, 200, 5), (4, 32, 32, 32, 70, 20), (3, 72, 20, 24, 100, 30)])
def test_learned_chain_steady_state_one_launch_vs_oracle(B, N, F, H, T, k):
    """A donated LearnedEdge chain from empty graphs carried past graph_size steps: the first N steps are cached steps,
    from step N on every step drops every graph's oldest node (gcm.py:263-271, 323-355) and runs
    gcm_learned_step_steady - selection, in-place roll and layer 1 of every row re-evaluated in ONE launch, the record
    of the two-launch form.  Against the oracle's per-step loop with the same injected gumbel draws: sampled adjacency
    bit exact, beliefs 1e-5, every gradient (GNN and edge network)."""
    from gcm.gcm import DenseGCM
    DenseGCM.did_warn = True
    hidden, mem = _run_both(B, N, F, H, T, k, seed=31 + N, count0=None, pick=list(range(B)), donate=True)
    assert int(hidden[3].min()) == N
    assert mem.learned_steady_steps_taken() == (T - N if (N % 4 == 0 and F % 4 == 0) else 0)


@pytest.mark.parametrize("B,N", [(7, 128), (5, 72), (3, 4), (64, 100)])
def test_adj_bits_through_the_c_abi(B, N):
    """gcm_adj_bits (the image a steady-state LearnedEdge chain keeps of its adjacency): bit (j & 31) of word (j >> 5)
    of row r = (adj[b, r, j] != 0), every other bit zero - against numpy, ragged N."""
    import numpy as np
    from gcm import _hip
    lib = _hip.lib()
    torch.manual_seed(B + N)
    adj = (torch.rand(B, N, N) < 0.2).float() * torch.randint(1, 3, (B, N, N)).float()   # non-zero, not only 1.0
    a_dev = adj.to(DEV)
    bits = torch.full((B, N, 4), -1, dtype=torch.int32, device=DEV)
    _hip.check(lib.gcm_adj_bits(a_dev.data_ptr(), bits.data_ptr(), B, N, _hip.stream()), "gcm_adj_bits")
    torch.cuda.synchronize()
    got = bits.cpu().numpy().astype(np.uint32)
    want = np.zeros((B, N, 4), dtype=np.uint32)
    nz = (adj != 0).numpy()
    for j in range(N):
        want[:, :, j >> 5] |= nz[:, :, j].astype(np.uint32) << np.uint32(j & 31)
    assert np.array_equal(got, want)
    assert lib.gcm_adj_bits(a_dev.data_ptr(), bits.data_ptr(), B, 130, _hip.stream()) != 0     # N > 128: refused


def test_learned_steady_chain_falls_back_when_the_caller_edits_the_state():
    """The steady-state step rolls the adjacency through the chain's own bit image - valid only while the donated state
    is the chain's.  A caller that edits the adjacency in place between two steps (torch sees it: the version counter)
    must get the general kernels from then on, working on the edited matrix: against the oracle given the same edit."""
    from gcm.gcm import DenseGCM
    DenseGCM.did_warn = True
    B, N, F, H, k = 4, 32, 32, 32, 4
    T1, T2 = N + 6, 5
    ref, net, g, sel, mem = _pair(F, H, N, k, seed=91, donate=True)
    gen = torch.Generator().manual_seed(92)
    obs = torch.rand(T1 + T2, B, F, generator=gen)
    noise = -torch.empty(T1 + T2, B, N).exponential_(generator=gen).log()
    step = {"t": 0}
    osel = od.LearnedEdge(net, num_edge_samples=k, noise_fn=lambda shape: noise[step["t"]][:, : shape[1]])
    sel.noise_fn = lambda like: noise[step["t"]].to(DEV)

    def edit(adj):           # a few edges the selector never sampled: rows 9 and 20 point at nodes 3 and 0
        adj[:, 9, 3] = 1.0
        adj[:, 20, 0] = 1.0
        adj[:, 20, 7] = 0.0

    hid_o, hid_p, outs_o, outs_p = None, None, [], []
    with torch.no_grad():
        for t in range(T1 + T2):
            step["t"] = t
            if t == T1:
                a = hid_o[1].clone()
                edit(a)
                hid_o = (hid_o[0], a, hid_o[2], hid_o[3])
                edit(hid_p[1])                     # in place, on the donated tensor
            mo, hid_o = od.dense_step(obs[t], hid_o, ref, graph_size=N, edge_selectors=osel)
            mp, hid_p = mem(obs[t].to(DEV), hid_p)
            outs_o.append(mo)
            outs_p.append(mp.cpu())
    mem.check_flags()
    assert mem.learned_steady_steps_taken() == T1 - N      # steady until the edit, the general kernels after it
    assert torch.equal(hid_p[1].cpu(), hid_o[1]) and torch.equal(hid_p[0].cpu(), hid_o[0])
    torch.testing.assert_close(torch.stack(outs_p), torch.stack(outs_o), rtol=RTOL, atol=2e-6)


@pytest.mark.parametrize("rollout", [False, True])
def test_learned_chain_does_not_depend_on_uninitialised_cache_rows(rollout):
    """The chain's caches (h1, agg1, nodes, U) are allocated without a zero fill: a cached step reads rows < cur only and
    the backward takes the rows of a 32-row block that lie behind the candidates as zeros.  With torch filling every
    uninitialised allocation with NaN (deterministic mode's fill_uninitialized_memory) the per-step chain and the
    time-parallel rollout must still match the oracle - T = 29 leaves three rows of the first block unwritten."""
    from gcm.gcm import DenseGCM
    DenseGCM.did_warn = True
    prev_det = torch.are_deterministic_algorithms_enabled()
    prev_fill = torch.utils.deterministic.fill_uninitialized_memory
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True
    try:
        _run_both(5, 128, 32, 32, 29, 4, seed=77, count0=None, pick=[0, 2, 4], donate=True, rollout=rollout)
        _run_both(4, 40, 20, 24, 29, 3, seed=78, count0=None, pick=[0, 1], donate=not rollout, rollout=rollout)
    finally:
        torch.utils.deterministic.fill_uninitialized_memory = prev_fill
        torch.use_deterministic_algorithms(prev_det)


def test_learned_round5_kernels_behind_the_ab_switches_match_the_oracle():
    """The four-wave cached step (GCM_STEP_FOUR_WAVES) and the 32-row-block pass B2 (GCM_BPTT_MLP_BLOCKS) stay behind the
    per-call flags for the A/B of tools/ab_cfg5.sh: the module reads GCM_LEARNED_FOUR_WAVES / GCM_BPTT_MLP_BLOCKS once per
    process, so the exact-shape oracle comparisons of this file run once more in a child that sets both - at those shapes
    the default run exercises k_learned_select8 / k_learned_bptt_mlp16 instead."""
    import subprocess
    import sys
    env = dict(os.environ, GCM_LEARNED_FOUR_WAVES="1", GCM_BPTT_MLP_BLOCKS="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run(
        [sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.abspath(__file__),
         "-k", "cfg5_timed_path or cached_steps_vs_oracle or steady_state_one_launch"],
        env=env, capture_output=True, timeout=900, cwd=root)
    tail = p.stdout.decode()[-1500:]
    assert p.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
