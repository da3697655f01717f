"""The live-row step with what sits between the selectors and the GNN folded in (SURVEY 8f rank 2,
gcm.py:290-306): a Linear preprocessor, index-writing aux selectors, PositionalEncoding.  Against the
reference's own vectors (G15), through forward() with functional and donated state and through
rollout().  Needs an MI355X."""
import pytest
import torch

from _golden import Fixture, FOLD_SPECS, fold_selector
from oracle import dense as od

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL, ATOL = 1e-5, 1e-6


def _build(fx, name, donate=False, fused=True):
    from gcm import nn as G
    from gcm.gcm import DenseGCM, PositionalEncoding
    from gcm.edge_selectors.temporal import TemporalBackedge
    from gcm.edge_selectors.dense import DenseEdge
    m = fx.meta
    ref = od.canonical_gnn(m["Fg"], m["H"])
    ref.load_state_dict(fx.group("param:"))
    g = G.Sequential("x, adj, weights, B, N", [
        (G.DenseGraphConv(m["Fg"], m["H"]), "x, adj -> x"), torch.nn.Tanh(),
        (G.DenseGraphConv(m["H"], m["H"]), "x, adj -> x"), torch.nn.Tanh()])
    g.load_state_dict(ref.state_dict())
    g = g.to(DEV)
    sp = fx.group("sel_param:")
    pre = None
    if m["pre_bias"] is not None:
        pre = torch.nn.Linear(m["F"], m["Fg"], bias=m["pre_bias"])
        pre.load_state_dict({k[len("pre."):]: v for k, v in sp.items() if k.startswith("pre.")})
        pre = pre.to(DEV)
    pe = None
    if m["mode"]:
        pe = PositionalEncoding(max_len=m["N"], mode=m["mode"], cat_dim=m["cat_dim"])
        if m["mode"] == "cat":           # the lazily built layer, with the reference's weights
            pe.run_once(torch.zeros(1, 1, m["Fg"], device=DEV))
            pe.load_state_dict({k[len("pe."):]: v for k, v in sp.items() if k.startswith("pe.")})
    sel, aux = (fold_selector(s, TemporalBackedge, DenseEdge) for s in FOLD_SPECS[name])
    mem = DenseGCM(g, preprocessor=pre, edge_selectors=sel, aux_edge_selectors=aux, positional_encoder=pe,
                   graph_size=m["N"], donate_state=donate, fused=fused)
    return mem, g, pre


def _fp64_reference(fx, name):
    """(beliefs of the same module evaluated in float64 by the oracle, tolerance): DenseEdge rows add
    up to N terms per aggregate, where two fp32 summation orders differ by more than 1e-5 relative
    (_golden.fp64_bound)."""
    import copy
    from _golden import fp64_bound
    m = fx.meta
    ref = od.canonical_gnn(m["Fg"], m["H"])
    ref.load_state_dict(fx.group("param:"))
    sp = fx.group("sel_param:")
    pre = None
    if m["pre_bias"] is not None:
        pre = torch.nn.Linear(m["F"], m["Fg"], bias=m["pre_bias"])
        pre.load_state_dict({k[len("pre."):]: v for k, v in sp.items() if k.startswith("pre.")})
        pre = copy.deepcopy(pre).double()
    pe = od.PositionalEncoding(max_len=m["N"], mode=m["mode"], cat_dim=m["cat_dim"]) if m["mode"] == "add" else None
    sel, aux = (fold_selector(s, od.TemporalBackedge, od.DenseEdge) for s in FOLD_SPECS[name])
    return fp64_bound(ref, fx["obs"], None, fx["mx"], graph_size=m["N"], edge_selectors=sel, preprocessor=pre,
                      aux_edge_selectors=aux, positional_encoder=pe)


def _check(fx, mxs, hidden, g, pre, tag):
    assert torch.equal(hidden[0].cpu(), fx["hT_nodes"]), tag      # stored nodes stay raw
    assert torch.equal(hidden[1].cpu(), fx["hT_adj"]), tag
    assert torch.equal(hidden[3].cpu(), fx["hT_num_nodes"]), tag
    name = tag.split()[0]
    if any(s is not None and s[0] == "dense" for s in FOLD_SPECS[name]):
        out64, atol = _fp64_reference(fx, name)
        torch.testing.assert_close(mxs.cpu().double(), out64, rtol=0, atol=atol, msg=lambda s: f"{tag}: {s}")
    else:
        torch.testing.assert_close(mxs.cpu(), fx["mx"], rtol=RTOL, atol=ATOL, msg=lambda s: f"{tag}: {s}")
    for k, p in g.named_parameters():
        want = fx["grad:" + k]
        torch.testing.assert_close(p.grad.cpu(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()),
                                   msg=lambda s, k=k: f"{tag} {k}: {s}")
    if pre is not None:
        for k, p in pre.named_parameters():
            want = fx["sel_grad:pre." + k]
            torch.testing.assert_close(p.grad.cpu(), want, rtol=1e-5, atol=1e-5 * float(want.abs().max()),
                                       msg=lambda s, k=k: f"{tag} pre.{k}: {s}")


@pytest.mark.parametrize("name", sorted(FOLD_SPECS))
@pytest.mark.parametrize("donate", [False, True])
def test_folded_step_matches_reference(name, donate):
    fx = Fixture(name)
    m = fx.meta
    mem, g, pre = _build(fx, name, donate=donate)
    obs = fx["obs"].to(DEV)
    hidden, mxs = None, []
    for t in range(m["T"]):
        mx, hidden = mem(obs[t], hidden)
        mxs.append(mx)
    cfg = mem._cfg_last[3]
    assert cfg.fold is not None and cfg.rows_ok           # the live-row kernel ran, not the layered path
    mxs = torch.stack(mxs)
    mxs.mean().backward()
    mem.check_flags()
    _check(fx, mxs.detach(), hidden, g, pre, f"{name} donate={donate}")


@pytest.mark.parametrize("name", sorted(FOLD_SPECS))
def test_folded_rollout_and_dx_dispatch(name):
    """rollout() on a folded module = the same per-step kernels; observations that need a gradient
    take the layered path (same numbers, plus d obs)."""
    fx = Fixture(name)
    mem, g, pre = _build(fx, name)
    mxs, hidden = mem.rollout(fx["obs"].to(DEV))
    mxs.mean().backward()
    mem.check_flags()
    _check(fx, mxs.detach(), hidden, g, pre, f"{name} rollout")
    g.zero_grad(set_to_none=True)
    if pre is not None:
        pre.zero_grad(set_to_none=True)
    obs = fx["obs"].to(DEV).requires_grad_(True)
    hidden, outs = None, []
    for t in range(fx.meta["T"]):
        mx, hidden = mem(obs[t], hidden)
        outs.append(mx)
    mxs = torch.stack(outs)
    mxs.mean().backward()
    _check(fx, mxs.detach(), hidden, g, pre, f"{name} layered")
    gs = float(fx["grad_obs"].abs().max())
    torch.testing.assert_close(obs.grad.cpu(), fx["grad_obs"], rtol=1e-5, atol=1e-5 * gs)


@pytest.mark.parametrize("kind", ["pre", "pe_add"])
def test_folded_step_full_size_vs_layered(kind):
    """cfg2's shape with the RLlib model's default preprocessor (ray_gcm.py:117) / a positional
    encoding: folded live-row step == layered path, through the overflow."""
    from gcm import nn as G
    from gcm.gcm import DenseGCM, PositionalEncoding
    from gcm.edge_selectors.temporal import TemporalBackedge
    B, N, F, Fg, H, T = 6, 128, 24, 32, 32, 140
    torch.manual_seed(5)
    obs = torch.rand(T, B, F, device=DEV)
    res = []
    for fused in (True, False):
        torch.manual_seed(6)
        fg = Fg if kind == "pre" else F
        g = G.Sequential("x, adj, weights, B, N", [
            (G.DenseGraphConv(fg, H), "x, adj -> x"), torch.nn.Tanh(),
            (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(DEV)
        pre = torch.nn.Linear(F, Fg).to(DEV) if kind == "pre" else None
        pe = PositionalEncoding(max_len=N, mode="add") if kind == "pe_add" else None
        # (DenseEdge + encoding at this size sums 128 O(1) terms per aggregate: covered with the
        # fp64 bound by g15_fold_pe_add_exact)
        aux = TemporalBackedge([3], direction="backward")
        mem = DenseGCM(g, preprocessor=pre, edge_selectors=TemporalBackedge([1, 2, 4]), aux_edge_selectors=aux,
                       positional_encoder=pe, graph_size=N, fused=fused)
        hidden, outs = None, []
        for t in range(T):
            mx, hidden = mem(obs[t], hidden)
            outs.append(mx)
        out = torch.stack(outs)
        (out * torch.linspace(0.5, 1.5, out.numel(), device=DEV).view_as(out)).sum().backward()
        mem.check_flags()
        if fused:
            assert mem._cfg_last[3].fold is not None
        grads = {k: p.grad.clone() for k, p in g.named_parameters()}
        if pre is not None:
            grads.update({"pre." + k: p.grad.clone() for k, p in pre.named_parameters()})
        res.append((out.detach(), hidden, grads))
    torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-5, atol=5e-6)
    assert torch.equal(res[0][1][0], res[1][1][0]) and torch.equal(res[0][1][1], res[1][1][1])
    for k in res[1][2]:
        scale = float(res[1][2][k].abs().max()) + 1e-12
        torch.testing.assert_close(res[0][2][k], res[1][2][k], rtol=1e-5, atol=2e-5 * scale, msg=k)
