"""Helpers shared by the parity tests: load a golden fixture, rebuild the
oracle objects it describes."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Fixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(str(z["meta"]))
        self.t = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}

    def __getitem__(self, k):
        return self.t[k]

    def __contains__(self, k):
        return k in self.t

    def group(self, prefix):
        return {k[len(prefix):]: v for k, v in self.t.items() if k.startswith(prefix)}

    def h0(self):
        if "h0_nodes" not in self.t:
            return None
        w = self.t["h0_weights"] if "h0_weights" in self.t else torch.zeros(0)
        return (self["h0_nodes"], self["h0_adj"], w, self["h0_num_nodes"])


def oracle_selector(meta, sel_params=None, noise=None):
    """Oracle selector object for a fixture's meta block."""
    from oracle import dense as od

    kind = meta["selector"]
    if kind == "temporal":
        return od.TemporalBackedge(meta["hops"], meta["direction"])
    if kind == "dense":
        return od.DenseEdge()
    dist_param = None
    if meta.get("learned"):
        dist_param = sel_params["dist_param"]
    if kind == "euclid":
        return od.EuclideanEdge(meta["max_distance"], dist_param=dist_param)
    if kind == "cosine":
        return od.CosineEdge(meta["max_distance"], dist_param=dist_param)
    if kind == "spatial":
        return od.SpatialEdge(meta["max_distance"], slice(*meta["a"]), slice(*meta["b"]),
                              dist_param=dist_param)
    raise KeyError(kind)


def fp64_bound(ref, obs, hidden, out32, **kw):
    """Summation-order noise: when an aggregate adds up to N terms of magnitude ~1, two fp32
    evaluations in different orders differ by more than 1e-5 relative.  The bound used instead:
    distance to the SAME computation in float64 (the oracle run in double), which must not exceed 3x
    the distance of the reference's own fp32 evaluation from it (floor 2e-6).  -> (out64, atol)."""
    import copy
    from oracle import dense as od
    ref64 = copy.deepcopy(ref).double()
    h64 = None if hidden is None else tuple(t.double() if t.is_floating_point() else t.clone() for t in hidden)
    with torch.no_grad():
        out64, _ = od.dense_rollout(obs.double(), h64, ref64, **kw)
    err32 = float((out32.detach().double() - out64).abs().max())
    return out64, max(2e-6, 3.0 * err32)


def fp64_grad_bound(ref, fx, sel, loss="mean", floor=5e-7, factor=3.0):
    """Gradient tolerances from the same rollout evaluated in float64 by the oracle: for every GNN
    parameter (and the observations) -> (g64, atol) with atol = max(factor x the error of the REFERENCE's
    own fp32 gradient against float64, floor x the gradient's scale).  In practice ~1e-6 of the
    gradient scale - two orders tighter than a 1e-4 rtol - and it scales with what fp32 can deliver for
    the case at hand (long sums: DenseEdge) instead of a fixed number."""
    import copy
    from oracle import dense as od
    m = fx.meta
    ref64 = copy.deepcopy(ref).double()
    h0 = fx.h0()
    h64 = None if h0 is None else tuple(t.double() if t.is_floating_point() else t.clone() for t in h0)
    obs64 = fx["obs"].double().requires_grad_(True)
    out64, _ = od.dense_rollout(obs64, h64, ref64, graph_size=m["N"], edge_selectors=sel)
    assert loss == "mean"
    out64.mean().backward()
    res = {}
    for k, p in list(ref64.named_parameters()) + [("obs", obs64)]:
        g64 = p.grad
        want32 = fx["grad_obs"] if k == "obs" else fx["grad:" + k]
        scale = float(g64.abs().max())
        err_ref = float((want32.double() - g64).abs().max())
        res[k] = (g64, max(factor * err_ref, floor * scale))
    return res


# ---- G15 (folded preprocessor / aux selectors / positional encoding): how each fixture's module
# was put together in tests/golden/make_golden.py -----------------------------------------------
FOLD_SPECS = {
    # name: (selector, aux selector): ("temporal", hops, direction) | ("dense",) | None
    "g15_fold_pre": (("temporal", [1, 2], "forward"), None),
    "g15_fold_pre_nobias": (("dense",), None),
    "g15_fold_pre_aux_cat": (("temporal", [1], "forward"), ("temporal", [3], "both")),
    "g15_fold_pe_add": (("temporal", [1], "forward"), ("temporal", [2], "forward")),
    "g15_fold_pre_exact": (("temporal", [1, 2, 4], "forward"), None),
    "g15_fold_pe_add_exact": (("temporal", [1, 2, 4], "forward"), ("dense",)),
}


def fold_selector(spec, temporal_cls, dense_cls):
    if spec is None:
        return None
    if spec[0] == "dense":
        return dense_cls()
    return temporal_cls(spec[1], direction=spec[2])


def fp64_rollout_bounds(ref, obs, hidden, weight, sel_factory, graph_size, factor=3.0, floor=5e-7):
    """The oracle on (obs, hidden) in fp32 and in float64, loss = sum(out * weight): ->
    (out32, final hidden32, {param name: (g64, atol)}, (out64, out_atol)) with
    atol = max(factor x |reference-fp32 - fp64|, floor x scale) - what fp32 can deliver for the case,
    instead of a fixed rtol.  sel_factory() -> a fresh oracle selector (stateless ones may be shared)."""
    import copy
    from oracle import dense as od
    ref.zero_grad(set_to_none=True)
    h32 = None if hidden is None else tuple(t.clone() for t in hidden)
    out32, hid32 = od.dense_rollout(obs, h32, ref, graph_size=graph_size, edge_selectors=sel_factory())
    (out32 * weight).sum().backward()
    ref64 = copy.deepcopy(ref).double()
    ref64.zero_grad(set_to_none=True)
    h64 = None if hidden is None else tuple(t.double() if t.is_floating_point() else t.clone() for t in hidden)
    out64, _ = od.dense_rollout(obs.double(), h64, ref64, graph_size=graph_size, edge_selectors=sel_factory())
    (out64 * weight.double()).sum().backward()
    bounds = {}
    for (k, p32), (_, p64) in zip(ref.named_parameters(), ref64.named_parameters()):
        scale = float(p64.grad.abs().max())
        err = float((p32.grad.double() - p64.grad).abs().max())
        bounds[k] = (p64.grad, max(factor * err, floor * scale))
    err_o = float((out32.detach().double() - out64.detach()).abs().max())
    return out32.detach(), hid32, bounds, (out64.detach(), max(2e-6, factor * err_o))
