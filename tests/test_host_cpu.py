"""CPU-side checks of the product package: the C ABI library loads and exports every
symbol include/gcm_hip.h declares, host logic (module wiring, state_dict keys, argument
validation) and the no-CPU-fallback rule.  No kernel is launched here."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gcm import _hip
    header = open(os.path.join(ROOT, "include", "gcm_hip.h")).read()
    declared = set(re.findall(r"\b(gcm_[a-z0-9_]+)\s*\(", header))
    assert declared, "no prototypes parsed"
    lib = _hip.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in gcm_hip.h but not exported"
    assert declared == set(_hip.PROTOTYPES), declared ^ set(_hip.PROTOTYPES)
    assert lib.gcm_version() >= 100
    assert lib.gcm_status_string(-2).decode().startswith("shape not supported")


def test_argument_errors_without_gpu():
    """Host-side validation returns GCM_EINVAL before anything touches a device."""
    from gcm import _hip
    lib = _hip.lib()
    assert lib.gcm_dense_graphconv_fwd(None, None, None, None, None, None, None, 1, 1, 1, 1, 0, None) == -1
    assert lib.gcm_edge_dense(None, None, 1, 1, None) == -1
    assert lib.gcm_dense_graphconv_bwd_workspace_bytes(256, 128, 32, 32) > 0
    assert lib.gcm_edge_distance_workspace_bytes(0, 256, 128, 64) == (256 * 64 + 256) * 4   # rows + norms


def test_no_cpu_fallback():
    from gcm.gcm import DenseGCM
    from gcm import nn as G, _hip
    g = G.Sequential("x, adj, weights, B, N", [(G.DenseGraphConv(4, 4), "x, adj -> x")])
    with pytest.raises(_hip.HipLibraryError, match="no CPU fallback"):
        DenseGCM(g, graph_size=4)(torch.zeros(2, 4), None)


def test_state_dict_keys_match_pyg_layout():
    from gcm import nn as G
    from oracle import pyg
    a = G.Sequential("x, adj, weights, B, N", [(G.DenseGraphConv(3, 5), "x, adj -> x"), torch.nn.Tanh(),
                                                (G.DenseGraphConv(5, 5), "x, adj -> x")])
    b = pyg.Sequential("x, adj, weights, B, N", [(pyg.DenseGraphConv(3, 5), "x, adj -> x"), torch.nn.Tanh(),
                                                  (pyg.DenseGraphConv(5, 5), "x, adj -> x")])
    assert list(a.state_dict()) == list(b.state_dict())
    assert "module_0.lin_rel.bias" in a.state_dict() and "module_0.lin_root.bias" not in a.state_dict()
    a.load_state_dict(b.state_dict())


def test_initial_hidden_state_and_asserts():
    from gcm.gcm import DenseGCM
    m = DenseGCM(torch.nn.Identity(), graph_size=6, edge_weights=True)
    nodes, adj, w, nn_ = m.get_initial_hidden_state(torch.zeros(3, 5))
    assert nodes.shape == (3, 6, 5) and adj.shape == (3, 6, 6) and w.shape == (3, 6, 6)
    assert nn_.dtype == torch.long and not nn_.any()
    m = DenseGCM(torch.nn.Identity(), graph_size=6)
    assert m.get_initial_hidden_state(torch.zeros(3, 5))[2].numel() == 0
    with pytest.raises(AssertionError):
        m(torch.zeros(3, 5, dtype=torch.float64), None)
    h = m.get_initial_hidden_state(torch.zeros(3, 5))
    with pytest.raises(AssertionError):
        m(torch.zeros(3, 5), (h[0], h[1], h[2], h[3].int()))


def test_selector_constructors():
    from gcm.edge_selectors.temporal import TemporalBackedge
    from gcm.edge_selectors.distance import EuclideanEdge, CosineEdge, SpatialEdge
    with pytest.raises(AssertionError):
        TemporalBackedge([1], direction="sideways")
    e = EuclideanEdge(2.0, learned=True)
    assert e.max_distance == 1.0 and float(e.dist_param) == 2.0
    assert CosineEdge(0.5).max_distance == 0.5
    s = SpatialEdge(1.0, slice(0, 3))
    assert s._slices(8) == ((0, 3), (0, 3))


def test_util_straight_through_helpers():
    """util.Spardmax / Hardmax / diff_or (util.py:29-56, 456-465): host-side torch ops, checked
    against the oracle's per-row sparsemax restatement."""
    from gcm import util
    from oracle import dense as od
    torch.manual_seed(0)
    z = torch.randn(5, 9, requires_grad=True)
    p = util.sparsemax(z)
    for r in range(5):
        torch.testing.assert_close(p[r], od.sparsemax(z[r].detach()), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(util.sparsemax(z.t(), 0).t(), p)
    hard = util.Spardmax()(z)
    assert torch.equal(hard.detach(), (p > 0).float())
    hard.sum().backward()                      # straight through: the sparsemax gradient (rows sum to 1 -> 0)
    assert float(z.grad.abs().max()) < 1e-6
    hm = util.Hardmax(cutoff=0.2)(z.detach())
    assert torch.equal(hm, (torch.softmax(z.detach(), -1) > 0.2).float())
    a, b, c = (torch.tensor(v) for v in ([0., 1., 0., 1.], [0., 0., 1., 1.], [0., 0., 0., 1.]))
    assert torch.equal(util.diff_or([a, b, c]), torch.tensor([0., 1., 1., 1.]))
