"""GNN building blocks the reference takes from torch_geometric (not vendored
by the reference; call sites README.md:52-62, tests/test_gcm.py:95-99,249-256,
tests/test_sparse_gcm.py:310-323).  Parameter names/layout follow modern PyG
(`lin_rel.{weight,bias}`, `lin_root.weight`) so state_dicts interchange between
the dense and sparse layers (tests/test_sparse_gcm.py:326-330).
"""
import torch

from . import _hip, _ops

_FUSABLE = {torch.nn.Tanh: _hip.ACT_TANH, torch.nn.ReLU: _hip.ACT_RELU}


class SkinnyLinear(torch.nn.Linear):
    """torch.nn.Linear (same parameters, same state_dict keys) for inputs with very many rows and
    <= 64 features in and out - the layers of LearnedEdge's default edge network (learned.py:38-51).
    Forward and input gradient are gcm_rows_linear (csrc/rows_linear.hip); the weight gradient runs as
    the row-split kernel gcm_skinny_wgrad.  Anything else (CPU tensors, wide layers, few rows, other dtypes)
    behaves exactly like nn.Linear."""

    MIN_ROWS = 2048

    def forward(self, x):
        if (x.is_cuda and x.dtype == torch.float32 and self.in_features <= 64 and self.out_features <= 64
                and x.numel() // self.in_features >= self.MIN_ROWS and torch.is_grad_enabled()
                and (self.weight.requires_grad or (self.bias is not None and self.bias.requires_grad))):
            return _ops.skinny_linear(x, self.weight, self.bias)
        return super().forward(x)


class DenseGraphConv(torch.nn.Module):
    """out = lin_rel(adj @ x) + lin_root(x), adj [B,N,N] float, x [B,N,F].
    Runs as one fused fp32-MFMA kernel (csrc/graphconv.hip)."""

    def __init__(self, in_channels, out_channels, aggr="add", bias=True):
        super().__init__()
        if aggr != "add":
            raise NotImplementedError("only aggr='add' (the reference's usage) is implemented")
        self.in_channels, self.out_channels, self.aggr = in_channels, out_channels, aggr
        self.lin_rel = torch.nn.Linear(in_channels, out_channels, bias=bias)
        self.lin_root = torch.nn.Linear(in_channels, out_channels, bias=False)

    def reset_parameters(self):
        self.lin_rel.reset_parameters()
        self.lin_root.reset_parameters()

    def forward(self, x, adj, mask=None, _act=_hip.ACT_NONE):
        squeeze = x.dim() == 2
        x = x.unsqueeze(0) if x.dim() == 2 else x
        adj = adj.unsqueeze(0) if adj.dim() == 2 else adj
        if adj.dtype != torch.float32:
            raise TypeError("adj must be float32 (gcm.py:203); got %s" % adj.dtype)
        if adj.shape[0] != x.shape[0]:
            adj = adj.expand(x.shape[0], -1, -1)
        if mask is not None and _act != _hip.ACT_NONE:
            raise ValueError("activation fusion is not available together with a mask")
        out = _ops.dense_graphconv(x, adj, self.lin_rel.weight, self.lin_rel.bias,
                                   self.lin_root.weight, _act)
        if mask is not None:
            out = out * mask.view(x.shape[0], x.shape[1], 1).to(x.dtype)
        return out

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels})"


class GraphConv(torch.nn.Module):
    """out[i] = lin_rel(sum_{(j->i)} w_ji * x_j) + lin_root(x_i);  edge_index [2,E] = (source,
    sink), x [M,F].  The neighbour reduction is a CSR gather fused with the two linears on
    the matrix cores (csrc/graphconv.hip, k_csr_graphconv_fwd).  When SparseGCM built the
    edge list it attaches a ready CSR (`edge_index.gcm_graph`); any other edge_index is
    indexed here on the device."""

    def __init__(self, in_channels, out_channels, aggr="add", bias=True):
        super().__init__()
        if aggr != "add":
            raise NotImplementedError("only aggr='add' (the reference's usage) is implemented")
        self.in_channels, self.out_channels, self.aggr = in_channels, out_channels, aggr
        self.lin_rel = torch.nn.Linear(in_channels, out_channels, bias=bias)
        self.lin_root = torch.nn.Linear(in_channels, out_channels, bias=False)

    def reset_parameters(self):
        self.lin_rel.reset_parameters()
        self.lin_root.reset_parameters()

    def forward(self, x, edge_index, edge_weight=None, _act=_hip.ACT_NONE):
        graph = getattr(edge_index, "gcm_graph", None)
        if graph is None or graph.M != x.shape[0]:
            graph = _ops.GraphIndex.from_edge_index(edge_index, x.shape[0])
        w = edge_weight
        if w is not None and (w.numel() != graph.E or getattr(w, "gcm_unit_weights", False)):
            # PyG: a weight vector of the wrong length is ignored.  Unit weights without a gradient
            # (what SparseGCM passes unless a learned selector is in play: sparse_gcm.py:160-164) multiply
            # by exactly 1: the kernel then skips the weight loads altogether
            w = None
        if w is not None and graph.csr_perm is not None:
            w = w[graph.csr_perm]
        return _ops.csr_graphconv(x, w, self.lin_rel.weight, self.lin_rel.bias,
                                  self.lin_root.weight, graph, _act)

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels})"


class Sequential(torch.nn.Module):
    """Stand-in for torch_geometric.nn.Sequential: a chain of modules wired by
    name, e.g. Sequential("x, adj, weights, B, N", [(conv, "x, adj -> x"), Tanh()]).
    Sub-modules are registered as module_0, module_1, ... (PyG's naming), so
    state_dicts are key compatible.  A DenseGraphConv immediately followed by a
    bare Tanh/ReLU is executed as ONE kernel (activation in the MFMA epilogue)."""

    def __init__(self, input_args, modules):
        super().__init__()
        self.arg_names = [a.strip() for a in input_args.split(",")]
        self._plan = []
        for i, entry in enumerate(modules):
            if isinstance(entry, (tuple, list)):
                mod, sig = entry
                lhs, rhs = sig.split("->")
                ins = [a.strip() for a in lhs.split(",")]
                outs = [a.strip() for a in rhs.split(",")]
            else:
                mod = entry
                prev = self._plan[-1][2] if self._plan else self.arg_names[:1]
                ins, outs = list(prev), list(prev)
            self.add_module(f"module_{i}", mod)
            self._plan.append((f"module_{i}", ins, outs))

    def stages(self):
        """[(module, inputs, outputs)] in execution order."""
        return [(getattr(self, n), i, o) for n, i, o in self._plan]

    def forward(self, *args):
        env = dict(zip(self.arg_names, args))
        plan, out, i = self._plan, None, 0
        while i < len(plan):
            name, ins, outs = plan[i]
            mod = getattr(self, name)
            fused = None
            if isinstance(mod, (DenseGraphConv, GraphConv)) and i + 1 < len(plan):
                nxt_name, nxt_in, nxt_out = plan[i + 1]
                nxt = getattr(self, nxt_name)
                if type(nxt) in _FUSABLE and nxt_in == outs and nxt_out == outs:
                    fused = _FUSABLE[type(nxt)]
            if fused is not None:
                out = mod(*[env[k] for k in ins], _act=fused)
                i += 2
            else:
                out = mod(*[env[k] for k in ins])
                i += 1
            if len(outs) == 1:
                env[outs[0]] = out
            else:
                for k, v in zip(outs, out):
                    env[k] = v
        return out
