"""Plain-torch restatement of the four torch_geometric symbols the hot path
calls.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

torch_geometric is a third-party dependency of the reference (setup.cfg:24,
`torch_geometric>=1.7.0`), absent from /root/reference and from this image.
The formulae below restate its published algorithm (PyG 2.x layout):

* DenseGraphConv  - call sites README.md:56-62, tests/test_gcm.py:95-99,249-256
* GraphConv       - call sites ray_sparse_gcm.py:37-39, tests/test_sparse_gcm.py:322-323
* coalesce        - call site  src/gcm/sparse_gcm.py:173-175
* k_hop_subgraph  - call site  src/gcm/sparse_gcm.py:192-198
* Sequential      - call sites src/gcm/gcm.py:158,165 (docs), every test

Parity versus real PyG numerics: UNPINNED in this container.
"""

import torch


class DenseGraphConv(torch.nn.Module):
    """out = lin_rel(adj @ x) + lin_root(x); lin_rel carries the bias
    (modern PyG parameter layout, key-compatible with GraphConv as
    tests/test_sparse_gcm.py:326-330 requires)."""

    def __init__(self, in_channels, out_channels, aggr="add", bias=True):
        super().__init__()
        assert aggr == "add"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_rel = torch.nn.Linear(in_channels, out_channels, bias=bias)
        self.lin_root = torch.nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, x, adj, mask=None):
        x = x.unsqueeze(0) if x.dim() == 2 else x
        adj = adj.unsqueeze(0) if adj.dim() == 2 else adj
        out = self.lin_rel(torch.matmul(adj, x))
        out = out + self.lin_root(x)
        if mask is not None:
            out = out * mask.view(x.shape[0], x.shape[1], 1).to(x.dtype)
        return out


class GraphConv(torch.nn.Module):
    """out[i] = lin_rel(sum_{(j->i) in E} w_ji * x_j) + lin_root(x_i);
    edge_index[0] = source j, edge_index[1] = target i."""

    def __init__(self, in_channels, out_channels, aggr="add", bias=True):
        super().__init__()
        assert aggr == "add"
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_rel = torch.nn.Linear(in_channels, out_channels, bias=bias)
        self.lin_root = torch.nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, x, edge_index, edge_weight=None):
        src, dst = edge_index[0], edge_index[1]
        msg = x.index_select(0, src)
        if edge_weight is not None and edge_weight.numel() == msg.shape[0]:
            msg = msg * edge_weight.view(-1, 1)
        agg = torch.zeros_like(x).index_add_(0, dst, msg)
        return self.lin_rel(agg) + self.lin_root(x)


def coalesce(edge_index, edge_attr=None, num_nodes=None, reduce="sum"):
    """Sort edges by (row, col), merge duplicates reducing edge_attr."""
    if num_nodes is None:
        num_nodes = int(edge_index.max()) + 1 if edge_index.numel() else 0
    key = edge_index[0] * num_nodes + edge_index[1]
    key, perm = torch.sort(key, stable=True)
    edge_index = edge_index[:, perm]
    first = torch.ones_like(key, dtype=torch.bool)
    first[1:] = key[1:] != key[:-1]
    if edge_attr is None:
        return edge_index[:, first]
    edge_attr = edge_attr[perm]
    if bool(first.all()):
        return edge_index, edge_attr
    seg = first.cumsum(0) - 1
    n_out = int(seg[-1]) + 1
    out = torch.zeros((n_out,) + edge_attr.shape[1:], dtype=edge_attr.dtype)
    out = out.index_add(0, seg, edge_attr)
    if reduce == "mean":
        cnt = torch.zeros(n_out, dtype=edge_attr.dtype).index_add(
            0, seg, torch.ones_like(key, dtype=edge_attr.dtype)
        )
        out = out / cnt.clamp(min=1).view((-1,) + (1,) * (out.dim() - 1))
    else:
        assert reduce in ("sum", "add")
    return edge_index[:, first], out


def k_hop_subgraph(node_idx, num_hops, edge_index, relabel_nodes=False, num_nodes=None):
    """BFS of `num_hops` steps from `node_idx` against the edge direction
    (flow = source_to_target).  Returns (subset, edge_index, inv, edge_mask)."""
    num_nodes = int(num_nodes)
    src, dst = edge_index[0], edge_index[1]
    frontier = [node_idx]
    node_mask = torch.zeros(num_nodes, dtype=torch.bool)
    for _ in range(num_hops):
        node_mask.fill_(False)
        node_mask[frontier[-1]] = True
        frontier.append(src[node_mask[dst]])
    subset, inv = torch.cat(frontier).unique(return_inverse=True)
    inv = inv[: node_idx.numel()]
    node_mask.fill_(False)
    node_mask[subset] = True
    edge_mask = node_mask[src] & node_mask[dst]
    edge_index = edge_index[:, edge_mask]
    if relabel_nodes:
        relabel = torch.full((num_nodes,), -1, dtype=torch.long)
        relabel[subset] = torch.arange(subset.numel())
        edge_index = relabel[edge_index]
    return subset, edge_index, inv, edge_mask


class Sequential(torch.nn.Module):
    """String-signature module chain: Sequential("x, adj, w, B, N",
    [(mod, "x, adj -> x"), act, ...]).  A bare module maps the first
    declared name to itself."""

    def __init__(self, input_args, modules):
        super().__init__()
        self.arg_names = [a.strip() for a in input_args.split(",")]
        self.plan = []
        for i, entry in enumerate(modules):
            if isinstance(entry, (tuple, list)):
                mod, sig = entry
                ins, outs = sig.split("->")
                ins = [a.strip() for a in ins.split(",")]
                outs = [a.strip() for a in outs.split(",")]
            else:
                mod = entry
                prev_out = self.plan[-1][2] if self.plan else self.arg_names[:1]
                ins, outs = list(prev_out), list(prev_out)
            self.add_module(f"module_{i}", mod)
            self.plan.append((f"module_{i}", ins, outs))

    def forward(self, *args):
        env = dict(zip(self.arg_names, args))
        out = None
        for name, ins, outs in self.plan:
            out = getattr(self, name)(*[env[k] for k in ins])
            if len(outs) == 1:
                env[outs[0]] = out
            else:
                for k, v in zip(outs, out):
                    env[k] = v
        return out
