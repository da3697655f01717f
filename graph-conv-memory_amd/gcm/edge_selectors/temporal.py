"""TemporalBackedge (reference: src/gcm/edge_selectors/temporal.py:17-94).

adj[b, i, j] = 1 means node i aggregates from node j (temporal.py:7-15)."""
from typing import List

import torch

from .. import _ops
from ..util import sparsemax as _sparsemax


class TemporalBackedge(torch.nn.Module):
    """Add temporal directional back edges, e.g. node_t <- node_{t-hop}."""

    def __init__(self, hops: List[int] = [1], direction="forward", learned=False,
                 learning_window=10, deterministic=False, num_samples=3):
        super().__init__()
        assert direction in ["forward", "backward", "both"]
        self.hops = list(hops)
        self.direction = direction
        self.learned = learned
        if learned:                   # temporal.py:44-50
            self.window = torch.nn.Parameter(torch.ones(learning_window))
            self.num_samples = num_samples
            self.deterministic = deterministic
        # test hook: callable(shape, device) -> standard gumbel noise (default: device RNG)
        self.noise_fn = None

    def native_desc(self):
        """Descriptor for the fused / rollout paths (struct gcm_selector_desc)."""
        from .. import _hip
        if self.learned or len(self.hops) > 16:   # learned windows / >16 hops: the layered path
            return None
        d = _hip.SelectorDesc(kind=_hip.SEL_TEMPORAL, n_hops=len(self.hops),
                              direction=_hip.DIR[self.direction])
        for i, h in enumerate(self.hops):
            d.hops[i] = h
        return d

    def learned_forward(self, nodes, adj_mats, edge_weights, num_nodes, B):
        """temporal.py:51-70, all graphs at once: every graph with n_b > 0 nodes draws `num_samples`
        straight-through gumbel one-hots over window[:n_b] (or one hard sparsemax when
        deterministic), ORs them (util.diff_or: res + t - res*t) and adds the result to
        adj[b, n_b, :n_b].  Like the reference this needs n_b <= learning_window (its slice
        assignment raises a shape RuntimeError beyond it)."""
        W = self.window.numel()
        N = adj_mats.shape[-1]
        if int(num_nodes.max()) > W:
            raise RuntimeError(f"TemporalBackedge(learned=True): a graph holds {int(num_nodes.max())} nodes, "
                               f"the learning window only {W}")
        Wn = min(W, N)
        cols = torch.arange(Wn, device=nodes.device)
        valid = cols[None, :] < num_nodes[:, None]                                   # [B, Wn]
        # an empty graph takes part with one dummy column so that its softmax stays finite; its
        # mask is zeroed below
        keep = valid | ((num_nodes == 0)[:, None] & (cols == 0)[None, :])
        logits = self.window[:Wn].to(nodes.device)[None, :].expand(B, Wn)
        if self.deterministic:
            soft = _sparsemax(logits.masked_fill(~keep, float("-inf")))
            mask = (soft > 0).float() - soft.detach() + soft                         # util.py:38-42
        else:
            S = self.num_samples
            if self.noise_fn is not None:
                g = self.noise_fn((S, B, Wn), nodes.device)
            else:                                                                    # F.gumbel_softmax's draw
                g = -torch.empty(S, B, Wn, device=nodes.device).exponential_().log()
            soft = torch.softmax((logits[None] + g).masked_fill(~keep[None], float("-inf")), dim=-1)
            hard = torch.zeros_like(soft).scatter_(-1, soft.argmax(-1, keepdim=True), 1.0)
            y = hard - soft.detach() + soft
            mask = torch.zeros_like(y[0])
            for s in range(S):                                                       # util.py:456-465
                mask = mask + y[s] - mask * y[s]
        mask = mask * valid
        b_idx = torch.arange(B, device=nodes.device)
        cur = num_nodes.clamp(max=N - 1)
        row = adj_mats[b_idx, cur]
        row = torch.cat([row[:, :Wn] + mask, row[:, Wn:]], dim=-1)
        adj_mats = adj_mats.index_put((b_idx, cur), row)
        return adj_mats, edge_weights

    def forward(self, nodes, adj_mats, edge_weights, num_nodes, B):
        """temporal.py:72-88: for every hop and every graph with num_nodes >= hop set
        adj[b, n, n-hop] (forward/both) and/or adj[b, n-hop, n] (backward/both)."""
        if self.learned:
            return self.learned_forward(nodes, adj_mats, edge_weights, num_nodes, B)
        if adj_mats.requires_grad:
            mask = _ops.edge_temporal_(torch.zeros_like(adj_mats), num_nodes, self.hops,
                                       self.direction)
            return torch.where(mask > 0, mask, adj_mats), edge_weights
        _ops.edge_temporal_(adj_mats, num_nodes, self.hops, self.direction)
        return adj_mats, edge_weights

