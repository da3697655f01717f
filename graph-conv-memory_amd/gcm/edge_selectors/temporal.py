"""TemporalBackedge (reference: src/gcm/edge_selectors/temporal.py:17-94).

adj[b, i, j] = 1 means node i aggregates from node j (temporal.py:7-15)."""
from typing import List

import torch

from .. import _ops


WINDOW_ERROR = "TemporalBackedge(learned=True): a graph holds more nodes than the learning window ({})"


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
        # the flag word of the DenseGCM that drives this selector (it lends it before each call)
        self._gcm_flags = None

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
        """temporal.py:51-70, all graphs in one launch (csrc/temporal_window.hip): every graph with
        n_b > 0 nodes draws `num_samples` straight-through gumbel one-hots over window[:n_b] (or one hard
        sparsemax when deterministic), ORs them (util.diff_or: res + t - res*t) and adds the result to
        adj[b, n_b, :n_b], in place like the reference.  Like the reference this needs
        n_b <= learning_window (its slice assignment raises a shape RuntimeError beyond it): the kernel
        raises a flag instead, which DenseGCM surfaces with its other flags (finite_check) - a selector
        called on its own checks it on the spot."""
        W = self.window.numel()
        N = adj_mats.shape[-1]
        Wn = min(W, N)
        noise = None
        if not self.deterministic:
            S = self.num_samples
            if self.noise_fn is not None:
                noise = self.noise_fn((S, B, Wn), nodes.device)
            else:                                                                    # F.gumbel_softmax's draw
                noise = -torch.empty(S, B, Wn, device=nodes.device).exponential_().log()
        flags, own = self._gcm_flags, False
        if flags is None or flags.device != adj_mats.device:
            flags, own = torch.zeros(1, dtype=torch.int32, device=adj_mats.device), True
        window = self.window if self.window.device == adj_mats.device else self.window.to(adj_mats.device)
        cur = num_nodes.clamp(max=N - 1)
        adj_mats = _ops.temporal_window_(adj_mats, window, cur, noise, self.num_samples, self.deterministic, flags)
        if own and int(flags.item()):
            raise RuntimeError(WINDOW_ERROR.format(W))
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

