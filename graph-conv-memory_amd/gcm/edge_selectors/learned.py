"""LearnedEdge (reference: src/gcm/edge_selectors/learned.py:7-125)."""
import torch

from .. import _ops, util


class LearnedEdge(torch.nn.Module):
    """An edge selector whose prior is learned: an MLP scores every candidate edge
    (new node <- earlier node), edges are sampled with a gumbel-softmax relaxation and
    made binary with a straight-through estimator, so the adjacency carries gradients.

    The candidate matrix is built on the device without nonzero()/max() round trips
    (one row per (graph, earlier node) slot, zero rows beyond the current node), the edge
    network runs on it as ordinary GEMMs, and gumbel-softmax + threshold + adjacency-row
    write are one fused kernel per direction.  `noise_fn(logits) -> gumbel noise [B, N]`
    may be set to inject the random draws (parity tests); the default draws them with the
    device RNG the way torch.nn.functional.gumbel_softmax does."""

    def __init__(self, input_size: int = 0, model: torch.nn.Sequential = None,
                 num_edge_samples: int = 5, deterministic: bool = False):
        super().__init__()
        self.deterministic = deterministic
        self.num_edge_samples = num_edge_samples
        assert input_size or model, "Must specify either input_size or model"
        self.edge_network = model if model else self.build_edge_network(input_size)
        if deterministic:
            self.sm = util.Spardmax()      # raises: dead at the reference HEAD as well
        self.ste = util.StraightThroughEstimator()
        self.noise_fn = None

    def build_edge_network(self, input_size: int) -> torch.nn.Sequential:
        """learned.py:38-51: (i || j) -> logit(edge(i, j)).  The linears are nn.Linear subclasses
        (same parameters and state_dict) whose weight gradient over the B*N candidate rows is a
        row-split kernel instead of a one-workgroup library GEMM."""
        from ..nn import SkinnyLinear
        return torch.nn.Sequential(
            SkinnyLinear(2 * input_size, input_size),
            torch.nn.ReLU(),
            torch.nn.LayerNorm(input_size),
            SkinnyLinear(input_size, input_size),
            torch.nn.ReLU(),
            torch.nn.LayerNorm(input_size),
            SkinnyLinear(input_size, 1),
        )

    def compute_new_adj(self, nodes, num_nodes, adj, B):
        """learned.py:53-113.  `adj` is rewritten in place (the caller hands over its own
        buffer, as DenseGCM does) and returned."""
        cutoff = 1 / (1 + self.num_edge_samples)
        if self.noise_fn is None:
            # default edge network + device RNG: the whole selector is one autograd node
            out = _ops.learned_edge_default(self.edge_network, nodes, adj, num_nodes, None, cutoff)
            if out is not None:
                return out
        pairs = _ops.learned_pairs(nodes, num_nodes)                  # [B, N, 2F]
        logits = _ops.edge_network_forward(self.edge_network, pairs).squeeze(-1)   # [B, N]
        if self.noise_fn is not None:
            noise = self.noise_fn(logits)
        else:
            noise = -torch.empty_like(logits).exponential_().log()
        return _ops.learned_select_(adj, logits, noise, num_nodes, cutoff)

    def forward(self, nodes, adj, weights, num_nodes, B):
        if self.edge_network[0].weight.device != nodes.device:
            self.edge_network = self.edge_network.to(nodes.device)
        if adj.is_leaf and adj.requires_grad:
            adj = adj.clone()
        return self.compute_new_adj(nodes, num_nodes, adj, B), weights
