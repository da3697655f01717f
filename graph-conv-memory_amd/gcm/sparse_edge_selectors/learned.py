"""Sparse LearnedEdge (reference: src/gcm/sparse_edge_selectors/learned.py:12-160)."""
import functools
from typing import Tuple, Union

import torch

from .. import _ops, util


class LearnedEdge(torch.nn.Module):
    """Sample incoming edges of the new nodes from a learned prior: an MLP scores every causal
    candidate pair (sink = new node, source = any earlier node inside `window`), a gumbel
    softmax runs over each sink's candidates and candidates above 1/(1+num_edge_samples)
    become edges whose weights keep the gradient path to the logits (value / value.detach()).

    Same constructor and return value as the reference (a torch.sparse_coo [B,N,N] with indices
    (batch, sink, source)).  Candidates are enumerated in closed form on the device, the pair
    matrix and its adjoint are gather kernels, the softmax over sink rows is one wave per row;
    the reference's default edge network (recognised by its structure) runs on the hand-written row kernels
    (gcm_rows_linear / gcm_skinny_wgrad / gcm_relu_layernorm_*: no library GEMM in the trace since round 3), any
    other user-supplied network is called as the torch module it is.  `noise_fn(logits) -> gumbel
    noise [E]` may be set to inject the random draws (parity tests)."""

    def __init__(self, input_size: int = 0, model: Union[None, torch.nn.Module] = None,
                 num_edge_samples: int = 5, deterministic: bool = False,
                 window: Union[int, None] = None, log_stats: bool = True,
                 softmax_temp: float = 1.0, learn_softmax_temp: bool = True,
                 temp_bounds: Tuple[float, float] = (0.001, 5), store_grads: bool = True):
        super().__init__()
        assert model or input_size, "Must specify either input_size or model"
        self.deterministic = deterministic
        self.num_edge_samples = num_edge_samples
        self.store_grads = store_grads
        self.edge_network = self.build_edge_network(input_size) if model is None else model
        self.ste = util.StraightThroughEstimator()
        self.window = window
        self.log_stats = log_stats
        self.stats = {}
        self.tau_param = torch.tensor([softmax_temp])
        self.temp_bounds = temp_bounds
        if learn_softmax_temp:
            self.tau_param = torch.nn.Parameter(self.tau_param)
        self.noise_fn = None

    def init_weights(self, m):
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.orthogonal_(m.weight)

    def grad_hook(self, p_name, grad):
        self.stats[f"gnorm_{p_name}"] = grad.norm().detach().item()

    def build_edge_network(self, input_size: int) -> torch.nn.Sequential:
        """learned.py:69-88: (i || j) -> logit(edge(i, j)), orthogonal init.  nn.Linear subclasses
        with a row-split weight-gradient kernel (the candidate list has O(sum T^2) rows)."""
        from ..nn import SkinnyLinear
        m = torch.nn.Sequential(
            SkinnyLinear(2 * input_size, input_size),
            torch.nn.ReLU(),
            torch.nn.LayerNorm(input_size),
            SkinnyLinear(input_size, input_size),
            torch.nn.ReLU(),
            torch.nn.LayerNorm(input_size),
            SkinnyLinear(input_size, 1),
        )
        m.apply(self.init_weights)
        if self.store_grads:
            for n, p in m.named_parameters():
                p.register_hook(functools.partial(self.grad_hook, n))
        return m

    new_sinks_only = True   # every sampled edge ends in a NEW node: SparseGCM merges without a sort

    def forward(self, nodes, T, taus, B):
        N = nodes.shape[1]
        if list(self.parameters())[0].device != nodes.device:
            self.to(nodes.device)
        if self.tau_param.device != nodes.device:
            self.tau_param = self.tau_param.to(nodes.device)
        edges = _ops.CausalEdges(T, taus, self.window)          # learned.py:117 (one readback)
        if edges.E == 0:                                        # learned.py:99-105
            return torch.sparse_coo_tensor(
                indices=torch.zeros(3, 0, dtype=torch.long, device=nodes.device),
                values=torch.zeros(0, device=nodes.device), size=(B, N, N))
        pairs = _ops.causal_pairs(nodes, edges)                 # [E, 2F]  learned.py:121-124
        logits = _ops.edge_network_forward(self.edge_network, pairs).squeeze(-1)
        cutoff = 1 / (1 + self.num_edge_samples)
        self.tau_param.data.clamp_(*self.temp_bounds)
        if self.deterministic:   # util.sparse_tempered_softmax: the same softmax without noise
            noise = torch.zeros_like(logits)
        elif self.noise_fn is not None:
            noise = self.noise_fn(logits)
        else:
            noise = -torch.empty_like(logits).exponential_().log()
        soft = _ops.segment_softmax(logits, self.tau_param, noise, edges)
        mask = soft > cutoff                                    # learned.py:143-151
        kept = soft[mask]
        # (a subset of the closed-form candidate list: still in coalesced order, duplicate free)
        adj = torch.sparse_coo_tensor(indices=edges.indices[:, mask], values=kept / kept.detach(),
                                      size=(B, N, N), is_coalesced=True)
        if self.log_stats:                                      # learned.py:153-159
            self.stats["edges_per_node"] = (kept.numel() / taus.sum().detach()).item()
            self.stats["edge_density"] = kept.numel() / edges.E
            self.stats["logits_mean"] = logits.detach().mean().item()
            self.stats["logits_var"] = logits.detach().var().item()
            self.stats["temperature"] = self.tau_param.detach().item()
        return adj
