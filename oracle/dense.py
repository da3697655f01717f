"""CPU restatement of the DenseGCM step and the dense edge selectors.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Each function names the reference lines it follows (paths are relative to the
reference checkout, src/gcm/...).  The restatement is functional: it never
mutates its inputs; the reference's one in-place side effect on the caller's
`num_nodes` (gcm.py:354) is reproduced in the RETURNED values only.
"""
import torch


# --------------------------------------------------------------------------
# state helpers
# --------------------------------------------------------------------------
def initial_hidden(x, graph_size, edge_weights=False):
    """gcm.py:194-211 - all-zero (nodes, adj, weights, num_nodes)."""
    B, F = x.shape
    dt = x.dtype                      # float32 like the reference; float64 for error bounds in tests
    nodes = torch.zeros(B, graph_size, F, dtype=dt)
    adj = torch.zeros(B, graph_size, graph_size, dtype=dt)
    weights = torch.zeros(B, graph_size, graph_size, dtype=dt) if edge_weights else torch.zeros(0, dtype=dt)
    return nodes, adj, weights, torch.zeros(B, dtype=torch.long)


def wrap_overflow(nodes, adj, weights, num_nodes):
    """gcm.py:323-355 - for graphs with num_nodes + 1 > N: clear node 0 and its
    row/col, rotate everything one slot towards index 0, num_nodes -= 1."""
    N = nodes.shape[1]
    full = num_nodes + 1 > N
    nodes, adj, num_nodes = nodes.clone(), adj.clone(), num_nodes.clone()
    weights = weights.clone()
    for b in torch.nonzero(full).flatten().tolist():
        nodes[b, 0] = 0
        adj[b, 0, :] = 0
        adj[b, :, 0] = 0
        nodes[b] = torch.roll(nodes[b], -1, 0)
        adj[b] = torch.roll(adj[b], (-1, -1), (0, 1))
        if weights.numel():
            weights[b, 0, :] = 0
            weights[b, :, 0] = 0
            weights[b] = torch.roll(weights[b], (-1, -1), (0, 1))
        num_nodes[b] -= 1
    return nodes, adj, weights, num_nodes


# --------------------------------------------------------------------------
# edge selectors: f(nodes, adj, weights, num_nodes, B) -> (adj, weights)
# --------------------------------------------------------------------------
def sparsemax(z):
    """Sparsemax of a 1-D tensor (Martins & Astudillo 2016, Algorithm 1) - what the third-party
    `sparsemax` package (absent from /root/reference; imported at util.py:5) computes for
    util.Spardmax (util.py:29-42).  Parity unpinned: published algorithm only."""
    zs, _ = torch.sort(z, descending=True)
    k = torch.arange(1, z.numel() + 1, dtype=z.dtype)
    csum = zs.cumsum(0)
    kz = int((1 + k * zs > csum).sum())
    tau = (csum[kz - 1] - 1) / kz
    return torch.clamp(z - tau, min=0)


class TemporalBackedge:
    """edge_selectors/temporal.py:72-88 (deterministic_forward) and :51-70 (learned_forward).

    learned=True: `window` is the learnable logit vector; `noise_fn(n)` returns the standard
    gumbel noise of one torch.nn.functional.gumbel_softmax call over n logits (calls come in the
    reference's order: graphs ascending, samples ascending) - default: drawn like torch does."""

    def __init__(self, hops=(1,), direction="forward", learned=False, learning_window=10,
                 deterministic=False, num_samples=3, noise_fn=None):
        assert direction in ("forward", "backward", "both")
        self.hops, self.direction = list(hops), direction
        self.learned = learned
        if learned:
            self.window = torch.ones(learning_window, requires_grad=True)
            self.num_samples, self.deterministic = num_samples, deterministic
            self.noise_fn = noise_fn or (lambda n: -torch.empty(n).exponential_().log())

    def _gumbel_hard(self, logits):
        # torch.nn.functional.gumbel_softmax(logits, tau=1, hard=True)
        soft = torch.softmax(logits + self.noise_fn(logits.numel()), dim=-1)
        hard = torch.zeros_like(soft)
        hard[int(soft.argmax())] = 1.0
        return hard - soft.detach() + soft

    def _learned(self, nodes, adj, weights, num_nodes, B):
        adj = adj.clone()
        for b in range(B):
            n = int(num_nodes[b])
            if n == 0:
                continue
            rel = self.window[:n]
            if self.deterministic:
                soft = sparsemax(rel)
                mask = (soft > 0).float() - soft.detach() + soft          # util.py:38-42
            else:
                mask = torch.zeros_like(rel)
                for _ in range(self.num_samples):                         # util.diff_or, util.py:456-465
                    y = self._gumbel_hard(rel)
                    mask = mask + y - mask * y
            adj[b, n, :n] = adj[b, n, :n] + mask          # (shape error when n > learning_window)
        return adj, weights

    def __call__(self, nodes, adj, weights, num_nodes, B):
        if self.learned:
            return self._learned(nodes, adj, weights, num_nodes, B)
        adj = adj.clone()
        for hop in self.hops:
            ok = torch.nonzero(num_nodes >= hop).flatten()
            cur = num_nodes[ok]
            if self.direction in ("forward", "both"):
                adj[ok, cur, cur - hop] = 1
            if self.direction in ("backward", "both"):
                adj[ok, cur - hop, cur] = 1
        return adj, weights


class DenseEdge:
    """edge_selectors/dense.py:11-23 - new node <-> every earlier node, plus
    a self edge."""

    def __call__(self, nodes, adj, weights, num_nodes, B):
        adj = adj.clone()
        for b in range(B):
            i = int(num_nodes[b])
            adj[b, i, : i + 1] = 1
            adj[b, :i, i] = 1
        return adj, weights


class _Distance:
    """edge_selectors/distance.py:18-39 - edge (n_b <- j) for every j < n_b
    whose dist_fn value is < max_distance."""

    def __init__(self, max_distance, bidirectional=False, dist_param=None):
        self.max_distance, self.bidirectional = max_distance, bidirectional
        # learned=True: nodes are divided by dist_param and the threshold is 1
        self.dist_param = dist_param
        if dist_param is not None:
            self.max_distance = 1.0

    def distances(self, nodes, num_nodes):
        if self.dist_param is not None:
            nodes = nodes / self.dist_param
        B = nodes.shape[0]
        cur = nodes[torch.arange(B), num_nodes]
        return self.dist_fn(cur, nodes)

    def __call__(self, nodes, adj, weights, num_nodes, B):
        adj = adj.clone()
        d = self.distances(nodes, num_nodes)
        N = nodes.shape[1]
        hit = (d < self.max_distance) & (torch.arange(N)[None, :] < num_nodes[:, None])
        b_idx, j_idx = torch.nonzero(hit, as_tuple=True)
        adj[b_idx, num_nodes[b_idx], j_idx] = 1
        if self.bidirectional:
            adj[b_idx, j_idx, num_nodes[b_idx]] = 1
        return adj, weights


class EuclideanEdge(_Distance):
    """distance.py:42-49 - NB cross-batch: d[b, j] = mean over ALL b' of
    ||cur[b'] - nodes[b, j]||  (cdist of [B,F] against [B,N,F] broadcasts to
    [B,B,N]; .mean(dim=1) averages the b' axis)."""

    def dist_fn(self, cur, nodes):
        return torch.cdist(cur, nodes).mean(dim=1)


class CosineEdge(_Distance):
    """distance.py:52-61 - cosine SIMILARITY (eps 1e-8) thresholded with '<'."""

    def dist_fn(self, cur, nodes):
        rep = cur.unsqueeze(1).expand(-1, nodes.shape[1], -1)
        return torch.nn.functional.cosine_similarity(rep, nodes, dim=2, eps=1e-8)


class SpatialEdge(_Distance):
    """distance.py:64-81 - per-graph L2 between pose slices."""

    def __init__(self, max_distance, a_pose_slice, b_pose_slice=None, **kw):
        super().__init__(max_distance, **kw)
        self.sa = a_pose_slice
        self.sb = b_pose_slice if b_pose_slice else a_pose_slice

    def dist_fn(self, cur, nodes):
        rep = torch.cat([cur.unsqueeze(1)] * nodes.shape[1], dim=1)
        return torch.cdist(rep[:, :, self.sa], nodes[:, :, self.sb]).mean(dim=1)


class _STE(torch.autograd.Function):
    """util.py:9-26 - forward (x > 0), backward identity."""

    @staticmethod
    def forward(ctx, x):
        return (x > 0).to(x.dtype)   # (`.float()` in the reference; dtype-generic for the float64 bound of the tests)

    @staticmethod
    def backward(ctx, g):
        return g


def build_edge_network(F):
    """learned.py:38-51."""
    L = torch.nn
    return L.Sequential(
        L.Linear(2 * F, F), L.ReLU(), L.LayerNorm(F),
        L.Linear(F, F), L.ReLU(), L.LayerNorm(F),
        L.Linear(F, 1),
    )


class LearnedEdge:
    """edge_selectors/learned.py:53-125 (non-deterministic branch).  The
    gumbel noise is drawn through `noise_fn(shape)` so a test can inject the
    draws the reference made (default: fresh torch-RNG gumbels, exactly
    torch.nn.functional.gumbel_softmax's `-log(Exp(1))`)."""

    def __init__(self, edge_network, num_edge_samples=5, noise_fn=None):
        self.net, self.k = edge_network, num_edge_samples
        self.noise_fn = noise_fn or (lambda shape: -torch.empty(shape).exponential_().log())

    def __call__(self, nodes, adj, weights, num_nodes, B):
        if int(num_nodes.max()) < 1:
            return adj, weights
        N = adj.shape[-1]
        past = torch.nonzero(torch.arange(N)[None, :] < num_nodes[:, None])
        b_idx, j_idx = past[:, 0], past[:, 1]
        i_idx = num_nodes[b_idx]
        pair = torch.cat((nodes[b_idx, i_idx], nodes[b_idx, j_idx]), dim=-1)
        logits = self.net(pair).squeeze()
        width = int(num_nodes.max())
        shaped = torch.full((B, width), -1e10, dtype=logits.dtype)   # (dtype-generic: the float64 bound of the tests)
        shaped = shaped.index_put((b_idx, j_idx), logits)
        soft = torch.softmax(shaped + self.noise_fn(shaped.shape), dim=-1)
        edges = _STE.apply(soft - 1.0 / (1 + self.k))
        new_adj = adj.clone()
        new_adj = new_adj.index_put(
            (b_idx, i_idx, j_idx), _STE.apply(edges[b_idx, j_idx] + adj[b_idx, i_idx, j_idx])
        )
        return new_adj, weights


def chain(*selectors):
    """Several selectors in sequence (the reference does this with
    torch_geometric.nn.Sequential, tests/test_gcm.py:646-658)."""

    def run(nodes, adj, weights, num_nodes, B):
        for s in selectors:
            adj, weights = s(nodes, adj, weights, num_nodes, B)
        return adj, weights

    return run


# --------------------------------------------------------------------------
# positional encoding (gcm.py:92-143)
# --------------------------------------------------------------------------
class PositionalEncoding:
    """gcm.py:92-143.  mode "add": rows 0..num_nodes[b] get pe[n, :F] added (the reference does
    it in place on the dirty node matrix, so the GNN sees the encoded rows too); mode "cat": the
    first cat_dim columns become pe, the rest a learned re-projection of the features."""

    def __init__(self, max_len=5000, mode="add", cat_dim=8, reproject=None):
        import math
        self.max_len, self.mode, self.cat_dim, self.reproject = max_len, mode, cat_dim, reproject
        self._math = math
        self.pe = None

    def table(self, F):
        math = self._math
        d_model = math.ceil(F / 2) * 2
        pos = torch.arange(self.max_len).unsqueeze(1)
        div = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(self.max_len, d_model)
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        return pe

    def __call__(self, x, num_nodes):
        B, N, F = x.shape
        if self.pe is None:
            self.pe = self.table(F)
        live = (torch.arange(N)[None, :] <= num_nodes[:, None]).unsqueeze(-1)
        if self.mode == "add":
            return x + live * self.pe[:N, :F].unsqueeze(0)
        enc = torch.cat((self.pe[:N, : self.cat_dim].unsqueeze(0).expand(B, -1, -1),
                         self.reproject(x)), dim=-1)
        return torch.where(live, enc, x)


# --------------------------------------------------------------------------
# the step
# --------------------------------------------------------------------------
def dense_step(x, hidden, gnn, graph_size=128, edge_selectors=None, preprocessor=None,
               aux_edge_selectors=None, positional_encoder=None, pooled=False,
               edge_weights=False):
    """gcm.py:213-321 - one DenseGCM.forward.  Returns (mx, new_hidden)."""
    if hidden is None:
        hidden = initial_hidden(x, graph_size, edge_weights)
    nodes, adj, weights, num_nodes = hidden
    assert x.dtype == nodes.dtype == weights.dtype and x.dtype in (torch.float32, torch.float64)
    assert num_nodes.dtype == torch.long and num_nodes.dim() == 1
    B, N = x.shape[0], nodes.shape[1]
    assert N == adj.shape[1] == adj.shape[2]
    rows = torch.arange(B)

    nodes = nodes.clone()
    if bool(torch.any(num_nodes + 1 > N)):                       # gcm.py:263
        nodes, adj, weights, num_nodes = wrap_overflow(nodes, adj, weights, num_nodes)
    nodes = nodes.index_put((rows, num_nodes), x)                # gcm.py:274
    dirty = nodes.clone()                                        # gcm.py:278
    if edge_selectors is not None:                               # gcm.py:284-287
        adj, weights = edge_selectors(dirty, adj.clone(), weights.clone(), num_nodes, B)
    if preprocessor is not None:                                 # gcm.py:290-291
        dirty = preprocessor(dirty)
    if aux_edge_selectors is not None:                           # gcm.py:294-306
        seen = dirty
        if positional_encoder is not None:
            seen = positional_encoder(dirty, num_nodes)
            if getattr(positional_encoder, "mode", None) == "add":
                dirty = seen        # the reference adds IN PLACE on dirty_nodes (gcm.py:131)
        adj, weights = aux_edge_selectors(seen, adj.clone(), weights.clone(), num_nodes, B)
    feats = gnn(dirty, adj, weights, B, N)                       # gcm.py:308
    mx = feats if pooled else feats[rows, num_nodes]             # gcm.py:309-314
    assert torch.all(torch.isfinite(mx)), "Got NaN in returned memory, try using tanh activation"
    return mx, (nodes, adj, weights, num_nodes + 1)


def dense_rollout(obs, hidden, gnn, **kw):
    """T calls of dense_step over obs[T, B, F]; returns (stack of mx [T,B,H], hidden)."""
    out = []
    for t in range(obs.shape[0]):
        mx, hidden = dense_step(obs[t], hidden, gnn, **kw)
        out.append(mx)
    return torch.stack(out), hidden


def canonical_gnn(F, H, act=torch.nn.Tanh, layers=2):
    """README.md:52-62 - DenseGraphConv + activation, `layers` times."""
    from . import pyg

    mods, cin = [], F
    for _ in range(layers):
        mods += [(pyg.DenseGraphConv(cin, H), "x, adj -> x"), act()]
        cin = H
    return pyg.Sequential("x, adj, weights, B, N", mods)
