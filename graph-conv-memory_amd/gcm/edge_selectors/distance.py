"""Distance-threshold edge selectors (reference: src/gcm/edge_selectors/distance.py).

One fused kernel computes the distances of the new node to the stored nodes,
thresholds them and writes the adjacency row (csrc/distance.hip); the [B,N]
distance matrix is never materialised."""
import torch

from .. import _hip, _ops


class Distance(torch.nn.Module):
    """distance.py:4-39 - edge (n_b <- j) for every j < n_b with dist < max_distance."""

    mode = None

    def __init__(self, max_distance, bidirectional=False, learned=False):
        super().__init__()
        self.max_distance = max_distance
        self.bidirectional = bidirectional
        self.learned = learned
        if learned:
            # distance.py:13-16 - the node matrix is divided by dist_param, threshold 1
            self.dist_param = torch.nn.Parameter(torch.Tensor([max_distance]))
            self.max_distance = 1.0

    def _slices(self, F):
        return (0, F), (0, F)

    def native_desc(self, F):
        a, b = self._slices(F)
        return _hip.SelectorDesc(
            kind=_hip.SEL_DISTANCE, mode=self.mode, max_distance=float(self.max_distance),
            dist_param=self.dist_param.data_ptr() if self.learned else None,
            a0=a[0], a1=a[1], b0=b[0], b1=b[1], bidirectional=int(self.bidirectional))

    def forward(self, nodes, adj_mats, edge_weights, num_nodes, B):
        a, b = self._slices(nodes.shape[-1])
        param = self.dist_param.detach() if self.learned else None
        target = torch.zeros_like(adj_mats) if adj_mats.requires_grad else adj_mats
        _ops.edge_distance_(nodes.detach(), target, num_nodes, self.mode, self.max_distance,
                            dist_param=param, a=a, b=b, bidirectional=self.bidirectional)
        if adj_mats.requires_grad:
            return torch.where(target > 0, target, adj_mats), edge_weights
        return adj_mats, edge_weights

    def distances(self, nodes, num_nodes):
        """The [B,N] matrix the threshold is applied to (debug / tests)."""
        a, b = self._slices(nodes.shape[-1])
        param = self.dist_param.detach() if self.learned else None
        B, N, _ = nodes.shape
        scratch = torch.zeros(B, N, N, device=nodes.device)
        _, d = _ops.edge_distance_(nodes.detach(), scratch, num_nodes, self.mode,
                                   self.max_distance, dist_param=param, a=a, b=b, want_dist=True)
        return d


class EuclideanEdge(Distance):
    """distance.py:42-49.  NB reference semantics: the distance of stored node (b, j) is the
    MEAN over all graphs b' of ||current[b'] - nodes[b, j]|| (cdist broadcast + mean(dim=1))."""

    mode = _hip.DIST_EUCLID_CROSSBATCH

    def __init__(self, max_distance, learned=False):
        super().__init__(max_distance, learned=learned)


class CosineEdge(Distance):
    """distance.py:52-61 - cosine SIMILARITY (eps 1e-8), edge when similarity < max_distance."""

    mode = _hip.DIST_COSINE_SIM

    def __init__(self, max_distance, learned=False):
        super().__init__(max_distance, learned=learned)


class SpatialEdge(Distance):
    """distance.py:64-81 - per-graph L2 between pose slices of the latent vectors."""

    mode = _hip.DIST_L2_PERGRAPH

    def __init__(self, max_distance, a_pose_slice, b_pose_slice=None, learned=False):
        super().__init__(max_distance, learned=learned)
        self.a_pose_slice = a_pose_slice
        self.b_pose_slice = b_pose_slice if b_pose_slice else a_pose_slice

    def _slices(self, F):
        def rng(s):
            start, stop, step = s.indices(F)
            if step != 1:
                raise NotImplementedError("pose slices must be contiguous")
            return start, stop
        return rng(self.a_pose_slice), rng(self.b_pose_slice)
