"""Distance-threshold edge selectors (reference: src/gcm/edge_selectors/distance.py).

One fused kernel computes the distances of the new node to the stored nodes,
thresholds them and writes the adjacency row (csrc/distance.hip); the [B,N]
distance matrix is never materialised."""
import torch

from .. import _hip, _ops


class Distance(torch.nn.Module):
    """distance.py:4-39 - edge (n_b <- j) for every j < n_b with dist < max_distance."""

    mode = None
    shard_group = None      # EuclideanEdge(shard_group=...): see there
    _rows = None            # [world * B, F] current nodes of every rank (persistent: its address is baked
                            # into the step configuration)

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

    def pointer_source(self):
        """(dist_param pointer | None, gathered current rows pointer | None, their count): the device
        addresses a step configuration bakes into its selector descriptor"""
        rows = self._rows
        return (self.dist_param.data_ptr() if self.learned else None,
                rows.data_ptr() if rows is not None else None, rows.shape[0] if rows is not None else 0)

    def native_desc(self, F):
        a, b = self._slices(F)
        p, rows, n_rows = self.pointer_source()
        return _hip.SelectorDesc(
            kind=_hip.SEL_DISTANCE, mode=self.mode, max_distance=float(self.max_distance), dist_param=p,
            a0=a[0], a1=a[1], b0=b[0], b1=b[1], bidirectional=int(self.bidirectional),
            cur_rows=rows, n_cur_rows=n_rows)

    def gather_current(self, cur):
        """Batch-sharded run (shard_group set): all-gather this rank's current nodes [B, F] into the
        persistent [world * B, F] buffer the kernels read.  -> True when the buffer was (re)allocated
        (the configurations that baked its address must re-read it).
        The equal-shard check (a size all_gather) runs when a rank (re)allocates its buffer, i.e. on the first
        call and when ITS batch size or feature width changes - a rank-local condition: a change of the batch
        size must therefore happen on every rank at the same step (as a sharded batch does by construction);
        otherwise the ranks would enter different collectives."""
        import torch.distributed as dist
        group = None if self.shard_group is True else self.shard_group
        world = dist.get_world_size(group)
        fresh = False
        if (self._rows is None or self._rows.shape != (world * cur.shape[0], cur.shape[1])
                or self._rows.device != cur.device):
            # the flat all-gather (and the kernels' [world * B, F] view of its result) need equal shards: checked
            # once per (re)allocation - unequal ones (parallel.shard_bounds when total % world != 0) would hang or
            # truncate the collective and skew the cross-batch mean
            mine = torch.tensor([cur.shape[0]], dtype=torch.int64, device=cur.device)
            sizes = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(sizes, mine, group=group)
            sizes = [int(s_) for s_ in sizes]
            if any(s_ != cur.shape[0] for s_ in sizes):
                raise ValueError(f"EuclideanEdge(shard_group=...): every rank must hold the same number of graphs, "
                                 f"got {sizes} - pad the batch or shard it evenly")
            self._rows = torch.empty(world * cur.shape[0], cur.shape[1], device=cur.device)
            fresh = True
        cur = cur.detach().contiguous()
        if dist.get_backend(group) == "gloo":   # (the CPU rehearsal backend has no flat all-gather for device tensors)
            dist.all_gather(list(self._rows.chunk(world)), cur, group=group)
        else:
            dist.all_gather_into_tensor(self._rows, cur, group=group)
        return fresh

    def forward(self, nodes, adj_mats, edge_weights, num_nodes, B):
        a, b = self._slices(nodes.shape[-1])
        param = self.dist_param.detach() if self.learned else None
        target = torch.zeros_like(adj_mats) if adj_mats.requires_grad else adj_mats
        rows = None
        if self.shard_group is not None:
            self.gather_current(nodes[torch.arange(B, device=nodes.device), num_nodes])
            rows = self._rows
        _ops.edge_distance_(nodes.detach(), target, num_nodes, self.mode, self.max_distance,
                            dist_param=param, a=a, b=b, bidirectional=self.bidirectional, cur_rows=rows)
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

    def __init__(self, max_distance, learned=False, shard_group=None):
        """shard_group (not in the reference, SURVEY 8e "Exception"): a torch.distributed process group
        (True: the default group) over which the batch is sharded.  The reference's mean runs over ALL
        graphs of the batch, so a rank that owns a shard needs every rank's current nodes: one
        all-gather of [B, F] rows per step (64 KB at cfg3) ahead of the distance kernel - results are
        then identical to the unsharded run on the concatenated batch.  None: the local batch only."""
        super().__init__(max_distance, learned=learned)
        self.shard_group = shard_group


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
