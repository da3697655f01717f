"""Data-parallel helpers: one process per GPU, the batch of graphs sharded over
ranks, parameters replicated.  Every graph of the batch is independent in the
DenseGCM / SparseGCM step (gcm.py:274,314 index per graph), so the forward pass
needs NO communication; the only collective is one all-reduce per backward over
one flat gradient bucket (RCCL over xGMI when the backend is "nccl").
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*.
    Returns (rank, local_rank, world).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    if os.environ.get("GCM_SINGLE_DEVICE") == "1":
        local_rank = 0          # test hook: every rank on cuda:0 (with GCM_DIST_BACKEND=gloo)
    backend = backend or os.environ.get("GCM_DIST_BACKEND")
    if use_cuda:
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"),
                                rank=rank, world_size=world)
    return rank, local_rank, world


def shard_bounds(total, rank, world):
    """[lo, hi) of the contiguous shard of `total` graphs owned by `rank` (remainder
    spread over the first ranks)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradBucket:
    """All parameter gradients of one or more modules as ONE flat fp32 buffer, so that a
    backward pass costs exactly one collective.  At F=H=32 the bucket is ~30 KB: the
    all-reduce is latency bound, link bandwidth is irrelevant.

    alias_grads=True: after the first call every `p.grad` IS its slice of the flat buffer (a view; the
    bucket keeps no reference of its own, so autograd accumulates into it in place): later backward
    passes write straight into the bucket and all_reduce_mean is the collective alone.  Needs
    `zero()` / `zero_grad(set_to_none=False)` between steps (set_to_none=True drops the views - still
    correct, the gather then runs again) and must stay off when the backward pass is replayed from a
    HIP graph (the graph writes the gradient tensors it was captured with)."""

    def __init__(self, *modules, alias_grads=False):
        self.params = [p for m in modules for p in m.parameters() if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.alias_grads = alias_grads
        self.flat = None
        self.launches = 0      # device launches of the last all_reduce_mean besides the collective (tests)

    def _view(self, i):
        off, p = self.offsets[i], self.params[i]
        return self.flat[off:off + p.numel()].view_as(p)

    def _ensure_flat(self):
        dev = self.params[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.zeros(self.numel, device=dev)
            self.offsets, off = [], 0
            for p in self.params:
                self.offsets.append(off)
                off += p.numel()

    def attach(self):
        """Make every `p.grad` its (zeroed) slice of the flat buffer NOW - before the first backward, e.g. ahead of a
        HIP-graph capture of `bucket.zero(); loss.backward(); bucket.all_reduce_mean(w)`: autograd then accumulates
        into the bucket in place (AccumulateGrad adds into an existing .grad), the captured graph needs no gather /
        copy-back launch, and the collective itself can sit inside the graph (all_reduce_mean is capture-safe once
        the gradients alias the bucket: a scale and the collective, no allocation).

        The aliasing lives in `p.grad`: `Module.zero_grad()` / `Optimizer.zero_grad()` with torch's default
        `set_to_none=True` DROP it, and a replayed graph then keeps accumulating into the bucket while `p.grad` is
        None - the optimizer would silently skip every parameter.  Between steps use `bucket.zero()` (one launch) or
        `zero_grad(set_to_none=False)`, and call `ensure_attached()` before each replay / optimizer step (cheap: pointer
        compares on the host; it re-attaches when a caller dropped the views)."""
        if not self.params:
            return self
        self._ensure_flat()
        self.flat.zero_()
        for i, p in enumerate(self.params):
            p.grad = self._view(i)
        self.alias_grads = True
        return self

    def ensure_attached(self):
        """Re-establish `p.grad` = slice of the flat bucket if a caller dropped it (zero_grad(set_to_none=True));
        -> True when the gradients were still aliased.  Host-side pointer compares only; the bucket's contents are
        kept (a replayed graph may already have written this step's gradient into it)."""
        if not self.params:
            return True
        self._ensure_flat()
        if self.aliased():
            return True
        for i, p in enumerate(self.params):
            p.grad = self._view(i)
        self.alias_grads = True
        return False

    def aliased(self):
        if self.flat is None:
            return False
        base = self.flat.data_ptr()
        return all(p.grad is not None and p.grad.is_contiguous() and p.grad.data_ptr() == base + 4 * o
                   for p, o in zip(self.params, self.offsets))

    def zero(self):
        """zero the bucket (and with it every aliased gradient) in one launch"""
        if self.flat is not None:
            self.flat.zero_()

    def all_reduce_mean(self, local_weight=1.0, force_collective=False):
        """grad <- sum_r local_weight_r * grad_r  (pass local_weight = B_local / B_global to
        get the gradient of the global-batch mean loss from per-rank local-mean losses).
        Launches per call: one multi-tensor gather into the flat buffer (none when the gradients
        alias it), the weight (folded into the collective as an average on RCCL when the shards are
        equal), the collective, one multi-tensor copy back (none with alias_grads); nothing at all in
        a one-rank job with weight 1."""
        self.launches = 0
        if not self.params:
            return
        world = dist.get_world_size() if dist.is_initialized() else 1
        if world == 1 and local_weight == 1.0 and not (force_collective and dist.is_initialized()):
            return
        self._ensure_flat()
        aliased = self.aliased()
        if not aliased:
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
            torch._foreach_copy_([self._view(i) for i in range(len(self.params))], grads)
            self.launches += 1
        avg = (world > 1 and abs(local_weight * world - 1.0) < 1e-12 and dist.get_backend() == "nccl")
        if local_weight != 1.0 and not avg:
            self.flat.mul_(local_weight)
            self.launches += 1
        if world > 1 or (force_collective and dist.is_initialized()):   # (force_collective: a one-rank communicator - tests)
            dist.all_reduce(self.flat, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM)
        if self.alias_grads:
            if not aliased:
                for i, p in enumerate(self.params):
                    p.grad = self._view(i)
            return
        for p in self.params:
            if p.grad is None:
                p.grad = torch.empty_like(p)
        torch._foreach_copy_([p.grad for p in self.params], [self._view(i) for i in range(len(self.params))])
        self.launches += 1
