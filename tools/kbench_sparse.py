#!/usr/bin/env python3
"""Micro-benchmark of the CSR GraphConv kernels at the cfg4 one-shot shape (B=512 chains of 512 nodes,
F=H=32): back-to-back launches timed with events, beside plain streaming copies of the same byte
counts (what the memory system gives a trivially coalesced kernel).  Dev tool."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch  # noqa: E402
from gcm import _hip  # noqa: E402

B, N, F, H = 512, 512, 32, 32
dev = "cuda:0"
lib = _hip.lib()
p, st = _hip.ptr, _hip.stream()
M = B * N
torch.manual_seed(0)
x = torch.rand(M, F, device=dev)
# chain graph per batch: node i <- i-1
deg = torch.ones(M, dtype=torch.long, device=dev)
deg[::N] = 0
row_ptr = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), deg.cumsum(0)])
col = (torch.arange(M, device=dev) - 1)[deg.bool()].contiguous()
E = col.numel()
w_rel, w_root, b = torch.randn(H, F, device=dev) * .1, torch.randn(H, F, device=dev) * .1, torch.randn(H, device=dev)
out, agg = torch.empty(M, H, device=dev), torch.empty(M, F, device=dev)
row_ptr0 = torch.zeros(M + 1, dtype=torch.long, device=dev)


def timeit(name, fn, nbytes, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(e) / iters * 1e3
    print(f"{name:44s} {us:8.2f} us   {nbytes / us / 1e6:6.2f} TB/s of {nbytes / 1e6:.1f} MB")


row = M * F * 4
timeit("csr fwd (train: x, agg, out)", lambda: lib.gcm_csr_graphconv_fwd(
    p(x), p(row_ptr), p(col), None, None, p(w_rel), p(b), p(w_root), p(out), p(agg), M, F, H, 1, st), 3 * row)
timeit("csr fwd (inference: x, out)", lambda: lib.gcm_csr_graphconv_fwd(
    p(x), p(row_ptr), p(col), None, None, p(w_rel), p(b), p(w_root), p(out), None, M, F, H, 1, st), 2 * row)
timeit("csr fwd, no edges (train)", lambda: lib.gcm_csr_graphconv_fwd(
    p(x), p(row_ptr0), p(col), None, None, p(w_rel), p(b), p(w_root), p(out), p(agg), M, F, H, 1, st), 3 * row)
y = torch.empty_like(x)
timeit("copy_ x -> y (1 read + 1 write)", lambda: y.copy_(x), 2 * row)
z = torch.empty(2, M, F, device=dev)
timeit("x -> 2 outputs (1 read + 2 writes)", lambda: torch.add(x.unsqueeze(0), 1.0, out=z), 3 * row)
big = torch.rand(8 * M, F, device=dev)
big2 = torch.empty_like(big)
timeit("copy_ 268 MB (1 read + 1 write)", lambda: big2.copy_(big), 16 * row)
