#!/usr/bin/env python3
"""Back-to-back timing of the live-row step kernel (k_step_rows) through the C ABI: donated against
functional state, mid-rollout states (no overflow) and steady-state overflow, beside plain copies of
the same byte count.  Dev tool.   python tools/kbench_rows.py [B N F H]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch  # noqa: E402
from gcm import _hip  # noqa: E402

B, N, F, H = (int(v) for v in (sys.argv[1:5] + [256, 128, 32, 32][len(sys.argv) - 1:]))
dev = "cuda:0"
lib = _hip.lib()
p, st = _hip.ptr, _hip.stream()
torch.manual_seed(0)
P = lib.gcm_dense_gnn2_param_count(F, H, H)
params = torch.randn(P, device=dev) * 0.1
desc = _hip.SelectorDesc()
desc.kind, desc.n_hops, desc.direction = _hip.SEL_TEMPORAL, 3, 1
for i, h in enumerate((1, 2, 4)):
    desc.hops[i] = h
arr = (_hip.SelectorDesc * 1)(desc)
lay = (ctypes.c_size_t * 6)()
lib.gcm_dense_rows_layout(B, N, F, H, H, ctypes.addressof(lay))
saved = torch.empty(lay[0], device=dev)
flags = torch.zeros(1, dtype=torch.int32, device=dev)
obs = torch.rand(B, F, device=dev)


def state(count):
    nodes = torch.rand(B, N, F, device=dev)
    adj = torch.zeros(B, N, N, device=dev)
    i = torch.arange(1, N, device=dev)
    adj[:, i, i - 1] = 1.0
    return nodes, adj, torch.full((B,), count, dtype=torch.long, device=dev)


def step(nodes, adj, cnt, n2, a2, c2, sv=True):
    rc = lib.gcm_dense_rows_step_fwd(p(obs), p(nodes), p(adj), p(cnt), p(n2), p(a2), p(c2), None,
                                     ctypes.addressof(arr), 1, p(params), 3, 1, 1, p(saved),
                                     p(saved) if sv else None, p(flags), B, N, F, H, H, st)
    assert rc == 0, rc


def timeit(name, fn, iters=200, nbytes=None):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / iters * 1e3
    extra = f"   {nbytes / us / 1e6:5.2f} TB/s of {nbytes / 1e6:.1f} MB" if nbytes else ""
    print(f"{name:52s} {us:8.2f} us/launch{extra}")


copy_bytes = 2 * 4 * B * (N * N + N * F)
for count, label in ((N // 2, "count = N/2"), (N, "count = N: every graph rolls")):
    nodes, adj, cnt = state(count)
    n2, a2, c2 = torch.empty_like(nodes), torch.empty_like(adj), torch.empty_like(cnt)
    timeit(f"functional, {label}", lambda: step(nodes, adj, cnt, n2, a2, c2), nbytes=copy_bytes)
    timeit(f"functional, inference, {label}", lambda: step(nodes, adj, cnt, n2, a2, c2, sv=False), nbytes=copy_bytes)
    keep = cnt.clone()

    def donated():
        step(nodes, adj, cnt, nodes, adj, cnt)
        cnt.copy_(keep)     # (a 2 KB copy kernel between launches: part of the figure)
    timeit(f"donated (+ count reset kernel), {label}", donated)
nodes, adj, cnt = state(N // 2)
n2, a2 = torch.empty_like(nodes), torch.empty_like(adj)
timeit("torch copy_ of nodes + adj (2 kernels)", lambda: (n2.copy_(nodes), a2.copy_(adj)), nbytes=copy_bytes)
buf = torch.empty(copy_bytes // 8, device=dev)
buf2 = torch.empty_like(buf)
timeit("torch copy_ of the same bytes (1 kernel)", lambda: buf2.copy_(buf), nbytes=copy_bytes)
