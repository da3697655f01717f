#!/usr/bin/env python3
"""Phase breakdown of the one-kernel forward step (k_step_fwd_live) from in-kernel stamps
(diagnostic build: make -C graph-conv-memory_amd/csrc stamps).  Dev tool."""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "graph-conv-memory_amd", "gcm", "_lib", "libgcm_hip_stamps.so"))
B, N, F, H = 256, 128, 32, 32
dev = "cuda:0"
torch.manual_seed(0)


class Sel(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("n_hops", ctypes.c_int), ("hops", ctypes.c_int32 * 16),
                ("direction", ctypes.c_int), ("mode", ctypes.c_int), ("max_distance", ctypes.c_float),
                ("dist_param", ctypes.c_void_p), ("a0", ctypes.c_int), ("a1", ctypes.c_int),
                ("b0", ctypes.c_int), ("b1", ctypes.c_int), ("bidirectional", ctypes.c_int)]


sel = Sel()
sel.kind, sel.n_hops, sel.direction = 1, 3, 1
for i, h in enumerate([1, 2, 4]):
    sel.hops[i] = h
CUR = 20   # workgroup 0's new node lands in tile 0: wave 0 (the stamped wave) is on the critical path
nodes = torch.rand(B, N, F, device=dev)
adj = torch.zeros(B, N, N, device=dev)
for i in range(1, CUR):
    for h in (1, 2, 4):
        if i - h >= 0:
            adj[:, i, i - h] = 1
count = torch.full((B,), CUR, dtype=torch.int64, device=dev)
obs = torch.rand(B, F, device=dev)
n_out, a_out = torch.empty_like(nodes), torch.empty_like(adj)
ibuf = torch.empty(2, B, dtype=torch.int64, device=dev)
flags = torch.zeros(1, dtype=torch.int32, device=dev)
W = [torch.randn(H, F, device=dev) * 0.1, torch.randn(H, device=dev) * 0.1, torch.randn(H, F, device=dev) * 0.1,
     torch.randn(H, H, device=dev) * 0.1, torch.randn(H, device=dev) * 0.1, torch.randn(H, H, device=dev) * 0.1]
mx = torch.empty(B, H, device=dev)
h1 = torch.empty(B, N, H, device=dev)
agg1 = torch.empty(B, N, F, device=dev)
agg2 = torch.empty(B, H, device=dev)
V = ctypes.c_void_p
p = lambda t: V(t.data_ptr())
st = V(torch.cuda.current_stream().cuda_stream)
names = ["issue loads (x, W1, adj rows, W2, biases)", "wait x: nodes_out stores", "x, W1, W2 -> LDS",
         "edits, adj_out stores, tiles -> LDS, mask", "barrier", "tile mask + live flags", "live-tile GNN + layer-2 row"]
acc = [0.0] * 7
R = 20
for it in range(R + 3):
    rc = lib.gcm_dense_step_fused_fwd(p(obs), p(nodes), p(adj), p(count), p(n_out), p(a_out), V(ibuf.data_ptr()),
                                      V(ibuf.data_ptr() + 8 * B), ctypes.byref(sel), 1, p(W[0]), p(W[1]), p(W[2]), 1,
                                      p(W[3]), p(W[4]), p(W[5]), 1, p(mx), p(h1), p(agg1), p(agg2), p(flags),
                                      B, N, F, H, H, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    lib.gcm_debug_read_stamps(out, 32)
    if it >= 3:
        for i in range(7):
            acc[i] += (out[i + 1] - out[i]) / R
print("k_step_fwd_live, workgroup 0, wave 0            cycles")
for i in range(7):
    print(f"  {names[i]:44s} {acc[i]:8.1f}")
print(f"  total (kernel entry to last instruction)     {sum(acc):8.1f}")
