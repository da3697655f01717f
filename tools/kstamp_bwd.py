#!/usr/bin/env python3
"""Phase breakdown of the per-step backward kernel (k_gnn2_row_bwd) from in-kernel stamps
(diagnostic build: make -C graph-conv-memory_amd/csrc stamps4).  Dev tool."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
_LIB = os.path.join(ROOT, "graph-conv-memory_amd", "gcm", "_lib")
lib = ctypes.CDLL(os.path.join(_LIB, "libgcm_hip_stamps4.so"))
B, N, F, H = 256, 128, 32, 32
dev = "cuda:0"
torch.manual_seed(0)
CUR = 20   # tile 0 is the live one: wave 0 (stamped) carries the row work
nodes = torch.rand(B, N, F, device=dev)
adj = torch.zeros(B, N, N, device=dev)
for i in range(1, CUR + 1):
    for h in (1, 2, 4):
        if i - h >= 0:
            adj[:, i, i - h] = 1
cur = torch.full((B,), CUR, dtype=torch.int64, device=dev)
W = [torch.randn(H, F, device=dev) * 0.1, torch.randn(H, device=dev) * 0.1, torch.randn(H, F, device=dev) * 0.1,
     torch.randn(H, H, device=dev) * 0.1, torch.randn(H, device=dev) * 0.1, torch.randn(H, H, device=dev) * 0.1]
mx = torch.rand(B, H, device=dev)
h1 = torch.rand(B, N, H, device=dev)
agg1 = torch.rand(B, N, F, device=dev)
agg2 = torch.rand(B, H, device=dev)
g_mx, g_no = torch.randn(B, H, device=dev), torch.randn(B, N, F, device=dev)
g_ni, g_obs = torch.empty(B, N, F, device=dev), torch.empty(B, F, device=dev)
P = 2 * H * F + H + 2 * H * H + H
slabs = torch.empty(B, P, device=dev)
V = ctypes.c_void_p
p = lambda t: V(t.data_ptr())
st = V(torch.cuda.current_stream().cuda_stream)
names = ["issue every load", "W2, d2, v -> LDS (waits for their loads)", "barrier", "u partials, layer-2 dW (slab RMW)",
         "h1, W1, adj tiles -> LDS + tile flags (waits adj)", "barrier, sU, g_live, barrier", "G1 + db1 partials",
         "barrier, db1; dW1 jobs (x/agg1 loads + MFMA, LDS reduce, slab)", "dAgg / root MFMA + gno loads",
         "barrier; dX MFMA", "epilogue stores"]
acc = [0.0] * 10
R = 20
for it in range(R + 3):
    rc = lib.gcm_dense_gnn2_row_bwd(p(g_mx), p(g_no), p(nodes), p(adj), p(cur), p(cur), p(W[0]), p(W[1]), p(W[2]), 1,
                                    p(W[3]), p(W[4]), p(W[5]), 1, p(mx), p(h1), p(agg1), p(agg2), p(g_ni), p(g_obs),
                                    p(slabs), 0, B, N, F, H, H, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    lib.gcm_debug_read_stamps(out, 32)
    if it >= 3:
        for i in range(10):
            acc[i] += (out[i + 1] - out[i]) / R
print("k_gnn2_row_bwd, workgroup 0, wave 0                         cycles")
for i in range(10):
    print(f"  {names[i]:58s} {acc[i]:8.1f}")
print(f"  total                                                      {sum(acc):8.1f}")
