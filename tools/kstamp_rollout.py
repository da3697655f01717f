#!/usr/bin/env python3
"""Phase breakdown of ONE step of the persistent rollout kernel from in-kernel stamps
(diagnostic build: make -C graph-conv-memory_amd/csrc stamps2).  Dev tool."""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "graph-conv-memory_amd", "gcm", "_lib", "libgcm_hip_stamps2.so"))
B, N, F, H, T = 256, 128, 32, 32, 128
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
obs = torch.rand(T, B, F, device=dev)
nodes = torch.zeros(T + 1, B, N, F, device=dev)
adj = torch.zeros(T + 1, B, N, N, device=dev)
count = torch.zeros(T + 1, B, dtype=torch.int64, device=dev)
cur = torch.zeros(T, B, dtype=torch.int64, device=dev)
flags = torch.zeros(1, dtype=torch.int32, device=dev)
W = [torch.randn(H, F, device=dev) * 0.1, torch.randn(H, device=dev) * 0.1, torch.randn(H, F, device=dev) * 0.1,
     torch.randn(H, H, device=dev) * 0.1, torch.randn(H, device=dev) * 0.1, torch.randn(H, H, device=dev) * 0.1]
mx = torch.empty(T, B, H, device=dev)
h1 = torch.empty(T, B, N, H, device=dev)
agg1 = torch.empty(T, B, N, F, device=dev)
agg2 = torch.empty(T, B, H, device=dev)
V = ctypes.c_void_p
p = lambda t: V(t.data_ptr())
st = V(torch.cuda.current_stream().cuda_stream)
names = ["insert+edits+barrier", "live mask + node stores", "adj tiles: lds read, stores, agg mfma",
         "agg -> LDS/HBM", "linears mfma", "act + h1 stores", "barrier wait", "layer 2 row", "end barrier"]
acc = [0.0] * 9
R = 10
fn = lib.gcm_dense_rollout_persistent_fwd
for it in range(R + 2):
    rc = fn(p(obs), p(nodes), p(adj), p(count), p(cur), ctypes.byref(sel), 1, p(W[0]), p(W[1]), p(W[2]), 1,
            p(W[3]), p(W[4]), p(W[5]), 1, p(mx), p(h1), p(agg1), p(agg2), p(flags), 1, T, B, N, F, H, H, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    lib.gcm_debug_read_stamps(out, 32)
    if it >= 2:
        for i in range(9):
            acc[i] += (out[i + 1] - out[i]) / R
print("step 20, workgroup 0, wave 0 (the live wave)      ticks of 10 ns")
for i in range(9):
    print(f"  {names[i]:40s} {acc[i]:8.1f}")
print(f"  total                                    {sum(acc):8.1f}")
