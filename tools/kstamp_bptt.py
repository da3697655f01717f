#!/usr/bin/env python3
"""Phase breakdown of ONE item of the time-parallel BPTT kernel from in-kernel stamps
(diagnostic build: make -C graph-conv-memory_amd/csrc stamps3).  Dev tool."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
from gcm.gcm import DenseGCM
from gcm import nn as G
from gcm.edge_selectors.temporal import TemporalBackedge

lib = ctypes.CDLL(os.path.join(ROOT, "graph-conv-memory_amd", "gcm", "_lib", "libgcm_hip_stamps3.so"))
B, N, F, H, T = 256, 128, 32, 32, 128
dev = "cuda"
g = G.Sequential("x, adj, weights, B, N", [(G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
                                           (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(dev)
mem = DenseGCM(g, edge_selectors=TemporalBackedge([1, 2, 4]), graph_size=N)
obs = torch.rand(T, B, F, device=dev)
out, hid = mem.rollout(obs)
nodes_all, adj_all, count_all, cur_all, mx_all, h1_all, agg1_all, agg2_all, packed = out.grad_fn.saved_tensors
g_mx = torch.rand(T, B, H, device=dev)
P = 2 * H * F + H + 2 * H * H + H
items = T * B
n_slabs = 768
Q = torch.empty(T, B, N, F, device=dev)
pobs = torch.empty(T, B, F, device=dev)
slabs = torch.empty(n_slabs, P, device=dev)
V = ctypes.c_void_p
p = lambda t: V(t.data_ptr())
base = packed.data_ptr()
w = [base, base + 4 * 2 * H * F, base + 4 * H * F, base + 4 * (2 * H * F + H), base + 4 * (2 * H * F + H + 2 * H * H),
     base + 4 * (2 * H * F + H + H * H)]
st = V(torch.cuda.current_stream().cuda_stream)
names = ["phase-0 loads (row of cur) + LDS", "barrier", "live, h1 loads issued, u, dW2", "barrier + sU + barrier", "G1",
         "barrier", "dW1 blocks (16x16x4)", "dAgg / root blocks", "barrier", "adj strip loads + dX MFMA + epilogue"]
acc = [0.0] * 10
R = 10
for it in range(R + 2):
    rc = lib.gcm_dense_bptt_batched(p(g_mx), None, V(nodes_all.data_ptr() + 4 * B * N * F), V(adj_all.data_ptr() + 4 * B * N * N),
                                    p(cur_all), p(count_all), V(w[0]), V(w[1]), V(w[2]), 1, V(w[3]), V(w[4]), V(w[5]), 1,
                                    p(mx_all), p(h1_all), p(agg1_all), p(agg2_all), p(Q), p(pobs), p(slabs), n_slabs, items,
                                    N, F, H, H, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    o = (ctypes.c_ulonglong * 32)()
    lib.gcm_debug_read_stamps(o, 32)
    if it >= 2:
        for i in range(10):
            acc[i] += (o[i + 1] - o[i]) / R
print("item 15360 (t=60, b=0), workgroup 0, wave 0          cycles")
for i in range(10):
    print(f"  {names[i]:44s} {acc[i]:8.1f}")
print(f"  total                                        {sum(acc):8.1f}")
