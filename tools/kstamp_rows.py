#!/usr/bin/env python3
"""Phase breakdown of the live-row step kernel (k_step_rows) from in-kernel stamps
(diagnostic build: make -C graph-conv-memory_amd/csrc stamps5).  Dev tool."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "graph-conv-memory_amd", "gcm", "_lib", "libgcm_hip_stamps5.so"))
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
CUR = int(os.environ.get("CUR", 100))
nodes0 = torch.rand(B, N, F, device=dev)
adj0 = torch.zeros(B, N, N, device=dev)
for i in range(1, CUR):
    for h in (1, 2, 4):
        if i - h >= 0:
            adj0[:, i, i - h] = 1
obs = torch.rand(B, F, device=dev)
flags = torch.zeros(1, dtype=torch.int32, device=dev)
P = 2 * H * F + H + 2 * H * H + H
params = torch.randn(P, device=dev) * 0.1
lay = (ctypes.c_size_t * 6)()
lib.gcm_dense_rows_layout(B, N, F, H, H, ctypes.byref(lay))
saved = torch.empty(lay[0], device=dev)
V = ctypes.c_void_p
p = lambda t: V(t.data_ptr())
st = V(torch.cuda.current_stream().cuda_stream)
names = {1: "count-independent loads issued, count arrived", 2: "(roll) row cur, candidates, ahead rows issued",
         3: "x image -> LDS, row-cur edits (loads landed)", 4: "barrier #1", 5: "live list",
         6: "C: rows -> LDS", 7: "barrier #3", 8: "D: agg1 MFMA", 9: "barrier #4",
         10: "E: linears MFMA + act + agg2", 11: "barrier #5", 12: "saved rows, layer 2, mx"}
NS = 12
for mode in ("donated", "functional"):
    acc = {}
    R = 20
    for it in range(R + 3):
        nodes, adj = nodes0.clone(), adj0.clone()
        count = torch.full((B,), CUR, dtype=torch.int64, device=dev)
        if mode == "donated":
            n_out, a_out, c_out = nodes, adj, count
        else:
            n_out, a_out, c_out = torch.empty_like(nodes), torch.empty_like(adj), torch.empty_like(count)
        torch.cuda.synchronize()
        rc = lib.gcm_dense_rows_step_fwd(p(obs), p(nodes), p(adj), p(count), p(n_out), p(a_out), p(c_out), None,
                                         ctypes.byref(sel), 1, p(params), 3, 1, 1, p(saved), p(saved), p(flags),
                                         B, N, F, H, H, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 32)()
        lib.gcm_debug_read_stamps(out, 32)
        if it >= 3:
            for i in range(NS):
                acc[i] = acc.get(i, 0.0) + (out[i + 1] - out[i]) / R
    print(f"k_step_rows ({mode}), workgroup 0, wave 0            cycles")
    for i in range(NS):
        print(f"  {i:2d} -> {i+1:2d}  {names.get(i+1, ''):44s} {acc[i]:8.1f}")
    print(f"  total (kernel entry to last stamp)                   {sum(acc.values()):8.1f}")
