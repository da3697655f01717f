#!/usr/bin/env python3
"""Phase breakdown of the cached live-row step (k_step_rows_cached) at cfg2's shape from in-kernel stamps
(diagnostic build: make -C graph-conv-memory_amd/csrc stamps8).  Dev tool."""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "graph-conv-memory_amd", "gcm", "_lib", "libgcm_hip_stamps8.so"))
B, N, F, H = 256, 128, 32, 32
CUR = int(os.environ.get("CUR", 60))
dev = "cuda:0"
torch.manual_seed(0)


class Sel(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("n_hops", ctypes.c_int), ("hops", ctypes.c_int32 * 16),
                ("direction", ctypes.c_int), ("mode", ctypes.c_int), ("max_distance", ctypes.c_float),
                ("dist_param", ctypes.c_void_p), ("a0", ctypes.c_int), ("a1", ctypes.c_int),
                ("b0", ctypes.c_int), ("b1", ctypes.c_int), ("bidirectional", ctypes.c_int),
                ("cur_rows", ctypes.c_void_p), ("n_cur_rows", ctypes.c_int)]


sel = Sel()
sel.kind, sel.n_hops, sel.direction = 1, 3, 1
for i, h in enumerate([1, 2, 4]):
    sel.hops[i] = h
nodes = torch.rand(B, N, F, device=dev)
adj = torch.zeros(B, N, N, device=dev)
count = torch.full((B,), CUR, dtype=torch.int64, device=dev)
obs = torch.rand(B, F, device=dev)
P = 2 * H * F + H + 2 * H * H + H
params = torch.randn(P, device=dev) * 0.1
cH, cA, cX = (torch.rand(B, N, d, device=dev) for d in (H, F, F))
lay = (ctypes.c_size_t * 5)()
lib.gcm_dense_rows_cached_layout(B, N, F, H, H, lay)
saved = torch.empty(lay[0], device=dev)
flags = torch.zeros(1, dtype=torch.int32, device=dev)
V = ctypes.c_void_p
p = lambda t: V(t.data_ptr())
st = V(torch.cuda.current_stream().cuda_stream)
names = ["weight loads issued, S", "input loads issued", "LDS images (zero, barrier, pieces, barrier)", "layer 1 row",
         "layer 2 row", "state / cache / record stores issued"]
for host_cur in (-1, CUR):
    acc, R = [0.0] * 6, 20
    for it in range(R + 3):
        count.fill_(CUR)
        torch.cuda.synchronize()
        rc = lib.gcm_dense_rows_step_cached(p(obs), p(nodes), p(adj), p(count), ctypes.byref(sel), 1, p(params), None, 3, 1, 1,
                                            p(cH), p(cA), p(cX), p(saved), 1, host_cur, p(flags), B, N, F, H, H, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 32)()
        lib.gcm_debug_read_stamps(out, 32)
        if it >= 3:
            for i in range(6):
                acc[i] += (out[i + 1] - out[i]) / R
    print(f"k_step_rows_cached (cur from {'the host' if host_cur >= 0 else 'the count'}), workgroup 0, wave 0     cycles")
    for i in range(6):
        print(f"  {i} -> {i + 1}  {names[i]:46s} {acc[i]:9.1f}")
    print(f"  total {sum(acc):9.1f}")
    print(f"  inside 0 -> 1: weight loads issued after {out[8] - out[0]} cycles; inside 5 -> 6: state + cache stores "
          f"{out[9] - out[5]}, mx + v {out[10] - out[9]}, live list + header {out[6] - out[10]}")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(50):
        count.fill_(CUR)
        lib.gcm_dense_rows_step_cached(p(obs), p(nodes), p(adj), p(count), ctypes.byref(sel), 1, p(params), None, 3, 1, 1,
                                       p(cH), p(cA), p(cX), p(saved), 1, host_cur, p(flags), B, N, F, H, H, st)
    e1.record()
    torch.cuda.synchronize()
    print(f"  {e0.elapsed_time(e1) / 50 * 1e3:6.2f} us per (fill + launch), 50 back to back")
