#!/usr/bin/env python3
"""Phase breakdown of the matrix-core Euclidean selector kernel (k_euclid_mfma) at cfg3's shape from
in-kernel stamps (diagnostic build: make -C graph-conv-memory_amd/csrc stamps6).  Dev tool."""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.environ.get("STAMPLIB") or os.path.join(ROOT, "graph-conv-memory_amd", "gcm", "_lib", "libgcm_hip_stamps6.so"))
B, N, F = 256, 128, 64
CUR = int(os.environ.get("CUR", 100))
dev = "cuda:0"
torch.manual_seed(0)
nodes = torch.rand(B, N, F, device=dev)
obs = torch.rand(B, F, device=dev)
count = torch.full((B,), CUR, dtype=torch.int64, device=dev)
row = torch.empty(B, N, device=dev)
ws = torch.empty(B * F + B, device=dev)
V = ctypes.c_void_p
p = lambda t: V(t.data_ptr())
st = V(torch.cuda.current_stream().cuda_stream)
lib.gcm_edge_distance_pre.argtypes = [V, V, V, V, ctypes.c_int, ctypes.c_float, V] + [ctypes.c_int] * 4 + \
    [V, ctypes.c_size_t] + [ctypes.c_int] * 3 + [V]
names = ["node rows + chunk 0 -> LDS (+barrier)", "|n|^2, |c|^2 (+barrier)", "chunk 0: next chunk's loads issued, MFMA + sqrt",
         "chunk 1 -> LDS (+barrier)", "|c|^2, chunk 1: MFMA + sqrt", "reductions, column tiles, emit"]
acc, R = [0.0] * 6, 20
for it in range(R + 3):
    torch.cuda.synchronize()
    rc = lib.gcm_edge_distance_pre(p(nodes), p(count), p(obs), p(row), 0, 2.0, None, 0, 0, 0, 0, p(ws),
                                   ws.numel() * 4, B, N, F, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    lib.gcm_debug_read_stamps(out, 32)
    if it >= 3:
        for i in range(6):
            acc[i] += (out[i + 1] - out[i]) / R
print(f"k_euclid_mfma (B={B}, N={N}, F={F}, cur={CUR}), one workgroup, wave 0       stamp ticks")
for i in range(6):
    print(f"  {i} -> {i + 1}  {names[i]:48s} {acc[i]:9.1f}")
print(f"  total {sum(acc):9.1f}")

# kernel time against the fill level of the graphs (the tiles of rows >= cur are skipped)
for cur in (1, 20, 40, 70, 100, 127):
    count.fill_(cur)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(3):
        lib.gcm_edge_distance_pre(p(nodes), p(count), p(obs), p(row), 0, 2.0, None, 0, 0, 0, 0, p(ws), ws.numel() * 4, B, N, F, st)
    e0.record()
    for it in range(50):
        lib.gcm_edge_distance_pre(p(nodes), p(count), p(obs), p(row), 0, 2.0, None, 0, 0, 0, 0, p(ws), ws.numel() * 4, B, N, F, st)
    e1.record()
    torch.cuda.synchronize()
    print(f"  cur = {cur:3d}: {e0.elapsed_time(e1) / 50 * 1e3:6.2f} us per launch (50 back to back)")
