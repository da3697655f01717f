#!/usr/bin/env python3
"""Phase breakdown of the fused forward kernel from in-kernel cycle stamps (diagnostic build:
make -C graph-conv-memory_amd/csrc stamps).  Dev tool."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "graph-conv-memory_amd", "gcm", "_lib", "libgcm_hip_stamps.so"))
B, N, F, H = 256, 128, 32, 32
dev = "cuda:0"
torch.manual_seed(0)
nodes = torch.rand(B, N, F, device=dev)
adj = (torch.rand(B, N, N, device=dev) < 0.03).float()
cur = torch.randint(0, N, (B,), device=dev)
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
names = ["issue loads", "wait x,W1", "store x,W1", "barrier", "phase A (4 tiles)", "agg->LDS/HBM",
         "phase B mfma", "epilogue act+stores", "barrier", "layer 2 row"]
acc = [0.0] * 10
R = 20
for it in range(R + 3):
    rc = lib.gcm_dense_gnn2_row_fwd(p(nodes), p(adj), p(cur), p(W[0]), p(W[1]), p(W[2]), 1, p(W[3]), p(W[4]), p(W[5]), 1,
                                    p(mx), p(h1), p(agg1), p(agg2), p(flags), B, N, F, H, H, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    lib.gcm_debug_read_stamps(out, 32)
    if it >= 3:
        for i in range(9):
            acc[i] += (out[i + 1] - out[i]) / R
print("fwd phase (wg 0, lane 0)        cycles(100MHz ticks?)")
for i in range(9):
    print(f"  {names[i]:26s} {acc[i]:10.0f}")
print(f"  total                      {sum(acc[:9]):10.0f}")
