#!/usr/bin/env python3
"""Phase breakdown of the LearnedEdge step backward (k_learned_step_bwd) at cfg5's per-GPU shape from
in-kernel stamps (diagnostic build: make -C graph-conv-memory_amd/csrc stamps7).  Dev tool."""
import ctypes
import os
import sys

os.environ["GCM_NO_TORCH_EXT"] = "1"            # the Python autograd Functions: ctypes calls into the library below
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch  # noqa: E402
from gcm import _hip  # noqa: E402
_hip._LIB_PATH = os.path.join(ROOT, "graph-conv-memory_amd", "gcm", "_lib", "libgcm_hip_stamps7.so")
from gcm import nn as G  # noqa: E402
from gcm.gcm import DenseGCM  # noqa: E402
from gcm.edge_selectors.learned import LearnedEdge  # noqa: E402

dev = "cuda:0"
B, N, F, H = 256, 128, 32, 32
WARM = int(os.environ.get("CUR", 100))         # nodes in the graph when the measured step runs
torch.manual_seed(0)
g = G.Sequential("x, adj, weights, B, N", [(G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
                                           (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(dev)
mem = DenseGCM(g, edge_selectors=LearnedEdge(F).to(dev), graph_size=N)
lib = _hip.lib()
names = ["loads -> LDS", "layer-2 adjoint, dW2 slabs", "live rows, G1", "dW1 on the live rows", "dAgg1, GA rows",
         "g_sel, softmax adjoint", "chain buffer: undo the state advance", "edge network recomputed (2 GEMMs, LN)",
         "column sums: dw2, dgamma1, dbeta1", "LayerNorm-1 adjoint", "dW1, gH0 (2 GEMMs)",
         "dgamma0 / dbeta0, LayerNorm-0 adjoint", "dW0 (GEMM), slabs"]
acc, R = [0.0] * 13, 10
for it in range(R + 2):
    hidden = None
    with torch.no_grad():
        for t in range(WARM):
            _, hidden = mem(torch.rand(B, F, device=dev), hidden)
    mx, hidden = mem(torch.rand(B, F, device=dev), tuple(hidden))
    mx.sum().backward()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    lib.gcm_debug_read_stamps(out, 32)
    if it >= 2:
        for i in range(13):
            acc[i] += (out[i + 1] - out[i]) / R
print(f"k_learned_step_bwd (B={B}, N={N}, F={F}, cur={WARM}), workgroup 0, thread 0        stamp ticks")
for i in range(13):
    print(f"  {i:2d} -> {i + 1:2d}  {names[i]:44s} {acc[i]:9.1f}")
print(f"  total {sum(acc):9.1f}")
