#!/usr/bin/env python3
"""Phase breakdown of pass B2 of the LearnedEdge backward (k_learned_bptt_mlp: the edge network per 32-row block,
one block per wave) from in-kernel stamps: the LAST block wave 0 of workgroup 0 processed (shader clocks).
Diagnostic build (GPU box only):
    make -C graph-conv-memory_amd/csrc stamps7 STAMP_FLAGS=-DGCM_STAMPS_B2 && cp .../libgcm_hip_stamps7.so \\
        .../libgcm_hip.so && python tools/kstamp_learned_b2.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from gcm import _hip  # noqa: E402

dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS["cfg5"])
mem, gnn, sel = bench.build_memory(dev, donate=True, selector="learned", cfg=c)
obs = bench.make_obs(c, 0, dev)
names = ["stage X block, x_cur, g_logit -> LDS", "P0 (2 products)", "LayerNorm 0 -> H0", "P1 (product)",
         "LayerNorm 1 statistics", "column sums dw2 dgamma1 dbeta1", "LayerNorm-1 adjoint",
         "c_b1, dW1, gH0 (2 products)", "dgamma0 dbeta0", "LayerNorm-0 adjoint", "c_b0, dW0b dW0a (2 products)"]
lib = _hip.lib()
acc, R = [0.0] * len(names), 5
for it in range(R + 1):
    bench.rollout(mem, obs)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    lib.gcm_debug_read_stamps(out, 32)
    if it >= 1:
        for i in range(len(names)):
            acc[i] += (out[1 + i] - out[i]) / R
    for m in (gnn, sel):
        m.zero_grad(set_to_none=True)
print("k_learned_bptt_mlp, last block of workgroup 0 / wave 0        shader clocks")
for i, n in enumerate(names):
    print(f"  {i:2d} -> {i + 1:2d}  {n:44s} {acc[i]:9.1f}")
print(f"  total {sum(acc):9.1f}")
