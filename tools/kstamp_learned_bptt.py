#!/usr/bin/env python3
"""Phase breakdown of pass B of the time-parallel LearnedEdge backward (k_learned_bptt_b) at cfg5's per-GPU
shape from in-kernel stamps: the LAST item workgroup 0 processed.  Diagnostic build of the whole library
in place of the product one (run on the GPU box only):
    make -C graph-conv-memory_amd/csrc stamps7 && cp graph-conv-memory_amd/gcm/_lib/libgcm_hip_stamps7.so \\
        graph-conv-memory_amd/gcm/_lib/libgcm_hip.so && python tools/kstamp_learned_bptt.py"""
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
c["T"] = int(os.environ.get("T", 64))
mem, gnn, sel = bench.build_memory(dev, donate=False, selector="learned", cfg=c)
obs = bench.make_obs(c, 0, dev)
names = ["slot search (hdr, live lists)", "dAgg1 gather, X / h1 -> LDS", "running sums over the later steps",
         "g_sel", "softmax adjoint | P0 (2 products)", "LayerNorm 0 -> H0", "P1 (product)", "LayerNorm 1 statistics",
         "column sums dw2 dgamma1 dbeta1", "LayerNorm-1 adjoint", "dW1, gH0 (2 products), X again",
         "dgamma0 dbeta0", "LayerNorm-0 adjoint", "dW0b dW0a (2 products)"]
lib = _hip.lib()
acc, R = [0.0] * 14, 5
for it in range(R + 1):
    bench.rollout(mem, obs)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    lib.gcm_debug_read_stamps(out, 32)
    if it >= 1:
        for i in range(14):
            acc[i] += (out[15 + i] - out[14 + i]) / R
    for m in (gnn, sel):
        m.zero_grad(set_to_none=True)
print("k_learned_bptt_b, last item of workgroup 0, thread 0        stamp ticks (100 MHz)")
for i in range(14):
    print(f"  {14 + i:2d} -> {15 + i:2d}  {names[i]:44s} {acc[i]:9.1f}")
print(f"  total {sum(acc):9.1f}")
