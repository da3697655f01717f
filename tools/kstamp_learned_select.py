#!/usr/bin/env python3
"""Phase breakdown of the cached LearnedEdge step (k_learned_select<2, true>) at cfg5's per-GPU shape from in-kernel
stamps (s_memtime of workgroup 0 / thread 0, shader clocks): the LAST step of a 64-step rollout.  Diagnostic build of
the whole library in place of the product one (run on the GPU box only):
    make -C graph-conv-memory_amd/csrc stamps7 && cp graph-conv-memory_amd/gcm/_lib/libgcm_hip_stamps7.so \\
        graph-conv-memory_amd/gcm/_lib/libgcm_hip.so && python tools/kstamp_learned_select.py"""
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
names = ["count / addresses, every load issued", "loads landed -> LDS images (+ barrier)", "c0, P0 (product)",
         "LayerNorm 0", "P1 (product)", "LayerNorm 1", "logits (+ fence, barrier)", "gumbel-softmax, adjacency row",
         "GNN tail on row cur, stores"]
lib = _hip.lib()
acc, R = [0.0] * len(names), 5
for it in range(R + 1):
    with torch.no_grad():
        hidden = None
        for t in range(c["T"]):
            mx, hidden = mem(obs[t], hidden)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    lib.gcm_debug_read_stamps(out, 32)
    if it >= 1:
        for i in range(len(names)):
            acc[i] += (out[15 + i] - out[14 + i]) / R
print("k_learned_select<2, true>, step 63 of a rollout, workgroup 0 / thread 0        shader clocks")
for i, n in enumerate(names):
    print(f"  {14 + i:2d} -> {15 + i:2d}  {n:44s} {acc[i]:9.1f}")
print(f"  total {sum(acc):9.1f}")
print("  inside the tail: 22 -> 24 gather of the selected rows %.0f, 24 -> 25 layer 1 on row cur %.0f, 25 -> 26 layer 2 %.0f, "
      "26 -> 23 stores %.0f" % (out[24] - out[22], out[25] - out[24], out[26] - out[25], out[23] - out[26]))
