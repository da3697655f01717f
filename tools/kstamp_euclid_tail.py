#!/usr/bin/env python3
"""Phase breakdown of the one-launch EuclideanEdge step (k_euclid_mfma2<2, true>: distances + the cached step as
wave 0's tail) in situ at cfg3's shape: the LAST step of a rollout, the workgroup of graph B / 2, wave 0 (shader
clocks).  Diagnostic build (GPU box only):
    make -C graph-conv-memory_amd/csrc stamps9 && cp .../libgcm_hip_stamps9.so .../libgcm_hip.so && \\
        python tools/kstamp_euclid_tail.py"""
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
c = dict(bench.CONFIGS["cfg3"])
mem, gnn, sel = bench.build_memory(dev, donate=True, selector="euclid", cfg=c)
obs = bench.make_obs(c, 0, dev)
names = ["node rows + chunk 0 -> LDS (+barrier)", "(norms)", "chunk 0: MFMA + sqrt", "chunk 1 -> LDS (+barrier)",
         "chunk 1: MFMA + sqrt", "reductions, decisions", "tail: barrier, masks", "tail: gather of the selected rows",
         "tail: layer 1 on row cur", "tail: layer 2", "tail: stores, record"]
lib = _hip.lib()
acc, R = [0.0] * len(names), 5
for it in range(R + 1):
    with torch.no_grad():
        hidden = None
        for t in range(int(os.environ.get("T_STOP", c["T"]))):     # (T_STOP: the step whose launch is read, e.g. 20)
            mx, hidden = mem(obs[t], hidden)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    lib.gcm_debug_read_stamps(out, 32)
    if it >= 1:
        for i in range(len(names)):
            acc[i] += (out[i + 1] - out[i]) / R
print("k_euclid_mfma2<2, true>, last step of a cfg3 rollout, graph B/2, wave 0        shader clocks")
for i, n in enumerate(names):
    print(f"  {i:2d} -> {i + 1:2d}  {n:44s} {acc[i]:9.1f}")
print(f"  total {sum(acc):9.1f}")
