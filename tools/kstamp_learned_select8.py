#!/usr/bin/env python3
"""Phase breakdown of the eight-wave cached LearnedEdge step (k_learned_select8) at cfg5's per-GPU shape from in-kernel
stamps (workgroup 0 / thread 0, shader clocks): the LAST step of a 64-step rollout.  Diagnostic build in place of the
product library (GPU box only):
    make -C graph-conv-memory_amd/csrc stamps7 && cp graph-conv-memory_amd/gcm/_lib/libgcm_hip_stamps7.so \\
        graph-conv-memory_amd/gcm/_lib/libgcm_hip.so && python tools/kstamp_learned_select8.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-conv-memory_amd")]
import torch  # noqa: E402
import bench  # noqa: E402
from gcm import _hip  # noqa: E402

dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS["cfg5"])
mem, gnn, sel = bench.build_memory(dev, donate=True, selector="learned", cfg=c)
obs = bench.make_obs(c, 0, dev)
names = ["every load issued", "c0 / U[cur] / node row (waves 5 - 7; wave 0: nothing)", "barrier 1", "ReLU + LayerNorm 0 (U rows arrived)",
         "P1 (product)", "LayerNorm 1, logits", "barrier 2", "gumbel-softmax, adjacency row", "selected rows: list, gather",
         "layer 1 / layer 2 on row cur", "stores"]
lib = _hip.lib()
acc, R = [0.0] * len(names), 5
acc7 = [0.0] * 16     # every wave's arrival at barrier 1, relative to wave 0's first stamp
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
            acc[i] += (out[1 + i] - out[i]) / R
        for i in range(16):
            acc7[i] += (float(out[16 + i]) - float(out[0])) / R
print("k_learned_select8, step 63 of a rollout, workgroup 0 / thread 0        shader clocks")
for i, n in enumerate(names):
    print(f"  {i:2d} -> {i + 1:2d}  {n:56s} {acc[i]:9.1f}")
print(f"  total {sum(acc):9.1f}")
print("  first instruction, clocks relative to wave 0's first stamp (after its kernel arguments arrived), waves 0 - 7: "
      + " ".join("%.0f" % v for v in acc7[8:]))
print("  arrival at barrier 1, waves 0 - 7: " + " ".join("%.0f" % v for v in acc7[:8]))
