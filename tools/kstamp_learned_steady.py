#!/usr/bin/env python3
"""Phase breakdown of the steady-state LearnedEdge step (k_learned_select<2, 2>: full graphs, roll + selection + the
GNN in one launch) at cfg5's per-GPU shape from in-kernel stamps (s_memtime of
workgroup 0 / thread 0, shader clocks): the LAST step of a 2 N-step chain.  Diagnostic build of the whole library in
place of the product one (run on the GPU box only):
    make -C graph-conv-memory_amd/csrc stamps7 && cp graph-conv-memory_amd/gcm/_lib/libgcm_hip_stamps7.so \\
        graph-conv-memory_amd/gcm/_lib/libgcm_hip.so && python tools/kstamp_learned_steady.py"""
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
c["T"] = 2 * c["N"]
mem, gnn, sel = bench.build_memory(dev, donate=True, selector="learned", cfg=c)
obs = bench.make_obs(c, 0, dev)
seq = [(14, 28, "count / addresses, every load issued, node image -> LDS, bit images"), (28, 29, "every load landed"),
       (29, 30, "barrier"), (30, 31, "previous layer 1 requested; which 16-byte pieces of the adjacency change"),
       (31, 15, "stores part 0"), (15, 16, "LDS images (+ barrier)"),
       (16, 17, "stores part 1; c0, P0 (product)"), (17, 18, "stores part 2; LayerNorm 0"), (18, 19, "stores part 3; P1 (product)"),
       (19, 20, "LayerNorm 1"), (20, 21, "logits (+ fence, barrier)"), (21, 22, "gumbel-softmax, adjacency row"),
       (22, 27, "row cur bits, barrier (waves 1-3: tiles that lost a source)"), (27, 24, "bit image out; selected rows gathered"),
       (24, 25, "h1[cur]"), (25, 26, "layer 2"), (26, 23, "stores")]
lib = _hip.lib()
acc, R = [0.0] * len(seq), 5
for it in range(R + 1):
    with torch.no_grad():
        hidden = None
        for t in range(c["T"]):
            mx, hidden = mem(obs[t], hidden)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    lib.gcm_debug_read_stamps(out, 32)
    if it >= 1:
        for i, (a, b, _) in enumerate(seq):
            acc[i] += (out[b] - out[a]) / R
print(f"steady steps taken: {mem.learned_steady_steps_taken()}")
print("k_learned_select<2, 2>, last step of a 2 N chain, workgroup 0 / thread 0        shader clocks")
for (a, b, n), v in zip(seq, acc):
    print(f"  {a:2d} -> {b:2d}  {n:58s} {v:9.1f}")
print(f"  total {sum(acc):9.1f}")
