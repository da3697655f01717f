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
seq = [(14, 28, "addresses, every load issued, node image -> LDS, bit images, changed adjacency pieces stored"), (28, 29, "every load landed"),
       (29, 30, "barrier"), (30, 31, "previous layer 1 requested"),
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
B = c["B"]
sp = (ctypes.c_ulonglong * (2 * B))()
lib.gcm_debug_read_spans.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.gcm_debug_read_spans(sp, B)
st = [sp[2 * i] for i in range(B)]
en = [sp[2 * i + 1] for i in range(B)]
t0 = min(st)
dur = sorted((e - s_) / 100.0 for s_, e in zip(st, en))
off = sorted((s_ - t0) / 100.0 for s_ in st)
print(f"every workgroup of the last launch (s_memrealtime, us): first start -> last end {(max(en) - t0) / 100.0:.2f}; own duration min "
      f"{dur[0]:.2f} median {dur[B // 2]:.2f} p90 {dur[int(B * 0.9)]:.2f} max {dur[-1]:.2f}; start offsets median {off[B // 2]:.2f} "
      f"p90 {off[int(B * 0.9)]:.2f} max {off[-1]:.2f}")
