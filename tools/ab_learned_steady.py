#!/usr/bin/env python3
"""Steady-state LearnedEdge step (k_learned_select<2, 2>) at cfg5's per-GPU shape: microseconds a step over steps
N .. 2 N of a donated chain (HIP events around the eager loop, which is GPU-bound here), and the density of the sampled
adjacency.  With `make -C graph-conv-memory_amd/csrc exp_ls`, run once per variant with that library copied over
libgcm_hip.so (GPU box only):
    for v in "" _ls1 _ls2 _ls4 _ls7; do cp graph-conv-memory_amd/gcm/_lib/libgcm_hip$v.so /tmp/x.so; ...; done"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS["cfg5"])
c["T"] = 2 * c["N"]
mem, gnn, sel = bench.build_memory(dev, donate=True, selector="learned", cfg=c)
obs = bench.make_obs(c, 0, dev)
N = c["N"]
best = 1e9
for it in range(6):
    with torch.no_grad():
        hidden = None
        for t in range(N):
            mx, hidden = mem(obs[t], hidden)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(N, 2 * N):
            mx, hidden = mem(obs[t], hidden)
        e1.record()
    torch.cuda.synchronize()
    if it:
        best = min(best, e0.elapsed_time(e1) * 1e3 / N)
def chain():
    with torch.no_grad():
        h = None
        for t in range(2 * N):
            _, h = mem(obs[t], h)


prof = bench.profile_kernels(chain, reps=2)
kern = {k: round(d["avg_us"], 2) for k, d in prof.items() if "k_learned_select" in k or "k_adj_bits" in k}
print("kernel durations (device activity trace), us:", kern)
adj = hidden[1]
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'product'}: steady step {best:7.2f} us (best of 5 chains), "
      f"steady steps {mem.learned_steady_steps_taken()}, adjacency density {float((adj != 0).float().mean()):.4f}")
