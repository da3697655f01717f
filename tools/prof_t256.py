#!/usr/bin/env python3
"""Per-kernel durations of a rollout of 2 x graph_size steps (the second half in the steady state: every step drops every
graph's oldest node) - the captured HIP graph of `rollout + backward` replayed under torch.profiler's device tracing.
bench.py prices the steady-state step from two graph times instead: kineto's trace teardown crashes now and then on
these graphs (a segmentation fault in 5 of 25 runs of cfg5 inside bench.py, and in this tool's first run), and a tool may
die where the bench line may not.
usage: rocprofv3 --kernel-trace --stats -d out -- python3 tools/prof_t256.py cfg5        (replays only; the profiler's table)
       python tools/prof_t256.py cfg5 --kineto                                            (in-process table; may crash)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-conv-memory_amd")]
import torch  # noqa: E402
import bench  # noqa: E402

name = next((a for a in sys.argv[1:] if not a.startswith("-")), "cfg5")
dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS[name])
N, T2 = c["N"], 2 * c["N"]
from gcm.gcm import DenseGCM  # noqa: E402
DenseGCM.did_warn = True
obs2 = bench.make_obs(dict(c, T=T2), 0, dev)
mem, gnn, sel = bench.build_memory(dev, donate=True, selector=c["selector"], cfg=c)
mods = [gnn] + ([sel] if c["selector"] == "learned" else [])


def zero():
    for q in mods:
        q.zero_grad(set_to_none=True)


g = bench.capture(lambda: bench.rollout(mem, obs2), zero)
t = bench.event_time(g.replay, 5)
print("%s, T = %d: replay %.3f ms = %.2f M belief-states/s" % (name, T2, t, c["B"] * T2 / t / 1e3))
if "--kineto" in sys.argv:
    prof = bench.profile_kernels(g.replay, reps=2)
    rows, total = bench.kernel_table(prof, top=12)
    print("%.1f us of kernels per rollout" % total)
    for r in rows:
        print("  %-90s n=%7.1f avg=%8.2f us share=%.3f" % (r["kernel"][:90], r["launches_per_step"], r["avg_us"], r["share_of_gpu_time"]))
