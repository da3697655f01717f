"""Does torch.profiler (kineto over roctracer) see the C-ABI kernels - launched eagerly and from a replayed HIP graph -
on this box?  Prints per-kernel average durations."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-conv-memory_amd")]
import torch
from torch.profiler import profile, ProfilerActivity

import bench

dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS["cfg2"])
mem, gnn, sel = bench.build_memory(dev, donate=True, selector="temporal", cfg=c)
obs = bench.make_obs(c, 0, dev)


def zero():
    gnn.zero_grad(set_to_none=True)


for _ in range(3):
    bench.rollout(mem, obs)
    zero()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        bench.rollout(mem, obs)
        zero()
    torch.cuda.synchronize()
print("EAGER")
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:8]:
    print("%-90s n=%6d avg=%8.2f us" % (e.key[:90], e.count, e.device_time_total / max(1, e.count)))

g = bench.capture(lambda: bench.rollout(mem, obs), zero)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
print("GRAPH")
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:8]:
    print("%-90s n=%6d avg=%8.2f us" % (e.key[:90], e.count, e.device_time_total / max(1, e.count)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gcm_debuglib
print("launch floor (graph us/node, empty dispatch duration us, back-to-back cadence us):", gcm_debuglib.launch_floor())
