"""The kernels of ONE call of a bench workload in launch order, with durations and the gaps between them (torch.profiler
device activity).  usage: python tools/kernel_sequence.py cfg4 | cfg2 | cfg3 | cfg5"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-conv-memory_amd")]
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS[name])
if c["kind"] == "sparse":
    from gcm import nn as G
    from gcm.sparse_gcm import SparseGCM
    from gcm.sparse_edge_selectors.temporal import TemporalEdge
    B, N, F, H = c["B"], c["N"], c["F"], c["H"]
    torch.manual_seed(0)
    g = G.Sequential("x, edges, weights", [(G.GraphConv(F, H), "x, edges, weights -> x"), torch.nn.Tanh(),
                                           (G.GraphConv(H, H), "x, edges, weights -> x"), torch.nn.Tanh()]).to(dev)
    mem = SparseGCM(g, edge_selectors=TemporalEdge([1]), graph_size=N)
    x = torch.rand(B, N, F, device=dev)
    taus = torch.full((B,), N, dtype=torch.long, device=dev)

    def call():
        out, _ = mem(x, taus, None)
        out.mean().backward()
        g.zero_grad(set_to_none=True)
else:
    if len(sys.argv) > 2:
        c["T"] = int(sys.argv[2])
    mem, gnn, sel = bench.build_memory(dev, donate=True, selector=c["selector"], cfg=c)
    obs = bench.make_obs(c, 0, dev)
    mods = [gnn] + ([sel] if c["selector"] == "learned" else [])

    def call():
        bench.rollout(mem, obs)
        for m in mods:
            m.zero_grad(set_to_none=True)

for _ in range(3):
    call()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    call()
    torch.cuda.synchronize()
evs = sorted((e for e in prof.events() if e.device_time_total > 0), key=lambda e: e.time_range.start)
t0 = evs[0].time_range.start
prev_end = t0
print("%4s %9s %8s %8s  %s" % ("#", "start_us", "dur_us", "gap_us", "kernel"))
for i, e in enumerate(evs[:int(os.environ.get("MAXK", "80"))]):
    s, d = e.time_range.start - t0, e.time_range.end - e.time_range.start
    print("%4d %9.1f %8.2f %8.2f  %s" % (i, s, d, e.time_range.start - prev_end, bench.short_kernel_name(e.name)[:100]))
    prev_end = e.time_range.end
print("kernels: %d, span %.1f us, busy %.1f us" % (len(evs), evs[-1].time_range.end - t0,
                                                   sum(e.time_range.end - e.time_range.start for e in evs)))
