"""Same-box A/B aid: the cfg2 timed region (per-step loop + backward, donated, one captured HIP graph) replayed under
HIP events - run from two checkouts (e.g. a `git worktree` of an older commit built in-tree) to compare them on ONE
lease: boxes of the pool differ by more than most kernel changes.  `python tools/ab_bptt.py`"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "graph-conv-memory_amd"))
import torch, bench
dev = torch.device("cuda", 0)
c = bench.CONFIGS["cfg2"]
mem, gnn, _ = bench.build_memory(dev, donate=True)
obs = bench.make_obs(c, 0, dev)
def zero(): gnn.zero_grad(set_to_none=True)
g = bench.capture(lambda: bench.rollout(mem, obs), zero)
for rep in range(3):
    ms = bench.event_time(g.replay, 200, warm=20)
    print("rollout %.4f ms  %.2f M" % (ms, c["B"] * c["T"] / ms / 1e3))
