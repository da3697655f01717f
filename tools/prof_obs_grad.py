import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graph-conv-memory_amd")):
    sys.path.insert(0, p)
import torch, bench
dev = torch.device("cuda", 0)
c = bench.CONFIGS["cfg2"]
mem, gnn, _ = bench.build_memory(dev, donate=False)
obs = bench.make_obs(c, 0, dev)
xs = [obs[t].clone().requires_grad_(True) for t in range(obs.shape[0])]
for _ in range(4):
    hidden, outs = None, []
    for x in xs:
        mx, hidden = mem(x, hidden); outs.append(mx)
    torch.stack(outs).mean().backward()
    gnn.zero_grad(set_to_none=True)
    for x in xs: x.grad = None
torch.cuda.synchronize()
