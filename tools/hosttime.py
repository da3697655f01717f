"""Host-side timing of the per-step loop on cfg2: eager with functional / donated state (fwd+bwd, forward only
under no_grad and in grad mode) and the HIP-graph replay.  Dev tool: `T=128 python tools/hosttime.py`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graph-conv-memory_amd")):
    sys.path.insert(0, p)
import torch
import bench

B, N, F, H, T = 256, 128, 32, 32, int(os.environ.get("T", 128))
dev = torch.device("cuda", 0)


def build(donate):
    mem, gnn = bench.build_memory(dev)
    mem.donate_state = donate
    return mem, gnn


def rollout(mem, obs):
    hidden, outs = None, []
    for t in range(obs.shape[0]):
        mx, hidden = mem(obs[t], hidden)
        outs.append(mx)
    loss = torch.stack(outs).mean()
    loss.backward()
    return loss


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


obs = torch.rand(T, B, F).to(dev)
for donate in (False, True):
    mem, gnn = build(donate)
    dt = timeit(lambda: (rollout(mem, obs), gnn.zero_grad(set_to_none=True)))
    print(f"eager donate={donate}: {dt*1e3:.3f} ms/rollout  {B*T/dt/1e6:.2f} M states/s")
    # forward only timing (no_grad)
    def fwd():
        with torch.no_grad():
            h = None
            for t in range(T):
                _, h = mem(obs[t], h)
    dt = timeit(fwd)
    print(f"   fwd-only no_grad: {dt*1e3:.3f} ms  {B*T/dt/1e6:.2f} M states/s")
    # fwd with grad, without backward
    def fwd_g():
        h, outs = None, []
        for t in range(T):
            mx, h = mem(obs[t], h)
            outs.append(mx)
        return outs
    dt = timeit(fwd_g)
    print(f"   fwd-only grad mode: {dt*1e3:.3f} ms")

# graph replay
mem, gnn = build(True)
mem.finite_check = "off"
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        gnn.zero_grad(set_to_none=True)
        rollout(mem, obs)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
gnn.zero_grad(set_to_none=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = rollout(mem, obs)
dt = timeit(lambda: g.replay(), n=20)
print(f"graph replay donated: {dt*1e3:.3f} ms/rollout  {B*T/dt/1e6:.2f} M states/s")
