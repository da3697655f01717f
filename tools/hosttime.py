"""Host-side cost of the per-step loop on cfg2: for each mode the time the interpreter needs to ISSUE a T-step
forward loop (no device sync inside), the time until the device has finished it, and the same for
loop + backward.  issue ~ done: host-bound; issue << done: GPU-bound.  Dev tool: `T=128 python tools/hosttime.py`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graph-conv-memory_amd")):
    sys.path.insert(0, p)
import torch
import bench

T = int(os.environ.get("T", 128))
dev = torch.device("cuda", 0)
c = bench.CONFIGS["cfg2"]
obs = torch.rand(T, c["B"], c["F"], device=dev)
rows = []
for donate in (True, False):
    mem, gnn, _ = bench.build_memory(dev, donate=donate)
    for grad in (False, True):
        res = []
        for it in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.set_grad_enabled(grad):
                hidden, outs = None, []
                for t in range(T):
                    mx, hidden = mem(obs[t], hidden)
                    outs.append(mx)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            tb0 = tb1 = tb2 = t2
            if grad:
                loss = torch.stack(outs).mean()
                tb0 = time.perf_counter()
                loss.backward()
                tb1 = time.perf_counter()
                torch.cuda.synchronize()
                tb2 = time.perf_counter()
                gnn.zero_grad(set_to_none=True)
            res.append((t1 - t0, t2 - t0, tb0 - t2, tb1 - tb0, tb2 - tb0))
        res = res[3:]
        m = [sum(r[i] for r in res) / len(res) for i in range(5)]
        print(f"donate={donate!s:5} grad={grad!s:5}  fwd issue {m[0] / T * 1e6:6.2f} us/step, done {m[1] / T * 1e6:6.2f} us/step"
              + (f" | stack+mean {m[2] * 1e6:6.0f} us, backward issue {m[3] * 1e6:6.0f} us, done {m[4] * 1e6:6.0f} us" if grad else ""))

# what the caller's own loop costs (no module call): obs[t] + list append
res = []
for it in range(6):
    t0 = time.perf_counter()
    outs = []
    for t in range(T):
        outs.append(obs[t])
    res.append(time.perf_counter() - t0)
print(f"caller's loop alone (obs[t] + append): {min(res) / T * 1e6:.2f} us/step")
xs = [obs[t] for t in range(T)]
for donate in (True, False):
    mem, gnn, _ = bench.build_memory(dev, donate=donate)
    res = []
    for it in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hidden, outs = None, []
        for x in xs:
            mx, hidden = mem(x, hidden)
            outs.append(mx)
        res.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        torch.stack(outs).mean().backward()
        gnn.zero_grad(set_to_none=True)
    print(f"donate={donate!s:5} pre-indexed observations: fwd issue {min(res[3:]) / T * 1e6:.2f} us/step")

# segments of RowsFast.step (extension built with GCM_HOST_PROF=1): us per step
from gcm import _ext
ext = _ext.module()
if hasattr(ext, "host_prof"):
    mem, gnn, _ = bench.build_memory(dev, donate=True)
    for it in range(4):
        ext.host_prof()
        t0 = time.perf_counter()
        hidden, outs = None, []
        for x in xs:
            mx, hidden = mem(x, hidden)
            outs.append(mx)
        dt = time.perf_counter() - t0
        v = ext.host_prof()
        torch.cuda.synchronize()
        torch.stack(outs).mean().backward()
        gnn.zero_grad(set_to_none=True)
    n = max(v[8], 1)
    names = ["checks", "record alloc", "launch", "mx alias", "autograd edge + rec", "wrap + tuple"]
    print("RowsFast.step segments (us/step, %d fast steps of %d): " % (n, T) + ", ".join("%s %.2f" % (nm, v[i] / n) for i, nm in enumerate(names))
          + "; sum %.2f of %.2f per loop iteration" % (sum(v[:6]) / n, dt / T * 1e6))

# the same loop on a non-default stream (the null stream's launches carry the legacy synchronisation bookkeeping)
side = torch.cuda.Stream()
for donate in (True, False):
    mem, gnn, _ = bench.build_memory(dev, donate=donate)
    res = []
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for it in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hidden, outs = None, []
            for x in xs:
                mx, hidden = mem(x, hidden)
                outs.append(mx)
            res.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
            torch.stack(outs).mean().backward()
            gnn.zero_grad(set_to_none=True)
    print(f"donate={donate!s:5} pre-indexed observations, side stream: fwd issue {min(res[3:]) / T * 1e6:.2f} us/step")
