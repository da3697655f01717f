import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "graph-conv-memory_amd")):
    sys.path.insert(0, p)
import torch
import bench
T = 128
dev = torch.device("cuda", 0)
c = bench.CONFIGS["cfg2"]
obs = torch.rand(T, c["B"], c["F"], device=dev)
xs = [obs[t] for t in range(T)]
def t_(f, n=5):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); f(); dt = time.perf_counter() - t0
        best = min(best, dt)
    return best / T * 1e6
print("obs[t] select          %.2f us" % t_(lambda: [obs[t] for t in range(T)]))
for donate in (True, False):
    for grad in (False, True):
        mem, gnn, _ = bench.build_memory(dev, donate=donate)
        with torch.set_grad_enabled(grad):
            mx, hid = mem(xs[0], None)
            fast = mem._fast[0]
            st = [hid]
            def direct():
                h = st[0]
                for t in range(T):
                    r = fast(xs[t], h); h = r[1]
                st[0] = h
            def call():
                h = st[0]
                for t in range(T):
                    mx, h = mem(xs[t], h)
                st[0] = h
            def full():
                h = None; outs = []
                for t in range(T):
                    mx, h = mem(obs[t], h); outs.append(mx)
            print(f"donate={donate} grad={grad}: fast.step {t_(direct):.2f}  mem() {t_(call):.2f}  full loop {t_(full):.2f} us/step (issue only)")
import ctypes
from gcm import _hip
lib=_hip.lib()
hip = ctypes.CDLL("libamdhip64.so")
