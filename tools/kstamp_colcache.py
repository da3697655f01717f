#!/usr/bin/env python3
"""The column-write cached step (k_step_colcache, csrc/rows_colcache.hip) alone at cfg2's shapes with DenseEdge:
duration per launch at fixed cur (HIP events around back-to-back launches, record on / off) and, with the diagnostic
build (make -C graph-conv-memory_amd/csrc stamps11), the phase breakdown of workgroup 0 / wave 0 from in-kernel
cycle stamps.  Dev tool."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-conv-memory_amd")]
from gcm import _hip  # noqa: E402

B, N, F, H = 256, 128, int(os.environ.get("F", 32)), 32
dev = "cuda:0"
torch.manual_seed(0)
sel = _hip.SelectorDesc(kind=_hip.SEL_DENSE)
V = ctypes.c_void_p
p = lambda t: V(t.data_ptr())
st = V(torch.cuda.current_stream().cuda_stream)
nodes = torch.rand(B, N, F, device=dev)
adj = torch.zeros(B, N, N, device=dev)
count = torch.zeros(B, dtype=torch.int64, device=dev)
obs = torch.rand(B, F, device=dev)
P = 2 * H * F + H + 2 * H * H + H
params = torch.randn(P, device=dev) * 0.1
cA, cR = torch.rand(B, N, F, device=dev), torch.rand(B, N, H, device=dev)
lay = (ctypes.c_size_t * 8)()
_hip.lib().gcm_dense_rows_layout(B, N, F, H, H, ctypes.addressof(lay))
saved = torch.empty(lay[0], device=dev)
flags = torch.zeros(1, dtype=torch.int32, device=dev)


FOUR = 256   # GCM_STEP_FOUR_WAVES: the four-wave kernel (the one the stamps are in) where the eight-wave form exists


def run(lib, cur, record, four=False):
    count.fill_(cur)
    rc = lib.gcm_dense_rows_step_colcache(p(obs), p(nodes), p(adj), p(count), ctypes.byref(sel), 1, p(params),
                                          3 | (FOUR if four else 0), 1, 1, p(cA), p(cR), p(saved), record, cur, p(flags),
                                          B, N, F, H, H, st)
    assert rc == 0, rc


lib = _hip.lib()
for four in (False, True):
    for record in (1, 0):
        for cur in (0, 31, 63, 95, 127):
            for _ in range(3):
                run(lib, cur, record, four)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                run(lib, cur, record, four)
            e1.record()
            torch.cuda.synchronize()
            print(f"{'four' if four else 'eight'} waves record={record} cur={cur:3d}: {e0.elapsed_time(e1) / 50 * 1e3:6.2f} us per (fill + launch)")
assert int(flags.item()) == 0, int(flags.item())
sp = os.path.join(ROOT, "graph-conv-memory_amd", "gcm", "_lib", "libgcm_hip_stamps11.so")
if os.path.exists(sp):
    ls = ctypes.CDLL(sp)
    names = ["loads issued", "count arrived", "new row's partial sums, root row", "barrier 1", "row sums, barrier 2",
             "A operand assembled (cache rows arrived), cache / record stores", "matrix product", "activation, agg2, record h1",
             "barrier 3", "layer 2", "record v / hdr"]
    for cur in (31, 63, 127):
        acc, R = [0.0] * 11, 10
        for it in range(R + 3):
            run(ls, cur, 1, True)
            torch.cuda.synchronize()
            out = (ctypes.c_ulonglong * 32)()
            ls.gcm_debug_read_stamps(out, 32)
            if it >= 3:
                for i in range(11):
                    acc[i] += (out[i + 1] - out[i]) / R
        print(f"k_step_colcache cur={cur}, workgroup 0, wave 0     cycles (100 MHz s_memtime ticks x 24 at 2.4 GHz)")
        for i in range(11):
            print(f"  {i:2d} -> {i + 1:2d}  {names[i]:66s} {acc[i]:9.1f}")
        print(f"  total {sum(acc):9.1f}")
