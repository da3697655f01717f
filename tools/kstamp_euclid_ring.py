#!/usr/bin/env python3
"""Phase stamps of the steady-state EuclideanEdge step (k_euclid_mfma2<.., 2>, gcm_edge_distance_step_ring) at cfg3's
shape on a state built from clustered observations (make -C graph-conv-memory_amd/csrc stamps9).  Dev tool."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-conv-memory_amd")]
os.environ.setdefault("GCM_HIP_LIB", os.path.join(ROOT, "graph-conv-memory_amd", "gcm", "_lib", "libgcm_hip_stamps9.so"))
import torch  # noqa: E402
import bench  # noqa: E402
from gcm import _hip  # noqa: E402

lib = _hip.lib()
dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS["cfg3"])
B, N, F, H = c["B"], c["N"], c["F"], c["H"]
T = 2 * N
obs = bench.make_obs(dict(c, T=T), 0, dev)
mem, gnn, _ = bench.build_memory(dev, donate=True, selector="euclid", cfg=c)
hid = None
with torch.no_grad():
    for t in range(N + 40):
        _, hid = mem(obs[t], hid)
torch.cuda.synchronize()
names = {0: "start", 1: "staging + barrier", 2: "", 3: "chunk 0", 4: "chunk 1 -> LDS", 5: "chunk 1", 6: "sums, decisions",
         12: "sDec/bits barrier", 13: "row N-1, bits out", 14: "weights -> regs", 15: "live rows (this wave's)",
         16: "barrier", 17: "row cur: agg over the selected", 18: "layer 1", 19: "layer 2, record"}
# the product path built the state; the stamped launches go through the C ABI of the diagnostic library
nodes, adj, _, count = hid
w = (adj.view(B, N, 4, 32) > 0).to(torch.int64) << torch.arange(32, device=dev)
abits = w.sum(-1).to(torch.int32).contiguous()            # [B, N, 4]: bit j of row i = adj[i, j]  (bit 31 wraps into the sign)
params = mem._packed_cache[1].detach().contiguous()
wimg = torch.empty(lib.gcm_dense_rows_cached_weight_image_floats(), device=dev)
st = _hip.stream()
assert lib.gcm_dense_rows_cached_weight_image(params.data_ptr(), wimg.data_ptr(), F, H, H, st) == 0
lay = (ctypes.c_size_t * 8)()
assert lib.gcm_dense_rows_layout(B, N, F, H, H, ctypes.addressof(lay)) == 0
saved = torch.empty(lay[0], device=dev)
flags = torch.zeros(1, dtype=torch.int32, device=dev)
acc = {}
R = 10
for it in range(R):
    rc = lib.gcm_edge_distance_step_ring(obs[N + 40 + it].data_ptr(), nodes.data_ptr(), adj.data_ptr(), count.data_ptr(),
                                         2.0, params.data_ptr(), wimg.data_ptr(), 1, 1, abits.data_ptr(), saved.data_ptr(),
                                         ctypes.addressof(lay), 1, flags.data_ptr(), B, N, F, H, H, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    lib.gcm_debug_read_stamps(out, 32)
    order = [0, 1, 2, 3, 4, 5, 6, 12, 13, 14, 15, 16, 17, 18, 19]
    for a, b_ in zip(order[:-1], order[1:]):
        acc[(a, b_)] = acc.get((a, b_), 0.0) + (out[b_] - out[a]) / R
print("selected per graph: %.1f" % float(adj[:, N - 1].sum(-1).mean()))
tot = 0.0
for (a, b_), v in acc.items():
    print("  %2d -> %2d  %-36s %9.1f" % (a, b_, names.get(b_, ""), v))
    tot += v
print("  total %9.1f cycles; rolled steps %d" % (tot, mem.rows_rolled_steps_taken()))
