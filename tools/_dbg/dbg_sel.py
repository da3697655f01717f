import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-conv-memory_amd")]
import torch
from gcm import _hip
lib = _hip.lib(); p = _hip.ptr; st = _hip.stream()
dev = "cuda:0"
B, N, F, H = 4, 16, 8, 16
torch.manual_seed(0)
P = lib.gcm_dense_gnn2_param_count(F, H, H)
params = torch.randn(P, device=dev) * 0.1
desc = _hip.SelectorDesc(); desc.kind = _hip.SEL_DISTANCE; desc.mode = 0; desc.max_distance = 2.0
arr = (_hip.SelectorDesc * 1)(desc)
wsb = lib.gcm_dense_rows_step_workspace_bytes(ctypes.addressof(arr), 1, B, N, F)
ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
lay = (ctypes.c_size_t * 6)(); lib.gcm_dense_rows_layout(B, N, F, H, H, ctypes.addressof(lay))
saved = torch.empty(lay[0], device=dev)
flags = torch.zeros(1, dtype=torch.int32, device=dev)
nodes = torch.zeros(B, N, F, device=dev); nodes[:, :4] = torch.tensor([0., 5, 10, 15], device=dev)[None, :, None]
adj = torch.zeros(B, N, N, device=dev)
cnt = torch.full((B,), 4, dtype=torch.long, device=dev)
obs = torch.zeros(B, F, device=dev) + 0.01   # close to node 0 only
for func in (False, True):
    n_in, a_in, c_in = nodes.clone(), adj.clone(), cnt.clone()
    n2, a2, c2 = (torch.empty_like(n_in), torch.empty_like(a_in), torch.empty_like(c_in)) if func else (n_in, a_in, c_in)
    rc = lib.gcm_dense_rows_step_fwd_ws(p(obs), p(n_in), p(a_in), p(c_in), p(n2), p(a2), p(c2), None, ctypes.addressof(arr), 1,
                                        p(params), 3, 1, 1, p(saved), p(saved), p(flags), p(ws), wsb, B, N, F, H, H, st)
    torch.cuda.synchronize()
    sel = ws[: B * N * 4].view(torch.float32).view(B, N)
    print("func", func, "rc", rc, "sel row g0:", sel[0, :6].tolist(), "adj row cur g0:", a2[0, 4, :6].tolist(), "count", c2.tolist(), "node cur", n2[0, 4, :2].tolist())
