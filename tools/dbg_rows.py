import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "graph-conv-memory_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from oracle import dense as od
from test_rows_gpu import _mk, DEV

def run(B, N, F, H1, H2, T, sel, donate=False, seed=0):
    torch.manual_seed(seed)
    ref, g, mem, osel = _mk(B, N, F, H1, H2, sel, donate)
    obs = torch.rand(T, B, F)
    out_c, hid_c = od.dense_rollout(obs, None, ref, graph_size=N, edge_selectors=osel)
    wgt = torch.rand(T, B, H2)
    (out_c * wgt).sum().backward()
    hid, outs = None, []
    for t in range(T):
        mx, hid = mem(obs[t].to(DEV), hid)
        outs.append(mx)
    out_d = torch.stack(outs)
    (out_d * wgt.to(DEV)).sum().backward()
    print("case", B, N, F, H1, H2, T, sel, "out err", float((out_d.cpu() - out_c).abs().max()))
    for (k, pc), (_, pd) in zip(ref.named_parameters(), g.named_parameters()):
        e = (pd.grad.cpu() - pc.grad).abs().max() / (pc.grad.abs().max() + 1e-12)
        print("   ", k, "rel err %.3g" % float(e), "norm ref %.3g dev %.3g" % (float(pc.grad.norm()), float(pd.grad.norm())))

run(1, 8, 4, 32, 32, 1, ("temporal", [1], "forward"))
run(1, 8, 4, 32, 32, 2, ("temporal", [1], "forward"))
run(2, 8, 4, 32, 32, 5, ("temporal", [1], "forward"))
run(4, 32, 8, 32, 32, 20, ("temporal", [1], "forward"))
run(4, 32, 8, 32, 32, 40, ("temporal", [1], "forward"))
