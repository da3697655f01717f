import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-conv-memory_amd"), os.path.join(ROOT, "tests")]
import torch
from _golden import Fixture
from oracle import dense as od
from test_dense_gpu import dev_gnn_from, product_selector, DEV
from gcm.gcm import DenseGCM
fx = Fixture("g3_euclid"); m = fx.meta
print(m)
ref = od.canonical_gnn(m["F"], m["H"]); ref.load_state_dict(fx.group("param:"))
res = {}
for donate in (True, False):
    g = dev_gnn_from(ref, [(m["F"], m["H"], torch.nn.Tanh), (m["H"], m["H"], torch.nn.Tanh)])
    mem = DenseGCM(g, edge_selectors=product_selector(m, fx.group("sel_param:")), graph_size=m["N"], donate_state=donate)
    obs = fx["obs"].to(DEV)
    h0 = fx.h0()
    hidden = None if h0 is None else tuple(t.to(DEV).clone() for t in h0)
    adjs = []
    with torch.no_grad():
        for t in range(m["T"]):
            mx, hidden = mem(obs[t], hidden)
            adjs.append((hidden[1].clone().cpu(), hidden[0].clone().cpu(), hidden[3].clone().cpu(), mx.clone().cpu()))
    res[donate] = (adjs, hidden[3].cpu())
for t in range(m["T"]):
    for q, nm in ((1, "nodes"), (2, "count"), (3, "mx")):
        if not torch.equal(res[True][0][t][q], res[False][0][t][q]):
            print("step", t, nm, "differs", float((res[True][0][t][q].float() - res[False][0][t][q].float()).abs().max()))
    a, b = res[True][0][t][0], res[False][0][t][0]
    if not torch.equal(a, b):
        d = (a != b).nonzero()
        print("step", t, "differs at", d[:10].tolist(), "donated", [float(a[tuple(i)]) for i in d[:10]], "functional", [float(b[tuple(i)]) for i in d[:10]])
        break
else:
    print("functional == donated at every step")
print(torch.equal(res[True][0][-1][0], fx["hT_adj"]), torch.equal(res[False][0][-1][0], fx["hT_adj"]))
