#!/usr/bin/env python3
"""Calibration of bench.py's `cpu_baseline` (kind "port"): wall time of oracle/dense.py against the
IMPORTED reference on identical inputs, in the build container only (needs /root/reference; the
reference never travels to the GPU box).  Prints one JSON line; the ratio is recorded in BASELINE.md.

  python tools/calibrate_cpu_baseline.py [T] [threads]
"""
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if not os.path.isdir("/root/reference/src"):
    sys.exit("needs /root/reference (build container only)")
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
sys.argv = sys.argv[:1] + ["__nothing__"] + sys.argv[1:]     # make_golden's fixture filter: match nothing
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
import torch  # noqa: E402

torch.set_num_threads(threads)
# only the placeholder installer is used from the generator (its main body is guarded)
src = open(os.path.join(ROOT, "tests", "golden", "make_golden.py")).read()
ns = {"__name__": "calib", "__file__": os.path.join(ROOT, "tests", "golden", "make_golden.py")}
exec(compile(src.split("ONLY = ")[0], "make_golden_head", "exec"), ns)
ns["install_placeholders"]()
from gcm.gcm import DenseGCM as RefGCM                                  # noqa: E402  (the reference)
from gcm.edge_selectors.temporal import TemporalBackedge as RefTB       # noqa: E402
from oracle import dense as od                                          # noqa: E402

B, N, F, H, HOPS = 256, 128, 32, 32, [1, 2, 4]
torch.manual_seed(0)
gnn = od.canonical_gnn(F, H)
obs = torch.rand(T, B, F)


def run_ref():
    mem = RefGCM(gnn, edge_selectors=RefTB(HOPS), graph_size=N)
    hidden, outs = None, []
    for t in range(T):
        mx, hidden = mem(obs[t], hidden)
        outs.append(mx)
    torch.stack(outs).mean().backward()
    gnn.zero_grad(set_to_none=True)


def run_port():
    out, _ = od.dense_rollout(obs, None, gnn, graph_size=N, edge_selectors=od.TemporalBackedge(HOPS))
    out.mean().backward()
    gnn.zero_grad(set_to_none=True)


def best(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


tr, tp = best(run_ref), best(run_port)
print(json.dumps({"config": f"cfg2 B={B} N={N} F={F} H={H} hops={HOPS} T={T} fwd+bwd", "threads": threads,
                  "reference_states_per_s": B * T / tr, "port_states_per_s": B * T / tp,
                  "port_over_reference_time": tp / tr}))
