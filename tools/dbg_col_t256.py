import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-conv-memory_amd")]
if len(sys.argv) == 1:
    for v in ("eager_prof:256", "graph_prof:256", "graph_prof:129", "graph_noprof:256", "graph_prof_nobwd:256", "eager_prof_nobwd:256"):
        r = subprocess.run([sys.executable, "-X", "faulthandler", __file__, v], capture_output=True, text=True)
        print(v, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1], flush=True)
        if r.returncode:
            print("\n".join(r.stderr.strip().splitlines()[-12:]), flush=True)
    sys.exit(0)
import torch
import bench
from gcm.gcm import DenseGCM
DenseGCM.did_warn = True
mode, T = sys.argv[1].split(":")
T = int(T)
dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS["dense_edge"], T=T)
obs = bench.make_obs(c, 0, dev)
mem, gnn, sel = bench.build_memory(dev, donate=True, selector="dense", cfg=c)
bwd = "nobwd" not in mode

def run():
    if bwd:
        bench.rollout(mem, obs)
    else:
        with torch.no_grad():
            h = None
            for t in range(T):
                _, h = mem(obs[t], h)

def zero():
    gnn.zero_grad(set_to_none=True)

fn = run
if mode.startswith("graph"):
    g = bench.capture(run, zero)
    fn = g.replay
fn(); torch.cuda.synchronize()
if "noprof" in mode:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
else:
    p = bench.profile_kernels(fn, reps=2)
    print(len(p), "kernels")
print("ok", mem.rows_col_steps_taken())
