#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc SQ_* passes (tools/collect_profiles.sh part `sq`): every MFMA kernel of the
bench configurations, few launches each (counter passes serialise dispatches).
  cfg3: k_euclid_mfma2 alone on full graphs (20 launches), one donated rollout T = 128 fwd + bwd (k_euclid_mfma2<..,TAIL>
        in situ, k_bptt_rows<64,..>), one rollout() call (the time-parallel distance contraction when built)
  cfg5: one donated rollout T = 64 (k_learned_select, k_learned_bptt_mlp / _sel, k_bptt_rows<..,2>), one rollout() call
        (k_learned_roll_logits / _pick / _l2)
  cfg2: one rollout() call (k_rollout_tp_l1 / _l2, k_bptt_rows<..,3>), the layered path T = 4 (k_layer_fwd / k_layer_bwd)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-conv-memory_amd"), os.path.join(ROOT, "tools")]
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
which = sys.argv[1:] or ["cfg3", "cfg5", "cfg2"]
from gcm.gcm import DenseGCM  # noqa: E402
DenseGCM.did_warn = True
if "cfg3" in which:
    launch = bench.euclid_launcher(bench.CONFIGS["cfg3"])
    for _ in range(20):
        launch()
for name in ("cfg3", "cfg5"):
    if name in which:
        c = bench.CONFIGS[name]
        obs = bench.make_obs(c, 0, dev)
        mem, gnn, sel = bench.build_memory(dev, donate=True, selector=c["selector"], cfg=c)
        bench.rollout(mem, obs)
        mem_f, _, _ = bench.build_memory(dev, donate=False, selector=c["selector"], cfg=c)
        bench.rollout_api(mem_f, obs)
        # 16 steps past graph_size: the steady-state steps k_euclid_mfma2<.., 2> / k_learned_select<2, 2, ..>
        bench.rollout(mem, bench.make_obs(dict(c, T=c["N"] + 16), 0, dev))
        torch.cuda.synchronize()
if "cfg2" in which:
    c = bench.CONFIGS["cfg2"]
    obs = bench.make_obs(c, 0, dev)
    mem_f, _, _ = bench.build_memory(dev, donate=False, selector="temporal", cfg=c)
    bench.rollout_api(mem_f, obs)
    import prof_layered
    for kind in ("three_layer",):
        mem, gnn = prof_layered.build(kind, dev, c)
        bench.rollout(mem, obs[:4])
torch.cuda.synchronize()
print("done")
