#!/usr/bin/env python3
"""Short workloads for rocprofv3 --pmc passes (HBM traffic of the dominant kernels of every bench config),
T shortened where the kernel's traffic does not depend on it.  Run once per counter:
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_run.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_run.py
then tools/pmc_summarise.py writes profiles/<tag>_traffic_detail.json and profiles/traffic.json.
  cfg2: 2 eager per-step rollouts fwd+bwd donated with cached steps (k_step_rows_cached, k_bptt_rows<..,3>) + one of
        T = 256 (k_step_rows_cached_roll), 2 without
        (k_step_rows<32,..,false>, k_bptt_rows), 2 functional
        (k_step_rows<32,..,true>), 2 rollout-API calls, 2 with observation gradients donated (k_rows_dx_all), T=128
  cfg3: 1 rollout donated, T=128 (k_euclid_mfma, k_step_rows<64,...>)
  cfg5: 1 rollout donated, T=64 (k_learned_select, k_gnn2_row_fwd, k_learned_bptt_sel, k_learned_bptt_mlp, k_bptt_rows mode 2)
  cfg4: 1 one-shot call fwd+bwd (k_csr_fwd3, k_csr_bwd3, packing kernels)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graph-conv-memory_amd"))
import torch  # noqa: E402
import bench  # noqa: E402

which = sys.argv[1:] or ["cfg2", "cfg3", "cfg5", "cfg4"]
dev = torch.device("cuda", 0)
if "cfg2" in which:
    c = bench.CONFIGS["cfg2"]
    obs = torch.rand(128, c["B"], c["F"]).to(dev)
    obs256 = torch.rand(256, c["B"], c["F"]).to(dev)
    from gcm.gcm import DenseGCM
    DenseGCM.did_warn = True
    for donate, cached in ((True, True), (True, False), (False, False)):
        mem, gnn, _ = bench.build_memory(dev, donate=donate)
        mem.rows_cached_steps = cached      # (True: k_step_rows_cached, what the bench's timed region runs)
        for _ in range(2):
            bench.rollout(mem, obs)
            gnn.zero_grad(set_to_none=True)
        if donate and cached:               # T = 2N: the steady-state cached step (k_step_rows_cached_roll)
            bench.rollout(mem, obs256)
            gnn.zero_grad(set_to_none=True)
        if not donate:
            for _ in range(2):
                bench.rollout_api(mem, obs)
                gnn.zero_grad(set_to_none=True)
    mem, gnn, _ = bench.build_memory(dev, donate=True)
    for _ in range(2):
        xs = [obs[t].clone().requires_grad_(True) for t in range(128)]
        hidden, outs = None, []
        for x in xs:
            mx, hidden = mem(x, hidden)
            outs.append(mx)
        torch.stack(outs).mean().backward()
        gnn.zero_grad(set_to_none=True)
for name, T in (("cfg3", 128), ("cfg5", 64)):
    if name in which:
        c = bench.CONFIGS[name]
        obs = bench.make_obs(dict(c, T=T), 0, dev)
        mem, gnn, sel = bench.build_memory(dev, donate=True, selector=c["selector"], cfg=c)
        bench.rollout(mem, obs)
        # the time-batched entry: cfg5 k_learned_roll_logits / _pick / _l2; cfg3 k_euclid_tp / k_euclid_tp_gnn (round 5)
        bench.rollout_api(mem, obs)
        if name == "cfg3":                  # the selector alone on full graphs (bench.py's roofline kernel: k_euclid_mfma2<.., 0>)
            launch = bench.euclid_launcher(c)
            for _ in range(8):
                launch()
        if name == "cfg3":                  # ... and 32 steps past graph_size: the steady-state step k_euclid_mfma2<.., 2>
            from gcm.gcm import DenseGCM
            DenseGCM.did_warn = True
            bench.rollout(mem, bench.make_obs(dict(c, T=T + 32), 0, dev))
        if name == "cfg5":                  # ... the LearnedEdge chain 32 steps past graph_size: k_learned_select<2, 2, ..>
            from gcm.gcm import DenseGCM
            DenseGCM.did_warn = True
            bench.rollout(mem, bench.make_obs(dict(c, T=c["N"] + 32), 0, dev))
        torch.cuda.synchronize()
if "cfg4" in which:
    from gcm import nn as G
    from gcm.sparse_gcm import SparseGCM
    from gcm.sparse_edge_selectors.temporal import TemporalEdge
    c = bench.CONFIGS["cfg4"]
    B, N, F, H = c["B"], c["N"], c["F"], c["H"]
    g = G.Sequential("x, edges, weights", [(G.GraphConv(F, H), "x, edges, weights -> x"), torch.nn.Tanh(),
                                           (G.GraphConv(H, H), "x, edges, weights -> x"), torch.nn.Tanh()]).to(dev)
    mem = SparseGCM(g, edge_selectors=TemporalEdge([1]), graph_size=N)
    x = torch.rand(B, N, F, device=dev)
    taus = torch.full((B,), N, dtype=torch.long, device=dev)
    for _ in range(2):
        out, _ = mem(x, taus, None)
        out.mean().backward()
torch.cuda.synchronize()
print("done")
