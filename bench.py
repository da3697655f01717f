#!/usr/bin/env python3
"""bench.py - belief-states/sec of the DenseGCM hot path on MI355X.

Metric (BASELINE.json): belief-states/sec = B*T / wall time of
    reset state; for t in range(T): mx_t, m = gcm(obs[t], m); loss = stack(mx).mean();
    loss.backward(); all-reduce grads (N>1); synchronize
on cfg2: DenseGCM + TemporalBackedge([1,2,4]), B=256 per GPU, graph_size=128, obs=hidden=32,
2 x DenseGraphConv + tanh.  One "step" of this bench = one such rollout (B*T belief states).
Batch-sharded over ranks (weak scaling: every rank owns B graphs), one RCCL all-reduce of the
flat gradient bucket per backward.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "graph-conv-memory_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix == fp32 vector peak; no TF32 on gfx950
B, N, F, H = 256, 128, 32, 32
HOPS = [1, 2, 4]


def build_memory(device):
    from gcm import nn as G
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.temporal import TemporalBackedge

    torch.manual_seed(0)
    gnn = G.Sequential("x, adj, weights, B, N", [
        (G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
        (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(device)
    return DenseGCM(gnn, edge_selectors=TemporalBackedge(HOPS), graph_size=N), gnn


def rollout(mem, obs, bucket, weight):
    hidden, outs = None, []
    for t in range(obs.shape[0]):
        mx, hidden = mem(obs[t], hidden)
        outs.append(mx)
    loss = torch.stack(outs).mean()
    loss.backward()
    bucket.all_reduce_mean(weight)
    return loss


def rollout_api(mem, obs, bucket, weight):
    """Same work through the additive time-batched entry DenseGCM.rollout."""
    out, _ = mem.rollout(obs, None)
    loss = out.mean()
    loss.backward()
    bucket.all_reduce_mean(weight)
    return loss


def time_dominant_kernels(mem, obs, reps=200):
    """Mean launch duration of the one-kernel forward step (k_step_fwd_live) and of the GNN-only
    kernels k_gnn2_row_fwd / k_gnn2_row_bwd on the real end-of-rollout state."""
    from gcm import _hip, _ops

    lib = _hip.lib()
    with torch.no_grad():
        hidden = None
        for t in range(obs.shape[0]):
            _, hidden = mem(obs[t], hidden)
    nodes, adj, _, count = hidden
    dev = nodes.device
    cur = (count - 1).contiguous()
    cfg = mem._fused_plan(nodes, adj, torch.zeros(0, device=dev), F)
    w = cfg.unpack_ptrs(mem._packed_params(cfg).detach())
    P = cfg.P
    mx = torch.empty(B, H, device=dev)
    h1 = torch.empty(B, N, H, device=dev)
    agg1 = torch.empty(B, N, F, device=dev)
    agg2 = torch.empty(B, H, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    g_mx, g_no = torch.randn(B, H, device=dev), torch.randn(B, N, F, device=dev)
    g_ni, g_obs = torch.empty(B, N, F, device=dev), torch.empty(B, F, device=dev)
    slabs = torch.empty(B, P, device=dev)
    p, st = _hip.ptr, _hip.stream()

    def fwd():
        return lib.gcm_dense_gnn2_row_fwd(p(nodes), p(adj), p(cur), w[0], w[1], w[2], 1, w[3], w[4], w[5], 1,
                                          p(mx), p(h1), p(agg1), p(agg2), p(flags), B, N, F, H, H, st)

    def bwd():
        return lib.gcm_dense_gnn2_row_bwd(p(g_mx), p(g_no), p(nodes), p(adj), p(cur), p(cur), w[0], w[1], w[2], 1,
                                          w[3], w[4], w[5], 1, p(mx), p(h1), p(agg1), p(agg2), p(g_ni),
                                          p(g_obs), p(slabs), 0, B, N, F, H, H, st)

    # the one-kernel forward step (advance + selector + GNN) on the state one step earlier
    with torch.no_grad():
        hid = None
        for t in range(obs.shape[0] - 1):
            _, hid = mem(obs[t], hid)
    n_in, a_in, _, c_in = hid
    n_out, a_out = torch.empty_like(n_in), torch.empty_like(a_in)
    ibuf = torch.empty(2, B, dtype=torch.int64, device=dev)
    x_last = obs[-1].contiguous()

    def step():
        return lib.gcm_dense_step_fused_fwd(p(x_last), p(n_in), p(a_in), p(c_in), p(n_out), p(a_out),
                                            ibuf.data_ptr(), ibuf.data_ptr() + 8 * B, cfg.arr_ptr, cfg.n_desc,
                                            w[0], w[1], w[2], 1, w[3], w[4], w[5], 1, p(mx), p(h1), p(agg1),
                                            p(agg2), p(flags), B, N, F, H, H, st)

    out = {}
    for name, fn in (("k_step_fwd_live", step), ("k_gnn2_row_fwd", fwd), ("k_gnn2_row_bwd", bwd)):
        for _ in range(10):
            assert fn() == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        out[name] = (reps, a.elapsed_time(b) / reps)
    return out


def cpu_baseline(T, budget_s=20.0):
    """The oracle (op-for-op eager-PyTorch restatement of the reference, kind "port") timed on
    this box's host cores on a BOUNDED sample of the same workload: the same B/N/F/H/selector,
    a rollout of T_s <= T steps fwd+bwd (T_s sized so the sample stays within ~budget_s).
    The thread count is calibrated first (8..64): eager torch on 256 threads is far slower
    than on 16-32 for these op sizes, so the CPU gets its best configuration."""
    from oracle import dense as od

    torch.manual_seed(0)
    gnn = od.canonical_gnn(F, H)
    sel = od.TemporalBackedge(HOPS)
    obs = torch.rand(T, B, F)

    def run(steps):
        t0 = time.perf_counter()
        out, _ = od.dense_rollout(obs[:steps], None, gnn, graph_size=N, edge_selectors=sel)
        out.mean().backward()
        gnn.zero_grad(set_to_none=True)
        return time.perf_counter() - t0

    best = None
    for th in [t for t in (8, 16, 32, 64) if t <= (os.cpu_count() or 8)] or [os.cpu_count() or 1]:
        torch.set_num_threads(th)
        run(2)
        dt = run(4)
        if best is None or dt < best[1]:
            best = (th, dt)
    torch.set_num_threads(best[0])
    per_step = best[1] / 4
    # per-step cost grows with t (autograd state), so size the sample conservatively
    T_s = int(max(8, min(T, budget_s / (2.5 * per_step))))
    dt = run(T_s)
    return {"value": B * T_s / dt, "unit": "belief-states/s", "cores": best[0], "kind": "port",
            "seconds": dt, "host_cpus": os.cpu_count(),
            "sample": f"1 rollout fwd+bwd, same workload (B={B}, N={N}, F={F}, H={H}, hops={HOPS}) "
                      f"truncated to T={T_s} steps, oracle/dense.py on {best[0]} torch threads "
                      f"(best of 8/16/32/64)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--T", type=int, default=128, help="rollout length (128 fills the graph)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from gcm import _ops, parallel

    rank, local_rank, world = parallel.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    device = torch.device("cuda", local_rank)
    T = args.T
    mem, gnn = build_memory(device)
    bucket = parallel.GradBucket(gnn)
    gen = torch.Generator().manual_seed(1000 + rank)
    obs = torch.rand(T, B, F, generator=gen).to(device)   # resident in HBM; like the reference's speed test and the CPU baseline, obs carries no grad
    weight = 1.0 / world

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        rollout(mem, obs, bucket, weight)
        gnn.zero_grad(set_to_none=True)
        obs.grad = None
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rollout(mem, obs, bucket, weight)
        gnn.zero_grad(set_to_none=True)
        obs.grad = None
    sync()
    dt = time.perf_counter() - t0
    mem.check_flags()
    t = torch.tensor([dt], device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # ---- the same work through the time-batched entry (reported beside `value`) ---------------
    def timed(fn, steps):
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn(mem, obs, bucket, weight)
            gnn.zero_grad(set_to_none=True)
            obs.grad = None
        sync()
        t = torch.tensor([time.perf_counter() - t0], device=device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    timed(rollout_api, 2)
    dt_roll = timed(rollout_api, args.steps)

    # forward only (inference, no autograd graph): SURVEY 8(d) asks for it beside fwd+bwd
    def fwd_step(mem, obs, bucket, weight):
        with torch.no_grad():
            hidden = None
            for t in range(obs.shape[0]):
                _, hidden = mem(obs[t], hidden)

    def fwd_roll(mem, obs, bucket, weight):
        with torch.no_grad():
            mem.rollout(obs, None)

    timed(fwd_step, 1)
    dt_fwd = timed(fwd_step, args.steps)
    timed(fwd_roll, 1)
    dt_fwd_roll = timed(fwd_roll, args.steps)

    # ---- kernel durations with HIP events on the launch stream -------------------------------
    # (a) in situ: event pairs around every C-ABI call of a repeated timed region (a step call
    #     enqueues state-advance + selector + fused GNN kernels, so these are per-call sums)
    _ops.TIMER = _ops.KernelTimer()
    for _ in range(min(args.steps, 3)):
        rollout(mem, obs, bucket, weight)
        gnn.zero_grad(set_to_none=True)
        obs.grad = None
    torch.cuda.synchronize()
    kern = _ops.TIMER.summary()
    # ... and around the two C-ABI calls of the rollout entry (persistent forward; backward =
    # time-parallel BPTT + reverse scan + one slab sum)
    _ops.TIMER = _ops.KernelTimer()
    for _ in range(min(args.steps, 3)):
        rollout_api(mem, obs, bucket, weight)
        gnn.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    kern_roll = _ops.TIMER.summary()
    _ops.TIMER = None
    # (b) the two dominant kernels alone: R back-to-back launches through the C ABI on the live
    #     state of this workload (graph after T steps), one event pair around the batch
    kern.update(time_dominant_kernels(mem, obs))

    if rank == 0:
        states = world * B * T * args.steps
        # full-dense algorithmic FLOPs per belief state (both layers on all N rows)
        fwd_full = 2 * N * N * (F + H) + 4 * N * (F * H + H * H)
        # what the fused kernels execute: layer 1 on all rows, layer 2 only on the kept row
        fwd_exec = 2 * N * N * F + 4 * N * F * H + 2 * N * H + 4 * H * H
        bwd_exec = 2 * N * N * F + 8 * N * F * H + 2 * N * H + 8 * H * H
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath))

        def mfma_view(kernel, full, execd):
            n_launch, ms = kern[kernel]
            sec = ms * 1e-3
            return {"bound": "mfma", "kernel": kernel, "achieved": B * full / sec / 1e12,
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": B * full / sec / 1e12 / PEAK_F32_MFMA_TFLOPS,
                    "traffic": traffic.get(kernel), "flops_per_launch": B * full,
                    "avg_launch_ms": ms, "launches_timed": n_launch,
                    "achieved_executed": B * execd / sec / 1e12,
                    "frac_executed": B * execd / sec / 1e12 / PEAK_F32_MFMA_TFLOPS}

        # Dominant kernel = k_step_fwd_live (one kernel per forward step).  It is HBM-bound
        # (AI 8-17 FLOP/B < ridge 20-25): SURVEY 8(d)'s compulsory bytes per belief state
        # (adj once, x once, obs in, belief out, adj-row write-back: 4N^2+4NF+4F+4H+4N = 82.7 KB at cfg2)
        # over its mean launch time; `functional` adds what the reference's functional state
        # semantics force through HBM (adj + nodes copied out every step, gcm.py:262-286) and the
        # activations saved for BPTT - only the live 32-row tiles of h1 / agg1 (1 of 4 on the timed
        # state: node 127 links to 126, 125, 123).
        alg_bytes = B * (4 * N * N + 4 * N * F + 4 * F + 4 * H + 4 * N)
        live_rows = 32
        func_bytes = B * (2 * 4 * N * N + 2 * 4 * N * F + 4 * F + 4 * H + 16
                          + 4 * live_rows * H + 4 * live_rows * F + 4 * H)
        n_launch, ms = kern["k_step_fwd_live"]
        sec = ms * 1e-3
        dominant = {"bound": "hbm", "kernel": "k_step_fwd_live", "achieved": alg_bytes / sec / 1e9,
                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": alg_bytes / sec / 1e9 / PEAK_HBM_GBS,
                    "traffic": traffic.get("k_step_fwd_live"), "bytes_per_launch": alg_bytes,
                    "avg_launch_ms": ms, "launches_timed": n_launch,
                    "achieved_functional": func_bytes / sec / 1e9,
                    "frac_functional": func_bytes / sec / 1e9 / PEAK_HBM_GBS,
                    "functional_bytes_per_launch": func_bytes,
                    "note": "bytes_per_launch = SURVEY 8(d) compulsory bytes (in-place state); the kernel "
                            "also copies adj+nodes out (functional hidden state, gcm.py:262-286) and saves "
                            "the live tiles of h1/agg1 for BPTT = functional_bytes_per_launch, which is what "
                            "`traffic` (PMC: 2*FETCH_SIZE+WRITE_SIZE) measures. avg_launch_ms: 200 "
                            "back-to-back launches on the end-of-rollout state, one HIP event pair"}
        note = ("flops: SURVEY 8(d) full-dense (2 layers x all N rows); *_executed: what the kernel can "
                "execute at most - layer 1 on all rows (GNN-only kernels) or on the live 32-row tile "
                "(k_step_fwd_live), row-only layer 2 (gcm.py:314); all-zero 32x32 adjacency tiles are skipped")
        fwd_live = 2 * live_rows * N * F + 4 * live_rows * F * H + 2 * N * H + 4 * H * H
        other = [dict(mfma_view("k_gnn2_row_bwd", 2 * fwd_full, bwd_exec), note=note),
                 dict(mfma_view("k_step_fwd_live", fwd_full, fwd_live), note=note),
                 dict(mfma_view("k_gnn2_row_fwd", fwd_full, fwd_exec), note=note)]
        from gcm import _ext
        line = {
            "metric": "belief-states/sec (BxT) DenseGCM fwd+bwd, graph_size=128 F=32",
            "value": states / dt, "unit": "belief-states/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "cfg2: DenseGCM + TemporalBackedge([1,2,4]), B=256/GPU, graph_size=128, "
                                   "obs=32, hidden=32, 2x DenseGraphConv+tanh, T=%d; one bench step = one rollout "
                                   "through the per-step drop-in API `for t: mx, m = gcm(obs[t], m)` + backward" % T,
                       "B_per_gpu": B, "graph_size": N, "obs": F, "hidden": H, "T": T,
                       "parallelism": f"dp{world} (batch-sharded, 1 flat-bucket all-reduce per backward)"},
            "forward_only": {"per_step_api": states / dt_fwd, "rollout_api": states / dt_fwd_roll,
                             "unit": "belief-states/s", "note": "torch.no_grad(): no history kept"},
            "host_path": "c++ autograd node (gcm/_lib/ext)" if _ext.module() is not None
                         else "python autograd function",
            "roofline": dominant, "roofline_mfma_view": other,
            "kernel_ms": {k: round(v[1], 5) for k, v in kern.items()},
            "rollout_api": {"value": states / dt_roll, "unit": "belief-states/s",
                            "ms_per_step": dt_roll / args.steps * 1e3,
                            "kernel_ms": {k: round(v[1], 5) for k, v in kern_roll.items()},
                            # SURVEY 8(d) bytes of T steps over the persistent forward kernel's time: an
                            # EFFECTIVE rate - the kernel keeps the state in LDS and does not move them
                            "forward_effective_GBps": (T * alg_bytes / (kern_roll["gcm_dense_rollout_fwd"][1] * 1e-3) / 1e9
                                                       if "gcm_dense_rollout_fwd" in kern_roll else None),
                            "note": "same workload and results through the additive DenseGCM.rollout(obs[T,B,F]) "
                                    "entry, one autograd node: persistent forward kernel (graph state resident "
                                    "in LDS for all T steps), time-parallel BPTT (one launch over T*B graph-steps "
                                    "+ reverse scan of the node gradient)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(T)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
