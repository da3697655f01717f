#!/usr/bin/env python3
"""bench.py - belief-states/sec of the DenseGCM hot path on MI355X.

Metric (BASELINE.json): belief-states/sec = B*T / wall time of
    reset state; for t in range(T): mx_t, m = gcm(obs[t], m); loss = stack(mx).mean();
    loss.backward(); all-reduce grads (N>1); synchronize
on cfg2: DenseGCM + TemporalBackedge([1,2,4]), B=256 per GPU, graph_size=128, obs=hidden=32,
2 x DenseGraphConv + tanh.  One "step" of this bench = one such rollout (B*T belief states)
through the per-step drop-in call surface.  Batch-sharded over ranks (weak scaling: every rank
owns B graphs), one RCCL all-reduce of the flat gradient bucket per backward.

`value` is measured with the module's donated-state mode (`DenseGCM(..., donate_state=True)`:
the step advances the hidden state in place instead of cloning it, same results) and with the
loop + backward captured once in a HIP graph and replayed (torch.cuda.CUDAGraph, in process);
the same loop run eagerly, with and without donation, is reported beside it (`variants`).

  python bench.py --gpus N --steps K --warmup W        # N > 1 without WORLD_SIZE: spawns N ranks
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "graph-conv-memory_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix == fp32 vector peak; no TF32 on gfx950
B, N, F, H = 256, 128, 32, 32
HOPS = [1, 2, 4]


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes BEFORE
    anything in this process touches the GPU (never re-exec a process that has), relay rank 0's
    JSON line and the worst exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = p.wait() or rc
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    sys.exit(rc)


def build_memory(device, donate=False, selector="temporal"):
    import torch
    from gcm import nn as G
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.temporal import TemporalBackedge

    torch.manual_seed(0)
    gnn = G.Sequential("x, adj, weights, B, N", [
        (G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
        (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(device)
    if selector == "learned":
        from gcm.edge_selectors.learned import LearnedEdge
        sel = LearnedEdge(F).to(device)
    else:
        sel = TemporalBackedge(HOPS)
    return DenseGCM(gnn, edge_selectors=sel, graph_size=N, donate_state=donate), gnn


def rollout(mem, obs, bucket=None, weight=1.0):
    import torch
    hidden, outs = None, []
    for t in range(obs.shape[0]):
        mx, hidden = mem(obs[t], hidden)
        outs.append(mx)
    loss = torch.stack(outs).mean()
    loss.backward()
    if bucket is not None:
        bucket.all_reduce_mean(weight)
    return loss


def rollout_api(mem, obs, bucket=None, weight=1.0):
    """Same work through the additive time-batched entry DenseGCM.rollout."""
    out, _ = mem.rollout(obs, None)
    loss = out.mean()
    loss.backward()
    if bucket is not None:
        bucket.all_reduce_mean(weight)
    return loss


def capture(mem, gnn, obs):
    """The per-step loop + backward as one HIP graph (captured in this process; the usual
    side-stream warm-up first).  Returns the graph; parameter .grad tensors are graph outputs."""
    import torch
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            gnn.zero_grad(set_to_none=True)
            rollout(mem, obs)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gnn.zero_grad(set_to_none=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rollout(mem, obs)
    return g


def time_step_kernel(mem, obs, reps=10):
    """Duration of the dominant kernel (k_step_rows, one launch per forward step) IN SITU: the T
    launches of a rollout through the C ABI on the evolving donated state, enqueued back to back (as the
    replayed graph does), a HIP event pair on the launch stream around every launch; mean over
    T x reps launches.  Also the time-parallel backward kernel on the records of one rollout."""
    import ctypes
    import torch
    from gcm import _hip

    lib = _hip.lib()
    dev = obs.device
    T = obs.shape[0]
    cfg = mem._fused_plan(*mem.get_initial_hidden_state(obs[0])[:3], F)
    params = mem._packed_params(cfg, head=True).detach()
    lay = (ctypes.c_size_t * 6)()
    lib.gcm_dense_rows_layout(B, N, F, H, H, ctypes.addressof(lay))
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    st, p = _hip.stream(), _hip.ptr
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]

    def new_event():
        e = ctypes.c_void_p()
        assert hip.hipEventCreate(ctypes.byref(e)) == 0
        return e

    evs = [(new_event(), new_event()) for _ in range(T)]
    spans = []
    saved_all = [torch.empty(lay[0], device=dev) for _ in range(T)]
    ev_a = (ctypes.c_void_p * T)(*[a for a, _ in evs])
    ev_b = (ctypes.c_void_p * T)(*[b for _, b in evs])
    sv_p = (ctypes.c_void_p * T)(*[t_.data_ptr() for t_ in saved_all])
    obs_c = obs.contiguous()
    for _ in range(reps + 1):
        nodes, adj, _, count = mem.get_initial_hidden_state(obs[0])
        # the T launches enqueued back to back from C (the cadence of the replayed graph that `value`
        # times), each bracketed by events recorded by the dispatch itself (kernel begin / end timestamps)
        rc = lib.gcm_debug_time_rows_rollout(p(obs_c), p(nodes), p(adj), p(count), cfg.arr_ptr, cfg.n_desc,
                                             p(params), cfg.has_bias, cfg.acts[0], cfg.acts[1], sv_p, p(flags),
                                             ev_a, ev_b, T, B, N, F, H, H, st)
        assert rc == 0
        torch.cuda.synchronize()
        ms = ctypes.c_float()
        row = []
        for a, b in evs:
            assert hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0
            row.append(ms.value)
        spans.append(row)
    for a, b in evs:
        hip.hipEventDestroy(a)
        hip.hipEventDestroy(b)
    step_ms = sum(sum(s) for s in spans[1:]) / (reps * T)
    # the backward kernel over the T records
    g_mx = torch.full((T, B, H), 1.0 / (T * B * H), device=dev)
    arr_s = (ctypes.c_void_p * T)(*[s.data_ptr() for s in saved_all])
    arr_g = (ctypes.c_void_p * T)(*[g_mx[t].data_ptr() for t in range(T)])
    ws_bytes = lib.gcm_dense_rows_bptt_workspace_bytes(T, B, F, H, H)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    gp = torch.empty(cfg.P, device=dev)
    times = []
    for _ in range(reps + 1):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = lib.gcm_dense_rows_bptt(arr_s, arr_g, T, H, 1, p(params), cfg.has_bias, cfg.acts[0], cfg.acts[1], None,
                                     p(gp), p(ws), ws_bytes, B, N, F, H, H, st)
        b.record()
        assert rc == 0
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
    return step_ms, sum(times[1:]) / reps


def cpu_baseline(T, budget_s=20.0):
    """The oracle (op-for-op eager-PyTorch restatement of the reference, kind "port") timed on
    this box's host cores on a BOUNDED sample of the same workload: the same B/N/F/H/selector,
    a rollout of T_s <= T steps fwd+bwd (T_s sized so the sample stays within ~budget_s).
    Thread policy: torch's own default is one thread per host CPU, which on a
    256-CPU host is several times SLOWER than 16-32 threads for these op sizes; the CPU gets its
    best configuration of {8, 16, 32, 64, os.cpu_count()} and the choice is recorded."""
    import torch
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

    ncpu = os.cpu_count() or 1
    tried = {}
    for th in sorted({t for t in (8, 16, 32, 64, ncpu) if t <= ncpu}):
        torch.set_num_threads(th)
        run(2)
        tried[th] = run(4)
    best = min(tried, key=tried.get)
    torch.set_num_threads(best)
    per_step = tried[best] / 4
    # per-step cost grows with t (autograd state), so size the sample conservatively
    T_s = int(max(8, min(T, budget_s / (2.5 * per_step))))
    dt = run(T_s)
    return {"value": B * T_s / dt, "unit": "belief-states/s", "cores": best, "kind": "port",
            "seconds": dt, "host_cpus": ncpu,
            "threads_tried_s_per_4_steps": {str(k): round(v, 3) for k, v in tried.items()},
            "sample": f"1 rollout fwd+bwd, same workload (B={B}, N={N}, F={F}, H={H}, hops={HOPS}) "
                      f"truncated to T={T_s} steps, oracle/dense.py on {best} torch threads "
                      f"(best of {sorted(tried)})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--T", type=int, default=128, help="rollout length (128 fills the graph)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="time the eager loop instead of the graph replay")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)

    import torch
    import torch.distributed as dist
    from gcm import parallel

    rank, local_rank, world = parallel.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    device = torch.device("cuda", local_rank)
    T = args.T
    mem, gnn = build_memory(device, donate=True)
    bucket = parallel.GradBucket(gnn)
    gen = torch.Generator().manual_seed(1000 + rank)
    # resident in HBM; like the reference's speed test and the CPU baseline, obs carries no grad
    obs = torch.rand(T, B, F, generator=gen).to(device)
    weight = 1.0 / world

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    graph = None if args.no_graph else capture(mem, gnn, obs)

    def step():
        if graph is not None:
            graph.replay()
            bucket.all_reduce_mean(weight)
        else:
            rollout(mem, obs, bucket, weight)
            gnn.zero_grad(set_to_none=True)

    def timed(fn, steps, warm):
        for _ in range(warm):
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        t = torch.tensor([time.perf_counter() - t0], device=device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    dt = timed(step, args.steps, args.warmup)
    # the flag word the kernels OR into (finite check, overflow, bad counts), read once
    flags = mem._flag_word(device)
    bits = int(flags.item())
    assert not (bits & 6), f"kernels flagged {bits}"

    # ---- the same work along the other paths (reported beside `value`) ---------------------------
    side = max(3, min(20, args.steps // 10))
    variants = {}
    mem_e, gnn_e = build_memory(device, donate=True)
    bucket_e = parallel.GradBucket(gnn_e)

    def eager(m, g, bk):
        def f():
            rollout(m, obs, bk, weight)
            g.zero_grad(set_to_none=True)
        return f

    variants["eager_donated"] = world * B * T * side / timed(eager(mem_e, gnn_e, bucket_e), side, 2)
    mem_f, gnn_f = build_memory(device, donate=False)
    bucket_f = parallel.GradBucket(gnn_f)
    variants["eager_functional"] = world * B * T * side / timed(eager(mem_f, gnn_f, bucket_f), side, 2)

    def roll():
        rollout_api(mem_f, obs, bucket_f, weight)
        gnn_f.zero_grad(set_to_none=True)

    variants["rollout_api"] = world * B * T * side / timed(roll, side, 2)

    def fwd_only():
        with torch.no_grad():
            hidden = None
            for t in range(T):
                _, hidden = mem_e(obs[t], hidden)

    variants["forward_only_eager_donated"] = world * B * T * side / timed(fwd_only, side, 1)
    mem_e.check_flags()
    mem_f.check_flags()

    step_ms, bptt_ms = time_step_kernel(mem_e, obs) if rank == 0 else (None, None)

    if rank == 0:
        states = world * B * T * args.steps
        traffic = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath))
        # Dominant kernel = k_step_rows: one launch per forward step, > 90 % of the GPU time of the
        # metric (the backward of the whole rollout is ONE launch of k_bptt_rows).  Bounding
        # roofline: HBM.  `achieved` = SURVEY 8(d)'s compulsory bytes per belief state (adj once, x
        # once, obs in, belief out, adj-row write-back: 4N^2+4NF+4F+4H+4N = 82.7 KB at cfg2) x B
        # over the mean launch time.  The kernel exploits "only row n_b is kept" (gcm.py:314): it
        # reads the node matrix, row cur and the live rows, not the [N,N] adjacency - `traffic` (PMC)
        # is what it actually moves, `achieved_moved` the rate of that.
        alg_bytes = B * (4 * N * N + 4 * N * F + 4 * F + 4 * H + 4 * N)
        sec = step_ms * 1e-3
        moved = traffic.get("k_step_rows")
        dominant = {"bound": "hbm", "kernel": "k_step_rows", "achieved": alg_bytes / sec / 1e9,
                    "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": alg_bytes / sec / 1e9 / PEAK_HBM_GBS,
                    "traffic": moved, "bytes_per_launch": alg_bytes, "avg_launch_ms": step_ms,
                    "launches_timed": 3 * T,
                    "achieved_moved": (moved / sec / 1e9) if moved else None,
                    "note": "bytes_per_launch = SURVEY 8(d) full-dense compulsory bytes (what the reference's "
                            "formulation must move); the live-row kernel moves `traffic` bytes (PMC: "
                            "2*FETCH_SIZE + WRITE_SIZE per launch, profiles/r02_traffic_detail.json) because only "
                            "the rows that reach the kept belief row are evaluated and the state is advanced "
                            "in place: the kernel is bound by its chain of dependent latencies (one wave per "
                            "SIMD at B = 256 graphs on 256 CUs), not by bytes. avg_launch_ms: the T launches "
                            "of a rollout in situ on the evolving state, enqueued back to back from C (the cadence of the "
                            "replayed graph the timed region runs), each bracketed by HIP events recorded by "
                            "the dispatch itself (hipExtLaunchKernelGGL start/stop events = the kernel begin/end "
                            "timestamps rocprofv3 --kernel-trace reports)"}
        fwd_full = 2 * N * N * (F + H) + 4 * N * (F * H + H * H)
        line = {
            "metric": "belief-states/sec (BxT) DenseGCM fwd+bwd, graph_size=128 F=32",
            "value": states / dt, "unit": "belief-states/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "cfg2: DenseGCM + TemporalBackedge([1,2,4]), B=256/GPU, graph_size=128, "
                                   "obs=32, hidden=32, 2x DenseGraphConv+tanh, T=%d; one bench step = one rollout "
                                   "through the per-step drop-in call surface `for t: mx, m = gcm(obs[t], m)` + "
                                   "backward" % T,
                       "B_per_gpu": B, "graph_size": N, "obs": F, "hidden": H, "T": T,
                       "state": "donated (DenseGCM(donate_state=True): hidden state advanced in place)",
                       "launch": "eager" if graph is None else "HIP graph of the loop + backward, captured once "
                                                               "in process, replayed per bench step",
                       "parallelism": f"dp{world} (batch-sharded, 1 flat-bucket all-reduce per backward)"},
            "variants": {k: round(v, 1) for k, v in variants.items()},
            "variants_note": "belief-states/s of the same workload: eager per-step loop with donated / "
                             "functional (reference-default) state, the additive DenseGCM.rollout entry, and the "
                             "forward loop alone under no_grad",
            "roofline": dominant,
            "roofline_mfma_view": {"kernel": "k_step_rows", "flops_per_launch_full_dense": B * fwd_full,
                                   "achieved_full_dense_TFLOPs": B * fwd_full / sec / 1e12,
                                   "frac_of_fp32_mfma_peak": B * fwd_full / sec / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                   "note": "SURVEY 8(d) full-dense FLOPs (2 layers x all N rows) over the same "
                                           "launch time; the kernel executes layer 1 on the live rows only"},
            "kernel_ms": {"k_step_rows": round(step_ms, 5), "k_bptt_rows(T=%d)" % T: round(bptt_ms, 5)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(T)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
