#!/usr/bin/env python3
"""bench.py - belief-states/sec of the DenseGCM / SparseGCM hot path on MI355X.

Metric (BASELINE.json): belief-states/sec = B*T / wall time of
    reset state; for t in range(T): mx_t, m = gcm(obs[t], m); loss = stack(mx).mean();
    loss.backward(); all-reduce grads (N>1); synchronize
Default workload = cfg2 (the configuration the metric is quoted on): DenseGCM + TemporalBackedge([1,2,4]),
B=256 per GPU, graph_size=128, obs=hidden=32, 2 x DenseGraphConv + tanh, T=128.  One "step" of this
bench = one such rollout (B*T belief states) through the per-step drop-in call surface.  Batch-sharded
over ranks (weak scaling: every rank owns B graphs), one RCCL all-reduce of the flat gradient bucket
per backward.

`value` (cfg2 / cfg3) is measured with the module's donated-state mode (`DenseGCM(..., donate_state=True)`:
the step advances the hidden state in place instead of cloning it, same results) and with the loop +
backward captured once in a HIP graph and replayed (torch.cuda.CUDAGraph, in process); the same loop run
eagerly, with donated and with functional (reference-default) state, is reported beside it (`variants`).

  python bench.py --gpus N --steps K --warmup W        # N > 1 without WORLD_SIZE: spawns N ranks
  python bench.py --config cfg3|cfg4|cfg5 ...          # the other BASELINE configs, same JSON contract
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import json
import os
import re
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "graph-conv-memory_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix == fp32 vector peak; no TF32 on gfx950
HOPS = [1, 2, 4]

# BASELINE.json configs[1..4]; B is per GPU
CONFIGS = {
    "cfg2": dict(kind="dense", B=256, N=128, F=32, H=32, T=128, selector="temporal",
                 text="cfg2: DenseGCM + TemporalBackedge([1,2,4]), B=256/GPU, graph_size=128, obs=32, hidden=32, "
                      "2x DenseGraphConv+tanh"),
    "cfg3": dict(kind="dense", B=256, N=128, F=64, H=32, T=128, selector="euclid",
                 text="cfg3: DenseGCM + EuclideanEdge(2.0) (cross-batch mean, reference-exact), B=256, "
                      "graph_size=128, obs=64, hidden=32, clustered observations (SURVEY 8d)"),
    "cfg5": dict(kind="dense", B=256, N=128, F=32, H=32, T=64, selector="learned",
                 text="cfg5: DenseGCM + LearnedEdge(32), B=256/GPU (B=2048 over 8 ranks), graph_size=128, "
                      "obs=hidden=32, gumbel noise from the per-rank device generator"),
    # the dense-materialised regime: the reference's own speed script runs DenseGCM + DenseEdge
    # (tests/test_speed.py:21-27, edge_selectors/dense.py:11-23) - every row <= cur is live, the step kernel does
    # the real cur^2 F aggregation on the fp32 MFMA
    "dense_edge": dict(kind="dense", B=256, N=128, F=32, H=32, T=128, selector="dense",
                       text="dense_edge: DenseGCM + DenseEdge (every earlier node both ways + self edge: the adjacency "
                            "is materialised dense), B=256, graph_size=128, obs=hidden=32, 2x DenseGraphConv+tanh"),
    "cfg4": dict(kind="sparse", B=512, N=512, F=32, H=32, T=512, selector="temporal_sparse",
                 text="cfg4: SparseGCM + TemporalEdge([1]), B=512/GPU, graph_size=512, obs=hidden=32, 2x GraphConv+tanh, "
                      "one call per episode (taus=512)"),
}


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
        # HSA_ENABLE_IPC_MODE_LEGACY=0: the hosts of this pool only support dmabuf IPC; without it RCCL's
        # intra-node transport setup fails with `hipIpcGetMemHandle: invalid argument` (it is exported
        # in the image already - kept explicit for ranks started from a scrubbed environment)
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


# ------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------
def dense_gnn(F, H, device):
    import torch
    from gcm import nn as G
    return G.Sequential("x, adj, weights, B, N", [
        (G.DenseGraphConv(F, H), "x, adj -> x"), torch.nn.Tanh(),
        (G.DenseGraphConv(H, H), "x, adj -> x"), torch.nn.Tanh()]).to(device)


def build_memory(device, donate=False, selector="temporal", cfg=None):
    """(DenseGCM, gnn, selector module) of a dense config (default: cfg2)."""
    import torch
    from gcm.gcm import DenseGCM
    from gcm.edge_selectors.temporal import TemporalBackedge

    c = cfg or CONFIGS["cfg2"]
    torch.manual_seed(0)
    gnn = dense_gnn(c["F"], c["H"], device)
    if selector == "learned":
        from gcm.edge_selectors.learned import LearnedEdge
        sel = LearnedEdge(c["F"]).to(device)
    elif selector == "euclid":
        from gcm.edge_selectors.distance import EuclideanEdge
        sel = EuclideanEdge(2.0)
    elif selector == "dense":
        from gcm.edge_selectors.dense import DenseEdge
        sel = DenseEdge()
    else:
        sel = TemporalBackedge(HOPS)
    mem = DenseGCM(gnn, edge_selectors=sel, graph_size=c["N"], donate_state=donate)
    return mem, gnn, sel


def make_obs(c, rank, device):
    """Synthetic observations resident in HBM; like the reference's speed test and the CPU
    baseline, obs carries no gradient."""
    import torch
    gen = torch.Generator().manual_seed(1000 + rank)
    T, B, F = c["T"], c["B"], c["F"]
    if c["selector"] == "euclid":      # 8 cluster centres shared across the batch, k_t = t mod 8
        centres = 4.0 * torch.randn(8, F, generator=torch.Generator().manual_seed(7))
        obs = centres[torch.arange(T) % 8][:, None, :] + 0.05 * torch.randn(T, B, F, generator=gen)
    else:
        obs = torch.rand(T, B, F, generator=gen)
    return obs.to(device)


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


def dist_backend():
    import torch.distributed as dist
    return dist.get_backend() if dist.is_initialized() else None


def capture(fn, zero):
    """`fn` as one HIP graph (captured in this process; the usual side-stream warm-up first)."""
    import torch
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            zero()
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    zero()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def event_time(fn, iters, warm=3):
    """ms per call of `fn` (launches on torch's current stream), HIP events around `iters` calls."""
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def short_kernel_name(name):
    """'void ns::k<32, 32>(float const*, ...)' -> 'ns::k<32, 32>' (argument list and return type dropped)"""
    name = name.strip()
    if name.endswith(")"):
        depth = 0
        for i in range(len(name) - 1, -1, -1):
            depth += name[i] == ")"
            depth -= name[i] == "("
            if depth == 0:
                name = name[:i]
                break
    return name[5:] if name.startswith("void ") else name


def external_profiler_attached():
    """rocprofv3 preloads its tool library into the process; kineto's own roctracer session then reports garbage
    durations (both trace the same dispatches) - the in-process profile is skipped and the line says so."""
    return ("rocprofiler-sdk-tool" in os.environ.get("LD_PRELOAD", "")
            or bool(os.environ.get("ROCP_TOOL_LIBRARIES")) or bool(os.environ.get("ROCPROFILER_LIBRARY_CTOR")))


def profile_kernels(fn, reps=3):
    """Per-kernel durations of `reps` calls of `fn`, measured LIVE in this process by torch.profiler's device
    activity tracing (kineto over roctracer: each kernel's begin / end timestamps on the stream it was launched on -
    what `rocprofv3 --kernel-trace --stats` reports; the committed profiles/ summaries of the same command must
    agree).  Works for kernels launched eagerly and for kernel nodes of a replayed HIP graph alike.  Independent of
    the wall-clock `value`.  -> {short kernel name: {"launches_per_call", "avg_us", "us_per_call"}}"""
    import torch
    from torch.profiler import ProfilerActivity, profile
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
    out = {}
    for e in prof.key_averages():
        if e.device_time_total <= 0 or e.count <= 0:
            continue
        k = short_kernel_name(e.key)
        d = out.setdefault(k, {"launches_per_call": 0.0, "us_per_call": 0.0})
        d["launches_per_call"] += e.count / reps
        d["us_per_call"] += e.device_time_total / reps
    for d in out.values():
        d["avg_us"] = d["us_per_call"] / max(d["launches_per_call"], 1e-9)
    return out


def kernel_table(prof, top=10):
    """the `top` kernels of a profile_kernels() result by time, with their share of the GPU time of a call"""
    total = sum(d["us_per_call"] for d in prof.values()) or 1.0
    rows = sorted(prof.items(), key=lambda kv: -kv[1]["us_per_call"])[:top]
    return [{"kernel": k, "launches_per_step": round(d["launches_per_call"], 2), "avg_us": round(d["avg_us"], 3),
             "us_per_step": round(d["us_per_call"], 2), "share_of_gpu_time": round(d["us_per_call"] / total, 4)}
            for k, d in rows], total


def find_kernel(prof, *prefixes):
    """the entry whose short name contains one of the given pieces (the longest-running one when several do)"""
    best = None
    for k, d in prof.items():
        if any(pf in k for pf in prefixes) and (best is None or d["us_per_call"] > best[1]["us_per_call"]):
            best = (k, d)
    return best


def forward_loop_cadence(mem, obs, iters=30):
    """Wall time per step-kernel launch: the T-step forward loop alone (grad mode: the records are written, as in the
    timed region) captured once as a HIP graph, HIP events on the launch stream around `iters` replays, / T.  An
    EXCLUSIVE time per launch (gaps between launches included): T x this <= ms_per_step by construction, unlike the
    begin-to-end durations a kernel tracer reports for consecutive nodes of a replayed graph, which overlap."""
    T = obs.shape[0]

    def fwd_only():
        hidden = None
        for t in range(T):
            _, hidden = mem(obs[t], hidden)

    g = capture(fwd_only, lambda: None)
    ms = event_time(g.replay, iters, warm=5) / T
    del g
    return ms


def committed_csv_avg(cfg_name, kernel_short):
    """The newest committed `rocprofv3 --kernel-trace --stats` summary of this bench command
    (profiles/rNN_bench_<cfg>_kernel_stats.csv) -> (file, calls, average ns) of the kernel whose name holds
    `kernel_short`; None when there is none.  A LOOKUP of a committed file, labelled as such in the line."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_%s_kernel_stats.csv" % cfg_name)))
    for path in reversed(files):
        try:
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    if kernel_short in short_kernel_name(row["Name"]):
                        return os.path.relpath(path, ROOT), int(row["Calls"]), float(row["AverageNs"])
        except Exception:
            continue
    return None


def launch_floor(grid, block, nodes=128):
    """What a chain of dependent launches costs on this box when the kernels do nothing (libgcm_hip_debug.so):
    us per node of a replayed HIP graph of `nodes` empty kernels of the step kernel's launch shape, the begin -> end
    duration of one empty dispatch and the begin-to-begin cadence of back-to-back empty launches outside a graph."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gcm_debuglib
    g_us, d_us, c_us = gcm_debuglib.launch_floor(grid=grid, block=block, nodes=nodes)
    return {"graph_of_%d_empty_kernels_us_per_node" % nodes: round(g_us, 3),
            "empty_dispatch_begin_to_end_us": round(d_us, 3), "empty_back_to_back_cadence_us": round(c_us, 3),
            "grid": grid, "block": block,
            "note": "cadence of a replayed HIP graph of empty kernels (same grid x block as the step kernel), one "
                    "after the other on one stream like the captured per-step loop: the launch floor the step "
                    "kernel's duration is to be read against; begin -> end of an empty kernel bracketed by "
                    "dispatch-recorded events (hipExtLaunchKernelGGL) beside it"}


# ------------------------------------------------------------------------------------------------
# in-situ duration of the dominant kernels
# ------------------------------------------------------------------------------------------------
def time_step_kernel(mem, obs, c, reps=10):
    """k_step_rows (one launch per forward step) IN SITU, two ways:
    (events) the T launches of a rollout through the C ABI on the evolving donated state, enqueued back
    to back from C, a HIP event pair recorded by the dispatch itself around every launch
    (hipExtLaunchKernelGGL start/stop events = the begin/end timestamps rocprofv3 --kernel-trace
    reports); mean over reps x T launches.  Sensitive to the host: when the box's CPUs are busy the
    launches are not back to back and the clocks sag (8.3 us against 5.7 us across two leases in round 2);
    (graph) the T-step forward loop alone captured in a HIP graph (training mode: records written),
    replay time / T - the cadence inside the graph `value` times, launch gaps included, so an UPPER bound
    on the kernel's own duration, independent of the host.
    Also the time-parallel backward kernel on the records of one rollout.
    -> (events ms, graph ms, bptt ms, launches timed by events)"""
    import torch
    from gcm import _hip

    lib = _hip.lib()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gcm_debuglib          # libgcm_hip_debug.so: the same kernels with dispatch-recorded events around them
    dbg = gcm_debuglib.lib()
    dev = obs.device
    T, B, N, F, H = obs.shape[0], c["B"], c["N"], c["F"], c["H"]
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

    saved_all = [torch.empty(lay[0], device=dev) for _ in range(T)]
    step_ev = None
    if not cfg.has_distance:     # (the C helper drives the plain step; a distance selector needs its workspace)
        evs = [(new_event(), new_event()) for _ in range(T)]
        spans = []
        ev_a = (ctypes.c_void_p * T)(*[a for a, _ in evs])
        ev_b = (ctypes.c_void_p * T)(*[b for _, b in evs])
        sv_p = (ctypes.c_void_p * T)(*[t_.data_ptr() for t_ in saved_all])
        obs_c = obs.contiguous()
        for _ in range(reps + 1):
            nodes, adj, _, count = mem.get_initial_hidden_state(obs[0])
            rc = dbg.gcm_debug_time_rows_rollout(p(obs_c), p(nodes), p(adj), p(count), cfg.arr_ptr, cfg.n_desc,
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
        step_ev = sum(sum(s) for s in spans[1:]) / (reps * T)

    # the cached step (csrc/rows_cached.hip: what a rollout from empty graphs runs on a donated state), same way
    step_ev_cached = None
    descs_ok = lib.gcm_dense_rows_cached_supported(cfg.arr_ptr, cfg.n_desc, cfg.has_bias, N, F, H, H)
    if descs_ok and T <= N and getattr(mem, "rows_cached_steps", False) and mem.donate_state:
        layc = (ctypes.c_size_t * 5)()
        lib.gcm_dense_rows_cached_layout(B, N, F, H, H, ctypes.addressof(layc))
        saved_c = [torch.empty(layc[0], device=dev) for _ in range(T)]
        sv_c = (ctypes.c_void_p * T)(*[t_.data_ptr() for t_ in saved_c])
        image = torch.empty(_hip.lib().gcm_dense_rows_cached_weight_image_floats(), device=dev)
        assert dbg.gcm_dense_rows_cached_weight_image(p(params), p(image), F, H, H, st) == 0
        evs = [(new_event(), new_event()) for _ in range(T)]
        ev_a = (ctypes.c_void_p * T)(*[a for a, _ in evs])
        ev_b = (ctypes.c_void_p * T)(*[b for _, b in evs])
        obs_c = obs.contiguous()
        spans = []
        for _ in range(reps + 1):
            nodes, adj, _, count = mem.get_initial_hidden_state(obs[0])
            cH, cA, cX = (torch.zeros(B, N, d, device=dev) for d in (H, F, F))
            rc = dbg.gcm_debug_time_cached_rollout(p(obs_c), p(nodes), p(adj), p(count), cfg.arr_ptr, cfg.n_desc,
                                                   p(params), p(image), cfg.has_bias, cfg.acts[0], cfg.acts[1], p(cH),
                                                   p(cA), p(cX), sv_c, p(flags), ev_a, ev_b, T, B, N, F, H, H, st)
            assert rc == 0, rc
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
        step_ev_cached = sum(sum(s_) for s_ in spans[1:]) / (reps * T)

    # the forward loop alone as a HIP graph (grad mode: the records are written, as in the timed region)
    def fwd_only():
        hidden = None
        for t in range(T):
            _, hidden = mem(obs[t], hidden)

    g = capture(fwd_only, lambda: None)
    step_graph = event_time(g.replay, 30, warm=5) / T
    del g

    # the backward kernel over the T records
    if cfg.has_distance:         # (records of a real rollout: the live-row counts matter)
        hidden = None
        recs = []
        keep = mem.rows_cached_steps
        mem.rows_cached_steps = False     # (live-row records: what gcm_dense_rows_bptt below reads; cached steps
        try:                              #  leave another layout, read through the chain's caches)
            for t in range(T):
                mx, hidden = mem(obs[t], hidden)
                recs.append(mx)
        finally:
            mem.rows_cached_steps = keep
        saved_ptrs = [m_.data_ptr() for m_ in recs]
    else:
        saved_ptrs = [s.data_ptr() for s in saved_all]
    g_mx = torch.full((T, B, H), 1.0 / (T * B * H), device=dev)
    arr_s = (ctypes.c_void_p * T)(*saved_ptrs)
    arr_g = (ctypes.c_void_p * T)(*[g_mx[t].data_ptr() for t in range(T)])
    ws_bytes = lib.gcm_dense_rows_bptt_workspace_bytes(T, B, F, H, H)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    gp = torch.empty(cfg.P, device=dev)

    def bptt():
        rc = lib.gcm_dense_rows_bptt(arr_s, arr_g, T, H, 1, p(params), cfg.has_bias, cfg.acts[0], cfg.acts[1], None,
                                     p(gp), p(ws), ws_bytes, B, N, F, H, H, st)
        assert rc == 0

    bptt_ms = event_time(bptt, reps, warm=1)
    return step_ev, step_graph, bptt_ms, reps * T, step_ev_cached


def euclid_launcher(c):
    """-> a closure launching k_euclid_mfma alone (gcm_edge_distance_pre through the C ABI) on FULL graphs (every row a
    candidate: 2*B*B*N*F flops per launch)"""
    import torch
    from gcm import _hip
    lib, p, st = _hip.lib(), _hip.ptr, _hip.stream()
    B, N, F = c["B"], c["N"], c["F"]
    dev = torch.device("cuda", torch.cuda.current_device())
    gen = torch.Generator().manual_seed(3)
    nodes = torch.randn(B, N, F, generator=gen).to(dev)
    count = torch.full((B,), N - 1, dtype=torch.long, device=dev)
    obs = torch.randn(B, F, generator=gen).to(dev)
    ws_bytes = lib.gcm_edge_distance_workspace_bytes(_hip.DIST_EUCLID_CROSSBATCH, B, N, F)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=dev)
    row = torch.empty(B, N, device=dev)

    def launch():
        rc = lib.gcm_edge_distance_pre(p(nodes), p(count), p(obs), p(row), _hip.DIST_EUCLID_CROSSBATCH, 2.0, None,
                                       0, 0, 0, 0, p(ws), ws_bytes, B, N, F, st)
        assert rc == 0
    launch.keep = (nodes, count, obs, ws, row)
    return launch


def time_euclid_kernel(c, iters=50):
    """`iters` back-to-back launches of the kernel alone on full graphs under the in-process kernel profiler
    -> profile_kernels() result."""
    launch = euclid_launcher(c)

    def burst():
        for _ in range(iters):
            launch()

    for _ in range(5):
        launch()
    return profile_kernels(burst, reps=1)


def time_csr_kernels(c, iters=50):
    """The CSR GraphConv forward kernel alone at the one-shot shape (B chains of N nodes: E = B*(N-1)
    edges, sum N = B*N rows), through the C ABI, HIP events around back-to-back launches."""
    import torch
    from gcm import _hip
    lib, p, st = _hip.lib(), _hip.ptr, _hip.stream()
    B, N, F, H = c["B"], c["N"], c["F"], c["H"]
    dev = torch.device("cuda", torch.cuda.current_device())
    M = B * N
    x = torch.rand(M, F, device=dev)
    deg = torch.ones(M, dtype=torch.long, device=dev)
    deg[::N] = 0
    row_ptr = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), deg.cumsum(0)])
    col = (torch.arange(M, device=dev) - 1)[deg.bool()].contiguous()
    w_rel, w_root = torch.randn(H, F, device=dev) * .1, torch.randn(H, F, device=dev) * .1
    b = torch.randn(H, device=dev)
    out, agg = torch.empty(M, H, device=dev), torch.empty(M, F, device=dev)

    def fwd():
        rc = lib.gcm_csr_graphconv_fwd(p(x), p(row_ptr), p(col), None, None, p(w_rel), p(b), p(w_root), p(out),
                                       p(agg), M, F, H, 1, st)
        assert rc == 0

    return event_time(fwd, iters, warm=5), int(col.numel()), M


# ------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (op-for-op eager-PyTorch restatement of the reference, kind "port")
# ------------------------------------------------------------------------------------------------
def cpu_baseline(c, budget_s=20.0):
    """The oracle timed on this box's host cores on a BOUNDED sample of the same workload: the same
    B/N/F/H/selector, a rollout of T_s <= T steps fwd+bwd (T_s sized so the sample stays within
    ~budget_s).  Thread policy: torch's own default is one thread per host CPU, which on a
    256-CPU host is several times SLOWER than 16-32 threads for these op sizes; the CPU gets its
    best configuration of {8, 16, 32, 64, os.cpu_count()} and the choice is recorded."""
    import torch
    from oracle import dense as od

    B, N, F, H, T = c["B"], c["N"], c["F"], c["H"], c["T"]
    torch.manual_seed(0)
    ncpu = os.cpu_count() or 1
    if c["kind"] == "sparse":
        from oracle import sparse as osp
        gnn = osp.canonical_gnn(F, H)
        x = torch.rand(B, N, F)
        taus = torch.full((B,), N, dtype=torch.long)

        def run(graphs):
            t0 = time.perf_counter()
            out, _ = osp.sparse_step(x[:graphs], taus[:graphs], None, gnn, graph_size=N,
                                     edge_selectors=osp.TemporalEdge([1]))
            out.mean().backward()
            gnn.zero_grad(set_to_none=True)
            return time.perf_counter() - t0

        tried = {}
        for th in sorted({t for t in (8, 16, 32, 64, ncpu) if t <= ncpu}):
            torch.set_num_threads(th)
            run(16)
            tried[th] = run(64)
        best = min(tried, key=tried.get)
        torch.set_num_threads(best)
        graphs = int(max(16, min(B, 64 * budget_s / (2.0 * tried[best]))))
        dt = run(graphs)
        return {"value": graphs * N / dt, "unit": "belief-states/s", "cores": best, "kind": "port",
                "seconds": dt, "host_cpus": ncpu,
                "threads_tried_s_per_64_graphs": {str(k): round(v, 3) for k, v in tried.items()},
                "sample": f"one SparseGCM call fwd+bwd on {graphs} of the {B} graphs (N={N} nodes each, F={F}, "
                          f"H={H}, TemporalEdge([1]), taus={N}), oracle/sparse.py on {best} torch threads "
                          f"(best of {sorted(tried)})"}
    gnn = od.canonical_gnn(F, H)
    if c["selector"] == "euclid":
        sel = od.EuclideanEdge(2.0)
        centres = 4.0 * torch.randn(8, F)
        obs = centres[torch.arange(T) % 8][:, None, :] + 0.05 * torch.randn(T, B, F)
        extra = []
    elif c["selector"] == "learned":
        net = od.build_edge_network(F)
        sel = od.LearnedEdge(net, num_edge_samples=5)
        obs = torch.rand(T, B, F)
        extra = [net]
    elif c["selector"] == "dense":
        sel = od.DenseEdge()
        obs = torch.rand(T, B, F)
        extra = []
    else:
        sel = od.TemporalBackedge(HOPS)
        obs = torch.rand(T, B, F)
        extra = []

    def run(steps):
        t0 = time.perf_counter()
        out, _ = od.dense_rollout(obs[:steps], None, gnn, graph_size=N, edge_selectors=sel)
        out.mean().backward()
        for m in [gnn] + extra:
            m.zero_grad(set_to_none=True)
        return time.perf_counter() - t0

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
            "sample": f"1 rollout fwd+bwd, same workload (B={B}, N={N}, F={F}, H={H}, selector={c['selector']}) "
                      f"truncated to T={T_s} steps, oracle/dense.py on {best} torch threads "
                      f"(best of {sorted(tried)})"}


# ------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 1000 (cfg2), 200 (cfg3/cfg5), 50 (cfg4)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--T", type=int, default=None, help="rollout length (cfg2: 128 fills the graph)")
    ap.add_argument("--repeats", type=int, default=5, help="extra timed blocks of K steps for value_spread")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="time the eager loop instead of the graph replay")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the variants (cfg4: the stepwise leg) - for profiles of the headline leg alone")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)

    import torch
    import torch.distributed as dist
    from gcm import parallel

    c = dict(CONFIGS[args.config])
    if args.T:
        c["T"] = args.T
    if args.steps is None:
        args.steps = {"cfg2": 1000, "cfg3": 200, "cfg5": 200, "cfg4": 50, "dense_edge": 100}[args.config]
    rank, local_rank, world = parallel.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    device = torch.device("cuda", local_rank)
    B, N, F, H, T = c["B"], c["N"], c["F"], c["H"], c["T"]
    weight = 1.0 / world

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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

    line = {"metric": "belief-states/sec (BxT) DenseGCM fwd+bwd, graph_size=128 F=32", "unit": "belief-states/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath))

    if c["kind"] == "sparse":
        bench_sparse(args, c, line, rank, world, device, timed, weight, traffic)
    else:
        bench_dense(args, c, line, rank, world, device, timed, weight, traffic)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def spread_of(times, states_per_block):
    vals = sorted(states_per_block / t for t in times)
    return {"min": vals[0], "median": statistics.median(vals), "max": vals[-1], "blocks": len(vals),
            "note": "belief-states/s of each timed block of `steps` bench steps (the first block is `value`)"}


def bench_dense(args, c, line, rank, world, device, timed, weight, traffic):
    import torch
    from gcm import parallel
    B, N, F, H, T = c["B"], c["N"], c["F"], c["H"], c["T"]
    name = args.config
    if c["selector"] == "euclid" and world > 1:
        raise SystemExit("cfg3: EuclideanEdge averages over the whole batch (distance.py:48-49); a sharded run "
                         "needs EuclideanEdge(shard_group=...) - run it on one GPU")
    can_donate = True
    mem, gnn, sel = build_memory(device, donate=can_donate, selector=c["selector"], cfg=c)
    mods = [gnn] + ([sel] if c["selector"] == "learned" else [])
    bucket = parallel.GradBucket(*mods)
    if c["selector"] == "learned":       # per-rank gumbel draws (seeds 0 + rank)
        torch.cuda.manual_seed(rank)
    obs = make_obs(c, rank, device)

    def zero():
        for m in mods:
            m.zero_grad(set_to_none=True)

    # N > 1: the gradients alias the flat bucket BEFORE the capture (GradBucket.attach: the backward accumulates into it
    # in place - no gather / copy-back launch per step), the bucket's zero fill is a node of the graph, and so is the
    # all-reduce when the backend can be captured (RCCL); else the collective follows each replay eagerly.
    comm = {"captured": False, "launches_outside_graph": 0, "bucket_floats": bucket.numel}
    graph = None
    if not args.no_graph and world > 1:
        bucket.attach()

        def body(with_comm):
            bucket.zero()
            rollout(mem, obs)
            if with_comm:
                bucket.all_reduce_mean(weight)
        if os.environ.get("GCM_BENCH_CAPTURE_COMM", "1") == "1" and dist_backend() == "nccl":
            try:
                graph = capture(lambda: body(True), bucket.zero)
                comm["captured"] = True
            except Exception as e:      # (reported; the eager collective is the fallback)
                comm["capture_error"] = "%s: %s" % (type(e).__name__, str(e)[:160])
                torch.cuda.synchronize()
                graph = None
            # every rank takes the same path: if the capture failed anywhere, nobody replays a captured collective
            import torch.distributed as dist
            ok = torch.tensor([1 if comm["captured"] else 0], device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and comm["captured"]:
                comm["captured"] = False
                comm["capture_error"] = "another rank failed to capture the collective"
                graph = None
        if graph is None:
            graph = capture(lambda: body(False), bucket.zero)
    elif not args.no_graph:
        graph = capture(lambda: rollout(mem, obs), zero)
    # what rank 0 profiles ALONE further down must hold no collective (the other ranks have left by then: a replayed
    # all-reduce would wait for them forever): with the collective captured, a twin of the graph without it
    graph_prof = graph
    if graph is not None and comm["captured"]:
        graph_prof = capture(lambda: body(False), bucket.zero)

    def step():
        if graph is not None:
            graph.replay()
            if world > 1 and not comm["captured"]:
                bucket.all_reduce_mean(weight)
        else:
            rollout(mem, obs, bucket, weight)
            zero()

    dt = timed(step, args.steps, args.warmup)
    blocks = [dt] + [timed(step, args.steps, 0) for _ in range(max(0, args.repeats))]
    flags = mem._flag_word(device)
    bits = int(flags.item())
    assert not (bits & 6), f"kernels flagged {bits}"
    if world > 1:
        # the collective alone (the flat bucket, ~30 KB: latency-bound), events around 20 calls on the launch stream
        comm["comm_us_per_step"] = round(event_time(lambda: bucket.all_reduce_mean(weight), 20, warm=3) * 1e3, 2)
        # (bucket.launches is reset by every all_reduce_mean call: what it holds is the LAST call's launches besides
        #  the collective - a per-step figure, measured on the call just timed)
        comm["launches_outside_graph"] = 0 if (graph is not None and comm["captured"]) else 1 + bucket.launches
        comm["gradients_aliased"] = bool(bucket.aliased())
        comm["note"] = ("per bench step besides the replayed graph: the all-reduce of the flat gradient bucket (and what "
                        "GradBucket.all_reduce_mean launches around it: nothing once the gradients alias the bucket and "
                        "the weight folds into ReduceOp.AVG); captured = the collective is a node of the HIP graph")
        line["comm"] = comm
    if external_profiler_attached():
        # under rocprofv3 the run is the timed region alone: its kernel-trace statistics are then those of the
        # headline's kernels (profiles/<tag>_bench_<cfg>_kernel_stats.csv), to be read against the unprofiled line
        if rank == 0:
            line.update({"value": world * B * T * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                         "config": {"workload": c["text"] + ", T=%d" % T},
                         "profiled_by": "an external profiler (rocprofv3): the timed region only; kernel durations are "
                                        "in its own kernel-trace statistics, not measured in process"})
            print(json.dumps(line))
        return

    # ---- the same work along the other paths (reported beside `value`) ---------------------------
    side = max(3, min(20, args.steps // 10))
    variants = {}

    def eager(m, mods_, bk):
        def f():
            rollout(m, obs, bk, weight)
            for q in mods_:
                q.zero_grad(set_to_none=True)
        return f

    def variant(donate, reps=3):
        m, g_, s_ = build_memory(device, donate=donate, selector=c["selector"], cfg=c)
        mods_ = [g_] + ([s_] if c["selector"] == "learned" else [])
        bk = parallel.GradBucket(*mods_)
        f = eager(m, mods_, bk)
        ts = [timed(f, side, 2 if i == 0 else 0) for i in range(reps)]
        m.check_flags()
        return m, g_, bk, [world * B * T * side / t for t in ts]

    if can_donate:
        mem_e, gnn_e, bucket_e, v = variant(True)
        variants["eager_donated"] = statistics.median(v)
        variants["eager_donated_min_max"] = [min(v), max(v)]
    mem_f, gnn_f, bucket_f, v = variant(False)
    variants["eager_functional"] = statistics.median(v)
    variants["eager_functional_min_max"] = [min(v), max(v)]
    # the additive time-batched entry DenseGCM.rollout on the FUNCTIONAL module: temporal / dense
    # selectors as one C call, LearnedEdge from empty graphs as the three-launch time-parallel forward, everything else
    # as the per-step loop on a state the call owns
    mods_f = [gnn_f] + ([s_ for s_ in [getattr(mem_f, "edge_selectors", None)] if c["selector"] == "learned"])

    def roll():
        rollout_api(mem_f, obs, bucket_f, weight)
        for q in mods_f:
            q.zero_grad(set_to_none=True)
    variants["rollout_api"] = world * B * T * side / timed(roll, side, 2)
    rollout_kernels = None
    if rank == 0 and world == 1 and c["selector"] == "euclid":   # (world == 1: `roll` reduces its gradients)
        # the time-parallel entry's own kernels (csrc/euclid_tp.hip): every step's decisions as one causal contraction
        pr = profile_kernels(roll, reps=2)
        kt = find_kernel(pr, "k_euclid_tp<")
        if kt is not None:
            pairs = sum(min(t, N - 1) for t in range(T))
            slot_rows = sum(32 * ((min(t, N) + 31) // 32) for t in range(1, T))
            alg, exe = 2.0 * B * B * F * pairs, 2.0 * B * B * (F + 2) * slot_rows
            sec = kt[1]["avg_us"] * 1e-6
            rollout_kernels = {
                "kernel": kt[0], "avg_launch_ms": kt[1]["avg_us"] * 1e-3, "launches_per_call": kt[1]["launches_per_call"],
                "flops_algorithmic": alg, "TFLOP/s": alg / sec / 1e12,
                "frac_of_fp32_mfma_peak": alg / sec / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "flops_executed": exe, "frac_executed": exe / sec / 1e12 / PEAK_F32_MFMA_TFLOPS,
                "table": kernel_table(pr, top=6)[0],
                "note": "DenseGCM.rollout(obs[T,B,F]) with EuclideanEdge: k_euclid_tp = the decisions of all T steps (2 B^2 F "
                        "flops per candidate (step, node) pair: algorithmic; executed: whole 32-slot blocks and the norm "
                        "k step), then the GNN of all steps in one launch per graph, the backward over the T records"}
    if c["selector"] == "temporal":

        def fwd_only():
            with torch.no_grad():
                hidden = None
                for t in range(T):
                    _, hidden = mem_e(obs[t], hidden)
        variants["forward_only_eager_donated"] = world * B * T * side / timed(fwd_only, side, 1)
        mem_e.check_flags()
        # observations that need a gradient (an encoder in front of the memory): the live-row kernels, dL/dx of
        # the whole chain in one time-parallel launch by the chain's node; eager and graph-replayed, functional
        # and donated state
        def rollout_xs(m, xs):      # per-step leaves (an encoder's outputs), not slices of one [T,B,F] leaf
            hidden, outs = None, []
            for x in xs:
                mx, hidden = m(x, hidden)
                outs.append(mx)
            torch.stack(outs).mean().backward()

        xs = [obs[t].clone().requires_grad_(True) for t in range(T)]

        def with_obs_grad():
            rollout_xs(mem_f, xs)
            gnn_f.zero_grad(set_to_none=True)
            for x in xs:
                x.grad = None
        variants["eager_functional_obs_grad"] = world * B * T * side / timed(with_obs_grad, side, 2)
        for state, don in (("functional", False), ("donated", True)):
            try:
                # (a fresh module and leaves: an AccumulateGrad node made on another stream breaks the capture)
                mem_g, gnn_g, _ = build_memory(device, donate=don, selector=c["selector"], cfg=c)
                xs2 = [obs[t].clone().requires_grad_(True) for t in range(T)]

                def zero_g():
                    gnn_g.zero_grad(set_to_none=True)
                    for x in xs2:
                        x.grad = None
                gg = capture(lambda: rollout_xs(mem_g, xs2), zero_g)
                variants[f"graph_{state}_obs_grad"] = world * B * T * side / timed(gg.replay, side, 2)
                del gg
            except Exception as e:      # (reported, not fatal: the headline does not depend on it)
                variants[f"graph_{state}_obs_grad"] = 0.0
                print("obs-grad graph capture failed:", type(e).__name__, str(e)[:200], file=sys.stderr)
                torch.cuda.synchronize()
        mem_f.check_flags()

    # ---- the layered path (a user GNN the fused step does not cover: three DenseGraphConv layers; pooled=True):
    # the general DenseGraphConv kernels, one launch per layer and direction -------------------------
    layered = None
    if c["selector"] == "temporal" and not args.headline_only and rank == 0:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import prof_layered
            layered = {}
            for kind in ("three_layer", "pooled"):
                lp, lT, _ = prof_layered.run(kind, T=16, reps=2, device=device)
                fw, bw = find_kernel(lp, "k_layer_fwd<", "k_graphconv_fwd<"), find_kernel(lp, "k_layer_bwd<", "k_graphconv_bwd")
                tot = sum(d_["us_per_call"] for d_ in lp.values())
                # per layer, training mode: adj once, x / agg / out once each
                lbytes = B * 4 * (N * N + N * F + N * F + N * H)
                ent = {"gpu_us_per_step": round(tot / lT, 2), "belief_states_per_s_gpu_time": B / (tot / lT * 1e-6)}
                if fw:
                    ent["layer_fwd"] = {"kernel": fw[0], "avg_us": round(fw[1]["avg_us"], 3), "bytes_per_launch": lbytes,
                                        "GB/s": lbytes / (fw[1]["avg_us"] * 1e-6) / 1e9,
                                        "frac_of_hbm_peak": lbytes / (fw[1]["avg_us"] * 1e-6) / 1e9 / PEAK_HBM_GBS}
                if bw:
                    ent["layer_bwd"] = {"kernel": bw[0], "avg_us": round(bw[1]["avg_us"], 3)}
                layered[kind] = ent
        except Exception as e:      # (reported, not fatal)
            layered = {"error": "%s: %s" % (type(e).__name__, str(e)[:160])}

    # ---- the steady-state regime ("also T=256"): after t >= graph_size every step drops every graph's
    # oldest node (gcm.py:263-271, 323-355) - the normal regime of a long RL rollout --------------------------------
    if T <= N and not args.headline_only:
        from gcm.gcm import DenseGCM
        DenseGCM.did_warn = True     # (the reference's one-time overflow notice is a print: stdout stays ONE JSON line)
        T2 = 2 * N
        obs2 = make_obs(dict(c, T=T2), rank, device)
        mem_2, gnn_2, sel_2 = build_memory(device, donate=True, selector=c["selector"], cfg=c)
        mods_2 = [gnn_2] + ([sel_2] if c["selector"] == "learned" else [])

        def zero_2():
            for q in mods_2:
                q.zero_grad(set_to_none=True)
        g2 = capture(lambda: rollout(mem_2, obs2), zero_2)
        t2_s = timed(g2.replay, side, 2) / side
        variants["T%d_graph_donated" % T2] = world * B * T2 / t2_s
        # No in-process kernel trace of this graph: kineto's trace teardown (stop_trace) crashes the process now and then
        # on the T = 2N graphs - a segmentation fault in 5 of 25 runs of cfg5, heap corruption three times in a row with
        # dense_edge, never seen with cfg2 - and a crash here would cost the whole line.  The steady-state step is
        # priced from the two graph times; tools/prof_t256.py prints the per-kernel table of the same graph (its own
        # process: what dies there is a tool).
        variants["T%d_steady_state_step_us" % T2] = round((t2_s - dt / args.steps) / (T2 - T) * 1e6, 3)
        variants["T%d_steady_state_step_note" % T2] = (
            "(replay time of the T=%d graph - replay time of the T=%d graph) / %d: the mean step past the first %d - %s"
            "forward + its share of the backward; per-kernel durations: tools/prof_t256.py under rocprofv3"
            % (T2, T, T2 - T, T, "" if T >= N else "%d of them still below graph_size, %d in the steady state - " % (N - T, T2 - N)))
        del g2

        def eager2(m, gn, bk):
            sels = [m.edge_selectors] if c["selector"] == "learned" else []

            def f():
                rollout(m, obs2, bk, weight)
                gn.zero_grad(set_to_none=True)
                for q in sels:
                    q.zero_grad(set_to_none=True)
            return f
        variants["T%d_eager_donated" % T2] = world * B * T2 * side / timed(eager2(mem_e, gnn_e, bucket_e), side, 1)
        variants["T%d_eager_functional" % T2] = world * B * T2 * side / timed(eager2(mem_f, gnn_f, bucket_f), side, 1)
        if c["selector"] not in ("temporal", "dense"):
            def roll2():
                rollout_api(mem_f, obs2, bucket_f, weight)
                for q in mods_f:
                    q.zero_grad(set_to_none=True)
            variants["T%d_rollout_api" % T2] = world * B * T2 * side / timed(roll2, side, 1)
        mem_e.check_flags()
        mem_f.check_flags()

    if rank != 0:
        return
    states = world * B * T * args.steps
    ms_per_step = dt / args.steps * 1e3
    fwd_full = 2 * N * N * (F + H) + 4 * N * (F * H + H * H)       # SURVEY 8(d), per belief state
    # ---- every kernel of the timed region, measured live (the replayed graph itself; the eager rollout with
    # --no-graph): the independent source of every kernel duration below ------------------------------------------
    prof_reps = 5
    if graph is not None:
        prof = profile_kernels(graph_prof.replay, reps=prof_reps)
    else:
        def one():
            rollout(mem, obs)
            zero()
        prof = profile_kernels(one, reps=prof_reps)
    table, gpu_us = kernel_table(prof)
    kernel_ms = {}
    src = ("torch.profiler device activity (kineto over roctracer: kernel begin / end timestamps, as rocprofv3 "
           "--kernel-trace reports them) over %d %s of the timed region, in this process"
           % (prof_reps, "replays of the captured HIP graph" if graph is not None else "eager rollouts"))

    def hbm_view(nbytes, sec):
        return None if not nbytes else {"bytes_per_launch": nbytes, "GB/s": nbytes / sec / 1e9,
                                        "frac_of_hbm_peak": nbytes / sec / 1e9 / PEAK_HBM_GBS}

    if c["selector"] == "temporal":
        # Dominant kernel = the step kernel: one launch per forward step, ~85 % of the GPU time of the metric
        # (the backward of the whole rollout is two launches of k_bptt_rows).
        # `achieved` = SURVEY 8(d)'s compulsory bytes per belief state (adj once, x once, obs in, belief out,
        # adj-row write-back: 4N^2+4NF+4F+4H+4N = 82.7 KB at cfg2) x B over the mean launch duration - an
        # EFFECTIVE rate: the kernel exploits "only row n_b is kept" (gcm.py:314) and, in a chain from empty
        # graphs, caches layer 1; `executed` is what it really moves and computes.
        k = find_kernel(prof, "k_step_rows_cached_img4", "k_step_rows_cached_img<", "k_step_rows_cached<", "k_step_rows<")
        kb = find_kernel(prof, "k_bptt_cached_graph", "k_bptt_rows<")
        step_kernel, kd = k
        sec = kd["avg_us"] * 1e-6
        alg_bytes = B * (4 * N * N + 4 * N * F + 4 * F + 4 * H + 4 * N)
        cached = "cached" in step_kernel
        tkey = (("k_step_rows_cached_img4b" if "img4b" in step_kernel else "k_step_rows_cached_img4")
                if "img4" in step_kernel else "k_step_rows_cached_img") if cached else "k_step_rows"
        moved = traffic.get(tkey, traffic.get("k_step_rows_cached_img4", traffic.get("k_step_rows_cached_img")) if cached else None)
        # executed flops per launch of the cached step: |S| row adds for both aggregates + four H x F matrix-vector
        # products per graph; the general kernel: layer 1 on its live rows (|S| + 1) over their non-zero chunks
        n_sel = len(HOPS)
        exec_flops = B * (n_sel * (F + H) + 2 * (2 * F * H) + 2 * (2 * H * H)) if cached else \
            B * ((n_sel + 1) * (n_sel * F + 2 * (2 * F * H)) + n_sel * H + 2 * (2 * H * H))
        cross = {}
        try:        # cross-checks (NOT the source of avg_launch_ms): dispatch-recorded events, graph cadence, value
            step_ev, step_graph, bptt_ev, n_ev, step_ev_cached = time_step_kernel(mem_e, obs, c)
            cross = {"events_back_to_back_from_C_ms": step_ev_cached if cached else step_ev,
                     "events_k_step_rows_general_ms": step_ev,
                     "captured_forward_loop_over_T_ms": step_graph,
                     "from_value_ms": (ms_per_step - (kb[1]["us_per_call"] * 1e-3 if kb else 0.0)) / T
                     if graph is not None else None,
                     "note": "events: the T launches of a rollout enqueued back to back from C (libgcm_hip_debug.so), "
                             "each bracketed by HIP events the dispatch itself records - an empty kernel reads "
                             "`launch_floor_us.empty_dispatch_begin_to_end_us` there; graph: replay time of the "
                             "captured forward loop / T (launch gaps included); from_value: (ms_per_step - "
                             "k_bptt_rows) / T, what the timed region itself leaves per step"}
        except Exception as e:      # (the cross-checks need libgcm_hip_debug.so; the line does not)
            cross = {"error": "%s: %s" % (type(e).__name__, str(e)[:160])}
        try:
            floor = launch_floor(B, 64 if cached else 256, nodes=T)
        except Exception as e:
            floor = {"error": "%s: %s" % (type(e).__name__, str(e)[:160])}
        ex = hbm_view(moved, sec) or {}
        ex.update({"flops_per_launch": exec_flops, "TFLOP/s": exec_flops / sec / 1e12,
                   "frac_of_fp32_mfma_peak": exec_flops / sec / 1e12 / PEAK_F32_MFMA_TFLOPS,
                   "note": "what the kernel really does per launch: bytes = PMC (2*FETCH_SIZE + WRITE_SIZE, "
                           "profiles/traffic.json), flops = its own formulation (row cur over the chain's caches); "
                           "at neither limit - a launch-to-retire latency chain, see launch_floor_us"})
        # The duration `frac` is computed from: the wall time the timed region spends per launch of the step kernel =
        # replay time of the captured forward loop / T, HIP events on the launch stream (the
        # begin-to-end durations of consecutive graph nodes overlap - their sum exceeded ms_per_step - so they stay a
        # side field, with the committed rocprofv3 CSV's figure of the same command next to them).
        try:
            cad_ms = forward_loop_cadence(mem_e, obs)
            cad_src = ("HIP events (torch.cuda.Event on the launch stream) around 30 replays of the T-step forward loop "
                       "captured as a HIP graph, / T: wall time per launch of the step kernel inside the loop the metric "
                       "times (launch gaps included; T x avg_launch_ms <= ms_per_step)")
        except Exception as e:      # (capture unavailable: fall back to what the timed region leaves per step)
            cad_ms = (ms_per_step - (kb[1]["us_per_call"] * 1e-3 if kb else 0.0)) / T
            cad_src = "(ms_per_step - k_bptt_rows) / T [forward-loop capture failed: %s]" % type(e).__name__
        csv_row = committed_csv_avg(name, step_kernel.split("::")[-1])
        b2e = {"avg_launch_ms": kd["avg_us"] * 1e-3, "frac": alg_bytes / sec / 1e9 / PEAK_HBM_GBS, "source": src,
               "sum_over_T_launches_ms": kd["avg_us"] * 1e-3 * T,
               "note": "begin -> end timestamps of each kernel node (what a kernel tracer reports); consecutive nodes "
                       "of a replayed graph overlap (node t + 1's begin stamp is taken while node t drains), so these "
                       "durations add up to more than the loop's wall time - NOT the duration `frac` uses"}
        if csv_row is not None:
            csv_frac = alg_bytes / (csv_row[2] * 1e-9) / 1e9 / PEAK_HBM_GBS
            b2e["rocprofv3_csv"] = {"file": csv_row[0], "calls": csv_row[1], "avg_ns": csv_row[2], "frac": csv_frac,
                                    "differs_from_in_process_pct": round(100.0 * (csv_row[2] * 1e-6 / (kd["avg_us"] * 1e-3) - 1.0), 2),
                                    "source": "committed file (rocprofv3 --kernel-trace --stats of `bench.py --config %s`), "
                                              "not measured in this run; frac = bytes_per_launch / avg_ns / 8 TB/s" % name}
        cad_sec = cad_ms * 1e-3
        ex_c = dict(ex)
        if moved:
            ex_c.update({"GB/s": moved / cad_sec / 1e9, "frac_of_hbm_peak": moved / cad_sec / 1e9 / PEAK_HBM_GBS})
        ex_c.update({"TFLOP/s": exec_flops / cad_sec / 1e12,
                     "frac_of_fp32_mfma_peak": exec_flops / cad_sec / 1e12 / PEAK_F32_MFMA_TFLOPS})
        line["roofline"] = {
            "bound": "hbm", "kernel": step_kernel, "achieved": alg_bytes / cad_sec / 1e9, "peak": PEAK_HBM_GBS,
            "unit": "GB/s", "frac": alg_bytes / cad_sec / 1e9 / PEAK_HBM_GBS, "traffic": moved,
            "traffic_source": "committed file profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over "
                              "tools/pmc_run.py, summarised by tools/pmc_summarise.py); not measured in this run",
            "bytes_per_launch": alg_bytes, "avg_launch_ms": cad_ms, "avg_launch_ms_source": cad_src,
            "launches_timed": 30 * T,
            "consistency": {"launches_per_step": T, "launches_x_avg_launch_ms": cad_ms * T, "ms_per_step": ms_per_step,
                            "holds": bool(cad_ms * T <= ms_per_step * 1.02) if graph is not None else None,
                            "whole_step_GB/s_upper_check": alg_bytes * T / (ms_per_step * 1e-3) / 1e9},
            "begin_to_end": b2e,
            "executed": ex_c, "launch_floor_us": floor, "cross_checks": cross,
            "note": "EFFECTIVE rate: bytes_per_launch = SURVEY 8(d)'s full-dense compulsory bytes (what the "
                    "reference's formulation must move per step) over the wall time per launch; the kernel "
                    "itself moves `traffic` bytes (`executed`): only the rows that reach the kept belief row are "
                    "evaluated, layer 1 of older rows comes from the chain's caches, the state is advanced in place"}
        sec = cad_sec      # (the MFMA view below is priced on the same duration)
        line["roofline_mfma_view"] = {
            "kernel": step_kernel, "flops_per_launch_full_dense": B * fwd_full,
            "achieved_full_dense_TFLOPs": B * fwd_full / sec / 1e12,
            "frac_of_fp32_mfma_peak": B * fwd_full / sec / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "note": "SURVEY 8(d) full-dense FLOPs (2 layers x all N rows) over the same duration: EFFECTIVE (> 1 means "
                    "the dense work is not performed, by design); executed flops: roofline.executed"}
        kernel_ms = {step_kernel: round(kd["avg_us"] * 1e-3, 6)}
        if kb:
            kernel_ms[kb[0]] = round(kb[1]["avg_us"] * 1e-3, 6)
    elif c["selector"] == "euclid":
        # Dominant kernel = k_euclid_mfma2 (the cross-batch distance contraction [N x F].[F x B] per graph on the
        # fp32 MFMA, in a donated chain from empty graphs with the cached step as its tail): 2*B*B*N*F flops per
        # launch on FULL graphs.  The roofline line is the kernel alone on full graphs; the
        # in-situ mean of the timed rollout (graphs fill up: blocks of 32 rows >= cur are skipped) beside it.
        eu_prof = time_euclid_kernel(c)
        ke = find_kernel(eu_prof, "k_euclid_mfma")
        eu_ms = ke[1]["avg_us"] * 1e-3
        flops = 2.0 * B * B * N * F
        ki = find_kernel(prof, "k_euclid_mfma")
        kb = find_kernel(prof, "k_bptt_cached_graph", "k_bptt_rows<")
        in_situ = None
        if ki is not None:
            rows_live = sum(min(N, 32 * (t // 32 + 1)) for t in range(T)) / T      # rows < cur, in 32-row blocks
            f_in = 2.0 * B * B * rows_live * F
            sec_i = ki[1]["avg_us"] * 1e-6
            in_situ = {"kernel": ki[0], "avg_launch_ms": ki[1]["avg_us"] * 1e-3,
                       "launches_timed": int(round(ki[1]["launches_per_call"] * prof_reps)),
                       "flops_per_launch_mean": f_in, "TFLOP/s": f_in / sec_i / 1e12,
                       "frac_of_fp32_mfma_peak": f_in / sec_i / 1e12 / PEAK_F32_MFMA_TFLOPS,
                       "traffic": traffic.get("k_euclid_mfma2_tail1", traffic.get("k_euclid_mfma2")),
                       "note": "the timed rollout from empty graphs: mean over its T launches (the cached step runs as "
                               "the tail of the same launch); flops = the live 32-row blocks' share of 2*B*B*N*F"}
            kernel_ms[ki[0] + " (in situ)"] = round(ki[1]["avg_us"] * 1e-3, 6)
        kernel_ms[ke[0] + " (full graphs, alone)"] = round(eu_ms, 6)
        if kb:
            kernel_ms[kb[0]] = round(kb[1]["avg_us"] * 1e-3, 6)
        line["roofline"] = {
            "bound": "mfma", "kernel": ke[0], "achieved": flops / (eu_ms * 1e-3) / 1e12,
            "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": flops / (eu_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, "traffic": traffic.get("k_euclid_mfma2"),
            "flops_per_launch": flops, "avg_launch_ms": eu_ms,
            "avg_launch_ms_source": "torch.profiler device activity over 50 back-to-back launches of the kernel alone "
                                    "(gcm_edge_distance_pre through the C ABI) on graphs that hold N-1 nodes",
            "in_situ": in_situ,
            "note": "fp32 MFMA (no TF32 on gfx950); executed flops = algorithmic flops here (every stored row is a "
                    "candidate)"}
    elif c["selector"] == "dense":
        # The dense-materialised regime (the reference's own speed script: tests/test_speed.py:21-27):
        # DenseEdge makes every row <= cur live and every live row's adjacency row full, so the step really does
        # the cur^2 F aggregation.  Priced on the fp32 MFMA peak, twice: EXECUTED flops of the kernel's own formulation
        # (per graph-step with L = cur + 1 live rows: 2 L^2 F aggregation + 4 L F H layer-1 linears + 2 L H + 4 H^2 for
        # row cur of layer 2), and SURVEY 8(d)'s full-dense flops (2 layers x all N rows) as the EFFECTIVE figure.
        k = find_kernel(prof, "k_step_colcache", "k_step_rows<")
        kb = find_kernel(prof, "k_bptt_dense<", "k_bptt_cached_graph", "k_bptt_rows<")
        step_kernel, kd = k
        sec = kd["avg_us"] * 1e-6
        Ls = [min(t, N - 1) + 1 for t in range(T)]
        recompute = "k_step_rows<" in step_kernel
        if recompute:
            per_graph = [2.0 * L * L * F + 4.0 * L * F * H + 2.0 * L * H + 4.0 * H * H for L in Ls]
        else:   # the cached dense step: rank-1 update of the layer-1 pre-activations (L H adds), tanh on L rows, row cur
            per_graph = [2.0 * L * H + 2.0 * L * H + 2.0 * (2 * F * H) + 2.0 * (2 * F * H) + 4.0 * H * H for L in Ls]
        exec_flops = B * sum(per_graph) / T
        exec_full = B * (2.0 * N * N * F + 4.0 * N * F * H + 2.0 * N * H + 4.0 * H * H)
        line["roofline"] = {
            "bound": "mfma", "kernel": step_kernel, "achieved": B * fwd_full / sec / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": B * fwd_full / sec / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "traffic": traffic.get("k_step_colcache" if not recompute else "k_step_rows_dense"),
            "flops_per_launch": B * fwd_full, "avg_launch_ms": kd["avg_us"] * 1e-3, "avg_launch_ms_source": src,
            "launches_timed": int(round(kd["launches_per_call"] * prof_reps)),
            "executed": {"flops_per_launch_mean": exec_flops, "TFLOP/s": exec_flops / sec / 1e12,
                         "frac_of_fp32_mfma_peak": exec_flops / sec / 1e12 / PEAK_F32_MFMA_TFLOPS,
                         "flops_per_launch_full_graphs": exec_full,
                         "note": "what the kernel's own formulation computes, mean over the T launches of a rollout from "
                                 "empty graphs (L = cur + 1 live rows at step cur)"},
            "note": "EFFECTIVE: SURVEY 8(d)'s full-dense forward flops per belief state (2 N^2 (F+H) + 4 N (FH + H^2): "
                    "both DenseGraphConv layers on all N rows, what the reference computes every step) x B over the "
                    "step kernel's measured mean duration; `executed` = the flops of the kernel's own formulation"}
        kernel_ms = {step_kernel: round(kd["avg_us"] * 1e-3, 6)}
        if kb:
            Lsum = sum(Ls)
            kernel_ms[kb[0]] = round(kb[1]["avg_us"] * 1e-3, 6)
            line["roofline"]["backward"] = {"kernel": kb[0], "launches_per_step": round(kb[1]["launches_per_call"], 2),
                                            "avg_launch_ms": kb[1]["avg_us"] * 1e-3, "us_per_step": round(kb[1]["us_per_call"], 1),
                                            "live_rows_per_rollout": B * Lsum}
    else:
        # cfg5: three kernels carry the step - the cached forward step (selection: the edge network on N candidate
        # rows + gumbel-softmax + the GNN on row cur), pass B of the backward (edge network recomputed and
        # differentiated per graph-step, persistent) and pass A (k_bptt_rows<..,2>: GNN parameter gradient).  Each
        # priced on the flops it EXECUTES (its own formulation) and on its PMC bytes; the dominant one is `roofline`.
        Fe = F
        # (the exact-shape cached step takes the first product from the chain's U cache: ONE N x F x F product)
        exact_u = N == 128 and F == 32 and H == 32
        sel_exec = B * ((1 if exact_u else 2) * (2 * N * Fe * Fe) + 2 * 2 * Fe * Fe + 2 * N * Fe + 2 * (2 * F * H) + 2 * (2 * H * H))
        sel_ref = B * (2 * N * (3 * Fe * Fe + Fe))           # learned.py:38-51 on N candidate pairs, forward
        # pass B2 works per 32-row block with a candidate row: step t of a rollout from empty graphs has t candidates
        live_blocks = sum((min(t, N - 1) + 31) // 32 for t in range(T))
        n_prod = 4 if exact_u else 5   # P1, dW1, gH0, dW0b (+ P0 without the U cache); c0 / dW0a are vector work
        bpb_exec = B * live_blocks * (n_prod * (2 * 32 * Fe * Fe))
        b2_note = ("pass B2: the edge network recomputed and differentiated per 32-row block that holds a candidate row "
                   "(one block per wave; 32 x F x F products on the fp32 MFMA: P1, dW1, gH0, dW0b - and P0 where the "
                   "chain keeps no U cache; c0 and dW0a are vector work since round 5 - LayerNorm passes not "
                   "counted); one persistent launch per chain")
        kk_b2 = find_kernel(prof, "k_learned_bptt_mlp")
        if kk_b2 and "mlp16" in kk_b2[0]:
            # the register-resident form (round 6): per 16-ROW tile with a candidate row, v_mfma_f32_16x16x4_f32 - P1, gH0,
            # dW1, dW0b and c0 (W0a x x_cur in every column: what it executes) as 16 x F x F products, dW0a as four
            # instructions on the tile's column sums
            live_tiles = sum((min(t, N - 1) + 15) // 16 for t in range(T))
            # (c0 comes from pass B1 where that ran per graph - k_learned_bptt_sel_graph - and is then no product here)
            n16 = 4 if find_kernel(prof, "k_learned_bptt_sel_graph") else 5
            bpb_exec = B * live_tiles * (n16 * (2 * 16 * Fe * Fe) + 4 * (2 * 16 * 16 * 4))
            b2_note = ("pass B2 in registers (round 6): the edge network recomputed and differentiated per 16-row TILE that "
                       "holds a candidate row, one tile per wave in the forward's lane layout, twelve waves per CU; flops: "
                       "the 16 x F x F products it executes on v_mfma_f32_16x16x4_f32 (P1, gH0, dW1, dW0b; c0 too where pass B1 did not "
                       "leave it) + dW0a's "
                       "four instructions - LayerNorm passes and adjoints (VALU) not counted; one persistent launch per chain")
        kinds = [("k_learned_select", ("k_learned_select<", "k_learned_select8"), sel_exec, sel_ref,
                  "selection + GNN tail (cached step); flops: the N x F x F products it executes (one with the U "
                  "cache of the exact shapes, else two), the W0b / W0a x[cur] and F -> 1 "
                  "layers, four matrix-vector products of the GNN tail; LayerNorm / softmax VALU work not counted; "
                  "flops_reference_formulation = 2 N (3F^2 + F) per graph (the reference's 2F-wide first layer on "
                  "every candidate pair)"),
                 ("k_learned_bptt_mlp", ("k_learned_bptt_mlp",), bpb_exec, None, b2_note),
                 ("k_learned_bptt_sel", ("k_learned_bptt_sel",), None, None,
                  "pass B1: per graph-step the gradient collected over the later steps that hold the node, selection "
                  "and softmax adjoint -> g_logit [T,B,N]"),
                 ("k_bptt_rows_learned", ("k_bptt_learned_graph", "k_bptt_rows<"), None, None,
                  "pass A: GNN parameter gradient over the live rows of every graph-step (k_bptt_learned_graph: per graph, "
                  "two live rows per fp32 MFMA, where every step of the backward is a donated cached step)")]
        rows = []
        for tkey, prefixes, fl_exec, fl_ref, note in kinds:
            kk = find_kernel(prof, *prefixes)
            if kk is None:
                continue
            sec = kk[1]["avg_us"] * 1e-6
            ent = {"kernel": kk[0], "launches_per_step": round(kk[1]["launches_per_call"], 2),
                   "avg_launch_ms": kk[1]["avg_us"] * 1e-3,
                   "share_of_gpu_time": round(kk[1]["us_per_call"] / gpu_us, 4), "note": note}
            if fl_exec:
                ent.update({"flops_per_launch": fl_exec, "TFLOP/s": fl_exec / sec / 1e12,
                            "frac_of_fp32_mfma_peak": fl_exec / sec / 1e12 / PEAK_F32_MFMA_TFLOPS})
            if fl_ref:
                ent["flops_reference_formulation"] = fl_ref
            short = re.search(r"\b(k_[A-Za-z0-9_]+)", kk[0])      # (tools/pmc_summarise.py keys by the kernel's own name)
            hv = hbm_view(traffic.get(short.group(1) if short else tkey, traffic.get(tkey)), sec)
            if hv:
                ent["hbm"] = hv
            rows.append(ent)
            kernel_ms[kk[0]] = round(kk[1]["avg_us"] * 1e-3, 6)
        dom = max((r for r in rows if "flops_per_launch" in r), key=lambda r: r["share_of_gpu_time"])
        per_state = 3 * fwd_full + 2 * N * N * (F + H) + 3 * N * 2 * (3 * F * F + F)
        step_s = dt / args.steps / T
        line["roofline"] = {
            "bound": "mfma", "kernel": dom["kernel"], "achieved": dom["TFLOP/s"], "peak": PEAK_F32_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": dom["frac_of_fp32_mfma_peak"],
            "traffic": dom.get("hbm", {}).get("bytes_per_launch"), "flops_per_launch": dom["flops_per_launch"],
            "avg_launch_ms": dom["avg_launch_ms"], "avg_launch_ms_source": src, "kernels": rows,
            "effective_whole_step": {"flops_per_step": B * per_state, "avg_step_ms": step_s * 1e3,
                                     "TFLOP/s": B * per_state / step_s / 1e12,
                                     "frac_of_fp32_mfma_peak": B * per_state / step_s / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                     "note": "SURVEY 8(d)'s full-dense flops of one fwd+bwd memory step (GNN 3x forward, "
                                             "dAdj, the edge network on N pairs fwd+bwd) over the wall time per step: "
                                             "EFFECTIVE, not executed - kept for continuity with round 3"},
            "note": "EXECUTED flops of the dominant kernel over its measured mean duration; every kernel of the step "
                    "in `kernels` (flops executed, PMC bytes where collected: profiles/traffic.json)"}
    line.update({
        "value": states / dt, "ms_per_step": ms_per_step,
        "value_spread": spread_of(blocks, world * B * T * args.steps),
        "config": {"workload": c["text"] + ", T=%d; one bench step = one rollout through the per-step drop-in call "
                                           "surface `for t: mx, m = gcm(obs[t], m)` + backward" % T,
                   "B_per_gpu": B, "graph_size": N, "obs": F, "hidden": H, "T": T,
                   "state": "donated (DenseGCM(donate_state=True): hidden state advanced in place)" if can_donate
                            else "functional",
                   "launch": "eager" if graph is None else "HIP graph of the loop + backward, captured once "
                                                           "in process, replayed per bench step",
                   "parallelism": f"dp{world} (batch-sharded, 1 flat-bucket all-reduce per backward)"},
        "variants": {k: (round(v, 1) if isinstance(v, float) else ([round(x, 1) for x in v] if isinstance(v, list) else v))
                     for k, v in variants.items()},
        "variants_note": "belief-states/s of the same workload: the eager per-step loop with donated / functional "
                         "(reference-default: what an unchanged caller of the reference gets) state, median [min, max] "
                         "of 3 blocks; the additive DenseGCM.rollout entry; the forward loop alone under no_grad; "
                         "T<2N>_*: rollouts of 2 x graph_size steps - the second half in the steady state, every step "
                         "dropping every graph's oldest node (gcm.py:323-355)",
        "kernel_ms": kernel_ms,
        "kernel_table": {"rows": table, "gpu_us_per_step": round(gpu_us, 2), "source": src},
    })
    # MFMA-pipe utilisation, LDS conflicts and LDS issue stalls from the committed SQ counter passes
    # (tools/collect_profiles.sh sq -> profiles/mfma_util.json): per kernel of this line that has an entry there
    upath = os.path.join(ROOT, "profiles", "mfma_util.json")
    if os.path.exists(upath):
        util = json.load(open(upath))

        def counters_of(kname):
            # (the longest key that names the kernel: "k_learned_bptt_mlp" is also a prefix of "k_learned_bptt_mlp16<true>")
            for key, ent in sorted(util.items(), key=lambda kv: -len(kv[0])):
                if key in kname and "mfma_busy_frac" in ent:
                    return dict({k: ent[k] for k in ("mfma_busy_frac", "lds_conflict_frac", "wait_inst_lds_frac",
                                                     "wait_inst_any_frac", "wait_any_frac", "avg_us_under_pmc") if k in ent},
                                source="committed file profiles/mfma_util.json (not measured in this run)")
            return None
        rf = line.get("roofline", {})
        cn = counters_of(rf.get("kernel", ""))
        if cn:
            rf["sq_counters"] = dict(cn, source="committed file profiles/mfma_util.json, not measured in this run: rocprofv3 --pmc SQ_* passes over tools/pmc_mfma_run.py, summarised by "
                                                "tools/pmc_sq_summarise.py: mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / "
                                                "(4 x SQ_BUSY_CU_CYCLES)")
        if isinstance(rf.get("in_situ"), dict):
            cn = counters_of(rf["in_situ"].get("kernel", ""))
            if cn:
                rf["in_situ"]["sq_counters"] = cn
        for ent in rf.get("kernels", []) or []:
            cn = counters_of(ent.get("kernel", ""))
            if cn:
                ent["sq_counters"] = cn
        if rollout_kernels is not None:
            cn = counters_of(rollout_kernels.get("kernel", ""))
            if cn:
                rollout_kernels["sq_counters"] = cn
    if rollout_kernels is not None:
        line["rollout_api_kernels"] = rollout_kernels
    if layered is not None:
        line["layered_path"] = dict(layered, note="the same shapes through GNNs the fused step does not cover (three "
                                                  "DenseGraphConv layers; the canonical two with pooled=True): one "
                                                  "DenseGraphConv kernel per layer and direction (csrc/fused_layer.hip), "
                                                  "per-step loop fwd + bwd, T = 16; layer_fwd priced on adj + x + agg + out "
                                                  "once each per layer (unfused AI = 16 FLOP/B: HBM-bound)")
    if name != "cfg2":
        line["metric"] = "belief-states/sec (BxT) DenseGCM fwd+bwd, graph_size=128 (%s)" % name
    info = {"world_size_seen": world, "device": torch.cuda.get_device_name(device), "torch": torch.__version__}
    try:
        info["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        info["rccl_version"] = None
    line["runtime"] = info
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(c)
    print(json.dumps(line))


def bench_sparse(args, c, line, rank, world, device, timed, weight, traffic):
    """cfg4: SparseGCM + TemporalEdge([1]).  One bench step = one call over whole episodes
    (x [B, N, F], taus = N: B*N belief states) + backward."""
    import torch
    from gcm import nn as G
    from gcm import parallel
    from gcm.sparse_gcm import SparseGCM
    from gcm.sparse_edge_selectors.temporal import TemporalEdge
    B, N, F, H = c["B"], c["N"], c["F"], c["H"]
    torch.manual_seed(0)
    g = G.Sequential("x, edges, weights", [(G.GraphConv(F, H), "x, edges, weights -> x"), torch.nn.Tanh(),
                                           (G.GraphConv(H, H), "x, edges, weights -> x"), torch.nn.Tanh()]).to(device)
    mem = SparseGCM(g, edge_selectors=TemporalEdge([1]), graph_size=N)
    bucket = parallel.GradBucket(g)
    gen = torch.Generator().manual_seed(1000 + rank)
    x = torch.rand(B, N, F, generator=gen).to(device)
    taus = torch.full((B,), N, dtype=torch.long, device=device)

    def oneshot():
        out, _ = mem(x, taus, None)
        out.mean().backward()
        bucket.all_reduce_mean(weight)
        g.zero_grad(set_to_none=True)

    dt = timed(oneshot, args.steps, args.warmup)
    blocks = [dt] + [timed(oneshot, args.steps, 0) for _ in range(max(0, args.repeats))]
    if external_profiler_attached():      # (see bench_dense)
        if rank == 0:
            line.update({"metric": "belief-states/sec (BxT) SparseGCM fwd+bwd, graph_size=512 F=32 (cfg4)",
                         "value": world * B * N * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                         "config": {"workload": c["text"]},
                         "profiled_by": "an external profiler (rocprofv3): the timed region only; kernel durations are "
                                        "in its own kernel-trace statistics, not measured in process"})
            print(json.dumps(line))
        return
    one = torch.ones(B, dtype=torch.long, device=device)
    n_sw = N      # stepwise taus=1 x 512

    xs = [x[:, t:t + 1].contiguous() for t in range(n_sw)]   # a stepwise caller's observations: one [B, 1, F] per call

    def stepwise():
        hid, outs = None, []
        for t in range(n_sw):
            o, hid = mem(xs[t], one, hid)
            outs.append(o)
        torch.cat(outs, 1).mean().backward()
        bucket.all_reduce_mean(weight)
        g.zero_grad(set_to_none=True)

    variants = {}
    if not args.headline_only:
        # (two warm-up rollouts: the caching allocator sees the 512 growing sizes of a rollout once before the clock starts)
        variants["stepwise_taus1_x%d" % n_sw] = world * B * n_sw * 3 / timed(stepwise, 3, 2)
        mem.stepwise_cache = False     # (A/B: every call runs both GraphConv layers over every stored node)
        variants["stepwise_taus1_x%d_general_path" % n_sw] = world * B * n_sw * 2 / timed(stepwise, 2, 1)
        mem.stepwise_cache = True
    if rank != 0:
        return
    fwd_ms_alone, E, M = time_csr_kernels(c)
    alg = E * (F * 4 + 16) + 2 * M * F * 4          # per layer, one-shot
    # every kernel of one call (forward + backward), measured live in this process
    prof_reps = 5

    def one_call():
        out, _ = mem(x, taus, None)
        out.mean().backward()
        g.zero_grad(set_to_none=True)

    prof = profile_kernels(one_call, reps=prof_reps)
    table, gpu_us = kernel_table(prof, top=16)
    n_launch = sum(d["launches_per_call"] for d in prof.values())
    src = ("torch.profiler device activity (kineto over roctracer: kernel begin / end timestamps, as rocprofv3 "
           "--kernel-trace reports them) over %d eager calls (forward + backward), in this process" % prof_reps)
    kf = find_kernel(prof, "k_csr_fwd3<", "k_csr_fwd2<", "k_csr_graphconv_fwd")
    fwd_ms = kf[1]["avg_us"] * 1e-3
    states = world * B * N * args.steps
    ms_per_step = dt / args.steps * 1e3
    # the whole call against the same roofline: 2 GraphConv layers x (forward + backward), each direction at least the
    # layer's forward bytes (x / agg / out once each, the edge list once) - what an ideal fused implementation moves
    whole_alg = 4 * alg
    pmc_sum = 0.0
    pmc_missing = []
    for k, d in prof.items():
        keys = [t for t in traffic if t in k]
        if keys:
            pmc_sum += traffic[max(keys, key=len)] * d["launches_per_call"]
        elif d["us_per_call"] > 0.02 * gpu_us:
            pmc_missing.append(k)
    line.update({
        "metric": "belief-states/sec (BxT) SparseGCM fwd+bwd, graph_size=512 F=32 (cfg4)",
        "value": states / dt, "ms_per_step": ms_per_step,
        "value_spread": spread_of(blocks, world * B * N * args.steps),
        "config": {"workload": c["text"] + "; one bench step = one SparseGCM call on [B, 512, F] + backward, eager",
                   "B_per_gpu": B, "graph_size": N, "obs": F, "hidden": H, "T": N,
                   "parallelism": f"dp{world} (batch-sharded, 1 flat-bucket all-reduce per backward)"},
        "variants": {k: round(v, 1) for k, v in variants.items()},
        "variants_note": "stepwise: the same episodes one node per call (taus = 1, %d calls from hidden = None) + one "
                         "backward, default settings (finite_check = 'sync'): every call on the chain's caches "
                         "(gcm_sparse_step_cached: the new node's rows alone; one time-parallel backward launch per "
                         "chain); _general_path: the same with stepwise_cache = False" % n_sw,
        "roofline": {"bound": "hbm", "kernel": kf[0], "achieved": alg / (fwd_ms * 1e-3) / 1e9,
                     "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": alg / (fwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                     "traffic": traffic.get("k_csr_fwd3"), "bytes_per_launch": alg, "avg_launch_ms": fwd_ms,
                     "avg_launch_ms_source": src,
                     "launches_timed": int(round(kf[1]["launches_per_call"] * prof_reps)),
                     "avg_launch_ms_alone_back_to_back": fwd_ms_alone,
                     "executed": {"bytes_per_launch": traffic.get("k_csr_fwd3"),
                                  "GB/s": (traffic.get("k_csr_fwd3") or 0) / (fwd_ms * 1e-3) / 1e9,
                                  "frac_of_hbm_peak": (traffic.get("k_csr_fwd3") or 0) / (fwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
                     "whole_call": {"algorithmic_bytes": whole_alg, "wall_ms": ms_per_step,
                                    "GB/s_over_wall": whole_alg / (ms_per_step * 1e-3) / 1e9,
                                    "frac_over_wall": whole_alg / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                    "gpu_ms": gpu_us * 1e-3,
                                    "GB/s_over_gpu_time": whole_alg / (gpu_us * 1e-6) / 1e9,
                                    "frac_over_gpu_time": whole_alg / (gpu_us * 1e-6) / 1e9 / PEAK_HBM_GBS,
                                    "kernel_launches_per_call": round(n_launch, 1),
                                    "pmc_bytes_per_call": pmc_sum or None,
                                    "pmc_over_algorithmic": (pmc_sum / whole_alg) if pmc_sum else None,
                                    "kernels_without_pmc_above_2pct": pmc_missing,
                                    "note": "the WHOLE call (forward + backward of both GraphConv layers, packing, loss) "
                                            "against 4 x the per-layer algorithmic bytes; pmc_bytes_per_call = sum over "
                                            "the call's kernels of their PMC bytes per launch (profiles/traffic.json)"},
                     "note": "SURVEY 8(d): E*(4F+16) + 2*sumN*4F bytes per GraphConv layer (E = %d edges, sumN = %d "
                             "rows); the forward kernel of one layer IN SITU (mean over the call's two layers)" % (E, M)},
        "kernel_ms": {kf[0]: round(fwd_ms, 5)},
        "kernel_table": {"rows": table, "gpu_us_per_step": round(gpu_us, 2), "source": src},
    })
    info = {"world_size_seen": world, "device": torch.cuda.get_device_name(device), "torch": torch.__version__}
    try:
        info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        info["rccl_version"] = None
    line["runtime"] = info
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(c)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
