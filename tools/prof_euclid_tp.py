#!/usr/bin/env python3
"""k_euclid_tp alone at cfg3's shape (B = 256 graphs, N = 128, F = 64): the decisions of all T steps of a rollout,
timed with HIP events on the launch stream; and, for T <= N, the whole forward (decisions + GNN).  Also usable under
rocprofv3 --kernel-trace --stats / --pmc (profiles/r05_euclid_tp_*).
  python3 tools/prof_euclid_tp.py [T ...]      default: 128 256"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-conv-memory_amd")]
import torch  # noqa: E402
import bench  # noqa: E402
from gcm import _hip  # noqa: E402

lib = _hip.lib()
dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS["cfg3"])
B, N, F = c["B"], c["N"], c["F"]
PEAK = bench.PEAK_F32_MFMA_TFLOPS
for T in [int(a) for a in sys.argv[1:]] or [128, 256]:
    obs = bench.make_obs(dict(c, T=T), 0, dev)
    bits = torch.empty(T, B, 4, dtype=torch.int32, device=dev)

    def run():
        rc = lib.gcm_euclid_rollout_tp_decide(obs.data_ptr(), 2.0, None, bits.data_ptr(), T, B, N, F, _hip.stream())
        assert rc == 0, rc
    ms = bench.event_time(run, 10, warm=2)
    pairs = sum(min(t, N - 1) for t in range(T))                  # candidate (step, node) pairs per graph
    blocks = sum(32 * ((min(t, N) + 31) // 32) for t in range(1, T))   # ... in live 32-slot blocks
    alg, exe = 2.0 * B * B * F * pairs, 2.0 * B * B * (F + 2) * blocks
    print("k_euclid_tp T=%d: %.3f ms  algorithmic %.1f GFLOP -> %.1f TFLOP/s = %.3f of fp32 MFMA peak; executed (32-slot "
          "blocks, norm step) %.1f GFLOP -> %.3f; %.1f M belief-states/s for the selection alone"
          % (T, ms, alg / 1e9, alg / ms / 1e9, alg / ms / 1e9 / PEAK, exe / 1e9, exe / ms / 1e9 / PEAK, B * T / ms / 1e3))
torch.cuda.synchronize()
