#!/usr/bin/env python3
"""Phase stamps of two rounds of k_euclid_tp (csrc/euclid_tp.hip) at cfg3's shape, T = 128: round 38 (step 20: one
live slot block) and round 238 (step 120: four) of workgroup 0, thread 0.  make -C graph-conv-memory_amd/csrc stamps10."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "graph-conv-memory_amd")]
os.environ.setdefault("GCM_HIP_LIB", os.path.join(ROOT, "graph-conv-memory_amd", "gcm", "_lib", "libgcm_hip_stamps10.so"))
import torch  # noqa: E402
import bench  # noqa: E402
from gcm import _hip  # noqa: E402

lib = _hip.lib()
dev = torch.device("cuda", 0)
c = dict(bench.CONFIGS["cfg3"])
B, N, F, T = c["B"], c["N"], c["F"], 128
obs = bench.make_obs(dict(c, T=T), 0, dev)
bits = torch.empty(T, B, 4, dtype=torch.int32, device=dev)
names = ["top: touch consumed, next chunk's loads issued, finalize, operand reload", "MFMA chain (+ pending epilogue)",
         "node copy", "store next chunk", "touch", "barrier"]
acc = [[0.0] * 6, [0.0] * 6]
R = 10
for it in range(R + 2):
    assert lib.gcm_euclid_rollout_tp_decide(obs.data_ptr(), 2.0, None, bits.data_ptr(), T, B, N, F, _hip.stream()) == 0
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 32)()
    lib.gcm_debug_read_stamps_tp(out, 32)
    if it >= 2:
        for r in range(2):
            for i in range(6):
                acc[r][i] += (out[8 * r + i + 1] - out[8 * r + i]) / R
for r, lab in enumerate(("round 38 (t = 20, 1 live block)", "round 238 (t = 120, 4 live blocks)")):
    print(lab)
    for i in range(6):
        print("  %d -> %d  %-78s %8.1f" % (i, i + 1, names[i], acc[r][i]))
    print("  total %8.1f cycles" % sum(acc[r]))
