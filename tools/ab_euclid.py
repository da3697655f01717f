#!/usr/bin/env python3
"""Same-box A/B of builds of the EuclideanEdge step kernel (csrc/distance.hip: k_euclid_mfma2 through
gcm_edge_distance_pre) at cfg3's shape: every library given on the command line is loaded in a child process of its own
and times the kernel alone with HIP events (100 back-to-back launches) at several fill levels.
  python3 tools/ab_euclid.py lib_a.so lib_b.so ...      (libraries built from distance.hip + state.hip suffice)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(path):
    import torch
    lib = ctypes.CDLL(path)
    P, I, F_, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    lib.gcm_edge_distance_pre.argtypes = [P, P, P, P, I, F_, P, I, I, I, I, P, Z, I, I, I, P]
    lib.gcm_edge_distance_pre.restype = I
    B, N, F = 256, 128, 64
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(3)
    nodes = torch.randn(B, N, F, generator=gen).to(dev)
    obs = torch.randn(B, F, generator=gen).to(dev)
    row = torch.zeros(B, N, device=dev)
    st = torch._C._cuda_getCurrentRawStream(0)
    out = []
    for fill in (127, 96, 64, 32):
        count = torch.full((B,), fill, dtype=torch.long, device=dev)

        def launch():
            rc = lib.gcm_edge_distance_pre(nodes.data_ptr(), count.data_ptr(), obs.data_ptr(), row.data_ptr(), 0, 2.0, None,
                                           0, 0, 0, 0, None, 0, B, N, F, st)
            assert rc == 0, rc
        for _ in range(10):
            launch()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            a.record()
            for _ in range(100):
                launch()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 10.0)
        out.append("%d: %.2f us" % (fill, best))
    print("%-40s %s   checksum %.1f" % (os.path.basename(path), "  ".join(out), float(row.sum())))


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for lib in sys.argv[1:]:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", os.path.abspath(lib)], check=False)
