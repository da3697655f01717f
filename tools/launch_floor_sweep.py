"""Launch floor of a replayed HIP graph of empty kernels against the launch shape (one MI355X): what a step kernel
could gain from a different grid, before any of its own work.  python tools/launch_floor_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402  (initialises the HIP runtime the debug library shares)
import gcm_debuglib  # noqa: E402

torch.zeros(1, device="cuda")
for grid, block in [(256, 128), (256, 64), (256, 256), (128, 256), (128, 128), (64, 512), (64, 256), (32, 1024),
                    (512, 64), (512, 128), (1024, 64), (8, 64), (1, 64)]:
    best = min(gcm_debuglib.launch_floor(grid, block, 128, 50) for _ in range(3))
    print(f"grid {grid:5d} x {block:4d} threads: {best[0]:.3f} us per graph node, {best[1]:.3f} us begin->end, "
          f"{best[2]:.3f} us cadence without a graph", flush=True)
