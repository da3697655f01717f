import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "graph-conv-memory_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle is eager PyTorch on small operands: torch's default of one thread per host CPU (256 on the GPU
    # boxes of this pool) makes it SLOWER by orders of magnitude than a few threads (bench.py's cpu_baseline: 18.8 s
    # against 0.03 s per four cfg2 steps at 256 / 16 threads) - and the oracle is most of the GPU suite's wall time.
    try:
        import torch
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    except Exception:       # (collection must not depend on it)
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
