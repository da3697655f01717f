"""The kernels through the ctypes binding ALONE: GCM_NO_TORCH_EXT=1 makes gcm/_ext.py refuse the
C++ extension, so every call below reaches libgcm_hip.so through gcm/_hip.py's prototypes and the Python
torch.autograd.Function wrappers of gcm/_ops.py - the golden vectors of the reference (g1 - g16: every dense selector,
wrap_overflow, positional encodings, pack / unpack, the sparse one-shot / stepwise / ragged / k-hop / aux runs) and the
kernel-level oracle checks of tests/test_dense_gpu.py and tests/test_sparse_gpu.py.  ONE child process (the
extension is decided at import time); the tests that pin behaviour of the C++ host path itself (the stepwise chain's
step counters, the aliases a one-call SparseGCM step returns) are left to the default run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_golden_and_kernel_suites_without_the_torch_extension():
    env = dict(os.environ, GCM_NO_TORCH_EXT="1")
    p = subprocess.run(
        [sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
         os.path.join(ROOT, "tests", "test_dense_gpu.py"), os.path.join(ROOT, "tests", "test_sparse_gpu.py"),
         "-k", "not stepwise_cached_chain and not returns_aliases_that_are_watched and not sizes_are_reused"],
        env=env, capture_output=True, timeout=1500, cwd=ROOT)
    tail = p.stdout.decode()[-1500:]
    assert p.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
    n = int(tail.strip().splitlines()[-1].split(" passed")[0].split()[-1])
    assert n >= 100, tail
