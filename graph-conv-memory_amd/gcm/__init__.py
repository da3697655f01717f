"""gcm - MI355X-native DenseGCM / SparseGCM hot path (drop-in for the same
module paths of proroklab/graph-conv-memory: gcm.gcm.DenseGCM,
gcm.edge_selectors.*, gcm.sparse_gcm.SparseGCM, gcm.sparse_edge_selectors.*).

Host code is Python on PyTorch-ROCm; all device work on the path runs in
hand-written gfx950 kernels reached through the C ABI in include/gcm_hip.h.
"""
__version__ = "0.1.0"
