"""CPU oracle for the DenseGCM / SparseGCM hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (eager PyTorch-CPU ops, op for op) of the
reference algorithm in proroklab/graph-conv-memory for the path named in
BASELINE.json `north_star`.  It exists to CHECK the HIP product path:

  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
    leg may import it;
  * the product package (`graph-conv-memory_amd/gcm`) never imports it and
    has no CPU fallback - it fails loudly when the HIP library is missing.

Pinning status
--------------
* `oracle.dense` / `oracle.sparse` (DenseGCM, SparseGCM, edge selectors) are
  pinned against golden vectors produced by importing the reference itself in
  the build container (`tests/golden/make_golden.py`, fixtures under
  `tests/golden/*.npz`) and against the known-answer bodies of the
  reference's own unit tests (tests/test_gcm.py, tests/test_sparse_gcm.py).
* `oracle.pyg` (DenseGraphConv, GraphConv, coalesce, k_hop_subgraph) restates
  the published algorithm of torch_geometric (pinned by the reference only as
  `torch_geometric>=1.7.0`, setup.cfg:24; not vendored, not installable here).
  Numerical parity versus a real PyG build is therefore **parity unpinned**;
  what pins it are the reference's own invariants that cross that boundary
  (identity-weight known answer tests/test_gcm.py:282-323, dense==sparse
  bit-equality tests/test_sparse_gcm.py:395-429).
"""
