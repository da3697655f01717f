"""ctypes binding of libgcm_hip.so (C ABI: include/gcm_hip.h).

There is NO CPU fallback: importing is always allowed (so host-side logic can
be inspected anywhere) but the first kernel call raises if the library is
missing or a tensor is not on a HIP device.
"""
import ctypes
import os

import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "libgcm_hip.so")
_lib = None

# constants mirrored from include/gcm_hip.h
ACT_NONE, ACT_TANH, ACT_RELU = 0, 1, 2
DIR = {"forward": 1, "backward": 2, "both": 3}
DIST_EUCLID_CROSSBATCH, DIST_L2_PERGRAPH, DIST_COSINE_SIM = 0, 1, 2
FLAG_WRAPPED, FLAG_BAD_COUNT, FLAG_NONFINITE = 1, 2, 4
FLAG_SPARSE_OVERFLOW, FLAG_ACAUSAL, FLAG_PACK_OVERFLOW, FLAG_MERGE_ORDER, FLAG_WINDOW = 8, 16, 32, 64, 128
GNN_HAS_DEG_TERM, GNN_HAS_PE_TABLE, GNN_RECORD_DX = 4, 8, 16      # has_bias bits of the live-row step (gcm_hip.h)
STEP_TWO_LAUNCH = 32      # ... of the cached step: a distance selector and the step as two launches (A/B)
STEP_IMG_V4 = 64          # ... its weights as 16-byte loads (the image's second layout)
BPTT_MANY_ROWS = 128      # gcm_dense_rows_bptt: records with many live rows per graph (DenseEdge)
STEP_ONE_WAVE = 512       # gcm_dense_rows_step_cached: the one-wave kernel where the two-wave form exists (A/B)
STEP_FOUR_WAVES = 256     # gcm_dense_rows_step_colcache: the four-wave kernel where the eight-wave form exists (A/B)

ABI_VERSION = 6           # include/gcm_hip.h: GCM_ABI_VERSION

_P, _I, _F, _Z, _L = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t,
                     ctypes.c_int64)

# name -> (restype, argtypes).  Kept in one table so tests can check that every
# symbol declared in include/gcm_hip.h is exported and bound.
PROTOTYPES = {
    "gcm_version": (_I, []),
    "gcm_abi_version": (_I, []),
    "gcm_status_string": (ctypes.c_char_p, [_I]),
    "gcm_state_advance_fwd": (_I, [_P] * 11 + [_I, _I, _I, _P]),
    "gcm_state_advance_bwd": (_I, [_P] * 6 + [_I, _I, _I, _P]),
    "gcm_gather_rows_fwd": (_I, [_P] * 4 + [_I, _I, _I, _P]),
    "gcm_gather_rows_bwd": (_I, [_P] * 3 + [_I, _I, _I, _P]),
    "gcm_edge_temporal": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "gcm_edge_dense": (_I, [_P, _P, _I, _I, _P]),
    "gcm_rows_linear": (_I, [_P] * 4 + [ctypes.c_int64, _I, _I, _I, _I, _P, _P, _F, _P, _P]),
    "gcm_posenc_cat_finish": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "gcm_posenc_cat_bwd": (_I, [_P] * 4 + [_I] * 4 + [_P]),
    "gcm_temporal_window_fwd": (_I, [_P] * 6 + [_I] * 5 + [_P, _P]),
    "gcm_temporal_window_bwd": (_I, [_P] * 4 + [_I] * 5 + [_P]),
    "gcm_edge_distance_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "gcm_edge_distance": (_I, [_P, _P, _P, _I, _F, _P, _I, _I, _I, _I, _I, _P, _P, _Z, _I, _I, _I, _P]),
    "gcm_edge_distance_ex": (_I, [_P, _P, _P, _I, _F, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _Z, _I, _I, _I, _P]),
    "gcm_dense_graphconv_fwd": (_I, [_P] * 7 + [_I] * 5 + [_P]),
    "gcm_dense_graphconv_bwd_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "gcm_dense_graphconv_bwd": (_I, [_P] * 13 + [_Z] + [_I] * 5 + [_P]),
    "gcm_sparse_plan": (_I, [_P] * 5 + [_I, _P]),
    "gcm_sparse_insert_fwd": (_I, [_P] * 6 + [_I] * 4 + [_P]),
    "gcm_sparse_insert_bwd": (_I, [_P] * 5 + [_I] * 4 + [_P]),
    "gcm_sparse_temporal_count": (_I, [_P, _P, _P, _I, _P, _I, _P]),
    "gcm_sparse_temporal_fill": (_I, [_P, _P, _P, _I, _P, _P, _L, _I, _P]),
    "gcm_sparse_temporal_fill_vals": (_I, [_P, _P, _P, _I, _P, _P, _P, _L, _I, _P]),
    "gcm_sparse_temporal_structure": (_I, [_P, _P, _I] + [_P] * 9 + [_L, _L, _I, _P]),
    "gcm_csr_graphconv_fwd_checked_supported": (_I, [_L, _I, _I]),
    "gcm_csr_graphconv_fwd_checked": (_I, [_P] * 10 + [_L, _I, _I, _I, _P, _P]),
    "gcm_sparse_flatten_fwd": (_I, [_P] * 5 + [_I, _I, _I, _L, _P]),
    "gcm_sparse_flatten_bwd": (_I, [_P] * 5 + [_I, _I, _I, _L, _P]),
    "gcm_sparse_edges_to_csr": (_I, [_P] * 5 + [_L, _L, _I, _P]),
    "gcm_ptr_from_sorted": (_I, [_P, _P, _L, _L, _P]),
    "gcm_csc_from_csr_batched": (_I, [_P] * 7 + [_I, _L, _L, _I, _P]),
    "gcm_coo_merge_segments": (_I, [_P] * 10 + [_L, _L, _I, _P]),
    "gcm_khop_mask": (_I, [_P] * 5 + [_I, _P, _P, _L, _I, _I, _P]),
    "gcm_sparse_extract_fwd": (_I, [_P] * 6 + [_I, _I, _I, _L, _P]),
    "gcm_sparse_extract_bwd": (_I, [_P] * 5 + [_I, _I, _I, _L, _P]),
    "gcm_csr_graphconv_fwd": (_I, [_P] * 10 + [_L, _I, _I, _I, _P]),
    "gcm_csr_graphconv_bwd_workspace_bytes": (_Z, [_L, _I, _I]),
    "gcm_csr_graphconv_bwd": (_I, [_P] * 19 + [_Z, _L, _L, _I, _I, _I, _P]),
    "gcm_learned_pairs_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "gcm_learned_pairs_bwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "gcm_learned_select_fwd": (_I, [_P, _P, _P, _F, _P, _P, _I, _I, _P]),
    "gcm_learned_select_bwd": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "gcm_causal_count": (_I, [_P, _P, _I, _P, _P, _I, _P]),
    "gcm_causal_fill": (_I, [_P, _P, _I, _P, _P, _P, _P, _L, _L, _I, _P]),
    "gcm_causal_pairs_fwd": (_I, [_P, _P, _P, _L, _I, _I, _I, _P]),
    "gcm_causal_pairs_bwd": (_I, [_P, _P, _P, _I, _P, _P, _L, _I, _I, _I, _P]),
    "gcm_segment_softmax_fwd": (_I, [_P] * 5 + [_L, _L, _P]),
    "gcm_segment_softmax_bwd": (_I, [_P] * 8 + [_L, _L, _P]),
    "gcm_posenc_add": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "gcm_pack_hidden": (_I, [_P] * 6 + [_L, _I, _I, _P]),
    "gcm_dense_gnn2_row_supported": (_I, [_I, _I, _I, _I]),
    "gcm_dense_gnn2_param_count": (_Z, [_I, _I, _I]),
    "gcm_dense_gnn2_row_fwd": (_I, [_P] * 6 + [_I] + [_P] * 3 + [_I] + [_P] * 5 + [_I] * 5 + [_P]),
    "gcm_dense_gnn2_row_bwd": (_I, [_P] * 9 + [_I] + [_P] * 3 + [_I] + [_P] * 7 + [_I] * 6 + [_P]),
    "gcm_relu_layernorm_fwd": (_I, [_P] * 4 + [_L, _I, ctypes.c_float, _P]),
    "gcm_relu_layernorm_bwd_workspace_bytes": (_Z, [_L, _I]),
    "gcm_relu_layernorm_bwd": (_I, [_P] * 6 + [_Z, _L, _I, ctypes.c_float, _P]),
    "gcm_skinny_wgrad_workspace_bytes": (_Z, [_I] * 3),
    "gcm_skinny_wgrad": (_I, [_P] * 4 + [_Z] + [_I] * 3 + [_P]),
    "gcm_sum_slabs": (_I, [_P, _I, _I, _P, _P]),
    "gcm_sum_slabs_acc": (_I, [_P, _I, _I, _P, _P, _P]),
    "gcm_dense_step_fused_fwd": (_I, [_P] * 9 + [_I] + [_P] * 3 + [_I] + [_P] * 3 + [_I] + [_P] * 5
                                 + [_I] * 5 + [_P]),
    "gcm_dense_step_fwd": (_I, [_P] * 9 + [_I] + [_P] + [_I] * 3 + [_P] * 6 + [_Z] + [_I] * 5 + [_P]),
    "gcm_dense_step_bwd": (_I, [_P] * 7 + [_I] * 3 + [_P] * 8 + [_Z] + [_I] * 5 + [_P]),
    "gcm_dense_step_bwd_slabs": (_I, [_P] * 7 + [_I] * 3 + [_P] * 7 + [_I] * 6 + [_P]),
    "gcm_dense_step_bwd_acc": (_I, [_P] * 7 + [_I] * 3 + [_P] * 9 + [_Z] + [_I] * 5 + [_P]),
    "gcm_dense_rows_supported": (_I, [_I] * 4),
    "gcm_dense_rows_layout": (_I, [_I] * 5 + [_P]),
    "gcm_dense_rows_layout_dx": (_I, [_I] * 5 + [_P]),
    "gcm_dense_rows_dx_supported": (_I, [_I] * 4),
    "gcm_dense_rows_bptt_dx_step": (_I, [_P, _P, ctypes.c_long, ctypes.c_long, _P, _P, _I, _I, _I, _P, _P, _P]
                                    + [_I] * 6 + [_P]),
    "gcm_dense_rows_cached_supported": (_I, [_P, _I, _I, _I, _I, _I, _I]),
    "gcm_dense_rows_cached_supported_ws": (_I, [_P, _I, _I, _I, _I, _I, _I]),
    "gcm_dense_rows_step_cached_ws": (_I, [_P] * 5 + [_I, _P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P, _Z]
                                      + [_I] * 5 + [_P]),
    "gcm_edge_distance_step_cached_supported": (_I, [_I] * 6),
    "gcm_dense_rows_cached_roll_supported": (_I, [_P, _I, _I, _I, _I, _I, _I]),
    "gcm_dense_rows_step_cached_roll": (_I, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P]
                                        + [_I] * 5 + [_P]),
    "gcm_dense_rows_colcache_supported": (_I, [_P, _I, _I, _I, _I, _I, _I]),
    "gcm_dense_rows_step_colcache_functional": (_I, [_P] * 8 + [_I, _P, _I, _I, _I, _P, _P, _P, _I, _I, _P] + [_I] * 5 + [_P]),
    "gcm_dense_rows_step_colcache": (_I, [_P] * 5 + [_I, _P, _I, _I, _I, _P, _P, _P, _I, _I, _P] + [_I] * 5 + [_P]),
    "gcm_dense_rows_cached_launches": (_I, [_P, _I, _I, _I, _I, _I, _I, _I]),
    "gcm_edge_distance_step_cached": (_I, [_P] * 4 + [_F, _P, _P, _I, _P, _P, _I, _I] + [_P] * 5 + [_I, _I, _P, _P]
                                      + [_I] * 5 + [_P]),
    "gcm_edge_distance_step_ring_supported": (_I, [_I] * 5),
    "gcm_edge_distance_step_ring": (_I, [_P] * 4 + [_F, _P, _P, _I, _I, _P, _P, _P, _I, _P] + [_I] * 5 + [_P]),
    "gcm_dense_rows_cached_layout": (_I, [_I] * 5 + [_P]),
    "gcm_sparse_step_plan": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _I, _P]),
    "gcm_sparse_chain_edges": (_I, [_P] * 6 + [_I, _P, _P, _L, _L, _I, _P]),
    "gcm_sparse_step_cached": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _I] + [_P] * 5 + [_I, _P] + [_I] * 5 + [_P]),
    "gcm_dense_rows_cached_weight_image": (_I, [_P, _P, _I, _I, _I, _P]),
    "gcm_dense_rows_cached_weight_image_floats": (_Z, []),
    "gcm_dense_rows_step_cached": (_I, [_P] * 5 + [_I, _P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P] + [_I] * 5 + [_P]),
    "gcm_dense_rows_bptt_cached": (_I, [_P, _P, _I, ctypes.c_long, ctypes.c_long, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _Z]
                                   + [_I] * 5 + [_P]),
    "gcm_dense_rows_bptt_dx_all_cached": (_I, [_P, _P, ctypes.c_long, ctypes.c_long, _I, _I, _P, _I, _I, _I, _P, _I, _P, _P]
                                          + [_I] * 5 + [_P]),
    "gcm_dense_rows_bptt_dx_all": (_I, [_P, _P, ctypes.c_long, ctypes.c_long, _P, _I, _I, _P, _I, _I, _I, _P, _P, _P]
                                   + [_I] * 5 + [_P]),
    "gcm_dense_rows_step_fwd": (_I, [_P] * 9 + [_I] + [_P] + [_I] * 3 + [_P] * 3 + [_I] * 5 + [_P]),
    "gcm_dense_rows_step_workspace_bytes": (_Z, [_P, _I, _I, _I, _I]),
    "gcm_dense_rows_step_fwd_ws": (_I, [_P] * 9 + [_I] + [_P] + [_I] * 3 + [_P] * 3 + [_P, _Z] + [_I] * 5 + [_P]),
    "gcm_edge_distance_pre": (_I, [_P] * 4 + [_I, _F, _P] + [_I] * 4 + [_P, _Z] + [_I] * 3 + [_P]),
    "gcm_edge_distance_pre_ex": (_I, [_P] * 4 + [_I, _F, _P] + [_I] * 4 + [_P, _I, _P, _Z] + [_I] * 3 + [_P]),
    "gcm_learned_step_supported": (_I, [_I] * 4),
    "gcm_learned_mlp_param_count": (_Z, [_I]),
    "gcm_learned_select_fused": (_I, [_P, _P, _P, _P, _I, _P, _F, _F, _F, _P, _I, _I, _I, _P]),
    "gcm_learned_advance_select_fused": (_I, [_P] * 5 + [_I, _P, _F, _F, _F] + [_P] * 6 + [_I, _I, _I, _P]),
    "gcm_learned_step_bwd": (_I, [_P] * 6 + [_I, _I] + [_P] * 6 + [_F, _F, _P, _P, _I] + [_I] * 5 + [_P]),
    "gcm_dense_rows_bptt_slabs": (_I, [_I, _I]),
    "gcm_learned_step_layout": (_I, [_I] * 6 + [_P]),
    "gcm_dense_rollout_tp_supported": (_I, [_P, _I, _I, _I, _I, _I, _I, _I]),
    "gcm_dense_rollout_tp_fwd": (_I, [_P, _P, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _P, _P]
                                 + [_I] * 7 + [_P]),
    "gcm_euclid_rollout_tp_supported": (_I, [_I] * 6),
    "gcm_euclid_rollout_tp_decide": (_I, [_P, _F, _P, _P, _I, _I, _I, _I, _P]),
    "gcm_euclid_rollout_tp_fwd": (_I, [_P, _F, _P, _P, _I, _I] + [_P] * 7 + [_Z, _I, _P, _P, _P] + [_I] * 7 + [_P]),
    "gcm_learned_step_steady": (_I, [_P] * 5 + [_I, _P, _I, _I, _I, _F, _F, _F] + [_P] * 13 + [_I] * 5 + [_P]),
    "gcm_adj_bits": (_I, [_P, _P, _I, _I, _P]),
    "gcm_learned_rollout_fwd": (_I, [_P, _P, _I, _P, _I, _I, _I, _F, _F, _F, _P, _P, _P, _P, _Z, _P, _P, _P, _P, _P]
                                + [_I] * 6 + [_P]),
    "gcm_learned_advance_select_inplace": (_I, [_P] * 5 + [_I, _P, _F, _F, _F] + [_P] * 6 + [_I, _I, _I, _P]),
    "gcm_learned_bptt_workspace_bytes": (_Z, [_I] * 6),
    "gcm_learned_bptt": (_I, [_P, _P, _I, ctypes.c_long, ctypes.c_long, _P, _I, _I, _F, _F, _I, _P, _P, _P, _Z]
                         + [_I] * 5 + [_P]),
    "gcm_learned_step_cached_functional": (_I, [_P] * 5 + [_I, _P, _I, _I, _I, _F, _F, _F] + [_P] * 12 + [_I] * 6 + [_P]),
    "gcm_learned_bptt_cached": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, ctypes.c_long, ctypes.c_long, _P, _I, _I, _F, _F, _I,
                                     _P, _P, _P, _Z] + [_I] * 5 + [_P]),
    "gcm_learned_step_cached": (_I, [_P] * 5 + [_I, _P, _I, _I, _I, _F, _F, _F] + [_P] * 11 + [_I] * 6 + [_P]),
    "gcm_dense_rollout_bwd_params_workspace_bytes": (_Z, [_I] * 5),
    "gcm_dense_rollout_bwd_params": (_I, [_P, ctypes.c_long, ctypes.c_long, ctypes.c_long] + [_P] * 4 + [_I, _I]
                                     + [_P] * 6 + [_Z] + [_I] * 6 + [_P]),
    "gcm_dense_rows_bptt_workspace_bytes": (_Z, [_I] * 5),
    "gcm_dense_rows_bptt": (_I, [_P, _P, _I, ctypes.c_long, ctypes.c_long, _P, _I, _I, _I, _P, _P, _P, _Z]
                            + [_I] * 5 + [_P]),
    "gcm_dense_rollout_fwd": (_I, [_P] * 6 + [_I] + [_P] * 3 + [_I] + [_P] * 3 + [_I] + [_P] * 6
                              + [_Z] + [_I] * 6 + [_P]),
    "gcm_dense_rollout_persistent_fwd": (_I, [_P] * 6 + [_I] + [_P] * 3 + [_I] + [_P] * 3 + [_I]
                                         + [_P] * 5 + [_I] * 7 + [_P]),
    "gcm_dense_bptt_batched_slabs": (_I, [_I]),
    "gcm_dense_bptt_batched": (_I, [_P] * 8 + [_I] + [_P] * 3 + [_I] + [_P] * 7 + [_I] * 6 + [_P]),
    "gcm_dense_gnodes_scan": (_I, [_P] * 7 + [_I] * 4 + [_P]),
    "gcm_dense_rollout_bwd_workspace_bytes": (_Z, [_I] * 5),
    "gcm_dense_rollout_bwd_batched_workspace_bytes": (_Z, [_I] * 6),
    "gcm_dense_rollout_bwd": (_I, [_P] * 9 + [_I] + [_P] * 3 + [_I] + [_P] * 8 + [_Z] + [_I] * 6
                              + [_P]),
}


class SelectorDesc(ctypes.Structure):
    """struct gcm_selector_desc (include/gcm_hip.h)."""
    _fields_ = [("kind", _I), ("n_hops", _I), ("hops", ctypes.c_int32 * 16), ("direction", _I),
                ("mode", _I), ("max_distance", _F), ("dist_param", _P), ("a0", _I), ("a1", _I),
                ("b0", _I), ("b1", _I), ("bidirectional", _I), ("cur_rows", _P), ("n_cur_rows", _I)]


SEL_TEMPORAL, SEL_DENSE, SEL_DISTANCE = 1, 2, 3


GCM_EUNSUPPORTED = -2   # include/gcm_hip.h


class HipLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the bound library; raises HipLibraryError if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise HipLibraryError(
                f"{_LIB_PATH} not found: build it with `python __graft_entry__.py` or "
                "`make -C graph-conv-memory_amd/csrc` (hipcc, --offload-arch=gfx950). "
                "This package has no CPU fallback."
            )
        # (GCM_HIP_LIB: a diagnostic build of the same library - csrc/Makefile `stamps*`, `exp` - for the
        #  dev tools under tools/; never set by the product)
        handle = ctypes.CDLL(os.environ.get("GCM_HIP_LIB") or _LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        got = handle.gcm_abi_version()
        if got != ABI_VERSION:     # (the prototype table below was written against include/gcm_hip.h's GCM_ABI_VERSION)
            raise HipLibraryError(f"{_LIB_PATH} has ABI revision {got}, this binding expects {ABI_VERSION}: "
                                  "rebuild the library (`python __graft_entry__.py`)")
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().gcm_status_string(rc).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {rc})")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    """Raw handle of the current HIP stream of the current device (what every kernel call takes).
    torch.cuda.current_stream() builds a Stream object per call (~2.5 us); this is the accessor
    underneath it."""
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def on_device(*tensors):
    """Validate that every given tensor lives on a HIP device and is contiguous."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise HipLibraryError(
                "gcm (MI355X build) runs on HIP devices only; got a tensor on "
                f"'{t.device}'. There is no CPU fallback."
            )
        if not t.is_contiguous():
            raise ValueError("gcm kernels need contiguous tensors")
        if t.device.index != torch.cuda.current_device():
            raise HipLibraryError(
                f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: the "
                "kernels launch on the current device's stream - wrap the call in torch.cuda.device(...)")
