/* gcm_hip.h - C ABI of libgcm_hip.so: the MI355X (gfx950) kernels behind the
 * DenseGCM / SparseGCM hot path of proroklab/graph-conv-memory.
 *
 * The reference is pure Python on PyTorch (+ torch_geometric); it has NO FFI for
 * this path.  Each entry point below therefore cites the reference Python lines
 * (relative to the reference checkout) whose device work it replaces; the
 * binding a maintainer adds is the ctypes stub shown in INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller unless the
 *     parameter is documented "host"; tensors are contiguous row-major,
 *     fp32 / int64 exactly like the reference's hidden state;
 *   - nothing is allocated inside; kernels that need scratch take
 *     `workspace` + `workspace_bytes`, sized by the matching *_workspace_bytes;
 *   - `stream` is a hipStream_t; calls are asynchronous, re-entrant, hold no
 *     global state and are HIP-graph capturable;
 *   - return 0 on success, GCM_E* (<0) for an argument error detected on the
 *     host, or a positive hipError_t from the launch.
 */
#ifndef GCM_HIP_H
#define GCM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gcm_stream_t; /* hipStream_t */

#define GCM_OK 0
#define GCM_EINVAL (-1)       /* null pointer / non-positive size            */
#define GCM_EUNSUPPORTED (-2) /* shape outside what the kernels are built for */
#define GCM_EWORKSPACE (-3)   /* workspace too small                          */

/* activation fused into the graph-conv epilogue */
#define GCM_ACT_NONE 0
#define GCM_ACT_TANH 1
#define GCM_ACT_RELU 2

/* TemporalBackedge direction (edge_selectors/temporal.py:41-42) */
#define GCM_DIR_FORWARD 1
#define GCM_DIR_BACKWARD 2
#define GCM_DIR_BOTH 3

/* Distance selector modes (edge_selectors/distance.py:42-81) */
#define GCM_DIST_EUCLID_CROSSBATCH 0 /* EuclideanEdge: mean over ALL graphs b' */
#define GCM_DIST_L2_PERGRAPH 1       /* SpatialEdge: per-graph L2 on slices   */
#define GCM_DIST_COSINE_SIM 2        /* CosineEdge: similarity, eps 1e-8      */

/* bits of the device `flags` word written by the state kernels */
#define GCM_FLAG_WRAPPED 1u    /* some graph overflowed and was rolled (gcm.py:263-271) */
#define GCM_FLAG_BAD_COUNT 2u  /* num_nodes outside [0, N]                            */
#define GCM_FLAG_NONFINITE 4u  /* belief state has NaN/Inf (gcm.py:316-318)            */

int gcm_version(void);
const char* gcm_status_string(int code);

/* ---- DenseGCM state ------------------------------------------------------ */

/* gcm.py:262-278 + wrap_overflow gcm.py:323-355.  For every graph b:
 * copy nodes/adj/weights to the *_out buffers; if num_nodes_in[b] + 1 > N first
 * clear node 0 and its row/col and rotate everything one slot towards index 0;
 * then write x[b] into row cur = (wrapped ? num_nodes_in[b]-1 : num_nodes_in[b]).
 * cur_idx_out[b] = cur, num_nodes_out[b] = cur + 1 (gcm.py:320).
 * weights_in/weights_out may both be NULL (reference: weights.numel()==0);
 * adj_in/adj_out may both be NULL (nodes only).  flags (uint32, 1 word) is OR-ed. */
int gcm_state_advance_fwd(const float* nodes_in, const float* adj_in, const float* weights_in,
                          const int64_t* num_nodes_in, const float* x, float* nodes_out,
                          float* adj_out, float* weights_out, int64_t* cur_idx_out,
                          int64_t* num_nodes_out, uint32_t* flags, int B, int N, int F,
                          gcm_stream_t stream);

/* Adjoint of the above for nodes (always) and one [B,N,N] plane (g_plane_*, may
 * be NULL): g_nodes_in, g_x and g_plane_in are overwritten. */
int gcm_state_advance_bwd(const float* g_nodes_out, const float* g_plane_out,
                          const int64_t* num_nodes_in, float* g_nodes_in, float* g_plane_in,
                          float* g_x, int B, int N, int F, gcm_stream_t stream);

/* gcm.py:309-314: out[b,:] = feats[b, cur_idx[b], :]; also ORs GCM_FLAG_NONFINITE
 * into flags when a gathered value is NaN/Inf (gcm.py:316-318). */
int gcm_gather_rows_fwd(const float* feats, const int64_t* cur_idx, float* out, uint32_t* flags,
                        int B, int N, int H, gcm_stream_t stream);
/* adjoint: g_feats (zero-filled by the kernel) gets g_out[b] at row cur_idx[b] */
int gcm_gather_rows_bwd(const float* g_out, const int64_t* cur_idx, float* g_feats, int B, int N,
                        int H, gcm_stream_t stream);

/* ---- dense edge selectors (plugin API #1); adj is modified IN PLACE --------- */

/* edge_selectors/temporal.py:72-88.  hops: HOST array of n_hops (<= 16) ints. */
int gcm_edge_temporal(float* adj, const int64_t* cur_idx, const int32_t* hops_host, int n_hops,
                      int direction, int B, int N, gcm_stream_t stream);

/* edge_selectors/dense.py:11-23. */
int gcm_edge_dense(float* adj, const int64_t* cur_idx, int B, int N, gcm_stream_t stream);

/* edge_selectors/distance.py:18-39 with dist_fn :48-49 / :59-61 / :77-81.
 * dist_param: device pointer to the learned scale (distance.py:13-16,21-22) or NULL.
 * [a0,a1) / [b0,b1): feature slices of the current node / the past nodes (mode
 * L2_PERGRAPH; other modes use the full row).  dist_out (may be NULL) receives the
 * [B,N] distance matrix the threshold was applied to. */
size_t gcm_edge_distance_workspace_bytes(int mode, int B, int N, int F);
int gcm_edge_distance(const float* nodes, float* adj, const int64_t* cur_idx, int mode,
                      float max_distance, const float* dist_param, int a0, int a1, int b0, int b1,
                      int bidirectional, float* dist_out, void* workspace, size_t workspace_bytes,
                      int B, int N, int F, gcm_stream_t stream);

/* ---- DenseGraphConv (PyG; call sites README.md:56-62) ---------------------- */

/* out = act( (adj @ x) @ w_rel^T + b_rel + x @ w_root^T ).
 * x [B,N,Fi], adj [B,N,N], w_rel/w_root [Fo,Fi], b_rel [Fo] or NULL, out [B,N,Fo].
 * agg (may be NULL) receives adj @ x [B,N,Fi] for the backward pass. */
int gcm_dense_graphconv_fwd(const float* x, const float* adj, const float* w_rel,
                            const float* b_rel, const float* w_root, float* out, float* agg, int B,
                            int N, int Fi, int Fo, int act, gcm_stream_t stream);

/* Backward of the above.  g_out/out [B,N,Fo].  Outputs (each may be NULL to skip):
 * g_x [B,N,Fi], g_adj [B,N,N], g_w_rel/g_w_root [Fo,Fi], g_b_rel [Fo] (overwritten,
 * summed over B).  agg as saved by the forward. */
size_t gcm_dense_graphconv_bwd_workspace_bytes(int B, int N, int Fi, int Fo);
int gcm_dense_graphconv_bwd(const float* g_out, const float* out, const float* x, const float* adj,
                            const float* agg, const float* w_rel, const float* w_root, float* g_x,
                            float* g_adj, float* g_w_rel, float* g_b_rel, float* g_w_root,
                            void* workspace, size_t workspace_bytes, int B, int N, int Fi, int Fo,
                            int act, gcm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GCM_HIP_H */
