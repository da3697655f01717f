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
 *   - `stream` is a hipStream_t; calls are asynchronous and HIP-graph capturable; the
 *     only process state is a one-time "dynamic LDS size allowed" attribute per kernel and
 *     the cached compute-unit count of the device (gcm_dense_bptt_batched_slabs);
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
/* ABI revision of this header: bumped whenever an EXISTING entry point's signature or the size of a caller-allocated
 * buffer changes (round 5 did both: the weight image of gcm_dense_rows_cached_weight_image grew from 16 384 to
 * gcm_dense_rows_cached_weight_image_floats() = 36 864 floats, and gcm_edge_distance_step_cached /
 * gcm_learned_step_cached(_functional) / gcm_learned_bptt_cached gained pointer arguments mid-signature).  A binding
 * compares gcm_abi_version() with the GCM_ABI_VERSION it was written against before any other call (gcm/_hip.py
 * does; INTEGRATION.md shows the stub) - stale ctypes prototypes would otherwise shift pointers silently. */
#define GCM_ABI_VERSION 6
int gcm_abi_version(void);
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

/* edge_selectors/temporal.py:51-70 (TemporalBackedge(learned=True)): every graph with 0 < n_b = cur_idx[b]
 * stored nodes draws S straight-through gumbel one-hots over window[:n_b] (noise [S, B, Wn], Wn = min(W, N),
 * standard gumbel draws) - or, deterministic, one hard sparsemax (util.py:29-42) - ORs them (util.py:456-465)
 * and ADDS the mask to adj[b, n_b, :n_b].  soft [S or 1, B, Wn] (the soft samples) is what the backward
 * needs; mask_ws [B, Wn] scratch.  n_b > W raises GCM_FLAG_WINDOW (the reference's slice assignment fails
 * there) and leaves the graph untouched.  S <= 32. */
#define GCM_FLAG_WINDOW 128u
int gcm_temporal_window_fwd(const float* window, const float* noise, const int64_t* cur_idx, float* adj,
                            float* soft, float* mask_ws, int B, int N, int W, int S, int deterministic,
                            uint32_t* flags, gcm_stream_t stream);
/* its adjoint w.r.t. the window logits: g_window_part [B, Wn], one row per graph (the caller sums over b);
 * g_adj [B, N, N] is the gradient of the adjacency the forward wrote (it passes through to adj unchanged). */
int gcm_temporal_window_bwd(const float* g_adj, const float* soft, const int64_t* cur_idx,
                            float* g_window_part, int B, int N, int W, int S, int deterministic,
                            gcm_stream_t stream);

/* ---- narrow Linear layers on many rows (edge_selectors/learned.py:38-51 edge network; gcm.py:133-140) ----
 * W is [O, I] row major (torch.nn.Linear.weight), I, O <= 64.
 *   transpose = 0:  y[M, O] = x[M, I] W^T + bias      (bias may be NULL)
 *   transpose = 1:  y[M, I] = x[M, O] W               (the input gradient; bias NULL)
 * ldy: leading dimension of y (0: dense).  gamma/beta/h non-NULL (transpose = 0 only): additionally
 * h[M, O] = LayerNorm(relu(y)) * gamma + beta over the O columns (biased variance, eps) - the
 * Linear-ReLU-LayerNorm stage of the default edge network in one pass over the rows. */
int gcm_rows_linear(const float* x, const float* w, const float* bias, float* y, int64_t M, int I, int O,
                    int transpose, int ldy, const float* gamma, const float* beta, float eps, float* h,
                    gcm_stream_t stream);
/* PositionalEncoding(mode="cat") (gcm.py:133-140): out [B,N,F] already holds the re-projected features in
 * its columns cat_dim.. (gcm_rows_linear with ldy = F); this writes the table columns pe[i, :cat_dim] of the
 * rows i <= num_nodes[b] and restores the rows beyond from x. */
int gcm_posenc_cat_finish(const float* x, const float* pe, int pe_ld, const int64_t* num_nodes, float* out,
                          int B, int N, int F, int cat_dim, gcm_stream_t stream);
/* adjoint: g_x [B,N,F] = the rows beyond num_nodes of g_out (zero elsewhere), g_proj [B*N, F - cat_dim] =
 * the encoded rows' columns cat_dim.. (zero elsewhere). */
int gcm_posenc_cat_bwd(const float* g_out, const int64_t* num_nodes, float* g_x, float* g_proj, int B, int N,
                       int F, int cat_dim, gcm_stream_t stream);

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
/* The same with the "current nodes" of GCM_DIST_EUCLID_CROSSBATCH supplied by the caller: cur_rows
 * [n_cur_rows, F] = the current nodes of every rank of a batch-sharded run, all-gathered (SURVEY 8e:
 * EuclideanEdge's mean couples all graphs of the GLOBAL batch, distance.py:48-49).  NULL / 0 = the
 * local graphs' own current nodes (gcm_edge_distance).  Workspace: gcm_edge_distance_workspace_bytes
 * with B = max(B, n_cur_rows). */
int gcm_edge_distance_ex(const float* nodes, float* adj, const int64_t* cur_idx, int mode,
                         float max_distance, const float* dist_param, int a0, int a1, int b0, int b1,
                         int bidirectional, float* dist_out, const float* cur_rows, int n_cur_rows,
                         void* workspace, size_t workspace_bytes, int B, int N, int F,
                         gcm_stream_t stream);

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

/* ---- SparseGCM (src/gcm/sparse_gcm.py:72-212) -------------------------------- */

/* bits OR-ed into `flags` by the sparse kernels */
#define GCM_FLAG_SPARSE_OVERFLOW 8u  /* T_b + tau_b > N  (sparse_gcm.py:120-121)   */
#define GCM_FLAG_ACAUSAL 16u         /* an edge with source >= sink (sparse_gcm.py:171) */

/* Per-call plan, replaces the Python loops over B of util.py:176-240:
 *   node_off[b]  = sum_{b'<b} (T+taus)[b']   (flat node offsets, util.get_batch_offsets), [B+1]
 *   new_off[b]   = sum_{b'<b} taus[b']       (offsets of the new nodes in output order), [B+1]
 *   totals[0] = node_off[B], totals[1] = new_off[B], totals[2] = max_b (T+taus)[b],
 *   totals[3] = max_b taus[b]                                                      */
int gcm_sparse_plan(const int64_t* T, const int64_t* taus, int64_t* node_off, int64_t* new_off,
                    int64_t* totals, int B, gcm_stream_t stream);

/* sparse_gcm.py:116-128: nodes_out = nodes_in with x[b, k, :] written to row T[b]+k for
 * k < taus[b] (x is [B, t_pad, F], zero padded).  ORs GCM_FLAG_SPARSE_OVERFLOW. */
/* (nodes_in == NULL: the incoming node matrix is all zeros - empty graphs - and is not read) */
int gcm_sparse_insert_fwd(const float* nodes_in, const float* x, const int64_t* T,
                          const int64_t* taus, float* nodes_out, uint32_t* flags, int B, int N,
                          int F, int t_pad, gcm_stream_t stream);
int gcm_sparse_insert_bwd(const float* g_nodes_out, const int64_t* T, const int64_t* taus,
                          float* g_nodes_in, float* g_x, int B, int N, int F, int t_pad,
                          gcm_stream_t stream);

/* SparseGCM called one node at a time (sparse_gcm.py:72-212 with x [B, 1, F]; ray_sparse_gcm.py's rollout loop):
 * everything such a call plans, in ONE launch - gcm_sparse_plan (util.py:176-208), gcm_sparse_temporal_count
 * (sparse_edge_selectors/temporal.py:18-63), T + taus (sparse_gcm.py:211), and the per-graph pointer of the merged
 * COO list.  plan: node_off [B+1] | new_off [B+1] | edge_off [B+1] | totals [4] {M, n_new, max_total, max_tau}.
 * old_bptr [B+1] (NULL: an empty list): per-graph pointer of the stored list; merged_bptr [B+1] (may be NULL) =
 * old_bptr + edge_off.  T_out [B] (may be NULL) = T + taus.  hops_host: HOST array, strictly descending. */
int gcm_sparse_step_plan(const int64_t* T, const int64_t* taus, const int32_t* hops_host, int n_hops,
                         const int64_t* old_bptr, int64_t* plan, int64_t* T_out, int64_t* merged_bptr, int B,
                         gcm_stream_t stream);
/* A call of such a chain: TemporalEdge's new entries (sparse_edge_selectors/temporal.py:18-63; one new node per
 * graph, taus in {0, 1}, every hop >= 1) generated straight into their places of the merged COO list
 * (sparse_gcm.py:132-139) - gcm_sparse_temporal_fill + gcm_coo_merge_segments as one launch, the number of new
 * entries read on the device (plan from gcm_sparse_step_plan), so that the host can enqueue the whole call before
 * it reads any size back.  out_idx: 3 rows (batch, sink, source), E = Ea + plan.edge_off[B] apart, in a buffer of at
 * least 3 * (Ea + max_new) elements; out_val [>= Ea + max_new]: unit weights.  max_new >= the new entries
 * (B * n_hops bounds them). */
int gcm_sparse_chain_edges(const int64_t* old_idx, const int64_t* old_bptr, const int64_t* plan, const int64_t* T,
                           const int64_t* taus, const int32_t* hops_host, int n_hops, int64_t* out_idx,
                           float* out_val, int64_t Ea, int64_t max_new, int B, gcm_stream_t stream);
/* sparse_edge_selectors/temporal.py:18-63, closed form.  hops_host: HOST array, must be
 * sorted DESCENDING and unique (so that sources ascend inside a sink: coalesced order).
 * count: edge_off[b] = number of edges of graphs < b, [B+1].
 * fill : indices [3, E] rows (batch, sink, source), E = edge_off[B]. */
int gcm_sparse_temporal_count(const int64_t* T, const int64_t* taus, const int32_t* hops_host,
                              int n_hops, int64_t* edge_off, int B, gcm_stream_t stream);
int gcm_sparse_temporal_fill(const int64_t* T, const int64_t* taus, const int32_t* hops_host,
                             int n_hops, const int64_t* edge_off, int64_t* indices, int64_t E,
                             int B, gcm_stream_t stream);
/* Whole episodes from EMPTY graphs (T = 0 for every graph): every index structure of the call in closed form, one
 * launch - the COO entries TemporalEdge adds (coo [3, E], vals [E] or NULL), the flat (source, sink) list edge_index
 * [2, E] with its CSR row pointers row_ptr [M + 1], and (col_ptr non-NULL) the CSC view col_ptr [M + 1] / rows [E] /
 * perm [E] of gcm_csc_from_csr_batched.  node_off / edge_off [B + 1]: gcm_sparse_plan / gcm_sparse_temporal_count (or
 * gcm_sparse_step_plan) on T = 0.  hops: distinct, descending, >= 1 (GCM_EUNSUPPORTED otherwise: the general kernels). */
int gcm_sparse_temporal_structure(const int64_t* taus, const int32_t* hops_host, int n_hops, const int64_t* node_off,
                                  const int64_t* edge_off, int64_t* coo, float* vals, int64_t* edge_index,
                                  int64_t* row_ptr, int64_t* col_ptr, int64_t* rows, int64_t* perm, int64_t E,
                                  int64_t M, int B, gcm_stream_t stream);
/* ... also writing the entries' unit weights vals [E] (NULL: indices only) */
int gcm_sparse_temporal_fill_vals(const int64_t* T, const int64_t* taus, const int32_t* hops_host, int n_hops,
                                  const int64_t* edge_off, int64_t* indices, float* vals, int64_t E, int B,
                                  gcm_stream_t stream);

/* util.flatten_nodes util.py:426-452: flat[node_off[b] + i] = nodes[b, i] for i < (T+taus)[b]. */
int gcm_sparse_flatten_fwd(const float* nodes, const int64_t* T, const int64_t* taus,
                           const int64_t* node_off, float* flat, int B, int N, int F, int64_t M,
                           gcm_stream_t stream);
int gcm_sparse_flatten_bwd(const float* g_flat, const int64_t* T, const int64_t* taus,
                           const int64_t* node_off, float* g_nodes, int B, int N, int F, int64_t M,
                           gcm_stream_t stream);

/* util.flatten_adj util.py:287-304 + flip (sparse_gcm.py:167-171): coo [3,E] (batch, sink,
 * source), sorted by (batch, sink, source) -> edge_index [2,E] = (source, sink) + node_off[b]
 * and CSR-by-destination row_ptr [M+1].  ORs GCM_FLAG_ACAUSAL when source >= sink. */
int gcm_sparse_edges_to_csr(const int64_t* coo, const int64_t* node_off, int64_t* edge_index,
                            int64_t* row_ptr, uint32_t* flags, int64_t E, int64_t M, int B,
                            gcm_stream_t stream);
/* sparse_gcm.py:132-139 (concatenate the stored and the new COO lists, coalesce) for the case every
 * shipped selector produces: the new entries of a graph sort BEHIND its stored ones (their sinks are
 * the new nodes).  Then the coalesced result is, per graph, the old entries followed by the new
 * ones - a segmented concatenation, no sort.  old_idx [3,Ea] / new_idx [3,Eb] coalesced (batch,
 * sink, source); old_bptr / new_bptr [B+1] = gcm_ptr_from_sorted over their batch rows; out_idx
 * [3,Ea+Eb]; out_val (may be NULL) gets the values, perm (may be NULL) the position of every
 * output entry in cat(old, new) (for values that carry gradients).  ORs GCM_FLAG_MERGE_ORDER when
 * a new entry does not sort behind the stored ones (the caller then falls back to a sort). */
#define GCM_FLAG_MERGE_ORDER 64u
int gcm_coo_merge_segments(const int64_t* old_idx, const int64_t* new_idx, const float* old_val,
                           const float* new_val, const int64_t* old_bptr, const int64_t* new_bptr,
                           int64_t* out_idx, float* out_val, int64_t* perm, uint32_t* flags,
                           int64_t Ea, int64_t Eb, int B, gcm_stream_t stream);

/* generic: ptr[r] = first position in sorted keys[0..E) with key >= r, r in [0, M]. */
int gcm_ptr_from_sorted(const int64_t* keys, int64_t* ptr, int64_t E, int64_t M,
                        gcm_stream_t stream);

/* torch_geometric.utils.k_hop_subgraph (sparse_gcm.py:192-198) without relabelling:
 * mask[i] = 1 for every node within `hops` steps AGAINST the edges from a new node.
 * row_ptr/col: CSR by destination (col = source).  mask is [M] bytes, overwritten;
 * scratch is 2*M bytes; t_pad >= max(taus) (the padded time length of x). */
int gcm_khop_mask(const int64_t* row_ptr, const int64_t* col, const int64_t* node_off,
                  const int64_t* T, const int64_t* taus, int hops, uint8_t* mask, uint8_t* scratch,
                  int64_t M, int B, int t_pad, gcm_stream_t stream);

/* sparse_gcm.py:176-208: out[b, k, :] = feats[node_off[b] + T[b] + k, :] for k < taus[b],
 * zero elsewhere; ORs GCM_FLAG_NONFINITE. */
int gcm_sparse_extract_fwd(const float* feats, const int64_t* T, const int64_t* taus,
                           const int64_t* node_off, float* out, uint32_t* flags, int B, int t_pad,
                           int H, int64_t M, gcm_stream_t stream);
int gcm_sparse_extract_bwd(const float* g_out, const int64_t* T, const int64_t* taus,
                           const int64_t* node_off, float* g_feats, int B, int t_pad, int H,
                           int64_t M, gcm_stream_t stream);

/* ---- GraphConv over CSR (PyG GraphConv; call sites ray_sparse_gcm.py:37-39) ------ */

/* out[i] = act( (sum_{e in row i} w[e] * x[col[e]]) @ w_rel^T + b_rel + x[i] @ w_root^T ).
 * x [M,Fi]; row_ptr [M+1], col [E] = CSR by destination; w [E] or NULL (=1);
 * mask [M] bytes or NULL: rows with mask 0 are skipped (out = 0) and so are edges whose
 * source has mask 0 (= the conv restricted to the k-hop subgraph).  agg (may be NULL)
 * receives the aggregated neighbours [M,Fi] for the backward pass. */
int gcm_csr_graphconv_fwd(const float* x, const int64_t* row_ptr, const int64_t* col,
                          const float* w, const uint8_t* mask, const float* w_rel,
                          const float* b_rel, const float* w_root, float* out, float* agg,
                          int64_t M, int Fi, int Fo, int act, gcm_stream_t stream);
/* The same with the finite check SparseGCM makes on the rows it returns (sparse_gcm.py:201-203) folded into the
 * layer's epilogue: GCM_FLAG_NONFINITE is raised in *flags when an output value is not finite - for callers that hand
 * `out` back as it is (every row a new node).  Fi = 32, Fo in {32, 64}, M >= 32 (the persistent-wave kernel);
 * GCM_EUNSUPPORTED otherwise (..._supported tells beforehand). */
int gcm_csr_graphconv_fwd_checked_supported(int64_t M, int Fi, int Fo);
int gcm_csr_graphconv_fwd_checked(const float* x, const int64_t* row_ptr, const int64_t* col, const float* w,
                                  const uint8_t* mask, const float* w_rel, const float* b_rel, const float* w_root,
                                  float* out, float* agg, int64_t M, int Fi, int Fo, int act, uint32_t* flags,
                                  gcm_stream_t stream);

/* Backward.  col_ptr/rows/perm: the same edges as CSC by source (perm[k] = position of CSC
 * entry k in the CSR edge order, for w and g_w).  Outputs may be NULL to skip. */
size_t gcm_csr_graphconv_bwd_workspace_bytes(int64_t M, int Fi, int Fo);
int gcm_csr_graphconv_bwd(const float* g_out, const float* out, const float* x, const float* agg,
                          const int64_t* row_ptr, const int64_t* col, const int64_t* col_ptr,
                          const int64_t* rows, const int64_t* perm, const float* w,
                          const uint8_t* mask, const float* w_rel, const float* w_root, float* g_x,
                          float* g_w, float* g_w_rel, float* g_b_rel, float* g_w_root,
                          void* workspace, size_t workspace_bytes, int64_t M, int64_t E, int Fi,
                          int Fo, int act, gcm_stream_t stream);

/* ---- LearnedEdge (src/gcm/edge_selectors/learned.py:53-125) ------------------- */

/* learned.py:66-72: pairs[b, j, :] = cat(nodes[b, cur_b], nodes[b, j]) for j < cur_b, zero rows
 * otherwise ([B, N, 2F]: the edge network's input, one row per candidate edge; replaces
 * util.idxs_up_to_num_nodes util.py:501-522 + two advanced-index gathers + cat). */
int gcm_learned_pairs_fwd(const float* nodes, const int64_t* cur_idx, float* pairs, int B, int N,
                          int F, gcm_stream_t stream);
/* adjoint: g_nodes [B,N,F] is overwritten. */
int gcm_learned_pairs_bwd(const float* g_pairs, const int64_t* cur_idx, float* g_nodes, int B,
                          int N, int F, gcm_stream_t stream);

/* Weight gradient of a skinny Linear (the LearnedEdge edge network, learned.py:38-51, scores B*N
 * candidate rows with <= 64-wide linears): dw_db = dW [O,I] | db [O] with dW = dY^T X, db = column
 * sums of dY; dY [M,O], X [M,I], O, I <= 64 (GCM_EUNSUPPORTED otherwise).  Rows are split over the
 * grid, partial slabs are summed in a fixed order. */
size_t gcm_skinny_wgrad_workspace_bytes(int M, int O, int I);
int gcm_skinny_wgrad(const float* dy, const float* x, float* dw_db, void* workspace,
                     size_t workspace_bytes, int M, int O, int I, gcm_stream_t stream);

/* y = LayerNorm(relu(x)) over the last dimension (F <= 64, biased variance, eps as torch.nn.LayerNorm):
 * the ReLU - LayerNorm pairs of the LearnedEdge edge network (learned.py:38-51) on M = B*N rows.
 * Backward: dx [M,F] and dgamma_dbeta = dgamma [F] | dbeta [F] (per-workgroup slabs, fixed-order
 * sum); x is the pre-activation saved by the caller. */
int gcm_relu_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                           int64_t M, int F, float eps, gcm_stream_t stream);
size_t gcm_relu_layernorm_bwd_workspace_bytes(int64_t M, int F);
int gcm_relu_layernorm_bwd(const float* dy, const float* x, const float* gamma, float* dx,
                           float* dgamma_dbeta, void* workspace, size_t workspace_bytes, int64_t M,
                           int F, float eps, gcm_stream_t stream);

/* learned.py:76-111 (non-deterministic branch), fused: per graph soft = softmax_j<cur(logits +
 * noise) (gumbel_softmax tau=1 with caller-supplied gumbel noise), edge_j = soft_j > cutoff
 * (STE forward, util.py:9-18), adj[b, cur, j] = (edge_j + adj[b, cur, j] > 0) for j < cur.
 * adj is updated IN PLACE; soft [B,N] is kept for the backward pass. */
int gcm_learned_select_fwd(const float* logits, const float* noise, const int64_t* cur_idx,
                           float cutoff, float* adj, float* soft, int B, int N,
                           gcm_stream_t stream);
/* STE backward is the identity, so d soft_j = g_adj[b, cur, j]; g_logits = softmax backward. */
int gcm_learned_select_bwd(const float* g_adj, const float* soft, const int64_t* cur_idx,
                           float* g_logits, int B, int N, gcm_stream_t stream);

/* ---- fused DenseGCM step for the canonical GNN (README.md:52-62, gcm.py:308-314) ------ */

/* 1 when the fused kernels cover this shape (N <= 128, F <= 64, H1 <= 64, H2 <= 256 and the
 * LDS images fit 160 KB); otherwise use the layered gcm_dense_graphconv_* path. */
int gcm_dense_gnn2_row_supported(int N, int F, int H1, int H2);
/* number of floats in one parameter-gradient slab:
 * dW_rel1 [H1*F] | dW_root1 [H1*F] | db1 [H1] | dW_rel2 [H2*H1] | dW_root2 [H2*H1] | db2 [H2] */
size_t gcm_dense_gnn2_param_count(int F, int H1, int H2);

/* mx[b] = act2(gc2(act1(gc1(x, adj)), adj))[b, cur_idx[b]]  - two DenseGraphConv layers in one
 * kernel, adj and x resident in LDS, the second layer evaluated only on the kept row.
 * Saved for backward (each may be NULL when no gradient is needed): h1 [B,N,H1] (post
 * activation), agg1 = adj@x [B,N,F], agg2 = adj[cur]@h1 [B,H1].  ORs GCM_FLAG_NONFINITE. */
int gcm_dense_gnn2_row_fwd(const float* x, const float* adj, const int64_t* cur_idx,
                           const float* w_rel1, const float* b_rel1, const float* w_root1, int act1,
                           const float* w_rel2, const float* b_rel2, const float* w_root2, int act2,
                           float* mx, float* h1, float* agg1, float* agg2, uint32_t* flags, int B,
                           int N, int F, int H1, int H2, gcm_stream_t stream);

/* Backward of the above with the adjoint of gcm_state_advance_fwd folded in:
 * g_nodes_out [B,N,F] (gradient w.r.t. the nodes returned by the step, NULL = 0) and g_mx [B,H2]
 * -> g_nodes_in [B,N,F], g_obs [B,F] and one parameter-gradient slab per graph
 * (slabs [B, gcm_dense_gnn2_param_count]; accumulate != 0 adds to what is there). */
int gcm_dense_gnn2_row_bwd(const float* g_mx, const float* g_nodes_out, const float* x,
                           const float* adj, const int64_t* cur_idx, const int64_t* num_nodes_in,
                           const float* w_rel1, const float* b_rel1, const float* w_root1, int act1,
                           const float* w_rel2, const float* b_rel2, const float* w_root2, int act2,
                           const float* mx, const float* h1, const float* agg1, const float* agg2,
                           float* g_nodes_in, float* g_obs, float* slabs, int accumulate, int B,
                           int N, int F, int H1, int H2, gcm_stream_t stream);

/* out[e] = sum_i slabs[i, e]  (fixed order, deterministic) */
int gcm_sum_slabs(const float* slabs, int n_slabs, int len, float* out, gcm_stream_t stream);
/* ... plus a running total: out[e] = prev[e] + sum_i slabs[i, e]  (prev may be NULL or == out) */
int gcm_sum_slabs_acc(const float* slabs, int n_slabs, int len, const float* prev, float* out,
                      gcm_stream_t stream);

/* One native edge selector of the per-step chain (host struct). */
#define GCM_SEL_TEMPORAL 1
#define GCM_SEL_DENSE 2
#define GCM_SEL_DISTANCE 3
typedef struct gcm_selector_desc {
  int kind;               /* GCM_SEL_*                                              */
  int n_hops;             /* temporal                                                */
  int32_t hops[16];       /* temporal                                                */
  int direction;          /* temporal: GCM_DIR_*                                     */
  int mode;               /* distance: GCM_DIST_*                                    */
  float max_distance;     /* distance                                                */
  const float* dist_param;/* distance: device pointer or NULL                        */
  int a0, a1, b0, b1;     /* distance: pose slices                                   */
  int bidirectional;      /* distance                                                */
  /* distance, GCM_DIST_EUCLID_CROSSBATCH in a batch-sharded run: the current nodes of EVERY rank,
   * all-gathered by the caller [n_cur_rows, F] (device pointer), so that the mean of distance.py:48-49
   * runs over the global batch; NULL / 0: the local batch's own current nodes                */
  const float* cur_rows;
  int n_cur_rows;
} gcm_selector_desc;

/* ---- one DenseGCM step per call (the per-step drop-in API, gcm.py:213-321) ------------ */

/* Packed parameter vector used by the step/rollout drivers (same layout as a gradient slab):
 * w_rel1 [H1*F] | w_root1 [H1*F] | b_rel1 [H1] | w_rel2 [H2*H1] | w_root2 [H2*H1] | b_rel2 [H2]. */

/* state advance + native selector chain + fused GNN, enqueued by ONE call (what
 * DenseGCM.forward does on the fused path).  has_bias: bit0 = layer 1, bit1 = layer 2. */
int gcm_dense_step_fwd(const float* obs, const float* nodes_in, const float* adj_in,
                       const int64_t* count_in, float* nodes_out, float* adj_out,
                       int64_t* cur_out, int64_t* count_out, const gcm_selector_desc* selectors,
                       int n_selectors, const float* params, int has_bias, int act1, int act2,
                       float* mx, float* h1, float* agg1, float* agg2, uint32_t* flags,
                       void* workspace, size_t workspace_bytes, int B, int N, int F, int H1,
                       int H2, gcm_stream_t stream);

/* ONE kernel per forward step: state advance (gcm.py:262-278, 323-355) + temporal/dense selector
 * writes + fused GNN (k_step_fwd).  Only the index-writing selectors can be folded
 * (GCM_SEL_TEMPORAL / GCM_SEL_DENSE) and N % 4 == F % 4 == 0 is required; returns
 * GCM_EUNSUPPORTED otherwise - gcm_dense_step_fwd / gcm_dense_rollout_fwd then use the
 * three-kernel sequence themselves. */
int gcm_dense_step_fused_fwd(const float* obs, const float* nodes_in, const float* adj_in,
                             const int64_t* count_in, float* nodes_out, float* adj_out,
                             int64_t* cur_out, int64_t* count_out,
                             const gcm_selector_desc* selectors, int n_selectors,
                             const float* w_rel1, const float* b_rel1, const float* w_root1,
                             int act1, const float* w_rel2, const float* b_rel2,
                             const float* w_root2, int act2, float* mx, float* h1, float* agg1,
                             float* agg2, uint32_t* flags, int B, int N, int F, int H1, int H2,
                             gcm_stream_t stream);

/* gcm_dense_gnn2_row_bwd + gcm_sum_slabs in one call: g_params [param_count] is overwritten.
 * workspace: B * param_count floats. */
int gcm_dense_step_bwd(const float* g_mx, const float* g_nodes_out, const float* nodes_out,
                       const float* adj_out, const int64_t* cur, const int64_t* count_in,
                       const float* params, int has_bias, int act1, int act2, const float* mx,
                       const float* h1, const float* agg1, const float* agg2, float* g_nodes_in,
                       float* g_obs, float* g_params, void* workspace, size_t workspace_bytes,
                       int B, int N, int F, int H1, int H2, gcm_stream_t stream);

/* gcm_dense_step_bwd with the parameter gradient of the LATER steps folded in:
 * g_params = g_params_prev (NULL = 0) + this step's gradient.  Lets a caller thread the parameter
 * gradient through the chain of step nodes instead of summing T separate [param_count] tensors. */
int gcm_dense_step_bwd_acc(const float* g_mx, const float* g_nodes_out, const float* nodes_out,
                           const float* adj_out, const int64_t* cur, const int64_t* count_in,
                           const float* params, int has_bias, int act1, int act2, const float* mx,
                           const float* h1, const float* agg1, const float* agg2,
                           float* g_nodes_in, float* g_obs, const float* g_params_prev,
                           float* g_params, void* workspace, size_t workspace_bytes, int B, int N,
                           int F, int H1, int H2, gcm_stream_t stream);

/* The step adjoint without the slab sum: slabs [B, param_count] receive (accumulate = 0) or
 * accumulate (accumulate != 0, read-modify-write) the per-graph parameter-gradient slabs of this
 * step.  A caller that owns a time loop keeps one slab array for all T steps and sums it once
 * (gcm_sum_slabs) instead of once per step. */
int gcm_dense_step_bwd_slabs(const float* g_mx, const float* g_nodes_out, const float* nodes_out,
                             const float* adj_out, const int64_t* cur, const int64_t* count_in,
                             const float* params, int has_bias, int act1, int act2, const float* mx,
                             const float* h1, const float* agg1, const float* agg2,
                             float* g_nodes_in, float* g_obs, float* slabs, int accumulate, int B,
                             int N, int F, int H1, int H2, gcm_stream_t stream);

/* ---- the DenseGCM step on the LIVE ROWS, donated or functional state (gcm.py:213-321) ---------
 *
 * DenseGCM keeps row cur of the last GNN layer (gcm.py:314), so layer 1 is needed only on the rows
 * j with adj[cur, j] != 0 and on row cur.  gcm_dense_rows_step_fwd evaluates exactly those rows
 * (exact: the rest is multiplied by zero in the reference) and does the whole step in one kernel:
 * overflow roll (gcm.py:323-355), node insert (:274), the index-writing selectors
 * (GCM_SEL_TEMPORAL / GCM_SEL_DENSE; others: GCM_EUNSUPPORTED), both DenseGraphConv layers, the
 * belief row and the finite flag.
 *
 * State ownership.  nodes_out == nodes_in AND adj_out == adj_in (count_out may equal count_in):
 * the caller DONATES the state; it is advanced in place and only the inserted node, the
 * selector's entries and the count are written - the reference's per-step clones
 * (gcm.py:262,278,286) are not materialised.  Distinct output buffers: functional semantics, the
 * kernel also streams the copy.  Results are identical either way.
 *
 * saved (may be NULL: inference): receives the record the time-parallel backward reads, laid out
 * by gcm_dense_rows_layout (float offsets {total, v, hdr, coef, rows} and the row width); mx must
 * then point at saved (the record starts with the [B,H2] belief states).
 * Shapes: N <= 128, N % 4 == 0, F % 4 == 0, F, H1, H2 <= 64 (gcm_dense_rows_supported).
 *
 * Folded node transforms (gcm.py:290-306: what sits between the selectors and the GNN).  A Linear
 * preprocessor x' = W_p x + b_p, applied by the reference to all N rows every step, commutes with
 * the aggregation: lin_rel1(adj @ x') = (W_rel1 W_p)(adj @ x) + rowsum(adj) * (W_rel1 b_p).  The caller
 * passes W_rel1 W_p / W_root1 W_p / b1 + W_root1 b_p in the layer-1 slots (F = the raw observation width)
 * and, with GCM_GNN_HAS_DEG_TERM set in has_bias, the vector c1 = W_rel1 b_p [H1] directly behind
 * the packed GNN parameters; the kernel adds rowsum(adj[j,:]) * c1 on every live row and records the
 * row sums, and gcm_dense_rows_bptt returns d c1 behind the GNN gradient (param_count + H1 floats).
 * GCM_GNN_HAS_PE_TABLE: a positional-encoding table pe [N, F] follows (behind c1 when both are
 * set); rows <= cur of the node image get pe[row] added before the GNN (PositionalEncoding mode
 * "add", gcm.py:120-131); the stored nodes stay raw. */
#define GCM_GNN_HAS_DEG_TERM 4
#define GCM_GNN_HAS_PE_TABLE 8
/* has_bias bit: the record also keeps what the gradient w.r.t. the observations / incoming nodes needs
 * (gcm_dense_rows_bptt_dx): the live rows' indices and adjacency rows (gcm_dense_rows_layout_dx). */
#define GCM_GNN_RECORD_DX 16
/* gcm_dense_rows_bptt only (a hint, never changes the result's meaning): the records hold MANY live rows per graph
 * (DenseEdge: every row <= cur) - the pass then fetches sixteen rows ahead instead of two (F = H1 = 32, H2 <= 32, no
 * folded terms; ignored elsewhere). */
#define GCM_BPTT_MANY_ROWS 128
/* gcm_dense_rows_step_colcache[_functional] and gcm_learned_step_cached only: the four-wave kernel where the eight-wave
 * form exists (column-write step: F = H1 = 32; LearnedEdge: N = 128, F = H1 = H2 = 32, cur_host >= 0) - the A/B of tests and
 * tools; a per-call argument, the library keeps no switch */
#define GCM_STEP_FOUR_WAVES 256
/* gcm_dense_rows_step_cached[_ws] only: the one-wave kernel where the two-wave form exists (F = H1 = 32, H2 <= 32,
 * cur_host >= 0, GCM_STEP_IMG_V4: a second wave does the state's entries and the record's live list) - A/B */
#define GCM_STEP_ONE_WAVE 512
/* gcm_dense_rows_step_cached_ws only: a distance selector and the cached step as two launches (see
 * gcm_dense_rows_cached_launches) */
#define GCM_STEP_TWO_LAUNCH 32
/* ... the cached temporal-hops step reads the SECOND layout of the weight image: the two matrices of a layer interleaved,
 * image2[layer][k][lane][rel | root] - one 8-byte load per pair, packed products without register moves (the module's
 * default since round 5: cfg2 58.2 M against 55.2 M; clear the bit for the A/B) */
#define GCM_STEP_IMG_V4 64
int gcm_dense_rows_supported(int N, int F, int H1, int H2);
int gcm_dense_rows_layout(int B, int N, int F, int H1, int H2, size_t* out6);
/* with GCM_GNN_RECORD_DX: float offsets {total, v, hdr, coef, rows, row width, live, arows} */
int gcm_dense_rows_layout_dx(int B, int N, int F, int H1, int H2, size_t* out8);
int gcm_dense_rows_step_fwd(const float* obs, const float* nodes_in, const float* adj_in,
                            const int64_t* count_in, float* nodes_out, float* adj_out,
                            int64_t* count_out, int64_t* cur_out /* may be NULL */,
                            const gcm_selector_desc* selectors, int n_selectors,
                            const float* params, int has_bias, int act1, int act2, float* mx,
                            float* saved, uint32_t* flags, int B, int N, int F, int H1, int H2,
                            gcm_stream_t stream);

/* CSC (by source) view of a CSR (by sink) edge list that is grouped by graph - what the backward's
 * transpose gather reads - without a sort (replaces the sort-based coalesce of the reference's
 * autograd path, torch_geometric GraphConv's scatter).  node_off [B+1]: the graphs' node ranges;
 * every edge stays inside its graph (SparseGCM's flat list, sparse_gcm.py:165-170); dst [E]: sink
 * of each CSR entry.  Out: col_ptr [M+1], rows [E] (sink of each CSC entry), perm [E] (its CSR
 * position).  One wave per graph with its source counters in LDS: GCM_EUNSUPPORTED when
 * max_nodes_per_graph > 8192 (the caller sorts instead). */
int gcm_csc_from_csr_batched(const int64_t* row_ptr, const int64_t* col, const int64_t* dst,
                             const int64_t* node_off, int64_t* col_ptr, int64_t* rows,
                             int64_t* perm, int B, int64_t M, int64_t E, int max_nodes_per_graph,
                             gcm_stream_t stream);

/* gcm_dense_rows_step_fwd for selector chains that contain ONE distance selector
 * (GCM_SEL_DISTANCE, not bidirectional; distance.py:18-39): the selector runs first, on the state
 * as it comes in (gcm_edge_distance_pre: the current node is obs[b], stored row j + 1 stands for
 * image row j when the graph is about to roll), and its decisions travel to the step kernel as a
 * row [B, N] inside `workspace` (gcm_dense_rows_step_workspace_bytes) - the adjacency is then written
 * once, by the step kernel.  Without a distance selector workspace may be NULL. */
size_t gcm_dense_rows_step_workspace_bytes(const gcm_selector_desc* selectors, int n_selectors, int B,
                                           int N, int F);
int gcm_dense_rows_step_fwd_ws(const float* obs, const float* nodes_in, const float* adj_in,
                               const int64_t* count_in, float* nodes_out, float* adj_out,
                               int64_t* count_out, int64_t* cur_out /* may be NULL */,
                               const gcm_selector_desc* selectors, int n_selectors,
                               const float* params, int has_bias, int act1, int act2, float* mx,
                               float* saved, uint32_t* flags, void* workspace, size_t workspace_bytes,
                               int B, int N, int F, int H1, int H2, gcm_stream_t stream);
int gcm_edge_distance_pre(const float* nodes_in, const int64_t* count_in, const float* obs,
                          float* sel_row, int mode, float max_distance, const float* dist_param,
                          int a0, int a1, int b0, int b1, void* workspace, size_t workspace_bytes,
                          int B, int N, int F, gcm_stream_t stream);
int gcm_edge_distance_pre_ex(const float* nodes_in, const int64_t* count_in, const float* obs,
                             float* sel_row, int mode, float max_distance, const float* dist_param,
                             int a0, int a1, int b0, int b1, const float* cur_rows, int n_cur_rows,
                             void* workspace, size_t workspace_bytes, int B, int N, int F,
                             gcm_stream_t stream);

/* Parameter gradient of n_steps recorded steps in one pass (time-parallel BPTT; valid when neither
 * the observations nor the incoming node matrix need a gradient, so step t's adjoint depends on
 * g_mx[t] and its own record only).  saved_host / gmx_host: HOST arrays of n_steps DEVICE pointers
 * (each step's record and its g_mx [B,H2], element strides gmx_stride_b / gmx_stride_h; 0 for an
 * expanded gradient).  g_params [param_count] = g_params_prev (NULL = 0) + gradient, laid out like
 * the packed parameter vector.  workspace: gcm_dense_rows_bptt_workspace_bytes. */
int gcm_dense_rows_bptt_slabs(int n_steps, int B);
size_t gcm_dense_rows_bptt_workspace_bytes(int n_steps, int B, int F, int H1, int H2);
int gcm_dense_rows_bptt(const float* const* saved_host, const float* const* gmx_host, int n_steps,
                        long gmx_stride_b, long gmx_stride_h, const float* params, int has_bias,
                        int act1, int act2, const float* g_params_prev, float* g_params,
                        void* workspace, size_t workspace_bytes, int B, int N, int F, int H1,
                        int H2, gcm_stream_t stream);

/* Records written with GCM_GNN_RECORD_DX, when the observations (and / or the node matrix the chain started
 * from) need a gradient too - the reference's own tests require gradients to reach the inputs
 * (tests/test_gcm.py:355-365).  x enters a step through layer 1 only, so per step the gradient w.r.t. the node
 * matrix is a handful of rows; row k of step s belongs to the node inserted at step s - (cur - k) of the chain.
 * The steps of a chain are handed over LAST TO FIRST, one launch each (one wave per graph), which adds those
 * rows straight into gx [T, B, F] (slot t: gradient w.r.t. the observation of step t, complete once step t
 * itself has run) and gn0 [B, N, F] (the initial nodes; may be NULL) - both zeroed by the caller before the
 * chain's first launch.  g_mx (strided) / g_nodes_out ([B,N,F], the gradient handed to the node matrix this
 * step returned) may be NULL.  The PARAMETER gradient of those steps is gcm_dense_rows_bptt over the same
 * records (their dx sections sit behind the others).  F <= 64, H1, H2 <= 32. */
int gcm_dense_rows_dx_supported(int N, int F, int H1, int H2);
int gcm_dense_rows_bptt_dx_step(const float* saved, const float* g_mx, long gmx_stride_b, long gmx_stride_h,
                                const float* g_nodes_out, const float* params, int has_bias, int act1, int act2,
                                const int64_t* count0, float* gx, float* gn0, int s_lin, int B, int N, int F,
                                int H1, int H2, gcm_stream_t stream);
/* The same for up to 128 consecutive steps of a chain in ONE launch (one wave per step and graph): saved /
 * g_mx / g_nodes_out are HOST arrays of n_steps device pointers (g_mx[s], g_nodes_out[s] NULL: none; every
 * g_mx with the same element strides), s0 the chain index of the first.  The steps run concurrently: gx / gn0
 * (zeroed by the caller) take hardware float atomic adds, so the order of summation - the last bits of the
 * result - is not fixed; gcm_dense_rows_bptt_dx_step is the ordered form. */
int gcm_dense_rows_bptt_dx_all(const float* const* saved, const float* const* g_mx, long gmx_stride_b,
                               long gmx_stride_h, const float* const* g_nodes_out, int n_steps, int s0,
                               const float* params, int has_bias, int act1, int act2, const int64_t* count0,
                               float* gx, float* gn0, int B, int N, int F, int H1, int H2, gcm_stream_t stream);

/* ---- cached live-row steps (round 3) --------------------------------------------------------------------
 * The DenseGCM step (gcm.py:262-321) for a chain of hidden states that started from EMPTY graphs, has made
 * fewer than N steps (nothing can have overflowed) and whose selectors only write row cur of the adjacency
 * (TemporalBackedge direction "forward", temporal.py:72-88; not DenseEdge, which also writes column cur).  There the rows of layer 1
 * are final once written, so the chain keeps h1 [B,N,H1], agg1 [B,N,F] and the node matrix [B,N,F] of every
 * node in caches (any contents at the chain's head: a row is read only behind the step that wrote it) and a step evaluates row cur alone, every input
 * fetched in one round trip behind the count.  The state (nodes, adj, count) is advanced IN PLACE.
 * saved: the step's record, gcm_dense_rows_cached_layout floats {total, v, hdr, coef, live} (mx [B,H2] at 0 - always
 * written; the rest only with record != 0).  cur_host >= 0: the row every graph's new node lands in, when the host
 * knows it (a chain from empty graphs: the number of steps made so far) - the kernel then does not wait for the
 * count; -1: read it.  weight_image (may be NULL): 2 x [4][64][64] + 4096 floats from gcm_dense_rows_cached_weight_image,
 * the four weight matrices lane-major [m][k][lane], behind them layer-interleaved [layer][k][lane][2] (GCM_STEP_IMG_V4)
 * and - used at F = H1 = 32, H2 <= 32 - the same split over the half-waves [layer][k % 16][(k / 16) * 32 + h][2]
 * (made once per chain: the parameters are fixed inside one) - with it a step is one
 * wave per graph whose every load, weights included, is issued at kernel start (no LDS staging, no barrier).
 * Widths: F, H1, H2 <= 64, N <= 128; 32 and 64 are compile-time widths, anything else runs padded to them and needs
 * the weight image (GCM_EUNSUPPORTED without it).  gcm_dense_rows_bptt_cached: gcm_dense_rows_bptt over such records. */
int gcm_dense_rows_cached_supported(const gcm_selector_desc* selectors, int n_selectors, int has_bias, int N,
                                    int F, int H1, int H2);
int gcm_dense_rows_cached_layout(int B, int N, int F, int H1, int H2, size_t* out5);
size_t gcm_dense_rows_cached_weight_image_floats(void);   /* the floats `image` must hold (36864) */
int gcm_dense_rows_cached_weight_image(const float* params, float* image, int F, int H1, int H2,
                                       gcm_stream_t stream);
int gcm_dense_rows_step_cached(const float* obs, float* nodes, float* adj, int64_t* count,
                               const gcm_selector_desc* selectors, int n_selectors, const float* params,
                               const float* weight_image, int has_bias, int act1, int act2, float* cache_h1, float* cache_agg1,
                               float* cache_nodes, float* saved, int record, int cur_host, uint32_t* flags, int B,
                               int N, int F, int H1, int H2, gcm_stream_t stream);
/* The cached step for selector chains that also hold ONE distance selector (GCM_SEL_DISTANCE, not bidirectional;
 * distance.py:18-39 - it writes row cur alone, so the rows of layer 1 stay final): the selector runs first, on the
 * state as it comes in (gcm_edge_distance_pre), its decisions reach the step as a row [B, N] at the head of
 * `workspace` (gcm_dense_rows_step_workspace_bytes), the selected rows are gathered eight per round trip.
 * weight_image is required on this path.  Without a distance selector workspace may be NULL. */
int gcm_dense_rows_cached_supported_ws(const gcm_selector_desc* selectors, int n_selectors, int has_bias, int N,
                                       int F, int H1, int H2);
int gcm_dense_rows_step_cached_ws(const float* obs, float* nodes, float* adj, int64_t* count,
                                  const gcm_selector_desc* selectors, int n_selectors, const float* params,
                                  const float* weight_image, int has_bias, int act1, int act2, float* cache_h1,
                                  float* cache_agg1, float* cache_nodes, float* saved, int record, int cur_host,
                                  uint32_t* flags, void* workspace, size_t workspace_bytes, int B, int N, int F,
                                  int H1, int H2, gcm_stream_t stream);
/* EuclideanEdge (distance.py:41-49, not bidirectional) as the ONLY selector of such a chain: the distance kernel and
 * the cached step as one launch (the step is the tail of the matrix-core distance kernel's first wave) -
 * gcm_dense_rows_step_cached_ws takes this path by itself when the shapes allow (>= 32 current rows, F in {32, 64},
 * N <= 128, H1, H2 <= 32); GCM_EUNSUPPORTED otherwise.  lay5: gcm_dense_rows_cached_layout.  cur_host >= 0: the row
 * every graph's new node lands in, when the host knows it (a chain from empty graphs: its step count) - the node rows
 * are then fetched without waiting for the count; -1: read it. */
/* The cached step in the STEADY STATE of such a chain (t_abs >= N steps made: every graph is full and every step
 * drops the oldest node, gcm.py:263-271, 323-355), selectors = forward temporal hops only, N > 2 max(hop).  There
 * the band adjacency is a fixed point of the overflow roll (the step leaves adj and count untouched - the values
 * the reference arrives at by rewriting all of it), the caches are rings (node t in slot t mod N), and the node
 * matrix is rolled in place by a second wave of the graph's workgroup.  saved: the GENERAL live-row record
 * (gcm_dense_rows_layout; written in full with record != 0, else mx [B,H2] only) - gcm_dense_rows_bptt reads it.
 * weight_image: gcm_dense_rows_cached_weight_image.  GCM_EUNSUPPORTED when the configuration has no such form
 * (gcm_dense_rows_cached_roll_supported). */
int gcm_dense_rows_cached_roll_supported(const gcm_selector_desc* selectors, int n_selectors, int has_bias, int N,
                                         int F, int H1, int H2);
int gcm_dense_rows_step_cached_roll(const float* obs, float* nodes, const gcm_selector_desc* selectors, int n_selectors,
                                    const float* params, const float* weight_image, int has_bias, int act1, int act2,
                                    float* cache_h1, float* cache_agg1, float* cache_nodes, float* saved, int record,
                                    int t_abs, uint32_t* flags, int B, int N, int F, int H1, int H2,
                                    gcm_stream_t stream);
/* The cached step for selector chains that also write COLUMN cur of the adjacency (round 6): DenseEdge
 * (edge_selectors/dense.py:16-21) and TemporalBackedge with direction "backward" / "both"
 * (edge_selectors/temporal.py:82-87) - alone or chained with forward hops.  Same preconditions as
 * gcm_dense_rows_step_cached (a chain from EMPTY graphs on a donated state, fewer than N steps made), plus cur_host >= 0
 * is required (every graph of such a chain holds cur_host nodes).  A column write is a rank-1 correction of the older
 * rows' layer-1 aggregate (agg1[j] += x[cur]), so the chain keeps cache_agg1 [B,N,F] (agg1 of every node) and
 * cache_root [B,N/4,H1,4] (W_root1 x[j] + b1 in quads of rows, final once written; chain-internal layout) - both
 * uninitialised at the chain's head - and a step is
 * the masked column sum for the new row, the rank-1 update, ONE [rows x F] . [F x H1] product of the live rows on the
 * fp32 matrix cores and layer 2 on row cur: cur F + 2 cur F H1 flops per graph where the general live-row kernel
 * re-aggregates cur^2 F.  The state (nodes, adj, count) is advanced IN PLACE; a graph whose count differs from cur_host
 * is left untouched and raises GCM_FLAG_BAD_COUNT.  saved: the GENERAL live-row record (gcm_dense_rows_layout; mx [B,H2]
 * at 0 - always written; the rest with record != 0), read by gcm_dense_rows_bptt.  F, H1 in {32, 64}, H2 <= 64,
 * N <= 128, has_bias without fold bits; _supported also returns 0 for chains that never write a column (they have
 * gcm_dense_rows_step_cached). */
int gcm_dense_rows_colcache_supported(const gcm_selector_desc* selectors, int n_selectors, int has_bias, int N,
                                      int F, int H1, int H2);
/* ... with FUNCTIONAL state (the reference's default, gcm.py:262,278,286: the inputs are never modified): the new state
 * (copy, overflow roll, the selectors' entries, the observation, the count) is written to the *_out buffers by extra
 * workgroups of the same launch, the chain's caches stay valid as long as the caller hands the state returned last back
 * in (a linear chain - the host checks it by tensor identity). */
int gcm_dense_rows_step_colcache_functional(const float* obs, const float* nodes_in, const float* adj_in,
                                            const int64_t* count_in, float* nodes_out, float* adj_out,
                                            int64_t* count_out, const gcm_selector_desc* selectors, int n_selectors,
                                            const float* params, int has_bias, int act1, int act2, float* cache_agg1,
                                            float* cache_root, float* saved, int record, int cur_host, uint32_t* flags,
                                            int B, int N, int F, int H1, int H2, gcm_stream_t stream);
int gcm_dense_rows_step_colcache(const float* obs, float* nodes, float* adj, int64_t* count,
                                 const gcm_selector_desc* selectors, int n_selectors, const float* params,
                                 int has_bias, int act1, int act2, float* cache_agg1, float* cache_root, float* saved,
                                 int record, int cur_host, uint32_t* flags, int B, int N, int F, int H1, int H2,
                                 gcm_stream_t stream);
int gcm_edge_distance_step_cached_supported(int n_cur_rows, int B, int N, int F, int H1, int H2);
/* -> the number of kernel launches ONE call of gcm_dense_rows_step_cached_ws makes for these arguments (0: not
 * supported; 1: forward temporal hops, or EuclideanEdge alone in the one-launch form; 2: a distance selector's kernel,
 * then the step).  GCM_STEP_TWO_LAUNCH in has_bias asks for the two-launch form where the one-launch form exists
 * (the A/B of tests and tools; a per-call argument - the library holds no switch). */
int gcm_dense_rows_cached_launches(const gcm_selector_desc* selectors, int n_selectors, int has_bias, int B, int N,
                                   int F, int H1, int H2);
int gcm_edge_distance_step_cached(const float* obs, float* nodes, float* adj, int64_t* count, float max_distance,
                                  const float* dist_param, const float* cur_rows, int n_cur_rows, const float* params,
                                  const float* weight_image, int act1, int act2, float* cache_h1, float* cache_agg1,
                                  float* cache_nodes, float* saved, const size_t* lay5, int record, int cur_host,
                                  uint32_t* adj_bits, uint32_t* flags, int B, int N, int F, int H1, int H2,
                                  gcm_stream_t stream);
/* ... and the same chain past graph_size steps (round 5) - the STEADY STATE of a long rollout: every graph holds N nodes
 * (the caller guarantees count[b] == N: a chain from empty graphs that has made >= N steps) and every step drops the
 * oldest one (gcm.py:263-271, 323-355), so the layer-1 rows of older nodes are no longer final.  Still ONE launch: the
 * distances on the matrix cores, the overflow roll of the donated state IN PLACE (nodes, adj; count stays N), the live
 * rows' layer 1 re-evaluated from the node image staged for the distances and from adj_bits - the chain's adjacency
 * as bits, [B][N][4] uint32 (row i: bit j = adj[i, j]), zero at the chain's head, kept current by
 * gcm_edge_distance_step_cached(adj_bits != NULL) and rolled here -, row cur, the belief.  saved: the GENERAL live-row
 * record (gcm_dense_rows_layout: lay6 = {total, v, hdr, coef, rows, rw}; mx [B,H2] at 0; in full with record != 0) that
 * gcm_dense_rows_bptt reads.  No `learned` divisor, the local batch's own current rows, B >= 32, F in {32, 64},
 * N <= 128, N % 4 == 0, H1, H2 <= 32. */
int gcm_edge_distance_step_ring_supported(int B, int N, int F, int H1, int H2);
int gcm_edge_distance_step_ring(const float* obs, float* nodes, float* adj, int64_t* count, float max_distance,
                                const float* params, const float* weight_image, int act1, int act2, uint32_t* adj_bits,
                                float* saved, const size_t* lay6, int record, uint32_t* flags, int B, int N, int F,
                                int H1, int H2, gcm_stream_t stream);
/* SparseGCM in stepwise use (sparse_gcm.py:72-212 called with x [B, 1, F], taus in {0, 1}) with a TemporalEdge selector
 * (sparse_edge_selectors/temporal.py:18-63; hops_host: HOST array, every hop >= 1), in a chain from empty graphs: the
 * new node's belief from the chain's caches (the layer-1 row of a node is final once written: its edges point at
 * older nodes only) - instead of flattening the batch, building CSR / CSC views and running both GraphConv layers
 * over every stored node.  T: node counts BEFORE the step; taus[b] = 0: no node (zero output row).  mx [B, H2]; saved:
 * the record (gcm_dense_rows_cached_layout; in full with record != 0), read by gcm_dense_rows_bptt_cached.  The state
 * itself (node matrix, COO adjacency, T) is advanced by the caller with the usual entry points. */
int gcm_sparse_step_cached(const float* x, const int64_t* T, const int64_t* taus, const int32_t* hops_host, int n_hops,
                           const float* params, const float* weight_image, int act1, int act2, float* cache_h1,
                           float* cache_agg1, float* cache_nodes, float* mx, float* saved, int record,
                           uint32_t* flags, int B, int N, int F, int H1, int H2, gcm_stream_t stream);
/* gcm_dense_rows_bptt_dx_all over the records of cached steps (a chain from empty graphs: s0 is also the row the
 * first of these steps' nodes landed in; live rows and their adjacency rows from the selectors' forward hops, their
 * h1 rows from the chain's cache). */
int gcm_dense_rows_bptt_dx_all_cached(const float* const* saved, const float* const* g_mx, long gmx_stride_b,
                                      long gmx_stride_h, int n_steps, int s0, const float* params, int has_bias,
                                      int act1, int act2, const gcm_selector_desc* selectors, int n_selectors,
                                      const float* cache_h1, float* gx, int B, int N, int F, int H1, int H2,
                                      gcm_stream_t stream);
/* gcm_dense_rows_bptt over the records of cached steps (their rows in the chain's caches).  Round 6: at F = 32 or 64, H1 = 32, H2 <= 32,
 * N <= 128 and n_steps <= 128 the pass runs per GRAPH (k_bptt_cached_graph: the caches' rows in LDS once, one slab per graph)
 * unless has_bias carries a bit other than the two bias bits and GCM_BPTT_MANY_ROWS - GCM_STEP_FOUR_WAVES is the A/B switch
 * for the per-item kernel.  Same workspace (gcm_dense_rows_bptt_workspace_bytes). */
int gcm_dense_rows_bptt_cached(const float* const* saved_host, const float* const* gmx_host, int n_steps,
                               long gmx_stride_b, long gmx_stride_h, const float* params, int has_bias, int act1,
                               int act2, const float* cache_nodes, const float* cache_h1,
                               const float* cache_agg1, const float* g_params_prev, float* g_params,
                               void* workspace, size_t workspace_bytes, int B, int N, int F, int H1, int H2,
                               gcm_stream_t stream);

/* Parameter gradient of a rollout from the history gcm_dense_rollout_fwd /
 * gcm_dense_rollout_persistent_fwd kept, when neither the observations nor the initial node matrix
 * need a gradient: all T*B graph-steps in ONE launch over the live rows (no reverse scan, no Q array).
 * g_mx_all: [T,B,H2] with element strides (0 for an expanded gradient). */
size_t gcm_dense_rollout_bwd_params_workspace_bytes(int T, int B, int F, int H1, int H2);
int gcm_dense_rollout_bwd_params(const float* g_mx_all, long gmx_stride_t, long gmx_stride_b,
                                 long gmx_stride_h, const float* nodes_all, const float* adj_all,
                                 const int64_t* cur_all, const float* params, int act1, int act2,
                                 const float* mx_all, const float* h1_all, const float* agg1_all,
                                 const float* agg2_all, float* g_params, void* workspace,
                                 size_t workspace_bytes, int T, int B, int N, int F, int H1, int H2,
                                 gcm_stream_t stream);

/* ---- DenseGCM + LearnedEdge, fused per graph (edge_selectors/learned.py:38-113) --------------
 *
 * The default edge network (learned.py:38-51: Linear(2F,F)-ReLU-LayerNorm-Linear(F,F)-ReLU-LayerNorm-
 * Linear(F,1)) as ONE kernel per direction.  Packed edge-network parameter vector (and gradient
 * slab) layout: W0 [F,2F] | b0 [F] | ln0.weight [F] | ln0.bias [F] | W1 [F,F] | b1 [F] | ln1.weight
 * [F] | ln1.bias [F] | w2 [F] | b2 [1]  (gcm_learned_mlp_param_count floats).
 * Shapes: N <= 128, F, H1, H2 <= 32 (gcm_learned_step_supported). */
int gcm_learned_step_supported(int N, int F, int H1, int H2);
size_t gcm_learned_mlp_param_count(int F);

/* learned.py:53-113 on an already advanced state: logits of all candidate pairs (cur, j), gumbel
 * softmax over j < cur with the given noise (gumbel draws, or exponential draws e with
 * noise_is_exp: g = -log e), straight-through threshold at `cutoff`, adjacency row cur rewritten IN
 * PLACE; soft [B,N] receives the softmax (what the backward needs). */
int gcm_learned_select_fused(const float* nodes, float* adj, const int64_t* cur_idx,
                             const float* noise, int noise_is_exp, const float* mlp_params,
                             float eps0, float eps1, float cutoff, float* soft, int B, int N, int F,
                             gcm_stream_t stream);

/* gcm_state_advance_fwd + gcm_learned_select_fused in ONE kernel (gcm.py:262-278 + learned.py:53-113):
 * the state copy, with the overflow roll folded in, travels through the registers of the workgroup that
 * runs the edge network on the same node rows; the observation lands in row cur, cur / count come out.
 * N % 4 == 0 and F % 4 == 0 (GCM_EUNSUPPORTED otherwise: call the two entry points one after the other). */
int gcm_learned_advance_select_fused(const float* obs, const float* nodes_in, const float* adj_in,
                                     const int64_t* count_in, const float* noise, int noise_is_exp,
                                     const float* mlp_params, float eps0, float eps1, float cutoff,
                                     float* nodes_out, float* adj_out, int64_t* cur_out,
                                     int64_t* count_out, float* soft, uint32_t* flags, int B, int N, int F,
                                     gcm_stream_t stream);

/* Backward of one DenseGCM + LearnedEdge step when the observations carry no gradient: GNN adjoint on
 * the live rows, the adjacency gradient in compact form, selection adjoint, edge-network adjoint
 * (forward recomputed).  nodes / adj: the step's OUTPUT state; h1 / agg1 [B,N,.], agg2, mx as saved
 * by gcm_dense_gnn2_row_fwd; soft from gcm_learned_select_fused; count_in = num_nodes before the
 * step.  GA [B,N,N] is the chain buffer of the adjacency gradient: in, as left by the step after this
 * one (zeros for the last step of a chain); out, for the step before (this step's live rows added,
 * row cur consumed, the state advance undone).  slabs
 * [B, gnn2_param_count + mlp_param_count]: per-graph parameter gradients, overwritten or
 * (accumulate != 0) added to; the caller sums them once (gcm_sum_slabs). */
int gcm_learned_step_bwd(const float* g_mx, const float* nodes, const float* adj,
                         const int64_t* cur_idx, const int64_t* count_in, const float* gnn_params,
                         int act1, int act2, const float* mx, const float* h1, const float* agg1,
                         const float* agg2, const float* soft, const float* mlp_params, float eps0,
                         float eps1, float* GA, float* slabs, int accumulate, int B, int N, int F,
                         int H1, int H2, gcm_stream_t stream);

/* ---- time-batched rollout (caller loop ray_gcm.py:200-202) --------- */


/* T DenseGCM steps in one call: for t in [0,T): state advance, the selector chain, fused GNN,
 * exactly what T calls of DenseGCM.forward do.  State arrays hold every step (needed by BPTT):
 * nodes_all [T+1,B,N,F], adj_all [T+1,B,N,N], count_all [T+1,B] with slot 0 = the incoming
 * hidden state (caller fills it); cur_all [T,B]; mx_all [T,B,H2]; h1_all [T,B,N,H1];
 * agg1_all [T,B,N,F]; agg2_all [T,B,H1] (the last three may be NULL when no backward follows).
 * workspace: gcm_edge_distance_workspace_bytes of the largest distance selector (or 0). */
int gcm_dense_rollout_fwd(const float* obs, float* nodes_all, float* adj_all, int64_t* count_all,
                          int64_t* cur_all, const gcm_selector_desc* selectors, int n_selectors,
                          const float* w_rel1, const float* b_rel1, const float* w_root1, int act1,
                          const float* w_rel2, const float* b_rel2, const float* w_root2, int act2,
                          float* mx_all, float* h1_all, float* agg1_all, float* agg2_all,
                          uint32_t* flags, void* workspace, size_t workspace_bytes, int T, int B,
                          int N, int F, int H1, int H2, gcm_stream_t stream);

/* The same T steps as ONE persistent launch: a workgroup per graph keeps the graph's adjacency and
 * node matrix in LDS for the whole rollout (no per-step state reads from HBM).  Index-writing
 * selectors only (temporal / dense) and N, F, H1, H2 multiples of 32 within the fused limits;
 * GCM_EUNSUPPORTED otherwise (gcm_dense_rollout_fwd tries this first and falls back itself).
 * history != 0: arrays as for gcm_dense_rollout_fwd, but of the intermediate slots 1..T-1 (and of
 *   h1_all / agg1_all) only the 32-row tiles BPTT reads are written - the tiles holding row cur and
 *   the non-zeros of adj[cur,:]; slot T (the returned hidden state) is complete.
 * history == 0 (inference): nodes_all / adj_all / count_all have TWO slots, 0 = incoming state,
 *   1 = state after T steps; cur_all, h1_all, agg1_all, agg2_all may be NULL. */
int gcm_dense_rollout_persistent_fwd(const float* obs, float* nodes_all, float* adj_all,
                                     int64_t* count_all, int64_t* cur_all,
                                     const gcm_selector_desc* selectors, int n_selectors,
                                     const float* w_rel1, const float* b_rel1, const float* w_root1,
                                     int act1, const float* w_rel2, const float* b_rel2,
                                     const float* w_root2, int act2, float* mx_all, float* h1_all,
                                     float* agg1_all, float* agg2_all, uint32_t* flags,
                                     int history, int T, int B, int N, int F, int H1, int H2,
                                     gcm_stream_t stream);

/* Reverse scan of the node gradient through the state advance (adjoint of gcm.py:262-278 /
 * 323-355), the sequential part of time-parallel BPTT:
 *   C_t = U_t(C_{t+1}) + Q_all[t],  g_obs_all[t] = C_{t+1}[cur_t] + pobs_all[t],  C_T = g_nodes_T
 * with Q_all [T,B,N,F] / pobs_all [T,B,F] = g_nodes_in / g_obs of gcm_dense_gnn2_row_bwd run over
 * all T*B graph-steps with no incoming node gradient.  N*F <= 8192. */
int gcm_dense_gnodes_scan(const float* Q_all, const float* pobs_all, const float* g_nodes_T,
                          const int64_t* cur_all, const int64_t* count_all, float* g_obs_all,
                          float* g_nodes_0, int T, int B, int N, int F, gcm_stream_t stream);

/* Time-parallel GNN adjoint over `items` = T*B independent graph-steps (arrays as for
 * gcm_dense_gnn2_row_bwd with B = items and no incoming node gradient): a persistent grid of
 * n_slabs workgroups (gcm_dense_bptt_batched_slabs(items)) walks the items, touching only the
 * 32-row tiles that can carry gradient (the tiles holding row cur and the non-zeros of
 * adj[cur,:]) and writing ONE parameter-gradient slab per workgroup: slabs [n_slabs, param_count].
 * g_nodes_out [items,N,F] = gradient w.r.t. each item's nodes from elsewhere (NULL = 0: the
 * time-parallel schedule adds it in the scan; the per-step backward passes the next step's).
 * Q [items,N,F] / pobs [items,F] feed gcm_dense_gnodes_scan.  N, F, H1, H2 multiples of 32 within
 * the fused limits, else GCM_EUNSUPPORTED. */
int gcm_dense_bptt_batched_slabs(int items);
int gcm_dense_bptt_batched(const float* g_mx, const float* g_nodes_out, const float* x,
                           const float* adj,
                           const int64_t* cur_idx, const int64_t* num_nodes_in,
                           const float* w_rel1, const float* b_rel1, const float* w_root1, int act1,
                           const float* w_rel2, const float* b_rel2, const float* w_root2, int act2,
                           const float* mx, const float* h1, const float* agg1, const float* agg2,
                           float* Q, float* pobs, float* slabs, int n_slabs, int items, int N,
                           int F, int H1, int H2, gcm_stream_t stream);

/* BPTT over the arrays written by gcm_dense_rollout_fwd.  g_mx_all [T,B,H2]; g_nodes_T
 * [B,N,F] = gradient w.r.t. the final nodes (NULL = 0).  Outputs: g_obs_all [T,B,F],
 * g_nodes_0 [B,N,F], g_params [gcm_dense_gnn2_param_count] (summed over graphs and steps).
 * workspace: 2*B*N*F floats (ping-pong) + B*param_count floats (slabs). */
size_t gcm_dense_rollout_bwd_workspace_bytes(int B, int N, int F, int H1, int H2);
/* A workspace of at least this size selects the time-parallel schedule: one gnn2_row_bwd launch
 * over T*B graph-steps, gcm_dense_gnodes_scan, one slab sum (T*B*(N*F + F + param_count) floats). */
size_t gcm_dense_rollout_bwd_batched_workspace_bytes(int T, int B, int N, int F, int H1, int H2);
int gcm_dense_rollout_bwd(const float* g_mx_all, const float* g_nodes_T, const float* nodes_all,
                          const float* adj_all, const int64_t* count_all, const int64_t* cur_all,
                          const float* w_rel1, const float* b_rel1, const float* w_root1, int act1,
                          const float* w_rel2, const float* b_rel2, const float* w_root2, int act2,
                          const float* mx_all, const float* h1_all, const float* agg1_all,
                          const float* agg2_all, float* g_obs_all, float* g_nodes_0,
                          float* g_params, void* workspace, size_t workspace_bytes, int T, int B,
                          int N, int F, int H1, int H2, gcm_stream_t stream);

/* ---- DenseGCM + LearnedEdge: the backward of a whole chain of steps, time-parallel ----------------------
 *
 * With observations that carry no gradient the steps are coupled only through the adjacency, and the
 * gradient w.r.t. the entries a step wrote collapses to one vector per inserted node,
 *     D_t = sum over later steps t' in which that node is a live row of dAgg1_t'[its row],
 *     g_sel_t[j] = dagg2_t . h1_t[j] + D_t . x[j]          (learned.py:96-110, both STEs identities)
 * so every graph-step is independent given two passes: A) GNN adjoint on the live rows (parameter
 * gradient, dagg2, dAgg1 per live row), B) D_t by a fixed-order scan of the <= N later steps of the graph,
 * selection adjoint, edge network recomputed and differentiated.  No [B,N,N] gradient tensor exists.
 * gcm_learned_step_layout: float offsets {total, adj, mx, h1, agg1, agg2, idx (cur | count_out, int64), soft} of
 * the buffer one forward step keeps (nodes at 0; what gcm_learned_advance_select_fused +
 * gcm_dense_gnn2_row_fwd write); compact: the `adj` section is row cur only, [B, N].  saved_host / gmx_host: HOST arrays of n_steps device pointers, the steps
 * of ONE chain of hidden states in order; gmx_host[t] == NULL: zero gradient.  params / g_params: GNN
 * (gcm_dense_gnn2_param_count) | edge network (gcm_learned_mlp_param_count), g_params = g_params_prev
 * (NULL = 0) + gradient. */
/* ---- time-parallel DenseGCM rollout (round 4) -------------------------------------------------------------
 * DenseGCM.rollout(obs [T,B,F]) from EMPTY graphs, selectors = forward temporal hops only (temporal.py:72-88),
 * observations without gradient, canonical two-layer GNN: no recurrence at all (node t is observation t, the band
 * adjacency is closed form and a fixed point of the overflow roll), so the whole forward is TWO launches - layer 1 of
 * every (step, graph) on the matrix cores into the caches [B, Tc, .] (Tc >= T rows per graph: cache_h1 [B,Tc,H1],
 * cache_agg1 / cache_nodes [B,Tc,F]) plus the final state (nodes [B,N,F]: zero on entry when T < N; adj [B,N,N]: ZERO on
 * entry; count [B]), then layer 2 -> mx_all [T,B,H2] and the T step records (gcm_dense_rows_cached_layout(B, Tc, ...)
 * - N := Tc -, rec_stride floats apart; record = 0: mx only) that gcm_dense_rows_bptt_cached(..., N := Tc) reads.
 * T > N needs N > 2 max(hop) (a live row must not have lost a source to the roll); F, H1 in {32, 64}, H2 <= 64. */
int gcm_dense_rollout_tp_supported(const gcm_selector_desc* selectors, int n_selectors, int has_bias, int T, int N,
                                   int F, int H1, int H2);
int gcm_dense_rollout_tp_fwd(const float* obs, const gcm_selector_desc* selectors, int n_selectors, const float* params,
                             int has_bias, int act1, int act2, float* nodes, float* adj, int64_t* count,
                             float* cache_h1, float* cache_agg1, float* cache_nodes, float* records, size_t rec_stride,
                             int record, float* mx_all, uint32_t* flags, int T, int B, int N, int Tc, int F, int H1,
                             int H2, gcm_stream_t stream);

/* ---- time-parallel DenseGCM rollout with EuclideanEdge (round 5; csrc/euclid_tp.hip) ----------------------------
 * DenseGCM.rollout(obs [T,B,F]) from EMPTY graphs with EuclideanEdge (edge_selectors/distance.py:18-49: the mean over
 * the B graphs' current nodes of the distance to a stored node, threshold max_distance; dist_param: the `learned`
 * divisor or NULL; not bidirectional) as the ONLY selector, observations without gradient: node j is observation j
 * and the current rows of step t are obs[t], so the T selections of a graph are one causal [N x F].[F x (T B)]
 * contraction on the fp32 MFMA instead of T dependent launches (replaces the loop of gcm.py:262-321 that
 * ray_gcm.py:186-209 drives).  Same arithmetic, value for value, as the per-step kernel behind gcm_edge_distance_pre /
 * gcm_edge_distance_step_cached: the same distances, hence the same decisions as T single steps.
 * _decide: decbits [T, B, 4] uint32 - bit s of step t = ring slot s selected, node n sitting in slot n mod N (for
 *   t < N: slot = node = adjacency column; at t >= N the slot t mod N is the node the roll of gcm.py:323-355 drops, never
 *   selected).  Any T >= 1.  B >= 32 (below: the per-step VALU form), N <= 128, N % 4 == 0, F <= 64, F % 4 == 0.
 * _fwd: T <= N.  Decisions, then the GNN of all T steps in one launch per graph (rows of layer 1 are final once
 *   written): caches [B, Tc, .] (Tc >= T), the T step records (gcm_dense_rows_cached_layout(B, Tc, ...), rec_stride
 *   floats apart; record = 0: mx only) for gcm_dense_rows_bptt_cached(..., N := Tc), mx_all [T,B,H2], and the state after
 *   the rollout written whole (nodes [B,N,F], adj [B,N,N], count [B] <- T; no zero-fill needed).  H1, H2 <= 64. */
int gcm_euclid_rollout_tp_supported(int T, int B, int N, int F, int H1, int H2);
int gcm_euclid_rollout_tp_decide(const float* obs, float max_distance, const float* dist_param, uint32_t* decbits,
                                 int T, int B, int N, int F, gcm_stream_t stream);
int gcm_euclid_rollout_tp_fwd(const float* obs, float max_distance, const float* dist_param, const float* params,
                              int act1, int act2, float* nodes, float* adj, int64_t* count, float* cache_h1,
                              float* cache_agg1, float* cache_nodes, float* records, size_t rec_stride, int record,
                              float* mx_all, uint32_t* decbits, uint32_t* flags, int T, int B, int N, int Tc, int F,
                              int H1, int H2, gcm_stream_t stream);

/* DenseGCM.rollout with LearnedEdge: the whole forward of T <= N steps from EMPTY graphs, observations without
 * gradient, in THREE launches - the selection of step t (learned.py:53-113) depends on raw observations and the given
 * gumbel draws only, so every (graph, step) is independent work: the edge network's logits per 32-row block that
 * holds a candidate row (one block per wave), then one wave per (graph, step) for gumbel-softmax, threshold, adjacency
 * row and layer 1 of the GNN on row cur into the caches, and the belief states in a third launch.
 * obs [T,B,F], noise [T,B,N]; nodes [B,N,F] / adj [B,N,N]: the state AFTER the rollout, ZERO on entry; count [B] <- T;
 * records: T step records (gcm_learned_step_layout, compact = 2) rec_stride floats apart; caches [B,N,.]; mx_all
 * [T,B,H2].  Backward: gcm_learned_bptt_cached over those records (n_cached = T, cached_layout = 2). */
int gcm_learned_rollout_fwd(const float* obs, const float* noise, int noise_is_exp, const float* params, int has_bias,
                            int act1, int act2, float eps0, float eps1, float cutoff, float* nodes, float* adj,
                            int64_t* count, float* records, size_t rec_stride, float* cache_h1, float* cache_agg1,
                            float* cache_nodes, float* mx_all, uint32_t* flags, int T, int B, int N, int F, int H1,
                            int H2, gcm_stream_t stream);
int gcm_learned_step_layout(int B, int N, int F, int H1, int H2, int compact, size_t* out8);
size_t gcm_learned_bptt_workspace_bytes(int n_steps, int B, int N, int F, int H1, int H2);
int gcm_learned_bptt(const float* const* saved_host, const float* const* gmx_host, int n_steps,
                     long gmx_stride_b, long gmx_stride_h, const float* params, int act1, int act2,
                     float eps0, float eps1, int compact, const float* g_params_prev, float* g_params,
                     void* workspace, size_t workspace_bytes, int B, int N, int F, int H1, int H2,
                     gcm_stream_t stream);
/* The whole forward step of such a chain PAST graph_size steps (round 5): the steady state - every graph holds N nodes
 * (count_in[b] == N, the caller's guarantee: a chain from empty graphs that has made >= N steps; GCM_FLAG_BAD_COUNT
 * otherwise) and each step drops the oldest one (gcm.py:263-271, 323-355).  gcm_learned_advance_select_inplace with the
 * GNN behind the selection in the SAME launch: layer 1 of every row re-evaluated on the matrix cores from the node image
 * staged for the edge network and a bit image of the advanced adjacency, row cur's layer 2.  Writes what
 * gcm_learned_advance_select_inplace + gcm_dense_gnn2_row_fwd write into the step's record (gcm_learned_step_layout,
 * compact = 1), which gcm_learned_bptt reads unchanged.
 * adj_bits [B][N][4] u32: the adjacency of the state as bits (gcm_adj_bits), that of the INPUT state on entry and of the
 * advanced state on exit - the roll shifts the image and writes the 16-byte pieces of the fp32 rows whose bits change;
 * the fp32 matrix is not read.
 * h1_prev [B,N,H1], agg1_prev [B,N,F]: layer 1 of the INPUT state's rows (the previous steady step's h1 / agg1, or the
 * caches of gcm_learned_step_cached in front of the first one); may be this step's h1 / agg1 (every load of a graph
 * lands before its first store).  Row r of the advanced graph is row r + 1 of the input graph: its layer 1 is copied
 * unless the row had the dropped node as a source - 32-row tiles with such a row are re-evaluated on the matrix cores
 * (agg1 = Adj X from the bit image).  The cost of a step therefore follows the adjacency: about the cached step's when
 * it is sparse, the full re-evaluation's when most rows point at the oldest node.
 * params: GNN | edge network.  N % 4 == 0, F % 4 == 0, F, H1, H2 <= 32. */
int gcm_learned_step_steady(const float* obs, float* nodes, float* adj, const int64_t* count_in, const float* noise,
                            int noise_is_exp, const float* params, int has_bias, int act1, int act2, float eps0,
                            float eps1, float cutoff, int64_t* cur_out, int64_t* count_out, float* soft,
                            float* nodes_snap, float* adj_row, float* mx, float* h1, float* agg1, float* agg2,
                            const float* h1_prev, const float* agg1_prev, uint32_t* adj_bits, uint32_t* flags, int B, int N, int F, int H1, int H2, gcm_stream_t stream);
/* adj [B,N,N] -> bits [B][N][4] u32: bit (j & 31) of word (j >> 5) of row r set where adj[b][r][j] != 0 (N <= 128,
 * N % 4 == 0).  Run once when a chain enters the steady state; gcm_learned_step_steady carries the image along. */
int gcm_adj_bits(const float* adj, uint32_t* bits, int B, int N, gcm_stream_t stream);
/* Cached steps.  Rows of h1 never change once written while a graph has not overflowed (row j's adjacency
 * entries and the rows it aggregates are final after step j), so a chain that starts from EMPTY graphs keeps
 * h1 [B,N,H1], agg1 [B,N,F] and the node matrix [B,N,F] of every node in per-chain caches (any contents at the
 * chain's head: rows < cur are written before they are read, and the backward takes block rows behind the candidates
 * as zeros) and a step computes row cur only - the whole forward step (gcm.py:262-321 with
 * learned.py:53-113) in ONE launch on a donated state, for the first N steps of such a chain.  The step's record
 * (gcm_learned_step_layout, compact = 2: no nodes / h1 / agg1 sections): adj_row [B,N], mx [B,H2], agg2 [B,H1],
 * cur | count, soft [B,N].  params: GNN | edge network, packed.  gcm_learned_step_cached_functional: the same on a
 * functional state (record layout compact = 3: nodes_out | adj_out | mx | agg2 | cur, count | soft).
 * cur_host >= 0: the row every graph's new node lands in, when the host knows it (the chain's step count - no count
 * load in front of the kernel's addresses; count_in is then only compared, GCM_FLAG_BAD_COUNT); -1: read count_in.
 * cache_u [B,N,F] (a fourth per-chain cache, any contents at the chain's head): the edge network's first-layer
 * product of every stored row, U[j] = W0[:, F:] x_j - x_j never changes once stored, so at the exact shapes N = 128,
 * F = H1 = H2 = 32 a step stages U instead of multiplying the node image by W0b again, adds W0a x_cur + b0 on the way
 * into the LayerNorm and writes U[cur]; other shapes leave it untouched.  At those shapes with cur_host >= 0 the step
 * runs as eight tile waves + a tail wave per graph (k_learned_select8: the edge network in registers on 16-row tiles; same
 * record, caches and state, logits equal to rounding) unless has_bias carries GCM_STEP_FOUR_WAVES.
 * gcm_learned_bptt_cached: gcm_learned_bptt for a chain whose first n_cached steps are such steps (records in
 * layout `cached_layout` = 2 | 3; the steps behind them: `compact` layout).  cache_u (may be NULL): the chain's U cache -
 * with it (F = 32) the edge network's backward takes P0 = U + c0 from it instead of multiplying the node rows by W0
 * again (32 of its 112 matrix instructions per 32-row block).  At N = 128, F = 32 that pass runs per 16-row tile in
 * registers (k_learned_bptt_mlp16: cached steps on the U cache, the steps behind them with U as one more product);
 * GCM_BPTT_MLP_BLOCKS or-ed into cached_layout keeps the 32-row-block kernel (the A/B: a per-call argument).  A backward whose
 * steps are ALL cached steps of one chain (n_cached == n_steps <= 128, the exact widths) runs pass B1 per graph
 * (k_learned_bptt_sel_graph) and, on donated records (cached_layout 2), pass A per graph on the matrix cores
 * (k_bptt_learned_graph); same results up to summation order, same workspace. */
#define GCM_BPTT_MLP_BLOCKS 256
int gcm_learned_step_cached(const float* obs, float* nodes, float* adj, const int64_t* count_in,
                            const float* noise, int noise_is_exp, const float* params, int has_bias, int act1,
                            int act2, float eps0, float eps1, float cutoff, int64_t* cur_out, int64_t* count_out,
                            float* soft, float* adj_row, float* mx, float* agg2, float* cache_h1,
                            float* cache_agg1, float* cache_nodes, float* cache_u, uint32_t* flags, int B, int N,
                            int F, int H1, int H2, int cur_host, gcm_stream_t stream);
int gcm_learned_step_cached_functional(const float* obs, const float* nodes_in, const float* adj_in,
                                       const int64_t* count_in, const float* noise, int noise_is_exp,
                                       const float* params, int has_bias, int act1, int act2, float eps0,
                                       float eps1, float cutoff, float* nodes_out, float* adj_out, int64_t* cur_out,
                                       int64_t* count_out, float* soft, float* mx, float* agg2, float* cache_h1,
                                       float* cache_agg1, float* cache_nodes, float* cache_u, uint32_t* flags, int B,
                                       int N, int F, int H1, int H2, int cur_host, gcm_stream_t stream);
int gcm_learned_bptt_cached(const float* const* saved_host, const float* const* gmx_host, int n_steps,
                            int n_cached, int cached_layout, const float* cache_nodes, const float* cache_h1,
                            const float* cache_agg1, const float* cache_u, long gmx_stride_b, long gmx_stride_h,
                            const float* params,
                            int act1, int act2, float eps0, float eps1, int compact, const float* g_params_prev,
                            float* g_params, void* workspace, size_t workspace_bytes, int B, int N, int F, int H1,
                            int H2, gcm_stream_t stream);

/* The forward's first kernel on a DONATED state (advanced in place, nothing copied without overflow):
 * nodes_snap [B,N,F] and adj_row [B,N] are the sections `nodes` / `adj` of the COMPACT step buffer
 * (gcm_learned_step_layout(..., compact = 1): row cur of the adjacency only).  gcm_dense_gnn2_row_fwd
 * then runs on the state itself. */
int gcm_learned_advance_select_inplace(const float* obs, float* nodes, float* adj, const int64_t* count_in,
                                       const float* noise, int noise_is_exp, const float* mlp_params,
                                       float eps0, float eps1, float cutoff, int64_t* cur_out,
                                       int64_t* count_out, float* soft, float* nodes_snap, float* adj_row,
                                       uint32_t* flags, int B, int N, int F, gcm_stream_t stream);

/* ---- SURVEY 8(f) "next" rows ---------------------------------------------------------- */

/* PositionalEncoding mode="add" (src/gcm/gcm.py:120-131, util.idxs_up_to_including_num_nodes
 * util.py:478-498): x[b, n, :] += pe[n, :F] for every n <= num_nodes[b], IN PLACE on x.
 * pe is the [max_len, d_model] sin/cos table (d_model = F rounded up to even). */
int gcm_posenc_add(float* x, const float* pe, const int64_t* num_nodes, int B, int N, int F,
                   int d_model, int max_len, gcm_stream_t stream);

/* util.pack_hidden (src/gcm/util.py:323-351): coalesced COO (batch, i, j) -> dense_edges
 * [B, 2, max_edges] (pre-filled by the caller with edge_fill) and dense_weights [B, 1, max_edges]
 * (pre-filled with weight_fill).  batch_ptr [B+1] = gcm_ptr_from_sorted over coo[0].  Edges
 * beyond max_edges-1 of a graph are dropped and GCM_FLAG_PACK_OVERFLOW is raised (the reference
 * asserts count < max_edges). */
#define GCM_FLAG_PACK_OVERFLOW 32u
int gcm_pack_hidden(const int64_t* coo, const float* values, const int64_t* batch_ptr,
                    int64_t* dense_edges, float* dense_weights, uint32_t* flags, int64_t E, int B,
                    int max_edges, gcm_stream_t stream);

/* ---- sparse LearnedEdge (src/gcm/sparse_edge_selectors/learned.py:90-160) -------------- */

/* util.get_causal_edges (util.py:242-282) in closed form: for graph b the candidate edges are
 * (sink i, source j) with i in [T_b, T_b+tau_b), lo_b <= j < i, lo_b = window < 0 ? 0 :
 * max(0, T_b - window); ordered by (batch, sink, source) = coalesced COO order.
 * count: edge_off [B+1] (edges before graph b) and seg_off [B+1] (sink rows before graph b).
 * fill : indices [3, E] rows (batch, sink, source) and seg_ptr [S+1] (S = seg_off[B] sink rows;
 *        seg_ptr[s] = first edge of sink row s). */
int gcm_causal_count(const int64_t* T, const int64_t* taus, int window, int64_t* edge_off,
                     int64_t* seg_off, int B, gcm_stream_t stream);
int gcm_causal_fill(const int64_t* T, const int64_t* taus, int window, const int64_t* edge_off,
                    const int64_t* seg_off, int64_t* indices, int64_t* seg_ptr, int64_t E,
                    int64_t S, int B, gcm_stream_t stream);

/* learned.py:121-124: pairs[e, :] = cat(nodes[b, sink_e], nodes[b, source_e])  ([E, 2F]). */
int gcm_causal_pairs_fwd(const float* nodes, const int64_t* indices, float* pairs, int64_t E,
                         int B, int N, int F, gcm_stream_t stream);
/* adjoint; the source half is gathered through the closed-form edge positions (no atomics).
 * g_nodes [B,N,F] is overwritten. */
int gcm_causal_pairs_bwd(const float* g_pairs, const int64_t* T, const int64_t* taus, int window,
                         const int64_t* edge_off, float* g_nodes, int64_t E, int B, int N, int F,
                         gcm_stream_t stream);

/* util.sparse_gumbel_softmax (util.py:89-113, hard=False) over the sink rows:
 * soft[e] = softmax over the row's edges of (logits[e] + noise[e]) / tau.  tau: device scalar. */
int gcm_segment_softmax_fwd(const float* logits, const float* noise, const float* tau,
                            const int64_t* seg_ptr, float* soft, int64_t S, int64_t E,
                            gcm_stream_t stream);
/* backward: g_logits [E]; g_tau_rows [S] = per-row contribution to d tau (caller sums). */
int gcm_segment_softmax_bwd(const float* g_soft, const float* soft, const float* logits,
                            const float* noise, const float* tau, const int64_t* seg_ptr,
                            float* g_logits, float* g_tau_rows, int64_t S, int64_t E,
                            gcm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GCM_HIP_H */
