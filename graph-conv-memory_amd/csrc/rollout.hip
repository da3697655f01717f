// Time-batched DenseGCM rollout: the canonical caller is a Python `for t in range(T)` loop
// over DenseGCM.forward (ray_gcm.py:200-202, README.md:79-82).  Here the T steps are enqueued
// from one C call - per step: state advance, the native selector chain, the fused GNN kernel -
// so the device never waits for host dispatch.  Pure launch sequencing; no new kernels.
#include "gcm_common.h"

namespace {
struct Unpacked {
  const float *w_rel1, *w_root1, *b1, *w_rel2, *w_root2, *b2;
};
Unpacked unpack(const float* p, int has_bias, int F, int H1, int H2) {
  Unpacked u;
  u.w_rel1 = p;
  u.w_root1 = u.w_rel1 + (size_t)H1 * F;
  const float* b1 = u.w_root1 + (size_t)H1 * F;
  u.w_rel2 = b1 + H1;
  u.w_root2 = u.w_rel2 + (size_t)H2 * H1;
  const float* b2 = u.w_root2 + (size_t)H2 * H1;
  u.b1 = (has_bias & 1) ? b1 : nullptr;
  u.b2 = (has_bias & 2) ? b2 : nullptr;
  return u;
}

int run_selectors(const gcm_selector_desc* selectors, int n_selectors, const float* nodes_out,
                  float* adj_out, const int64_t* cur, void* workspace, size_t workspace_bytes,
                  int B, int N, int F, gcm_stream_t stream) {
  for (int i = 0; i < n_selectors; ++i) {
    const gcm_selector_desc& d = selectors[i];
    int rc;
    if (d.kind == GCM_SEL_TEMPORAL)
      rc = gcm_edge_temporal(adj_out, cur, d.hops, d.n_hops, d.direction, B, N, stream);
    else if (d.kind == GCM_SEL_DENSE)
      rc = gcm_edge_dense(adj_out, cur, B, N, stream);
    else if (d.kind == GCM_SEL_DISTANCE)
      rc = gcm_edge_distance_ex(nodes_out, adj_out, cur, d.mode, d.max_distance, d.dist_param, d.a0,
                                d.a1, d.b0, d.b1, d.bidirectional, nullptr, d.cur_rows, d.n_cur_rows,
                                workspace, workspace_bytes, B, N, F, stream);
    else
      rc = GCM_EINVAL;
    if (rc) return rc;
  }
  return GCM_OK;
}
}  // namespace

extern "C" int gcm_dense_step_fwd(const float* obs, const float* nodes_in, const float* adj_in,
                                  const int64_t* count_in, float* nodes_out, float* adj_out,
                                  int64_t* cur_out, int64_t* count_out,
                                  const gcm_selector_desc* selectors, int n_selectors,
                                  const float* params, int has_bias, int act1, int act2, float* mx,
                                  float* h1, float* agg1, float* agg2, uint32_t* flags,
                                  void* workspace, size_t workspace_bytes, int B, int N, int F,
                                  int H1, int H2, gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes_in && adj_in && count_in && nodes_out && adj_out && cur_out &&
              count_out && params && mx && flags);
  GCM_REQUIRE(B > 0 && (selectors || n_selectors == 0));
  if (!gcm_dense_gnn2_row_supported(N, F, H1, H2)) return GCM_EUNSUPPORTED;
  const Unpacked u = unpack(params, has_bias, F, H1, H2);
  int rc = gcm_dense_step_fused_fwd(obs, nodes_in, adj_in, count_in, nodes_out, adj_out, cur_out,
                                    count_out, selectors, n_selectors, u.w_rel1, u.b1, u.w_root1,
                                    act1, u.w_rel2, u.b2, u.w_root2, act2, mx, h1, agg1, agg2,
                                    flags, B, N, F, H1, H2, stream);
  if (rc != GCM_EUNSUPPORTED) return rc;   // done in one kernel (or a real error)
  rc = gcm_state_advance_fwd(nodes_in, adj_in, nullptr, count_in, obs, nodes_out, adj_out,
                             nullptr, cur_out, count_out, flags, B, N, F, stream);
  if (rc) return rc;
  rc = run_selectors(selectors, n_selectors, nodes_out, adj_out, cur_out, workspace,
                     workspace_bytes, B, N, F, stream);
  if (rc) return rc;
  return gcm_dense_gnn2_row_fwd(nodes_out, adj_out, cur_out, u.w_rel1, u.b1, u.w_root1, act1,
                                u.w_rel2, u.b2, u.w_root2, act2, mx, h1, agg1, agg2, flags, B, N,
                                F, H1, H2, stream);
}

extern "C" int gcm_dense_step_bwd(const float* g_mx, const float* g_nodes_out,
                                  const float* nodes_out, const float* adj_out,
                                  const int64_t* cur, const int64_t* count_in, const float* params,
                                  int has_bias, int act1, int act2, const float* mx,
                                  const float* h1, const float* agg1, const float* agg2,
                                  float* g_nodes_in, float* g_obs, float* g_params,
                                  void* workspace, size_t workspace_bytes, int B, int N, int F,
                                  int H1, int H2, gcm_stream_t stream) {
  return gcm_dense_step_bwd_acc(g_mx, g_nodes_out, nodes_out, adj_out, cur, count_in, params,
                                has_bias, act1, act2, mx, h1, agg1, agg2, g_nodes_in, g_obs,
                                nullptr, g_params, workspace, workspace_bytes, B, N, F, H1, H2,
                                stream);
}

extern "C" int gcm_dense_step_bwd_acc(const float* g_mx, const float* g_nodes_out,
                                      const float* nodes_out, const float* adj_out,
                                      const int64_t* cur, const int64_t* count_in,
                                      const float* params, int has_bias, int act1, int act2,
                                      const float* mx, const float* h1, const float* agg1,
                                      const float* agg2, float* g_nodes_in, float* g_obs,
                                      const float* g_params_prev, float* g_params, void* workspace,
                                      size_t workspace_bytes, int B, int N, int F, int H1, int H2,
                                      gcm_stream_t stream) {
  GCM_REQUIRE(params && g_params && workspace);
  const size_t P = gcm_dense_gnn2_param_count(F, H1, H2);
  if (workspace_bytes < sizeof(float) * (size_t)B * P) return GCM_EWORKSPACE;
  const Unpacked u = unpack(params, has_bias, F, H1, H2);
  float* slabs = (float*)workspace;
  // (the live-tile kernel of the time-parallel schedule chains several dependent loads per item:
  // with ONE item per workgroup it is slower than this one, which issues every load up front)
  const int n_slabs = B;
  int rc = gcm_dense_gnn2_row_bwd(g_mx, g_nodes_out, nodes_out, adj_out, cur, count_in, u.w_rel1,
                                  u.b1, u.w_root1, act1, u.w_rel2, u.b2, u.w_root2, act2, mx, h1,
                                  agg1, agg2, g_nodes_in, g_obs, slabs, 0, B, N, F, H1, H2, stream);
  if (rc) return rc;
  return gcm_sum_slabs_acc(slabs, n_slabs, (int)P, g_params_prev, g_params, stream);
}

extern "C" int gcm_dense_step_bwd_slabs(const float* g_mx, const float* g_nodes_out,
                                        const float* nodes_out, const float* adj_out,
                                        const int64_t* cur, const int64_t* count_in,
                                        const float* params, int has_bias, int act1, int act2,
                                        const float* mx, const float* h1, const float* agg1,
                                        const float* agg2, float* g_nodes_in, float* g_obs,
                                        float* slabs, int accumulate, int B, int N, int F, int H1,
                                        int H2, gcm_stream_t stream) {
  GCM_REQUIRE(params && slabs);
  const Unpacked u = unpack(params, has_bias, F, H1, H2);
  return gcm_dense_gnn2_row_bwd(g_mx, g_nodes_out, nodes_out, adj_out, cur, count_in, u.w_rel1,
                                u.b1, u.w_root1, act1, u.w_rel2, u.b2, u.w_root2, act2, mx, h1,
                                agg1, agg2, g_nodes_in, g_obs, slabs, accumulate, B, N, F, H1, H2,
                                stream);
}

extern "C" int gcm_dense_rollout_fwd(const float* obs, float* nodes_all, float* adj_all,
                                     int64_t* count_all, int64_t* cur_all,
                                     const gcm_selector_desc* selectors, int n_selectors,
                                     const float* w_rel1, const float* b_rel1,
                                     const float* w_root1, int act1, const float* w_rel2,
                                     const float* b_rel2, const float* w_root2, int act2,
                                     float* mx_all, float* h1_all, float* agg1_all,
                                     float* agg2_all, uint32_t* flags, void* workspace,
                                     size_t workspace_bytes, int T, int B, int N, int F, int H1,
                                     int H2, gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes_all && adj_all && count_all && cur_all && mx_all && flags);
  GCM_REQUIRE(T > 0 && B > 0 && (selectors || n_selectors == 0));
  if (!gcm_dense_gnn2_row_supported(N, F, H1, H2)) return GCM_EUNSUPPORTED;
  {  // one persistent launch when the selectors are index writes and the shape is tile-exact
    const int rc = gcm_dense_rollout_persistent_fwd(
        obs, nodes_all, adj_all, count_all, cur_all, selectors, n_selectors, w_rel1, b_rel1,
        w_root1, act1, w_rel2, b_rel2, w_root2, act2, mx_all, h1_all, agg1_all, agg2_all, flags,
        /*history=*/1, T, B, N, F, H1, H2, stream);
    if (rc != GCM_EUNSUPPORTED) return rc;
  }
  const size_t nodes_sz = (size_t)B * N * F, adj_sz = (size_t)B * N * N;
  for (int t = 0; t < T; ++t) {
    const float* nodes_in = nodes_all + (size_t)t * nodes_sz;
    float* nodes_out = nodes_all + (size_t)(t + 1) * nodes_sz;
    const float* adj_in = adj_all + (size_t)t * adj_sz;
    float* adj_out = adj_all + (size_t)(t + 1) * adj_sz;
    int64_t* cur = cur_all + (size_t)t * B;
    float* mx_t = mx_all + (size_t)t * B * H2;
    float* h1_t = h1_all ? h1_all + (size_t)t * B * N * H1 : nullptr;
    float* agg1_t = agg1_all ? agg1_all + (size_t)t * nodes_sz : nullptr;
    float* agg2_t = agg2_all ? agg2_all + (size_t)t * B * H1 : nullptr;
    int rc = gcm_dense_step_fused_fwd(obs + (size_t)t * B * F, nodes_in, adj_in,
                                      count_all + (size_t)t * B, nodes_out, adj_out, cur,
                                      count_all + (size_t)(t + 1) * B, selectors, n_selectors,
                                      w_rel1, b_rel1, w_root1, act1, w_rel2, b_rel2, w_root2, act2,
                                      mx_t, h1_t, agg1_t, agg2_t, flags, B, N, F, H1, H2, stream);
    if (rc == GCM_OK) continue;              // the whole step ran as one kernel
    if (rc != GCM_EUNSUPPORTED) return rc;
    rc = gcm_state_advance_fwd(nodes_in, adj_in, nullptr, count_all + (size_t)t * B,
                               obs + (size_t)t * B * F, nodes_out, adj_out, nullptr, cur,
                               count_all + (size_t)(t + 1) * B, flags, B, N, F, stream);
    if (rc) return rc;
    rc = run_selectors(selectors, n_selectors, nodes_out, adj_out, cur, workspace, workspace_bytes,
                       B, N, F, stream);
    if (rc) return rc;
    rc = gcm_dense_gnn2_row_fwd(nodes_out, adj_out, cur, w_rel1, b_rel1, w_root1, act1, w_rel2,
                                b_rel2, w_root2, act2, mx_t, h1_t, agg1_t, agg2_t, flags, B, N, F,
                                H1, H2, stream);
    if (rc) return rc;
  }
  return GCM_OK;
}

extern "C" size_t gcm_dense_rollout_bwd_batched_workspace_bytes(int T, int B, int N, int F, int H1,
                                                                int H2) {
  if (T <= 0 || B <= 0 || N <= 0 || F <= 0 || H1 <= 0 || H2 <= 0) return 0;
  return sizeof(float) * (size_t)T * B *
         ((size_t)N * F + F + gcm_dense_gnn2_param_count(F, H1, H2));
}

extern "C" size_t gcm_dense_rollout_bwd_workspace_bytes(int B, int N, int F, int H1, int H2) {
  if (B <= 0 || N <= 0 || F <= 0 || H1 <= 0 || H2 <= 0) return 0;
  return sizeof(float) * (2 * (size_t)B * N * F + (size_t)B * gcm_dense_gnn2_param_count(F, H1, H2));
}

extern "C" int gcm_dense_rollout_bwd(const float* g_mx_all, const float* g_nodes_T,
                                     const float* nodes_all, const float* adj_all,
                                     const int64_t* count_all, const int64_t* cur_all,
                                     const float* w_rel1, const float* b_rel1,
                                     const float* w_root1, int act1, const float* w_rel2,
                                     const float* b_rel2, const float* w_root2, int act2,
                                     const float* mx_all, const float* h1_all,
                                     const float* agg1_all, const float* agg2_all,
                                     float* g_obs_all, float* g_nodes_0, float* g_params,
                                     void* workspace, size_t workspace_bytes, int T, int B, int N,
                                     int F, int H1, int H2, gcm_stream_t stream) {
  GCM_REQUIRE(g_mx_all && nodes_all && adj_all && count_all && cur_all && mx_all && h1_all &&
              agg1_all && agg2_all && g_obs_all && g_nodes_0 && g_params && workspace);
  GCM_REQUIRE(T > 0 && B > 0);
  if (!gcm_dense_gnn2_row_supported(N, F, H1, H2)) return GCM_EUNSUPPORTED;
  if (workspace_bytes < gcm_dense_rollout_bwd_workspace_bytes(B, N, F, H1, H2))
    return GCM_EWORKSPACE;
  const size_t nodes_sz = (size_t)B * N * F, adj_sz = (size_t)B * N * N;
  if (T > 1 && (size_t)N * F <= 8192 &&
      sizeof(float) * 2 * (size_t)N * F + sizeof(int) * (size_t)T <= 160 * 1024 &&   // scan's LDS
      workspace_bytes >= gcm_dense_rollout_bwd_batched_workspace_bytes(T, B, N, F, H1, H2) &&
      (size_t)T * B < (1u << 31)) {
    // time-parallel BPTT: the GNN adjoint of step t needs g_mx[t] only, so all T*B graph-steps go
    // in ONE launch (the per-step arrays are contiguous in t); what is sequential is the cheap
    // reverse scan of the node gradient through the state advance.
    const size_t Pn = gcm_dense_gnn2_param_count(F, H1, H2);
    float* Q_all = (float*)workspace;                       // [T,B,N,F]  U_t(dX_t)
    float* pobs = Q_all + (size_t)T * nodes_sz;             // [T,B,F]    dX_t[cur_t]
    float* slabs_all = pobs + (size_t)T * B * F;            // [T*B, Pn]
    // persistent live-tile kernel (one slab per workgroup) when the shape is tile-exact, else the
    // per-graph kernel over all T*B graph-steps (one slab each)
    int n_slabs = gcm_dense_bptt_batched_slabs(T * B);
    int rc = n_slabs > 0 ? gcm_dense_bptt_batched(g_mx_all, nullptr, nodes_all + nodes_sz,
                                                  adj_all + adj_sz,
                                                  cur_all, count_all, w_rel1, b_rel1, w_root1, act1,
                                                  w_rel2, b_rel2, w_root2, act2, mx_all, h1_all,
                                                  agg1_all, agg2_all, Q_all, pobs, slabs_all,
                                                  n_slabs, T * B, N, F, H1, H2, stream)
                         : GCM_EUNSUPPORTED;
    if (rc == GCM_EUNSUPPORTED) {
      n_slabs = T * B;
      rc = gcm_dense_gnn2_row_bwd(g_mx_all, nullptr, nodes_all + nodes_sz, adj_all + adj_sz,
                                  cur_all, count_all, w_rel1, b_rel1, w_root1, act1, w_rel2,
                                  b_rel2, w_root2, act2, mx_all, h1_all, agg1_all, agg2_all,
                                  Q_all, pobs, slabs_all, 0, T * B, N, F, H1, H2, stream);
    }
    if (rc) return rc;
    rc = gcm_dense_gnodes_scan(Q_all, pobs, g_nodes_T, cur_all, count_all, g_obs_all, g_nodes_0, T,
                               B, N, F, stream);
    if (rc) return rc;
    return gcm_sum_slabs(slabs_all, n_slabs, (int)Pn, g_params, stream);
  }
  float* ping[2] = {(float*)workspace, (float*)workspace + nodes_sz};
  float* slabs = (float*)workspace + 2 * nodes_sz;
  const size_t P = gcm_dense_gnn2_param_count(F, H1, H2);
  const float* g_next = g_nodes_T;  // gradient w.r.t. the nodes returned by step t
  for (int t = T - 1; t >= 0; --t) {
    float* g_prev = t == 0 ? g_nodes_0 : ping[t & 1];
    int rc = gcm_dense_gnn2_row_bwd(
        g_mx_all + (size_t)t * B * H2, g_next, nodes_all + (size_t)(t + 1) * nodes_sz,
        adj_all + (size_t)(t + 1) * adj_sz, cur_all + (size_t)t * B, count_all + (size_t)t * B,
        w_rel1, b_rel1, w_root1, act1, w_rel2, b_rel2, w_root2, act2, mx_all + (size_t)t * B * H2,
        h1_all + (size_t)t * B * N * H1, agg1_all + (size_t)t * nodes_sz,
        agg2_all + (size_t)t * B * H1, g_prev, g_obs_all + (size_t)t * B * F, slabs,
        /*accumulate=*/t != T - 1, B, N, F, H1, H2, stream);
    if (rc) return rc;
    g_next = g_prev;
  }
  return gcm_sum_slabs(slabs, B, (int)P, g_params, stream);
}
