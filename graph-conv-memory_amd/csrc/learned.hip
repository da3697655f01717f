// LearnedEdge kernels (edge_selectors/learned.py:53-125).  The edge network itself is a
// user-replaceable torch module (learned.py:28-33) and stays a stack of library GEMMs; what
// is fused here is everything around it: building the candidate-pair matrix without
// nonzero()/max() host syncs, and gumbel-softmax + straight-through threshold + adjacency
// row write (one wave per graph, row kept in registers, wavefront shuffles for the
// reductions).
#include "gcm_common.h"

namespace {

__global__ __launch_bounds__(256) void k_pairs_fwd(const float* __restrict__ nodes,
                                                   const int64_t* __restrict__ cur_idx,
                                                   float* __restrict__ pairs, int N, int F,
                                                   int rows_per_block) {
  const int b = blockIdx.y;
  int64_t cur = cur_idx[b];
  cur = cur < 0 ? 0 : (cur > N - 1 ? N - 1 : cur);
  const float* nb = nodes + (size_t)b * N * F;
  const int j0 = blockIdx.x * rows_per_block, j1 = min(N, j0 + rows_per_block);
  const int W = 2 * F;
  for (int e = threadIdx.x; e < (j1 - j0) * W; e += blockDim.x) {
    const int j = j0 + e / W, c = e % W;
    float v = 0.f;
    if (j < cur) v = c < F ? nb[cur * F + c] : nb[(size_t)j * F + (c - F)];
    pairs[((size_t)b * N + j) * W + c] = v;
  }
}

__global__ __launch_bounds__(256) void k_pairs_bwd(const float* __restrict__ g_pairs,
                                                   const int64_t* __restrict__ cur_idx,
                                                   float* __restrict__ g_nodes, int N, int F) {
  // one workgroup per graph: rows j < cur get their own half; row cur gets the column sums
  const int b = blockIdx.x;
  int64_t cur = cur_idx[b];
  cur = cur < 0 ? 0 : (cur > N - 1 ? N - 1 : cur);
  const int W = 2 * F;
  const float* gp = g_pairs + (size_t)b * N * W;
  float* gn = g_nodes + (size_t)b * N * F;
  for (int e = threadIdx.x; e < N * F; e += blockDim.x) {
    const int j = e / F, c = e % F;
    float v = 0.f;
    if (j < cur) {
      v = gp[(size_t)j * W + F + c];
    } else if (j == cur) {
      for (int k = 0; k < cur; ++k) v += gp[(size_t)k * W + c];
    }
    gn[e] = v;
  }
}

__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one wave per graph; lane l owns columns l, l+64, ... (N <= 64*MAXC)
constexpr int MAXC = 16;

__global__ __launch_bounds__(256) void k_select_fwd(const float* __restrict__ logits,
                                                    const float* __restrict__ noise,
                                                    const int64_t* __restrict__ cur_idx,
                                                    float cutoff, float* __restrict__ adj,
                                                    float* __restrict__ soft, int B, int N) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= B) return;
  int64_t cur = cur_idx[b];
  cur = cur < 0 ? 0 : (cur > N - 1 ? N - 1 : cur);
  float z[MAXC];
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int j = lane + 64 * c;
    z[c] = (j < cur) ? logits[(size_t)b * N + j] + noise[(size_t)b * N + j] : -INFINITY;
    m = fmaxf(m, z[c]);
  }
  m = wave_max(m);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    z[c] = (lane + 64 * c < cur) ? expf(z[c] - m) : 0.f;
    s += z[c];
  }
  s = wave_sum(s);
  const float inv = s > 0.f ? 1.f / s : 0.f;
  float* row = adj + ((size_t)b * N + cur) * N;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int j = lane + 64 * c;
    if (j >= N) break;
    const float p = z[c] * inv;
    soft[(size_t)b * N + j] = p;
    if (j < cur) {
      const float edge = (p - cutoff > 0.f) ? 1.f : 0.f;     // STE forward (util.py:12)
      row[j] = (edge + row[j] > 0.f) ? 1.f : 0.f;            // learned.py:108-110
    }
  }
}

__global__ __launch_bounds__(256) void k_select_bwd(const float* __restrict__ g_adj,
                                                    const float* __restrict__ soft,
                                                    const int64_t* __restrict__ cur_idx,
                                                    float* __restrict__ g_logits, int B, int N) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= B) return;
  int64_t cur = cur_idx[b];
  cur = cur < 0 ? 0 : (cur > N - 1 ? N - 1 : cur);
  const float* grow = g_adj + ((size_t)b * N + cur) * N;
  float p[MAXC], g[MAXC];
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int j = lane + 64 * c;
    const bool live = j < cur;
    p[c] = live ? soft[(size_t)b * N + j] : 0.f;
    g[c] = live ? grow[j] : 0.f;
    dot = fmaf(p[c], g[c], dot);
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int j = lane + 64 * c;
    if (j < N) g_logits[(size_t)b * N + j] = p[c] * (g[c] - dot);
  }
}

}  // namespace

extern "C" int gcm_learned_pairs_fwd(const float* nodes, const int64_t* cur_idx, float* pairs,
                                     int B, int N, int F, gcm_stream_t stream) {
  GCM_REQUIRE(nodes && cur_idx && pairs && B > 0 && N > 0 && F > 0);
  if (B > 65535) return GCM_EUNSUPPORTED;
  const int rpb = 16;
  hipLaunchKernelGGL(k_pairs_fwd, dim3((N + rpb - 1) / rpb, B), dim3(256), 0, (hipStream_t)stream,
                     nodes, cur_idx, pairs, N, F, rpb);
  return gcm_launch_status();
}

extern "C" int gcm_learned_pairs_bwd(const float* g_pairs, const int64_t* cur_idx, float* g_nodes,
                                     int B, int N, int F, gcm_stream_t stream) {
  GCM_REQUIRE(g_pairs && cur_idx && g_nodes && B > 0 && N > 0 && F > 0);
  hipLaunchKernelGGL(k_pairs_bwd, dim3(B), dim3(256), 0, (hipStream_t)stream, g_pairs, cur_idx,
                     g_nodes, N, F);
  return gcm_launch_status();
}

extern "C" int gcm_learned_select_fwd(const float* logits, const float* noise,
                                      const int64_t* cur_idx, float cutoff, float* adj,
                                      float* soft, int B, int N, gcm_stream_t stream) {
  GCM_REQUIRE(logits && noise && cur_idx && adj && soft && B > 0 && N > 0);
  if (N > 64 * MAXC) return GCM_EUNSUPPORTED;
  hipLaunchKernelGGL(k_select_fwd, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits,
                     noise, cur_idx, cutoff, adj, soft, B, N);
  return gcm_launch_status();
}

extern "C" int gcm_learned_select_bwd(const float* g_adj, const float* soft,
                                      const int64_t* cur_idx, float* g_logits, int B, int N,
                                      gcm_stream_t stream) {
  GCM_REQUIRE(g_adj && soft && cur_idx && g_logits && B > 0 && N > 0);
  if (N > 64 * MAXC) return GCM_EUNSUPPORTED;
  hipLaunchKernelGGL(k_select_bwd, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, g_adj,
                     soft, cur_idx, g_logits, B, N);
  return gcm_launch_status();
}
