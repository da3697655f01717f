// Sparse LearnedEdge kernels (src/gcm/sparse_edge_selectors/learned.py:90-160, util.py:89-113,
// 242-282): closed-form enumeration of the causal candidate edges (the reference loops over
// graphs building tril_indices), candidate-pair gather with a closed-form adjoint, and the
// gumbel softmax over every sink row as one wave per row (wavefront shuffles for max / sum).
#include "gcm_common.h"

namespace {

__device__ __forceinline__ int64_t lo_of(int64_t t0, int window) {
  if (window < 0) return 0;
  const int64_t l = t0 - window;
  return l > 0 ? l : 0;
}
// edges of graph b before sink row k = i - t0:  k*(t0-lo) + k(k-1)/2
__device__ __forceinline__ int64_t row_start(int64_t k, int64_t t0, int64_t lo) {
  return k * (t0 - lo) + k * (k - 1) / 2;
}

__global__ __launch_bounds__(256) void k_causal_count(const int64_t* __restrict__ T,
                                                      const int64_t* __restrict__ taus, int window,
                                                      int64_t* __restrict__ edge_off,
                                                      int64_t* __restrict__ seg_off, int B) {
  extern __shared__ int64_t buf[];   // [2*B] exclusive-in-chunk values
  __shared__ int64_t chunk_e[256], chunk_s[256];
  const int per = (B + 255) / 256;
  const int lo_b = min(B, (int)threadIdx.x * per), hi_b = min(B, lo_b + per);
  int64_t re = 0, rs = 0;
  for (int b = lo_b; b < hi_b; ++b) {
    const int64_t t0 = T[b], tau = taus[b] > 0 ? taus[b] : 0;
    buf[b] = re;
    buf[B + b] = rs;
    re += row_start(tau, t0, lo_of(t0, window));
    rs += tau;
  }
  chunk_e[threadIdx.x] = re;
  chunk_s[threadIdx.x] = rs;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t ae = 0, as = 0;
    for (int i = 0; i < 256; ++i) {
      const int64_t te = chunk_e[i], ts = chunk_s[i];
      chunk_e[i] = ae; chunk_s[i] = as;
      ae += te; as += ts;
    }
    edge_off[B] = ae;
    seg_off[B] = as;
  }
  __syncthreads();
  for (int b = lo_b; b < hi_b; ++b) {
    edge_off[b] = buf[b] + chunk_e[threadIdx.x];
    seg_off[b] = buf[B + b] + chunk_s[threadIdx.x];
  }
}

// graph of sink row s (seg_off is ascending, [B+1])
__device__ __forceinline__ int graph_of(const int64_t* seg_off, int B, int64_t s) {
  int lo = 0, hi = B;   // largest b with seg_off[b] <= s
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg_off[mid] <= s) lo = mid; else hi = mid;
  }
  return lo;
}

// one wave per sink row: writes the row's (batch, sink, source) entries and seg_ptr
__global__ __launch_bounds__(256) void k_causal_fill(const int64_t* __restrict__ T,
                                                     const int64_t* __restrict__ taus, int window,
                                                     const int64_t* __restrict__ edge_off,
                                                     const int64_t* __restrict__ seg_off,
                                                     int64_t* __restrict__ indices,
                                                     int64_t* __restrict__ seg_ptr, int64_t E,
                                                     int64_t S, int B) {
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (s > S) return;
  if (s == S) {
    if (lane == 0) seg_ptr[S] = E;
    return;
  }
  const int b = graph_of(seg_off, B, s);
  const int64_t t0 = T[b], lo = lo_of(t0, window);
  const int64_t k = s - seg_off[b], i = t0 + k;
  const int64_t start = edge_off[b] + row_start(k, t0, lo);
  if (lane == 0) seg_ptr[s] = start;
  const int64_t len = i - lo;
  for (int64_t q = lane; q < len; q += 64) {
    indices[start + q] = b;
    indices[E + start + q] = i;
    indices[2 * E + start + q] = lo + q;
  }
}

__global__ void k_causal_pairs_fwd(const float* __restrict__ nodes,
                                   const int64_t* __restrict__ indices, float* __restrict__ pairs,
                                   int64_t E, int N, int F) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int W = 2 * F;
  if (idx >= E * W) return;
  const int64_t e = idx / W;
  const int c = idx - e * W;
  const int64_t b = indices[e];
  const int64_t r = c < F ? indices[E + e] : indices[2 * E + e];
  pairs[idx] = nodes[((size_t)b * N + r) * F + (c < F ? c : c - F)];
}

// g_nodes[b, r, f] = [r is a new sink] sum over its row of g_pairs[., f]
//                  + sum over new sinks i > r (r >= lo) of g_pairs[pos(i, r), F + f]
__global__ void k_causal_pairs_bwd(const float* __restrict__ g_pairs, const int64_t* __restrict__ T,
                                   const int64_t* __restrict__ taus, int window,
                                   const int64_t* __restrict__ edge_off,
                                   float* __restrict__ g_nodes, int N, int F) {
  const int b = blockIdx.y;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * F) return;
  const int r = idx / F, f = idx - r * F;
  const int64_t t0 = T[b], tau = taus[b] > 0 ? taus[b] : 0, lo = lo_of(t0, window);
  const int64_t base = edge_off[b];
  const int W = 2 * F;
  float s = 0.f;
  if (r >= t0 && r < t0 + tau) {   // sink half: the row of sink r
    const int64_t start = base + row_start(r - t0, t0, lo), len = r - lo;
    for (int64_t q = 0; q < len; ++q) s += g_pairs[(size_t)(start + q) * W + f];
  }
  if (r >= lo) {                   // source half: every new sink above r
    const int64_t i0 = (r + 1 > t0) ? r + 1 : t0;
    for (int64_t i = i0; i < t0 + tau; ++i) {
      const int64_t pos = base + row_start(i - t0, t0, lo) + (r - lo);
      s += g_pairs[(size_t)pos * W + F + f];
    }
  }
  g_nodes[((size_t)b * N + r) * F + f] = s;
}

__device__ __forceinline__ float wmax(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wsum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(256) void k_segment_softmax_fwd(
    const float* __restrict__ logits, const float* __restrict__ noise,
    const float* __restrict__ tau, const int64_t* __restrict__ seg_ptr, float* __restrict__ soft,
    int64_t S) {
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (s >= S) return;
  const int64_t e0 = seg_ptr[s], e1 = seg_ptr[s + 1];
  const float t = tau[0];
  float m = -INFINITY;
  for (int64_t e = e0 + lane; e < e1; e += 64) m = fmaxf(m, (logits[e] + noise[e]) / t);
  m = wmax(m);
  float z = 0.f;
  for (int64_t e = e0 + lane; e < e1; e += 64) z += expf((logits[e] + noise[e]) / t - m);
  z = wsum(z);
  for (int64_t e = e0 + lane; e < e1; e += 64) soft[e] = expf((logits[e] + noise[e]) / t - m) / z;
}

__global__ __launch_bounds__(256) void k_segment_softmax_bwd(
    const float* __restrict__ g_soft, const float* __restrict__ soft,
    const float* __restrict__ logits, const float* __restrict__ noise,
    const float* __restrict__ tau, const int64_t* __restrict__ seg_ptr,
    float* __restrict__ g_logits, float* __restrict__ g_tau_rows, int64_t S) {
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (s >= S) return;
  const int64_t e0 = seg_ptr[s], e1 = seg_ptr[s + 1];
  const float t = tau[0];
  float dot = 0.f;
  for (int64_t e = e0 + lane; e < e1; e += 64) dot = fmaf(soft[e], g_soft[e], dot);
  dot = wsum(dot);
  float gt = 0.f;
  for (int64_t e = e0 + lane; e < e1; e += 64) {
    const float gz = soft[e] * (g_soft[e] - dot);      // d / d z_e,  z = (logit + noise) / tau
    g_logits[e] = gz / t;
    gt -= gz * (logits[e] + noise[e]) / (t * t);
  }
  gt = wsum(gt);
  if (lane == 0) g_tau_rows[s] = gt;
}

}  // namespace

extern "C" int gcm_causal_count(const int64_t* T, const int64_t* taus, int window,
                                int64_t* edge_off, int64_t* seg_off, int B, gcm_stream_t stream) {
  GCM_REQUIRE(T && taus && edge_off && seg_off && B > 0);
  if ((size_t)2 * B * sizeof(int64_t) > 60 * 1024) return GCM_EUNSUPPORTED;
  hipLaunchKernelGGL(k_causal_count, dim3(1), dim3(256), (size_t)2 * B * sizeof(int64_t),
                     (hipStream_t)stream, T, taus, window, edge_off, seg_off, B);
  return gcm_launch_status();
}

extern "C" int gcm_causal_fill(const int64_t* T, const int64_t* taus, int window,
                               const int64_t* edge_off, const int64_t* seg_off, int64_t* indices,
                               int64_t* seg_ptr, int64_t E, int64_t S, int B,
                               gcm_stream_t stream) {
  GCM_REQUIRE(T && taus && edge_off && seg_off && seg_ptr && E >= 0 && S >= 0 && B > 0);
  GCM_REQUIRE(indices || E == 0);
  hipLaunchKernelGGL(k_causal_fill, dim3((unsigned)((S + 1 + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, T, taus, window, edge_off, seg_off, indices, seg_ptr, E,
                     S, B);
  return gcm_launch_status();
}

extern "C" int gcm_causal_pairs_fwd(const float* nodes, const int64_t* indices, float* pairs,
                                    int64_t E, int B, int N, int F, gcm_stream_t stream) {
  GCM_REQUIRE(nodes && B > 0 && N > 0 && F > 0 && E >= 0);
  if (E == 0) return GCM_OK;
  GCM_REQUIRE(indices && pairs);
  const int64_t total = E * 2 * F;
  hipLaunchKernelGGL(k_causal_pairs_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, nodes, indices, pairs, E, N, F);
  return gcm_launch_status();
}

extern "C" int gcm_causal_pairs_bwd(const float* g_pairs, const int64_t* T, const int64_t* taus,
                                    int window, const int64_t* edge_off, float* g_nodes, int64_t E,
                                    int B, int N, int F, gcm_stream_t stream) {
  GCM_REQUIRE(T && taus && edge_off && g_nodes && B > 0 && N > 0 && F > 0 && E >= 0);
  GCM_REQUIRE(g_pairs || E == 0);
  if (B > 65535) return GCM_EUNSUPPORTED;
  hipLaunchKernelGGL(k_causal_pairs_bwd, dim3((N * F + 255) / 256, B), dim3(256), 0,
                     (hipStream_t)stream, g_pairs, T, taus, window, edge_off, g_nodes, N, F);
  return gcm_launch_status();
}

extern "C" int gcm_segment_softmax_fwd(const float* logits, const float* noise, const float* tau,
                                       const int64_t* seg_ptr, float* soft, int64_t S, int64_t E,
                                       gcm_stream_t stream) {
  GCM_REQUIRE(tau && seg_ptr && S >= 0 && E >= 0);
  if (S == 0 || E == 0) return GCM_OK;
  GCM_REQUIRE(logits && noise && soft);
  hipLaunchKernelGGL(k_segment_softmax_fwd, dim3((unsigned)((S + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, logits, noise, tau, seg_ptr, soft, S);
  return gcm_launch_status();
}

extern "C" int gcm_segment_softmax_bwd(const float* g_soft, const float* soft, const float* logits,
                                       const float* noise, const float* tau,
                                       const int64_t* seg_ptr, float* g_logits, float* g_tau_rows,
                                       int64_t S, int64_t E, gcm_stream_t stream) {
  GCM_REQUIRE(tau && seg_ptr && S >= 0 && E >= 0);
  if (S == 0 || E == 0) return GCM_OK;
  GCM_REQUIRE(g_soft && soft && logits && noise && g_logits && g_tau_rows);
  hipLaunchKernelGGL(k_segment_softmax_bwd, dim3((unsigned)((S + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, g_soft, soft, logits, noise, tau, seg_ptr, g_logits,
                     g_tau_rows, S);
  return gcm_launch_status();
}
