// TemporalBackedge(learned=True) (reference: src/gcm/edge_selectors/temporal.py:51-70; util.py:29-42
// Spardmax, util.py:456-465 diff_or): every graph with n_b > 0 stored nodes draws `S` straight-through
// gumbel one-hots over window[:n_b] (or takes one hard sparsemax), ORs them and adds the mask to
// adj[b, n_b, :n_b].  One wave per graph, forward and backward; no host readback (a graph that holds
// more nodes than the window raises GCM_FLAG_WINDOW instead of the reference's shape error).
#include "gcm_common.h"

namespace {

__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// (value, index) of the first maximum
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(v, o);
    const int i2 = __shfl_xor(i, o);
    const bool take = (v2 > v) | ((v2 == v) & (i2 < i));
    v = take ? v2 : v;
    i = take ? i2 : i;
  }
}

// the straight-through value (hard - soft.detach() + soft) and one diff_or step, rounded as torch does
__device__ __forceinline__ float st_value(float hard, float soft) { return __fadd_rn(__fsub_rn(hard, soft), soft); }
__device__ __forceinline__ float diff_or(float m, float y) { return __fsub_rn(__fadd_rn(m, y), __fmul_rn(m, y)); }

// sparsemax of z[0..n) (read through `zf`): lane-strided O(n^2) ranking - the window is a handful of
// entries.  -> tau and the support size; z_j - tau clamped at 0 is the result.
template <typename ZF>
__device__ __forceinline__ void sparsemax_tau(ZF zf, int n, int lane, float& tau, int& ksup) {
  int cnt = 0;
  float pick = 0.f;
  // first pass: support size
  for (int j = lane; j < n; j += 64) {
    const float zj = zf(j);
    int k = 1;
    float cs = 0.f;
    for (int i = 0; i < n; ++i) {
      const float zi = zf(i);
      const bool before = (zi > zj) | ((zi == zj) & (i < j));
      k += before ? 1 : 0;
      cs += (before | (i == j)) ? zi : 0.f;
    }
    cnt += (1.f + (float)k * zj > cs) ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  ksup = cnt < 1 ? 1 : cnt;
  // second pass: the cumulative sum at rank ksup
  for (int j = lane; j < n; j += 64) {
    const float zj = zf(j);
    int k = 1;
    float cs = 0.f;
    for (int i = 0; i < n; ++i) {
      const float zi = zf(i);
      const bool before = (zi > zj) | ((zi == zj) & (i < j));
      k += before ? 1 : 0;
      cs += (before | (i == j)) ? zi : 0.f;
    }
    pick += (k == ksup) ? cs : 0.f;
  }
  pick = wave_sum(pick);
  tau = (pick - 1.f) / (float)ksup;
}

// soft [S, B, Wn] (S = 1 when deterministic) is what the backward needs; mask_ws [B, Wn] scratch.
__global__ __launch_bounds__(64) void k_temporal_window_fwd(
    const float* __restrict__ window, const float* __restrict__ noise, const int64_t* __restrict__ cur,
    float* __restrict__ adj, float* __restrict__ soft, float* __restrict__ mask_ws, int B, int N, int W,
    int Wn, int S, int deterministic, uint32_t* __restrict__ flags) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int64_t n64 = cur[b];
  const int n_soft = deterministic ? 1 : S;
  if (n64 <= 0 || n64 > W || n64 >= N) {
    if (lane == 0 && n64 > 0) atomicOr(flags, n64 > W ? GCM_FLAG_WINDOW : GCM_FLAG_BAD_COUNT);
    for (int s = 0; s < n_soft; ++s)
      for (int j = lane; j < Wn; j += 64) soft[((size_t)s * B + b) * Wn + j] = 0.f;
    return;
  }
  const int n = (int)n64;
  float* row = adj + ((size_t)b * N + n) * N;
  float* mk = mask_ws + (size_t)b * Wn;
  if (deterministic) {
    float tau;
    int ksup;
    sparsemax_tau([&](int j) { return window[j]; }, n, lane, tau, ksup);
    for (int j = lane; j < Wn; j += 64) {
      float sj = 0.f;
      if (j < n) {
        sj = fmaxf(window[j] - tau, 0.f);
        row[j] += st_value(sj > 0.f ? 1.f : 0.f, sj);
      }
      soft[(size_t)b * Wn + j] = sj;
    }
    return;
  }
  for (int s = 0; s < S; ++s) {
    const float* g = noise + ((size_t)s * B + b) * Wn;
    float* so = soft + ((size_t)s * B + b) * Wn;
    float m = -INFINITY;
    for (int j = lane; j < n; j += 64) m = fmaxf(m, window[j] + g[j]);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) sum += expf(window[j] + g[j] - m);
    sum = wave_sum(sum);
    float best = -1.f;
    int arg = 0x7fffffff;
    for (int j = lane; j < n; j += 64) {
      const float p = expf(window[j] + g[j] - m) / sum;
      so[j] = p;
      if (p > best) { best = p; arg = j; }
    }
    wave_argmax(best, arg);
    for (int j = lane; j < Wn; j += 64) {
      if (j >= n) { so[j] = 0.f; continue; }
      const float y = st_value(j == arg ? 1.f : 0.f, so[j]);
      mk[j] = s == 0 ? diff_or(0.f, y) : diff_or(mk[j], y);
    }
  }
  for (int j = lane; j < n; j += 64) row[j] += mk[j];
}

#define GCM_WINDOW_MAX_SAMPLES 32

// g_window_part[b, j] = dL/dwindow[j] from graph b (summed over b by the caller, in a fixed order)
__global__ __launch_bounds__(64) void k_temporal_window_bwd(
    const float* __restrict__ g_adj, const float* __restrict__ soft, const int64_t* __restrict__ cur,
    float* __restrict__ g_part, int B, int N, int W, int Wn, int S, int deterministic) {
  __shared__ int sArg[GCM_WINDOW_MAX_SAMPLES];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int64_t n64 = cur[b];
  float* gp = g_part + (size_t)b * Wn;
  if (n64 <= 0 || n64 > W || n64 >= N) {
    for (int j = lane; j < Wn; j += 64) gp[j] = 0.f;
    return;
  }
  const int n = (int)n64;
  const float* grow = g_adj + ((size_t)b * N + n) * N;
  if (deterministic) {
    const float* so = soft + (size_t)b * Wn;
    float acc = 0.f;
    int cnt = 0;
    for (int j = lane; j < n; j += 64) {
      const bool sup = so[j] > 0.f;
      acc += sup ? grow[j] : 0.f;
      cnt += sup ? 1 : 0;
    }
    acc = wave_sum(acc);
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    const float v = acc / (float)(cnt < 1 ? 1 : cnt);
    for (int j = lane; j < Wn; j += 64) gp[j] = (j < n && so[j] > 0.f) ? grow[j] - v : 0.f;
    return;
  }
  for (int s = 0; s < S; ++s) {
    const float* so = soft + ((size_t)s * B + b) * Wn;
    float best = -1.f;
    int arg = 0x7fffffff;
    for (int j = lane; j < n; j += 64)
      if (so[j] > best) { best = so[j]; arg = j; }
    wave_argmax(best, arg);
    if (lane == 0) sArg[s] = arg;
  }
  __syncthreads();
  for (int j = lane; j < Wn; j += 64) gp[j] = 0.f;
  for (int s = 0; s < S; ++s) {
    const float* so = soft + ((size_t)s * B + b) * Wn;
    // dL/dy_s[j] = g * prod_{t>s}(1 - y_t) * (1 - m_{s-1}), the chain autograd walks through diff_or
    auto dy = [&](int j) {
      float m = 0.f;
      for (int t = 0; t < s; ++t)
        m = diff_or(m, st_value(j == sArg[t] ? 1.f : 0.f, soft[((size_t)t * B + b) * Wn + j]));
      float g = grow[j];
      for (int t = S - 1; t > s; --t)
        g = g * (1.f - st_value(j == sArg[t] ? 1.f : 0.f, soft[((size_t)t * B + b) * Wn + j]));
      return g * (1.f - m);
    };
    float dot = 0.f;
    for (int j = lane; j < n; j += 64) dot += dy(j) * so[j];
    dot = wave_sum(dot);
    for (int j = lane; j < n; j += 64) gp[j] += so[j] * (dy(j) - dot);
  }
}

}  // namespace

extern "C" int gcm_temporal_window_fwd(const float* window, const float* noise, const int64_t* cur_idx,
                                       float* adj, float* soft, float* mask_ws, int B, int N, int W, int S,
                                       int deterministic, uint32_t* flags, gcm_stream_t stream) {
  GCM_REQUIRE(window && cur_idx && adj && soft && mask_ws && flags && B > 0 && N > 0 && W > 0);
  GCM_REQUIRE(deterministic || (noise && S > 0));
  if (!deterministic && S > GCM_WINDOW_MAX_SAMPLES) return GCM_EUNSUPPORTED;
  const int Wn = W < N ? W : N;
  hipLaunchKernelGGL(k_temporal_window_fwd, dim3(B), dim3(64), 0, (hipStream_t)stream, window, noise, cur_idx,
                     adj, soft, mask_ws, B, N, W, Wn, S, deterministic, flags);
  return gcm_launch_status();
}

extern "C" int gcm_temporal_window_bwd(const float* g_adj, const float* soft, const int64_t* cur_idx,
                                       float* g_window_part, int B, int N, int W, int S, int deterministic,
                                       gcm_stream_t stream) {
  GCM_REQUIRE(g_adj && soft && cur_idx && g_window_part && B > 0 && N > 0 && W > 0);
  if (!deterministic && (S <= 0 || S > GCM_WINDOW_MAX_SAMPLES)) return GCM_EUNSUPPORTED;
  const int Wn = W < N ? W : N;
  hipLaunchKernelGGL(k_temporal_window_bwd, dim3(B), dim3(64), 0, (hipStream_t)stream, g_adj, soft, cur_idx,
                     g_window_part, B, N, W, Wn, S, deterministic);
  return gcm_launch_status();
}
