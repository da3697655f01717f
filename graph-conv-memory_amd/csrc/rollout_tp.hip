// Time-parallel DenseGCM rollout (round 4): DenseGCM.rollout(obs[T,B,F]) from EMPTY graphs with forward temporal
// hops as the only selectors (temporal.py:72-88) and observations that carry no gradient.
//
// In that setting there is no recurrence at all.  Node t of a graph IS observation t; the adjacency is the band
// adj[i, i - h] = 1 (closed form, and a fixed point of the overflow roll: rows_cached.hip); so with the canonical
// two-layer GNN (README.md:52-62)
//     agg1[t] = sum_{h valid at t} x[t - h] (+ x[t] with a hop of 0)      h1[t] = act1(W_rel1 agg1[t] + W_root1 x[t] + b1)
//     agg2[t] = sum_{h valid at t} h1[t - h] (+ h1[t])                    mx[t] = act2(W_rel2 agg2[t] + W_root2 h1[t] + b2)
// where "valid at t" is h <= min(t, N - 1) (gcm.py:274, temporal.py:74: the node sits in row min(t, N - 1)), and
// h1[t - h] as step t sees it equals h1[t - h] as computed at its own step as long as node t - h has not lost a source
// to the overflow roll in between: N > 2 max(hop) (checked by the host; T <= N needs nothing).  The reference walks
// the T steps one after the other, 2 x B x (N^2 + N F) floats of state per step; the per-step kernels of this library
// follow the same chain because the call surface hands them one observation at a time.  rollout() sees them all:
//
//   k_rollout_tp_l1   every (step, 32 graphs) tile: gather the observation rows, layer 1 on the matrix cores
//                     ([agg1 | x] [W_rel1 | W_root1]^T, K = 2F), h1 / agg1 / x into the chain's caches [B, Tc, .],
//                     the final state's node rows / adjacency rows / counts for the last min(T, N) steps
//   k_rollout_tp_l2   the same tiles once every h1 row exists: gather h1, layer 2, beliefs [T, B, H2] and the
//                     step records (gcm_dense_rows_cached_layout with N := Tc) gcm_dense_rows_bptt_cached reads
//
// Two launches for the whole forward instead of T; the backward is the usual time-parallel launch over the records.
#include "fused_common.h"
#include "gcm_common.h"
#include "rows_common.h"

namespace gcm_rtp {

using gcm_fused::acc_row;
using gcm_fused::mma32;

struct Hops {
  int n;          // distinct hops >= 1, DESCENDING (sources in ascending node order)
  int h[16];
  int self;       // a hop of 0: self loop
};

// rows b0 .. b0 + 31 of a [*, B, W] tensor at step s (W = 4 * W4 floats): lane loads W4 / 2 float4 (32 rows x W4 = 16 W4
// float4 per wave instruction group); piece i of lane: e4 = lane + 64 i, row = e4 / W4, col4 = e4 % W4
template <int W4>
__device__ __forceinline__ void load_rows(const float* __restrict__ base, size_t row_stride, int b0, int B, int lane,
                                          float4 (&v)[W4 / 2]) {
#pragma unroll
  for (int i = 0; i < W4 / 2; ++i) {
    const int e4 = lane + 64 * i, r = e4 / W4, c4 = e4 % W4;
    const int b = b0 + r < B ? b0 + r : B - 1;   // (clamped: an unconditional load)
    v[i] = *reinterpret_cast<const float4*>(base + (size_t)b * row_stride + 4 * c4);
  }
}
__device__ __forceinline__ void add4(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// ---------------------------------------------------------------------------------------------------------
template <int FP, int HP>
__global__ __launch_bounds__(256) void k_rollout_tp_l1(
    const float* __restrict__ obs, Hops hp, const float* __restrict__ params, int act1, float* __restrict__ cH,
    float* __restrict__ cA, float* __restrict__ cX, float* __restrict__ nodes_out, float* __restrict__ adj_out,
    int64_t* __restrict__ count_out, int B, int T, int N, int Tc, int n_tiles) {
  constexpr int F = FP, H1 = HP, F4 = FP / 4;
  constexpr int AS = 2 * FP + 1;          // A tile row stride (odd: conflict-free fragment reads)
  constexpr int WS = HP + 1;              // B operand [k][n] row stride
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  extern __shared__ float smem[];
  float* sW = smem;                                      // [2 FP][WS]: k < FP: W_rel1[n][k], else W_root1[n][k - FP]
  float* sA = sW + 2 * FP * WS + (size_t)wave * 32 * AS;   // this wave's [32][AS] tile: agg1 | x
  for (int e = tid; e < 2 * FP * HP; e += 256) {
    const int m = e / (FP * HP), rem = e - m * FP * HP, n = rem / FP, k = rem % FP;
    sW[(m * FP + k) * WS + n] = params[(size_t)m * H1 * F + (size_t)n * F + k];
  }
  float bias[HP / 32];
#pragma unroll
  for (int nt = 0; nt < HP / 32; ++nt) bias[nt] = params[2 * (size_t)H1 * F + nt * 32 + li];
  const int act_v = gcm_vgpr(act1);
  __syncthreads();
  const int nbt = (B + 31) / 32;
  const int t_state0 = T > N ? T - N : 0;   // the final state holds the nodes of steps t_state0 .. T - 1
#pragma unroll 1
  for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
    const int t = tile / nbt, b0 = (tile - t * nbt) * 32;
    const int cur = t < N ? t : N - 1;
    float4 xv[F4 / 2], ag[F4 / 2];
    load_rows<F4>(obs + (size_t)t * B * F, F, b0, B, lane, xv);
#pragma unroll
    for (int i = 0; i < F4 / 2; ++i) ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0; q < hp.n; ++q) {        // hops descending: sources in ascending node order
      int h = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) h = q == i ? hp.h[i] : h;
      if (h > cur) continue;                // (uniform) temporal.py:74
      float4 sv[F4 / 2];
      load_rows<F4>(obs + (size_t)(t - h) * B * F, F, b0, B, lane, sv);
#pragma unroll
      for (int i = 0; i < F4 / 2; ++i) add4(ag[i], sv[i]);
    }
    if (hp.self) {
#pragma unroll
      for (int i = 0; i < F4 / 2; ++i) add4(ag[i], xv[i]);
    }
    const bool in_state = t >= t_state0;
    const int r_state = t - t_state0;
#pragma unroll
    for (int i = 0; i < F4 / 2; ++i) {
      const int e4 = lane + 64 * i, r = e4 / F4, c = (e4 % F4) * 4;
      float* a = sA + r * AS + c;
      a[0] = ag[i].x; a[1] = ag[i].y; a[2] = ag[i].z; a[3] = ag[i].w;
      a[FP] = xv[i].x; a[FP + 1] = xv[i].y; a[FP + 2] = xv[i].z; a[FP + 3] = xv[i].w;
      const int b = b0 + r;
      if (b < B) {
        const size_t rc = ((size_t)b * Tc + t) * F + c;
        *reinterpret_cast<float4*>(cA + rc) = ag[i];
        *reinterpret_cast<float4*>(cX + rc) = xv[i];
        if (in_state) *reinterpret_cast<float4*>(nodes_out + ((size_t)b * N + r_state) * F + c) = xv[i];
      }
    }
    if (in_state && li + b0 < B) {          // the band row of the final adjacency, and the count behind the last step
      float* arow = adj_out + ((size_t)(b0 + li) * N + r_state) * N;
      for (int q = lh; q < hp.n; q += 2) {
        int h = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) h = q == i ? hp.h[i] : h;
        if (h <= r_state) arow[r_state - h] = 1.f;
      }
      if (hp.self && lh == 0) arow[r_state] = 1.f;
      if (t == T - 1 && lh == 0) count_out[b0 + li] = T < N ? T : N;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int nt = 0; nt < HP / 32; ++nt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      mma32(acc, sA, AS, 1, sW + nt * 32, WS, 1, 2 * FP, li, lh);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int b = b0 + acc_row(r, lh);
        if (b < B) cH[((size_t)b * Tc + t) * H1 + nt * 32 + li] = gcm_act_sel(acc[r] + bias[nt], act_v);
      }
    }
    __builtin_amdgcn_wave_barrier();        // the tile is rewritten by the next trip
  }
}

// ---------------------------------------------------------------------------------------------------------
template <int HP>
__global__ __launch_bounds__(256) void k_rollout_tp_l2(
    Hops hp, const float* __restrict__ params, int F, int act2, const float* __restrict__ cH,
    float* __restrict__ mx_all, float* __restrict__ rec0, size_t rec_stride, gcm_rows::CachedLayout lay, int record,
    uint32_t* __restrict__ flags, int B, int T, int N, int Tc, int H2, int n_tiles) {
  constexpr int H1 = HP, H4 = HP / 4;
  constexpr int AS = 2 * HP + 1, WS = 65;   // H2 <= 64
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  extern __shared__ float smem[];
  float* sW = smem;                                      // [2 HP][WS]: k < HP: W_rel2[n][k], else W_root2[n][k - HP]
  float* sA = sW + 2 * HP * WS + (size_t)wave * 32 * AS;   // [32][AS]: agg2 | h1[t]
  const float* w2 = params + 2 * (size_t)H1 * F + H1;
  for (int e = tid; e < 2 * HP * 64; e += 256) {
    const int m = e / (HP * 64), rem = e - m * HP * 64, n = rem / HP, k = rem % HP;
    sW[(m * HP + k) * WS + n] = n < H2 ? w2[(size_t)m * H2 * H1 + (size_t)n * H1 + k] : 0.f;
  }
  const float* b2 = w2 + 2 * (size_t)H2 * H1;
  float bias[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) bias[nt] = nt * 32 + li < H2 ? b2[nt * 32 + li] : 0.f;
  const int act_v = gcm_vgpr(act2);
  __syncthreads();
  const int nbt = (B + 31) / 32;
  const int n_out = (H2 + 31) / 32;
  bool bad = false;
#pragma unroll 1
  for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += gridDim.x * 4) {
    const int t = tile / nbt, b0 = (tile - t * nbt) * 32;
    const int cur = t < N ? t : N - 1;
    float4 hv[H4 / 2], ag[H4 / 2];
    load_rows<H4>(cH + (size_t)t * H1, (size_t)Tc * H1, b0, B, lane, hv);
#pragma unroll
    for (int i = 0; i < H4 / 2; ++i) ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    int n_valid = 0;
    for (int q = 0; q < hp.n; ++q) {
      int h = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) h = q == i ? hp.h[i] : h;
      if (h > cur) continue;
      ++n_valid;
      float4 sv[H4 / 2];
      load_rows<H4>(cH + (size_t)(t - h) * H1, (size_t)Tc * H1, b0, B, lane, sv);
#pragma unroll
      for (int i = 0; i < H4 / 2; ++i) add4(ag[i], sv[i]);
    }
    if (hp.self) {
#pragma unroll
      for (int i = 0; i < H4 / 2; ++i) add4(ag[i], hv[i]);
    }
    float* rec = rec0 + (size_t)t * rec_stride;
#pragma unroll
    for (int i = 0; i < H4 / 2; ++i) {
      const int e4 = lane + 64 * i, r = e4 / H4, c = (e4 % H4) * 4;
      float* a = sA + r * AS + c;
      a[0] = ag[i].x; a[1] = ag[i].y; a[2] = ag[i].z; a[3] = ag[i].w;
      a[HP] = hv[i].x; a[HP + 1] = hv[i].y; a[HP + 2] = hv[i].z; a[HP + 3] = hv[i].w;
      const int b = b0 + r;
      if (record && b < B) {                // v = agg2 | h1[cur]
        float* v = rec + lay.o_v + (size_t)b * 2 * H1 + c;
        *reinterpret_cast<float4*>(v) = ag[i];
        *reinterpret_cast<float4*>(v + H1) = hv[i];
      }
    }
    if (record && b0 + li < B && lh == 0) {   // the live list: the selected rows (ascending), row cur behind them
      const int b = b0 + li;
      int* live = reinterpret_cast<int*>(rec + lay.o_live) + (size_t)b * Tc;
      float* coef = rec + lay.o_coef + (size_t)b * Tc;
      int l = 0;
      for (int q = 0; q < hp.n; ++q) {
        int h = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) h = q == i ? hp.h[i] : h;
        if (h > cur) continue;
        live[l] = t - h;
        coef[l] = 1.f;
        ++l;
      }
      live[l] = t;
      coef[l] = hp.self ? 1.f : 0.f;
      int* hdr = reinterpret_cast<int*>(rec + lay.o_hdr) + 4 * b;
      hdr[0] = n_valid + 1; hdr[1] = n_valid; hdr[2] = cur; hdr[3] = t >= N ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int nt = 0; nt < n_out; ++nt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      mma32(acc, sA, AS, 1, sW + nt * 32, WS, 1, 2 * HP, li, lh);
      const int col = nt * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int b = b0 + acc_row(r, lh);
        const float v = gcm_act_sel(acc[r] + bias[nt & 1], act_v);
        if (b < B && col < H2) {
          mx_all[((size_t)t * B + b) * H2 + col] = v;
          rec[(size_t)b * H2 + col] = v;      // mx: the head of the record
          bad = bad || !isfinite(v);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (__any(bad) && lane == 0) atomicOr(flags, GCM_FLAG_NONFINITE);
}

static int collect_hops(const gcm_selector_desc* selectors, int n_selectors, int N, int T, Hops* out) {
  Hops hp{};
  int mx = 0;
  for (int i = 0; i < n_selectors; ++i) {
    const gcm_selector_desc& d = selectors[i];
    if (d.kind != GCM_SEL_TEMPORAL || d.direction != GCM_DIR_FORWARD) return 0;
    for (int k = 0; k < d.n_hops; ++k) {
      const int h = d.hops[k];
      if (h < 0 || h > N - 1) continue;       // (temporal.py:74: never valid in a graph of N nodes)
      if (h == 0) { hp.self = 1; continue; }
      bool seen = false;
      for (int q = 0; q < hp.n; ++q) seen = seen || hp.h[q] == h;
      if (seen) continue;
      if (hp.n == 16) return 0;
      hp.h[hp.n++] = h;
      mx = h > mx ? h : mx;
    }
  }
  if (T > N && N <= 2 * mx) return 0;         // a live row would have lost a source to the overflow roll
  for (int a = 0; a < hp.n; ++a)              // descending
    for (int b = a + 1; b < hp.n; ++b)
      if (hp.h[b] > hp.h[a]) { const int t = hp.h[a]; hp.h[a] = hp.h[b]; hp.h[b] = t; }
  *out = hp;
  return 1;
}

}  // namespace gcm_rtp

extern "C" int gcm_dense_rollout_tp_supported(const gcm_selector_desc* selectors, int n_selectors, int has_bias, int T,
                                              int N, int F, int H1, int H2) {
  if (T <= 0 || N <= 0 || (F != 32 && F != 64) || (H1 != 32 && H1 != 64) || H2 <= 0 || H2 > 64) return 0;
  if (has_bias & (GCM_GNN_HAS_DEG_TERM | GCM_GNN_HAS_PE_TABLE | GCM_GNN_RECORD_DX)) return 0;
  gcm_rtp::Hops hp;
  return gcm_rtp::collect_hops(selectors, n_selectors, N, T, &hp);
}

extern "C" int gcm_dense_rollout_tp_fwd(const float* obs, const gcm_selector_desc* selectors, int n_selectors,
                                        const float* params, int has_bias, int act1, int act2, float* nodes, float* adj,
                                        int64_t* count, float* cache_h1, float* cache_agg1, float* cache_nodes,
                                        float* records, size_t rec_stride, int record, float* mx_all, uint32_t* flags,
                                        int T, int B, int N, int Tc, int F, int H1, int H2, gcm_stream_t stream) {
  GCM_REQUIRE(obs && params && nodes && adj && count && cache_h1 && cache_agg1 && cache_nodes && records && mx_all &&
              flags && B > 0 && Tc >= T && (selectors || n_selectors == 0));
  if (!gcm_dense_rollout_tp_supported(selectors, n_selectors, has_bias, T, N, F, H1, H2)) return GCM_EUNSUPPORTED;
  if ((size_t)B * Tc * 64 >= ((size_t)1 << 40)) return GCM_EUNSUPPORTED;
  gcm_rtp::Hops hp;
  gcm_rtp::collect_hops(selectors, n_selectors, N, T, &hp);
  const gcm_rows::CachedLayout lay = gcm_rows::make_cached_layout(B, Tc, H1, H2);
  GCM_REQUIRE(rec_stride >= (record ? lay.total : gcm_rows::pad64((size_t)B * H2)));
  const int nbt = (B + 31) / 32;
  const long tiles_l = (long)T * nbt;
  if (tiles_l > 2147483647L) return GCM_EUNSUPPORTED;
  const int n_tiles = (int)tiles_l;
  const int cap = 2 * gcm_cu_count();
  const int grid = (n_tiles + 3) / 4 < cap ? (n_tiles + 3) / 4 : cap;
  hipStream_t s = (hipStream_t)stream;
#define GCM_TP1(a, b_)                                                                                            \
  if (F == a && H1 == b_) {                                                                                       \
    auto k1 = gcm_rtp::k_rollout_tp_l1<a, b_>;                                                                    \
    const size_t lds1 = sizeof(float) * ((size_t)2 * a * (b_ + 1) + (size_t)4 * 32 * (2 * a + 1));               \
    gcm_allow_dynamic_lds((const void*)k1, lds1);                                                                 \
    hipLaunchKernelGGL(k1, dim3(grid), dim3(256), lds1, s, obs, hp, params, act1, cache_h1, cache_agg1,           \
                       cache_nodes, nodes, adj, count, B, T, N, Tc, n_tiles);                                     \
  }
  GCM_TP1(32, 32) GCM_TP1(64, 32) GCM_TP1(32, 64) GCM_TP1(64, 64)
#undef GCM_TP1
  int rc = gcm_launch_status();
  if (rc) return rc;
#define GCM_TP2(b_)                                                                                               \
  if (H1 == b_) {                                                                                                 \
    auto k2 = gcm_rtp::k_rollout_tp_l2<b_>;                                                                       \
    const size_t lds2 = sizeof(float) * ((size_t)2 * b_ * 65 + (size_t)4 * 32 * (2 * b_ + 1));                   \
    gcm_allow_dynamic_lds((const void*)k2, lds2);                                                                 \
    hipLaunchKernelGGL(k2, dim3(grid), dim3(256), lds2, s, hp, params, F, act2, cache_h1, mx_all, records,        \
                       rec_stride, lay, record, flags, B, T, N, Tc, H2, n_tiles);                                 \
  }
  GCM_TP2(32) GCM_TP2(64)
#undef GCM_TP2
  return gcm_launch_status();
}
