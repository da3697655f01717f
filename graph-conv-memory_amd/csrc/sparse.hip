// SparseGCM packing / index kernels (src/gcm/sparse_gcm.py:72-212 and the util.py helpers
// it calls).  Everything here is HBM-bound index and row movement: the per-graph Python
// loops of the reference (util.py:181-187,202-207,222-230,442-451; temporal.py:35-36)
// become closed-form kernels over (graph, node) / (graph, edge) with coalesced row copies.
#include "gcm_common.h"

namespace {

// ---------------------------------------------------------------------------
// plan: exclusive prefix sums over B (one workgroup; B is at most a few thousand)
// ---------------------------------------------------------------------------
__device__ __forceinline__ int64_t shfl_up_i64(int64_t v, int d, int lane) {
  const int lo = __shfl_up((int)(v & 0xffffffffll), d), hi = __shfl_up((int)(v >> 32), d);
  const int64_t r = ((int64_t)hi << 32) | (uint32_t)lo;
  return lane >= d ? r : 0;
}
__device__ __forceinline__ int64_t shfl_xor_i64(int64_t v, int d) {
  const int lo = __shfl_xor((int)(v & 0xffffffffll), d), hi = __shfl_xor((int)(v >> 32), d);
  return ((int64_t)hi << 32) | (uint32_t)lo;
}

// Thread t owns `per` consecutive graphs (their values stay in registers when per <= 4: B <= 1024); the 256
// partial sums are scanned with wave shuffles (6 steps) and the four wave totals through LDS - the serial
// 256-element scan by thread 0 this replaced took 13 us per call, the fourth-largest kernel of a stepwise
// SparseGCM run.
__global__ __launch_bounds__(256) void k_sparse_plan(const int64_t* __restrict__ T,
                                                     const int64_t* __restrict__ taus,
                                                     int64_t* __restrict__ node_off,
                                                     int64_t* __restrict__ new_off,
                                                     int64_t* __restrict__ totals, int B) {
  __shared__ int64_t sW[4][4];   // per wave: sum a, sum c, max a, max c
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int per = (B + 255) / 256;
  const int lo = min(B, tid * per), hi = min(B, lo + per);
  constexpr int KEEP = 4;
  int64_t tv[KEEP], uv[KEEP];
#pragma unroll
  for (int i = 0; i < KEEP; ++i) {   // every load in flight before the first use
    const int b = lo + i;
    tv[i] = b < hi ? T[b] : 0;
    uv[i] = b < hi ? taus[b] : 0;
  }
  int64_t a = 0, c = 0, ma = 0, mc = 0;
#pragma unroll
  for (int i = 0; i < KEEP; ++i) {
    const int64_t n = tv[i] + uv[i];
    a += n;
    c += uv[i];
    ma = n > ma ? n : ma;
    mc = uv[i] > mc ? uv[i] : mc;
  }
  for (int b = lo + KEEP; b < hi; ++b) {
    const int64_t n = T[b] + taus[b];
    a += n;
    c += taus[b];
    ma = n > ma ? n : ma;
    mc = taus[b] > mc ? taus[b] : mc;
  }
  // inclusive scan inside the wave, maxima by butterfly
  int64_t ia = a, ic = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    ia += shfl_up_i64(ia, d, lane);
    ic += shfl_up_i64(ic, d, lane);
    const int64_t oa = shfl_xor_i64(ma, d), oc = shfl_xor_i64(mc, d);
    ma = oa > ma ? oa : ma;
    mc = oc > mc ? oc : mc;
  }
  if (lane == 63) { sW[wave][0] = ia; sW[wave][1] = ic; sW[wave][2] = ma; sW[wave][3] = mc; }
  __syncthreads();
  int64_t ba = 0, bc = 0, ta = 0, tc = 0, xa = 0, xc = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) { ba += sW[w][0]; bc += sW[w][1]; }
    ta += sW[w][0]; tc += sW[w][1];
    xa = sW[w][2] > xa ? sW[w][2] : xa;
    xc = sW[w][3] > xc ? sW[w][3] : xc;
  }
  if (tid == 0) {
    node_off[B] = ta; new_off[B] = tc;
    totals[0] = ta; totals[1] = tc; totals[2] = xa; totals[3] = xc;
  }
  a = ba + ia - a;   // exclusive prefix of this thread's first graph
  c = bc + ic - c;
#pragma unroll
  for (int i = 0; i < KEEP; ++i) {
    const int b = lo + i;
    if (b < hi) { node_off[b] = a; new_off[b] = c; }
    a += tv[i] + uv[i];
    c += uv[i];
  }
  for (int b = lo + KEEP; b < hi; ++b) {
    node_off[b] = a; new_off[b] = c;
    a += T[b] + taus[b];
    c += taus[b];
  }
}

// The packing kernels below move whole feature rows: VT = float4 (rows of F / 4 vectors, every
// pointer 16-byte aligned) moves them 16 bytes per lane, VT = float is the general form.  F counts VTs.
__device__ __forceinline__ float vt_zero(float) { return 0.f; }
__device__ __forceinline__ float4 vt_zero(float4) { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ bool vt_nonfinite(float v) { return !isfinite(v); }
__device__ __forceinline__ bool vt_nonfinite(const float4& v) {
  return !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w));
}
static inline bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr) {
  return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0;
}

// ---------------------------------------------------------------------------
// insert new observations into the node matrix (sparse_gcm.py:116-128) + adjoint
// ---------------------------------------------------------------------------
template <bool BWD, typename VT>
__global__ __launch_bounds__(256) void k_sparse_insert(
    const VT* __restrict__ src, const VT* __restrict__ x, const int64_t* __restrict__ T,
    const int64_t* __restrict__ taus, VT* __restrict__ dst, VT* __restrict__ g_x,
    uint32_t* __restrict__ flags, int N, int F, int t_pad, int rows_per_block) {
  const int b = blockIdx.y;
  const int64_t t0 = T[b];
  int64_t tau = taus[b];
  if (!BWD && blockIdx.x == 0 && threadIdx.x == 0 && (t0 < 0 || tau < 0 || t0 + tau > N))
    atomicOr(flags, GCM_FLAG_SPARSE_OVERFLOW);
  if (tau < 0) tau = 0;
  if (tau > t_pad) tau = t_pad;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
  const int total = (r1 - r0) * F;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int r = r0 + e / F, f = e % F;
    const int64_t k = r - t0;
    const bool fresh = k >= 0 && k < tau;
    const size_t at = ((size_t)b * N + r) * F + f;
    if (!BWD) {
      dst[at] = fresh ? x[((size_t)b * t_pad + k) * F + f] : (src ? src[at] : vt_zero(VT()));   // (src NULL: empty graphs)
    } else {
      dst[at] = fresh ? vt_zero(VT()) : src[at];
    }
  }
  if (BWD && g_x) {  // g_x[b, k] = g_nodes_out[b, T+k] for k < tau, else 0 (only for rows in range)
    const int k0 = blockIdx.x * rows_per_block, k1 = min(t_pad, k0 + rows_per_block);
    const int tot = max(0, k1 - k0) * F;
    for (int e = threadIdx.x; e < tot; e += blockDim.x) {
      const int k = k0 + e / F, f = e % F;
      const int64_t r = t0 + k;
      const bool ok = k < tau && r >= 0 && r < N;
      g_x[((size_t)b * t_pad + k) * F + f] = ok ? src[((size_t)b * N + r) * F + f] : vt_zero(VT());
    }
  }
}

// ---------------------------------------------------------------------------
// TemporalEdge closed form (sparse_edge_selectors/temporal.py:18-63)
// ---------------------------------------------------------------------------
struct Hops16 {
  int32_t h[16];
};

// edges of new node t (absolute index): hops h with t - h >= 0, and t > 0
__device__ __forceinline__ int temporal_degree(int64_t t, const Hops16& hops, int n_hops) {
  if (t <= 0) return 0;
  int d = 0;
  for (int i = 0; i < n_hops; ++i) d += (hops.h[i] >= 0 && t - hops.h[i] >= 0) ? 1 : 0;
  return d;
}

__global__ __launch_bounds__(256) void k_temporal_count(const int64_t* __restrict__ T,
                                                        const int64_t* __restrict__ taus,
                                                        Hops16 hops, int n_hops,
                                                        int64_t* __restrict__ edge_off, int B) {
  // one workgroup: per-graph counts in closed form (per hop: new nodes t >= max(h, 1)), then a
  // two-level scan (per-thread chunks, 256 partials by thread 0)
  extern __shared__ int64_t cnt[];
  __shared__ int64_t chunk_sum[256];
  const int per = (B + 255) / 256;
  const int lo = min(B, (int)threadIdx.x * per), hi = min(B, lo + per);
  int64_t run = 0;
  for (int b = lo; b < hi; ++b) {
    const int64_t t0 = T[b], tau = taus[b] > 0 ? taus[b] : 0;
    int64_t c = 0;
    for (int i = 0; i < n_hops; ++i) {
      const int64_t h = hops.h[i];
      if (h < 0) continue;
      const int64_t first = (h > 1 ? h : 1) > t0 ? (h > 1 ? h : 1) : t0;
      const int64_t n = t0 + tau - first;
      c += n > 0 ? n : 0;
    }
    cnt[b] = run;      // exclusive within the chunk
    run += c;
  }
  chunk_sum[threadIdx.x] = run;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t acc = 0;
    for (int i = 0; i < 256; ++i) {
      const int64_t t = chunk_sum[i];
      chunk_sum[i] = acc;
      acc += t;
    }
    edge_off[B] = acc;
  }
  __syncthreads();
  for (int b = lo; b < hi; ++b) edge_off[b] = cnt[b] + chunk_sum[threadIdx.x];
}

// SparseGCM called one node at a time (x [B, 1, F]): everything a call plans, in ONE launch instead of three and a
// pointer rebuild - k_sparse_plan's offsets / totals, k_temporal_count's edge offsets, T + taus, and the per-graph
// pointer of the merged COO list (old_bptr + edge offsets: the list only ever grows behind each graph's stored
// entries, so the pointer of the state a chain hands on needs no k_ptr_from_sorted over the entries).
// out: node_off [B+1] | new_off [B+1] | edge_off [B+1] | totals [4]  (the layout sparse_temporal_step reads back)
__device__ __forceinline__ int64_t temporal_count_of(int64_t t0, int64_t tau, const Hops16& hops, int n_hops) {
  int64_t c = 0;
  for (int i = 0; i < n_hops; ++i) {
    const int64_t h = hops.h[i];
    if (h < 0) continue;
    const int64_t first = (h > 1 ? h : 1) > t0 ? (h > 1 ? h : 1) : t0;
    const int64_t n = t0 + tau - first;
    c += n > 0 ? n : 0;
  }
  return c;
}
__global__ __launch_bounds__(256) void k_sparse_step_plan(const int64_t* __restrict__ T,
                                                          const int64_t* __restrict__ taus, Hops16 hops,
                                                          int n_hops, const int64_t* __restrict__ old_bptr,
                                                          int64_t* __restrict__ out, int64_t* __restrict__ T_out,
                                                          int64_t* __restrict__ merged_bptr, int B) {
  __shared__ int64_t sW[4][5];   // per wave: sum n, sum tau, sum edges, max n, max tau
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int per = (B + 255) / 256;
  const int lo = min(B, tid * per), hi = min(B, lo + per);
  int64_t* node_off = out;
  int64_t* new_off = out + (B + 1);
  int64_t* edge_off = new_off + (B + 1);
  int64_t* totals = edge_off + (B + 1);
  int64_t a = 0, c = 0, e = 0, ma = 0, mc = 0;
  for (int b = lo; b < hi; ++b) {
    const int64_t t0 = T[b], tau = taus[b] > 0 ? taus[b] : 0;
    a += t0 + taus[b];
    c += taus[b];
    e += temporal_count_of(t0, tau, hops, n_hops);
    ma = t0 + taus[b] > ma ? t0 + taus[b] : ma;
    mc = taus[b] > mc ? taus[b] : mc;
  }
  int64_t ia = a, ic = c, ie = e;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    ia += shfl_up_i64(ia, d, lane);
    ic += shfl_up_i64(ic, d, lane);
    ie += shfl_up_i64(ie, d, lane);
    const int64_t oa = shfl_xor_i64(ma, d), oc = shfl_xor_i64(mc, d);
    ma = oa > ma ? oa : ma;
    mc = oc > mc ? oc : mc;
  }
  if (lane == 63) { sW[wave][0] = ia; sW[wave][1] = ic; sW[wave][2] = ie; sW[wave][3] = ma; sW[wave][4] = mc; }
  __syncthreads();
  int64_t ba = 0, bc = 0, be = 0, ta = 0, tc = 0, te = 0, xa = 0, xc = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < wave) { ba += sW[w][0]; bc += sW[w][1]; be += sW[w][2]; }
    ta += sW[w][0]; tc += sW[w][1]; te += sW[w][2];
    xa = sW[w][3] > xa ? sW[w][3] : xa;
    xc = sW[w][4] > xc ? sW[w][4] : xc;
  }
  if (tid == 0) {
    node_off[B] = ta; new_off[B] = tc; edge_off[B] = te;
    totals[0] = ta; totals[1] = tc; totals[2] = xa; totals[3] = xc;
    if (merged_bptr) merged_bptr[B] = (old_bptr ? old_bptr[B] : 0) + te;
  }
  a = ba + ia - a;   // exclusive prefixes of this thread's first graph
  c = bc + ic - c;
  e = be + ie - e;
  for (int b = lo; b < hi; ++b) {
    const int64_t t0 = T[b], tau = taus[b] > 0 ? taus[b] : 0;
    node_off[b] = a; new_off[b] = c; edge_off[b] = e;
    if (merged_bptr) merged_bptr[b] = (old_bptr ? old_bptr[b] : 0) + e;
    if (T_out) T_out[b] = t0 + taus[b];
    a += t0 + taus[b];
    c += taus[b];
    e += temporal_count_of(t0, tau, hops, n_hops);
  }
}

__global__ __launch_bounds__(256) void k_temporal_fill(const int64_t* __restrict__ T,
                                                       const int64_t* __restrict__ taus,
                                                       Hops16 hops, int n_hops,
                                                       const int64_t* __restrict__ edge_off,
                                                       int64_t* __restrict__ indices, int64_t E,
                                                       float* __restrict__ vals) {
  // one workgroup per graph; thread k walks new node T+k.  Position inside the graph =
  // sum of degrees of earlier new nodes; degrees are < n_hops only for the first max(hops)
  // nodes of an episode, so the prefix is closed form after a short serial head.
  const int b = blockIdx.x;
  const int64_t t0 = T[b], tau = taus[b];
  const int64_t base = edge_off[b];
  for (int64_t k = threadIdx.x; k < tau; k += blockDim.x) {
    // prefix: sum_{k'<k} degree(t0+k')
    int64_t pos = 0;
    for (int i = 0; i < n_hops; ++i) {
      const int64_t h = hops.h[i];
      if (h < 0) continue;
      // nodes t in [t0, t0+k) with t >= max(h, 1)
      const int64_t first = (h > 1 ? h : 1) > t0 ? (h > 1 ? h : 1) : t0;
      const int64_t n = t0 + k - first;
      pos += n > 0 ? n : 0;
    }
    const int64_t t = t0 + k;
    if (t <= 0) continue;
    int64_t w = base + pos;
    for (int i = 0; i < n_hops; ++i) {  // hops descending -> sources ascending
      const int64_t h = hops.h[i];
      if (h < 0 || t - h < 0) continue;
      if (w < E) {
        indices[w] = b;
        indices[E + w] = t;
        indices[2 * E + w] = t - h;
        if (vals) vals[w] = 1.f;   // (the unit weights of the list, sparse_gcm.py:160-164)
      }
      ++w;
    }
  }
}

// ---------------------------------------------------------------------------
// Whole episodes from EMPTY graphs (T = 0 everywhere: the one-shot use of SparseGCM, cfg4): every index structure of
// the call in closed form, in ONE launch - the COO entries TemporalEdge adds (sparse_edge_selectors/temporal.py:18-63)
// with their unit weights, the flat (source, sink) list with CSR row pointers (util.py:287-304 + the sort inside
// coalesce), and the CSC view the backward's transpose gather reads.  Replaces k_temporal_fill, the fill of the
// values, k_edges_flat, k_ptr_from_sorted and k_csc_batched (33 us of launches at cfg4) - nothing here depends on
// data, only on taus and the hop set H (distinct, descending, h >= 1):
//   node t of graph b (offset o = node_off[b], edge base e0 = edge_off[b], tau = taus[b]) has the sources t - h, h <= t:
//     CSR   row_ptr[o + t] = e0 + P(t),  P(t) = sum_h max(0, t - h);  its entries in hop order (sources ascending)
//     CSC   col_ptr[o + s] = e0 + Q(s),  Q(s) = sum_h min(s, max(0, tau - h));  the entries of source s are the
//           sinks s + h < tau, h ascending;  perm = the CSR position of (s + h, s) = e0 + P(s + h) + #{h' > h, h' <= s + h}
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_temporal_structure(
    const int64_t* __restrict__ taus, Hops16 hops, int n_hops, const int64_t* __restrict__ node_off,
    const int64_t* __restrict__ edge_off, int64_t* __restrict__ coo, float* __restrict__ vals,
    int64_t* __restrict__ edge_index, int64_t* __restrict__ row_ptr, int64_t* __restrict__ col_ptr,
    int64_t* __restrict__ rows, int64_t* __restrict__ perm, int64_t E, int64_t M, int B) {
  const int b = blockIdx.x;
  const int64_t tau = taus[b] > 0 ? taus[b] : 0;
  const int64_t o = node_off[b], e0 = edge_off[b];
  if (b == 0 && threadIdx.x == 0) {
    row_ptr[M] = E;
    if (col_ptr) col_ptr[M] = E;
  }
  for (int64_t t = threadIdx.x; t < tau; t += blockDim.x) {
    if (o + t >= M) break;
    int64_t P = 0, Q = 0;
    for (int i = 0; i < n_hops; ++i) {
      const int64_t h = hops.h[i];
      P += t > h ? t - h : 0;
      const int64_t lim = tau > h ? tau - h : 0;
      Q += t < lim ? t : lim;
    }
    row_ptr[o + t] = e0 + P;
    int64_t w = e0 + P;
    for (int i = 0; i < n_hops; ++i) {   // hops descending -> sources ascending
      const int64_t h = hops.h[i];
      if (h > t) continue;
      if (w < E) {
        coo[w] = b;
        coo[E + w] = t;
        coo[2 * E + w] = t - h;
        if (vals) vals[w] = 1.f;
        edge_index[w] = o + t - h;
        edge_index[E + w] = o + t;
      }
      ++w;
    }
    if (col_ptr) {
      col_ptr[o + t] = e0 + Q;
      int64_t k = e0 + Q;
      for (int i = n_hops - 1; i >= 0; --i) {   // hops ascending -> sinks ascending
        const int64_t h = hops.h[i], sink = t + h;
        if (sink >= tau) continue;
        int64_t Ps = 0;
        int above = 0;
        for (int j = 0; j < n_hops; ++j) {
          const int64_t hj = hops.h[j];
          Ps += sink > hj ? sink - hj : 0;
          above += (hj > h && hj <= sink) ? 1 : 0;
        }
        if (k < E) {
          rows[k] = o + sink;
          perm[k] = e0 + Ps + above;
        }
        ++k;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// flatten nodes (util.py:426-452) + adjoint
// ---------------------------------------------------------------------------
template <bool BWD, typename VT>
__global__ __launch_bounds__(256) void k_flatten(const VT* __restrict__ src,
                                                 const int64_t* __restrict__ T,
                                                 const int64_t* __restrict__ taus,
                                                 const int64_t* __restrict__ node_off,
                                                 VT* __restrict__ dst, int N, int F, int64_t M,
                                                 int rows_per_block) {
  const int b = blockIdx.y;
  int64_t live = T[b] + taus[b];
  live = live < 0 ? 0 : (live > N ? N : live);
  const int64_t off = node_off[b];
  const int r0 = blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
  const int total = (r1 - r0) * F;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int r = r0 + e / F, f = e % F;
    const bool ok = r < live && off + r < M;
    if (!BWD) {
      if (ok) dst[(size_t)(off + r) * F + f] = src[((size_t)b * N + r) * F + f];
    } else {
      dst[((size_t)b * N + r) * F + f] = ok ? src[(size_t)(off + r) * F + f] : vt_zero(VT());
    }
  }
}

// ---------------------------------------------------------------------------
// COO (batch, sink, source) -> flat (source, sink) edge list + CSR row_ptr by destination
// ---------------------------------------------------------------------------
__global__ void k_edges_flat(const int64_t* __restrict__ coo, const int64_t* __restrict__ node_off,
                             int64_t* __restrict__ edge_index, uint32_t* __restrict__ flags,
                             int64_t E, int B) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool bad = false;
  if (e < E) {
    int64_t b = coo[e];
    b = b < 0 ? 0 : (b >= B ? B - 1 : b);
    const int64_t sink = coo[E + e], src = coo[2 * E + e];
    const int64_t off = node_off[b];
    edge_index[e] = src + off;
    edge_index[E + e] = sink + off;
    bad = !(src < sink);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flags, GCM_FLAG_ACAUSAL);
}

__global__ void k_ptr_from_sorted(const int64_t* __restrict__ keys, int64_t* __restrict__ ptr,
                                  int64_t E, int64_t M) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > M) return;
  int64_t lo = 0, hi = E;  // first position with key >= r
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < r) lo = mid + 1; else hi = mid;
  }
  ptr[r] = lo;
}

// ---------------------------------------------------------------------------
// k-hop mask (sparse_gcm.py:192-198)
// ---------------------------------------------------------------------------
__global__ void k_khop_seed(const int64_t* __restrict__ node_off, const int64_t* __restrict__ T,
                            const int64_t* __restrict__ taus, uint8_t* __restrict__ mask,
                            uint8_t* __restrict__ frontier, int64_t M, int t_max) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= t_max || k >= taus[b]) return;
  const int64_t i = node_off[b] + T[b] + k;
  if (i >= 0 && i < M) {
    mask[i] = 1;
    frontier[i] = 1;
  }
}

// one BFS round against the edges: every source of a frontier row joins mask and next
__global__ void k_khop_round(const int64_t* __restrict__ row_ptr, const int64_t* __restrict__ col,
                             const uint8_t* __restrict__ frontier, uint8_t* __restrict__ next,
                             uint8_t* __restrict__ mask, int64_t M) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M || !frontier[i]) return;
  for (int64_t e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
    const int64_t s = col[e];
    if (s >= 0 && s < M) {
      next[s] = 1;
      mask[s] = 1;
    }
  }
}

// ---------------------------------------------------------------------------
// output rows (sparse_gcm.py:176-208) + adjoint
// ---------------------------------------------------------------------------
template <typename VT>
__global__ void k_extract_fwd(const VT* __restrict__ feats, const int64_t* __restrict__ T,
                              const int64_t* __restrict__ taus,
                              const int64_t* __restrict__ node_off, VT* __restrict__ out,
                              uint32_t* __restrict__ flags, int B, int t_pad, int H, int64_t M) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool nonfinite = false;
  if (i < (int64_t)B * t_pad * H) {
    const int f = i % H;
    const int k = (i / H) % t_pad;
    const int b = i / ((int64_t)H * t_pad);
    VT v = vt_zero(VT());
    const int64_t row = node_off[b] + T[b] + k;
    if (k < taus[b] && row >= 0 && row < M) {
      v = feats[(size_t)row * H + f];
      nonfinite = vt_nonfinite(v);
    }
    out[i] = v;
  }
  if (__any(nonfinite) && (threadIdx.x & 63) == 0) atomicOr(flags, GCM_FLAG_NONFINITE);
}

template <typename VT>
__global__ void k_extract_bwd(const VT* __restrict__ g_out, const int64_t* __restrict__ T,
                              const int64_t* __restrict__ taus,
                              const int64_t* __restrict__ node_off, VT* __restrict__ g_feats,
                              int B, int t_pad, int H, int64_t M) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * t_pad * H) return;
  const int f = i % H;
  const int k = (i / H) % t_pad;
  const int b = i / ((int64_t)H * t_pad);
  const int64_t row = node_off[b] + T[b] + k;
  if (k < taus[b] && row >= 0 && row < M) g_feats[(size_t)row * H + f] = g_out[i];
}

// util.pack_hidden (util.py:323-351): one thread per COO entry
__global__ void k_pack_hidden(const int64_t* __restrict__ coo, const float* __restrict__ values,
                              const int64_t* __restrict__ batch_ptr,
                              int64_t* __restrict__ dense_edges, float* __restrict__ dense_weights,
                              uint32_t* __restrict__ flags, int64_t E, int B, int max_edges) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool over = false;
  if (e < E) {
    int64_t b = coo[e];
    b = b < 0 ? 0 : (b >= B ? B - 1 : b);
    const int64_t pos = e - batch_ptr[b];
    // the reference requires count < max_edges (strict)
    over = batch_ptr[b + 1] - batch_ptr[b] >= max_edges;
    if (pos >= 0 && pos < max_edges) {
      dense_edges[((size_t)b * 2 + 0) * max_edges + pos] = coo[E + e];
      dense_edges[((size_t)b * 2 + 1) * max_edges + pos] = coo[2 * E + e];
      dense_weights[(size_t)b * max_edges + pos] = values[e];
    }
  }
  if (__any(over) && (threadIdx.x & 63) == 0) atomicOr(flags, GCM_FLAG_PACK_OVERFLOW);
}

Hops16 pack_hops(const int32_t* hops_host, int n_hops) {
  Hops16 h;
  for (int i = 0; i < 16; ++i) h.h[i] = i < n_hops ? hops_host[i] : -1;
  return h;
}

}  // namespace

extern "C" int gcm_sparse_plan(const int64_t* T, const int64_t* taus, int64_t* node_off,
                               int64_t* new_off, int64_t* totals, int B, gcm_stream_t stream) {
  GCM_REQUIRE(T && taus && node_off && new_off && totals && B > 0);
  hipLaunchKernelGGL(k_sparse_plan, dim3(1), dim3(256), 0, (hipStream_t)stream, T, taus, node_off,
                     new_off, totals, B);
  return gcm_launch_status();
}

extern "C" int gcm_sparse_insert_fwd(const float* nodes_in, const float* x, const int64_t* T,
                                     const int64_t* taus, float* nodes_out, uint32_t* flags, int B,
                                     int N, int F, int t_pad, gcm_stream_t stream) {
  GCM_REQUIRE(x && T && taus && nodes_out && flags);   // (nodes_in may be NULL: all zeros, not read)
  GCM_REQUIRE(B > 0 && N > 0 && F > 0 && t_pad > 0);
  if (B > 65535) return GCM_EUNSUPPORTED;
  const int rpb = 32;
  if (F % 4 == 0 && aligned16(nodes_in, x, nodes_out))
    hipLaunchKernelGGL((k_sparse_insert<false, float4>), dim3((N + rpb - 1) / rpb, B), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)nodes_in, (const float4*)x, T, taus,
                       (float4*)nodes_out, (float4*)nullptr, flags, N, F / 4, t_pad, rpb);
  else
    hipLaunchKernelGGL((k_sparse_insert<false, float>), dim3((N + rpb - 1) / rpb, B), dim3(256), 0,
                       (hipStream_t)stream, nodes_in, x, T, taus, nodes_out, (float*)nullptr, flags,
                       N, F, t_pad, rpb);
  return gcm_launch_status();
}

extern "C" int gcm_sparse_insert_bwd(const float* g_nodes_out, const int64_t* T,
                                     const int64_t* taus, float* g_nodes_in, float* g_x, int B,
                                     int N, int F, int t_pad, gcm_stream_t stream) {
  GCM_REQUIRE(g_nodes_out && T && taus && g_nodes_in);
  GCM_REQUIRE(B > 0 && N > 0 && F > 0 && t_pad > 0);
  if (B > 65535) return GCM_EUNSUPPORTED;
  const int rpb = 32;
  const int blocks = (max(N, t_pad) + rpb - 1) / rpb;
  if (F % 4 == 0 && aligned16(g_nodes_out, g_nodes_in, g_x))
    hipLaunchKernelGGL((k_sparse_insert<true, float4>), dim3(blocks, B), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)g_nodes_out, (const float4*)nullptr, T, taus, (float4*)g_nodes_in,
                       (float4*)g_x, (uint32_t*)nullptr, N, F / 4, t_pad, rpb);
  else
    hipLaunchKernelGGL((k_sparse_insert<true, float>), dim3(blocks, B), dim3(256), 0, (hipStream_t)stream,
                       g_nodes_out, (const float*)nullptr, T, taus, g_nodes_in, g_x,
                       (uint32_t*)nullptr, N, F, t_pad, rpb);
  return gcm_launch_status();
}

extern "C" int gcm_sparse_temporal_count(const int64_t* T, const int64_t* taus,
                                         const int32_t* hops_host, int n_hops, int64_t* edge_off,
                                         int B, gcm_stream_t stream) {
  GCM_REQUIRE(T && taus && hops_host && edge_off && B > 0 && n_hops > 0);
  if (n_hops > 16 || (size_t)B * sizeof(int64_t) > 64 * 1024) return GCM_EUNSUPPORTED;
  for (int i = 1; i < n_hops; ++i) GCM_REQUIRE(hops_host[i] < hops_host[i - 1]);
  hipLaunchKernelGGL(k_temporal_count, dim3(1), dim3(256), (size_t)B * sizeof(int64_t),
                     (hipStream_t)stream, T, taus, pack_hops(hops_host, n_hops), n_hops, edge_off,
                     B);
  return gcm_launch_status();
}

extern "C" int gcm_sparse_step_plan(const int64_t* T, const int64_t* taus, const int32_t* hops_host, int n_hops,
                                    const int64_t* old_bptr, int64_t* plan, int64_t* T_out, int64_t* merged_bptr,
                                    int B, gcm_stream_t stream) {
  GCM_REQUIRE(T && taus && hops_host && plan && B > 0 && n_hops > 0);
  if (n_hops > 16) return GCM_EUNSUPPORTED;
  for (int i = 1; i < n_hops; ++i) GCM_REQUIRE(hops_host[i] < hops_host[i - 1]);
  hipLaunchKernelGGL(k_sparse_step_plan, dim3(1), dim3(256), 0, (hipStream_t)stream, T, taus,
                     pack_hops(hops_host, n_hops), n_hops, old_bptr, plan, T_out, merged_bptr, B);
  return gcm_launch_status();
}

extern "C" int gcm_sparse_temporal_fill(const int64_t* T, const int64_t* taus,
                                        const int32_t* hops_host, int n_hops,
                                        const int64_t* edge_off, int64_t* indices, int64_t E,
                                        int B, gcm_stream_t stream) {
  return gcm_sparse_temporal_fill_vals(T, taus, hops_host, n_hops, edge_off, indices, nullptr, E, B, stream);
}

extern "C" int gcm_sparse_temporal_fill_vals(const int64_t* T, const int64_t* taus, const int32_t* hops_host,
                                             int n_hops, const int64_t* edge_off, int64_t* indices, float* vals,
                                             int64_t E, int B, gcm_stream_t stream) {
  GCM_REQUIRE(T && taus && hops_host && edge_off && B > 0 && n_hops > 0 && E >= 0);
  if (n_hops > 16) return GCM_EUNSUPPORTED;
  if (E == 0) return GCM_OK;
  GCM_REQUIRE(indices);
  hipLaunchKernelGGL(k_temporal_fill, dim3(B), dim3(256), 0, (hipStream_t)stream, T, taus,
                     pack_hops(hops_host, n_hops), n_hops, edge_off, indices, E, vals);
  return gcm_launch_status();
}

extern "C" int gcm_sparse_temporal_structure(const int64_t* taus, const int32_t* hops_host, int n_hops,
                                             const int64_t* node_off, const int64_t* edge_off, int64_t* coo,
                                             float* vals, int64_t* edge_index, int64_t* row_ptr, int64_t* col_ptr,
                                             int64_t* rows, int64_t* perm, int64_t E, int64_t M, int B,
                                             gcm_stream_t stream) {
  GCM_REQUIRE(taus && hops_host && node_off && edge_off && row_ptr && B > 0 && n_hops > 0 && E >= 0 && M >= 0);
  GCM_REQUIRE((coo && edge_index) || E == 0);
  GCM_REQUIRE(!col_ptr || E == 0 || (rows && perm));
  if (n_hops > 16) return GCM_EUNSUPPORTED;
  for (int i = 0; i < n_hops; ++i)   // distinct, descending, no self loops
    if (hops_host[i] < 1 || (i > 0 && hops_host[i] >= hops_host[i - 1])) return GCM_EUNSUPPORTED;
  hipLaunchKernelGGL(k_temporal_structure, dim3(B), dim3(256), 0, (hipStream_t)stream, taus,
                     pack_hops(hops_host, n_hops), n_hops, node_off, edge_off, coo, vals, edge_index, row_ptr, col_ptr,
                     rows, perm, E, M, B);
  return gcm_launch_status();
}

extern "C" int gcm_sparse_flatten_fwd(const float* nodes, const int64_t* T, const int64_t* taus,
                                      const int64_t* node_off, float* flat, int B, int N, int F,
                                      int64_t M, gcm_stream_t stream) {
  GCM_REQUIRE(nodes && T && taus && node_off && B > 0 && N > 0 && F > 0 && M >= 0);
  if (B > 65535) return GCM_EUNSUPPORTED;
  if (M == 0) return GCM_OK;
  GCM_REQUIRE(flat);
  const int rpb = 32;
  if (F % 4 == 0 && aligned16(nodes, flat))
    hipLaunchKernelGGL((k_flatten<false, float4>), dim3((N + rpb - 1) / rpb, B), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)nodes, T, taus, node_off, (float4*)flat, N, F / 4,
                       M, rpb);
  else
    hipLaunchKernelGGL((k_flatten<false, float>), dim3((N + rpb - 1) / rpb, B), dim3(256), 0,
                       (hipStream_t)stream, nodes, T, taus, node_off, flat, N, F, M, rpb);
  return gcm_launch_status();
}

extern "C" int gcm_sparse_flatten_bwd(const float* g_flat, const int64_t* T, const int64_t* taus,
                                      const int64_t* node_off, float* g_nodes, int B, int N, int F,
                                      int64_t M, gcm_stream_t stream) {
  GCM_REQUIRE(T && taus && node_off && g_nodes && B > 0 && N > 0 && F > 0 && M >= 0);
  GCM_REQUIRE(g_flat || M == 0);
  if (B > 65535) return GCM_EUNSUPPORTED;
  const int rpb = 32;
  if (F % 4 == 0 && aligned16(g_flat, g_nodes))
    hipLaunchKernelGGL((k_flatten<true, float4>), dim3((N + rpb - 1) / rpb, B), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)g_flat, T, taus, node_off, (float4*)g_nodes, N,
                       F / 4, M, rpb);
  else
    hipLaunchKernelGGL((k_flatten<true, float>), dim3((N + rpb - 1) / rpb, B), dim3(256), 0,
                       (hipStream_t)stream, g_flat, T, taus, node_off, g_nodes, N, F, M, rpb);
  return gcm_launch_status();
}

__global__ void k_coo_merge(const int64_t* __restrict__ old_idx, const int64_t* __restrict__ new_idx,
                            const float* __restrict__ old_val, const float* __restrict__ new_val,
                            const int64_t* __restrict__ old_bptr, const int64_t* __restrict__ new_bptr,
                            int64_t* __restrict__ out_idx, float* __restrict__ out_val,
                            int64_t* __restrict__ perm, uint32_t* __restrict__ flags, int64_t Ea,
                            int64_t Eb) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t E = Ea + Eb;
  if (e >= E) return;
  int64_t b, snk, src, pos;
  float v = 1.f;
  if (e < Ea) {   // a stored entry: shifted by the new entries of the graphs before its own
    b = old_idx[e]; snk = old_idx[Ea + e]; src = old_idx[2 * Ea + e];
    pos = e + new_bptr[b];
    if (old_val) v = old_val[e];
  } else {        // a new entry: behind every stored entry of its own graph
    const int64_t j = e - Ea;
    b = new_idx[j]; snk = new_idx[Eb + j]; src = new_idx[2 * Eb + j];
    pos = j + old_bptr[b + 1];
    if (new_val) v = new_val[j];
    const int64_t last = old_bptr[b + 1] - 1;
    if (last >= old_bptr[b]) {
      const int64_t ls = old_idx[Ea + last], lc = old_idx[2 * Ea + last];
      if (snk < ls || (snk == ls && src <= lc)) atomicOr(flags, GCM_FLAG_MERGE_ORDER);
    }
  }
  out_idx[pos] = b; out_idx[E + pos] = snk; out_idx[2 * E + pos] = src;
  if (out_val) out_val[pos] = v;
  if (perm) perm[pos] = e;
}

// A call of a one-node-per-call chain (gcm_sparse_step_plan): TemporalEdge's new entries (sparse_edge_selectors/
// temporal.py:18-63: sink T[b], sources T[b] - h ascending) generated straight into their places of the merged
// list (sparse_gcm.py:132-139: stored entries of graph b, then its new ones) - k_temporal_fill + k_coo_merge as one
// launch, with the number of new entries read on the DEVICE (plan: edge_off[B]) so that the host can enqueue the
// whole call before it reads any size back.  One thread per slot of the upper bound Ea + max_new; rows of the
// output are E = Ea + edge_off[B] apart.  Unit weights written alongside.
__global__ void k_chain_edges(const int64_t* __restrict__ old_idx, const int64_t* __restrict__ old_bptr,
                              const int64_t* __restrict__ edge_off, const int64_t* __restrict__ T,
                              const int64_t* __restrict__ taus, Hops16 hops, int n_hops,
                              int64_t* __restrict__ out_idx, float* __restrict__ out_val, int64_t Ea, int B) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t Eb = edge_off[B], E = Ea + Eb;
  if (e >= E) return;
  int64_t b, snk, src, pos;
  if (e < Ea) {
    b = old_idx[e]; snk = old_idx[Ea + e]; src = old_idx[2 * Ea + e];
    pos = e + edge_off[b];
  } else {
    const int64_t j = e - Ea;
    int lo = 0, hi = B;                 // the graph with edge_off[b] <= j < edge_off[b + 1]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (edge_off[mid] <= j) lo = mid; else hi = mid;
    }
    b = lo;
    int64_t k = j - edge_off[b];        // its k-th new entry: one new node (taus in {0, 1}), hops descending
    const int64_t t = T[b];
    snk = t; src = -1;
    for (int i = 0; i < n_hops; ++i) {
      const int64_t h = hops.h[i];
      if (h < 0 || t - h < 0 || t <= 0) continue;
      if (k == 0) { src = t - h; break; }
      --k;
    }
    pos = j + (old_bptr ? old_bptr[b + 1] : 0);
  }
  out_idx[pos] = b; out_idx[E + pos] = snk; out_idx[2 * E + pos] = src;
  out_val[pos] = 1.f;
}

extern "C" int gcm_sparse_chain_edges(const int64_t* old_idx, const int64_t* old_bptr, const int64_t* plan,
                                      const int64_t* T, const int64_t* taus, const int32_t* hops_host, int n_hops,
                                      int64_t* out_idx, float* out_val, int64_t Ea, int64_t max_new, int B,
                                      gcm_stream_t stream) {
  GCM_REQUIRE(plan && T && taus && hops_host && out_idx && out_val && (old_idx || Ea == 0) && (old_bptr || Ea == 0));
  GCM_REQUIRE(Ea >= 0 && max_new >= 0 && B > 0 && n_hops > 0);
  if (n_hops > 16) return GCM_EUNSUPPORTED;
  if (Ea + max_new == 0) return GCM_OK;
  hipLaunchKernelGGL(k_chain_edges, dim3((unsigned)((Ea + max_new + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     old_idx, old_bptr, plan + 2 * (size_t)(B + 1), T, taus, pack_hops(hops_host, n_hops), n_hops,
                     out_idx, out_val, Ea, B);
  return gcm_launch_status();
}

extern "C" int gcm_coo_merge_segments(const int64_t* old_idx, const int64_t* new_idx,
                                      const float* old_val, const float* new_val,
                                      const int64_t* old_bptr, const int64_t* new_bptr,
                                      int64_t* out_idx, float* out_val, int64_t* perm,
                                      uint32_t* flags, int64_t Ea, int64_t Eb, int B,
                                      gcm_stream_t stream) {
  GCM_REQUIRE(old_bptr && new_bptr && out_idx && flags && (old_idx || Ea == 0) && (new_idx || Eb == 0));
  GCM_REQUIRE(Ea >= 0 && Eb >= 0 && B > 0);
  const int64_t E = Ea + Eb;
  if (E == 0) return GCM_OK;
  hipLaunchKernelGGL(k_coo_merge, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     old_idx, new_idx, old_val, new_val, old_bptr, new_bptr, out_idx, out_val, perm, flags,
                     Ea, Eb);
  return gcm_launch_status();
}

// ---------------------------------------------------------------------------
// CSC view of a batched CSR edge list without a sort (the backward's transpose gather).
// The flat edge list of SparseGCM is grouped by graph and an edge never leaves its graph, so the
// CSC entries of graph b's sources occupy exactly graph b's own slice [row_ptr[n0], row_ptr[n1]) of
// the edge array: a counting sort per graph, no global scan.  ONE WAVE owns a graph - counters of its
// (local) sources in LDS, edges visited in CSR order 64 at a time - so the result does not depend on
// the timing of other waves; entries of a column keep ascending sink order except among the edges of
// one 64-edge chunk that share a source, where the LDS unit's fixed lane order decides.
// ---------------------------------------------------------------------------
#define GCM_CSC_WAVES 4
__global__ __launch_bounds__(64 * GCM_CSC_WAVES) void k_csc_batched(
    const int64_t* __restrict__ row_ptr, const int64_t* __restrict__ col,
    const int64_t* __restrict__ dst, const int64_t* __restrict__ node_off,
    int64_t* __restrict__ col_ptr, int64_t* __restrict__ rows, int64_t* __restrict__ perm, int B,
    int64_t M, int n_cap) {
  extern __shared__ int s_cnt[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * GCM_CSC_WAVES + wave;
  if (b >= B) return;   // (no workgroup barrier below: waves are independent)
  int* cnt = s_cnt + (size_t)wave * n_cap;
  const int64_t n0 = node_off[b], n1 = node_off[b + 1];
  const int n = min((int)(n1 - n0), n_cap);
  const int64_t e0 = row_ptr[n0], e1 = n > 0 ? row_ptr[n1] : e0;
  // (a source outside the graph's node range cannot come from SparseGCM; clamped, never out of bounds)
  auto local = [&](int64_t e) { const int64_t j = col[e] - n0; return (int)(j < 0 ? 0 : (j >= n ? n - 1 : j)); };
  for (int j = lane; j < n; j += 64) cnt[j] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int64_t e = e0 + lane; e < e1; e += 64) atomicAdd(&cnt[local(e)], 1);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // exclusive scan of the counters -> column starts (global positions), 64 at a time
  int carry = 0;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    const int c = j < n ? cnt[j] : 0;
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    const int start = carry + incl - c;
    if (j < n) {
      cnt[j] = start;
      col_ptr[n0 + j] = e0 + start;
    }
    carry += __shfl(incl, 63);
  }
  if (b == B - 1 && lane == 0) col_ptr[M] = e1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // fill, in CSR order
  for (int64_t base = e0; base < e1; base += 64) {
    const int64_t e = base + lane;
    if (e < e1) {
      const int pos = atomicAdd(&cnt[local(e)], 1);
      rows[e0 + pos] = dst[e];
      perm[e0 + pos] = e;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

extern "C" int gcm_csc_from_csr_batched(const int64_t* row_ptr, const int64_t* col,
                                        const int64_t* dst, const int64_t* node_off,
                                        int64_t* col_ptr, int64_t* rows, int64_t* perm, int B,
                                        int64_t M, int64_t E, int max_nodes_per_graph,
                                        gcm_stream_t stream) {
  GCM_REQUIRE(row_ptr && node_off && col_ptr && B > 0 && M >= 0 && E >= 0 && max_nodes_per_graph > 0);
  GCM_REQUIRE((col && dst && rows && perm) || E == 0);
  if (max_nodes_per_graph > 8192) return GCM_EUNSUPPORTED;
  const size_t lds = sizeof(int) * (size_t)max_nodes_per_graph * GCM_CSC_WAVES;
  gcm_allow_dynamic_lds((const void*)k_csc_batched, lds);
  hipLaunchKernelGGL(k_csc_batched, dim3((unsigned)((B + GCM_CSC_WAVES - 1) / GCM_CSC_WAVES)),
                     dim3(64 * GCM_CSC_WAVES), lds, (hipStream_t)stream, row_ptr, col, dst, node_off, col_ptr,
                     rows, perm, B, M, max_nodes_per_graph);
  return gcm_launch_status();
}

extern "C" int gcm_ptr_from_sorted(const int64_t* keys, int64_t* ptr, int64_t E, int64_t M,
                                   gcm_stream_t stream) {
  GCM_REQUIRE(ptr && E >= 0 && M >= 0 && (keys || E == 0));
  hipLaunchKernelGGL(k_ptr_from_sorted, dim3((unsigned)((M + 1 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, keys, ptr, E, M);
  return gcm_launch_status();
}

extern "C" int gcm_sparse_edges_to_csr(const int64_t* coo, const int64_t* node_off,
                                       int64_t* edge_index, int64_t* row_ptr, uint32_t* flags,
                                       int64_t E, int64_t M, int B, gcm_stream_t stream) {
  GCM_REQUIRE(node_off && row_ptr && flags && E >= 0 && M >= 0 && B > 0);
  GCM_REQUIRE((coo && edge_index) || E == 0);
  if (E > 0) {
    hipLaunchKernelGGL(k_edges_flat, dim3((unsigned)((E + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, coo, node_off, edge_index, flags, E, B);
  }
  // sinks ascend in (batch, sink, source) order once the per-graph offsets are added
  return gcm_ptr_from_sorted(edge_index ? edge_index + E : nullptr, row_ptr, E, M, stream);
}

extern "C" int gcm_khop_mask(const int64_t* row_ptr, const int64_t* col, const int64_t* node_off,
                             const int64_t* T, const int64_t* taus, int hops, uint8_t* mask,
                             uint8_t* scratch, int64_t M, int B, int t_pad, gcm_stream_t stream) {
  GCM_REQUIRE(row_ptr && node_off && T && taus && mask && scratch && M > 0 && B > 0 && hops >= 0);
  GCM_REQUIRE(t_pad > 0 && (col || hops == 0));
  if (B > 65535) return GCM_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  uint8_t* fr[2] = {scratch, scratch + M};
  hipError_t e = hipMemsetAsync(mask, 0, (size_t)M, s);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(scratch, 0, 2 * (size_t)M, s);
  if (e != hipSuccess) return (int)e;
  const int t_cap = t_pad;
  hipLaunchKernelGGL(k_khop_seed, dim3((unsigned)((t_cap + 127) / 128), B), dim3(128), 0, s,
                     node_off, T, taus, mask, fr[0], M, (int)t_cap);
  for (int h = 0; h < hops; ++h) {
    hipLaunchKernelGGL(k_khop_round, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, row_ptr,
                       col, fr[h & 1], fr[(h + 1) & 1], mask, M);
    if (h + 1 < hops) {
      e = hipMemsetAsync(fr[h & 1], 0, (size_t)M, s);
      if (e != hipSuccess) return (int)e;
    }
  }
  return gcm_launch_status();
}

extern "C" int gcm_sparse_extract_fwd(const float* feats, const int64_t* T, const int64_t* taus,
                                      const int64_t* node_off, float* out, uint32_t* flags, int B,
                                      int t_pad, int H, int64_t M, gcm_stream_t stream) {
  GCM_REQUIRE(T && taus && node_off && out && flags && B > 0 && t_pad > 0 && H > 0 && M >= 0);
  GCM_REQUIRE(feats || M == 0);
  if (H % 4 == 0 && aligned16(feats, out)) {
    const int64_t total = (int64_t)B * t_pad * (H / 4);
    hipLaunchKernelGGL(k_extract_fwd<float4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)feats, T, taus, node_off, (float4*)out, flags, B,
                       t_pad, H / 4, M);
  } else {
    const int64_t total = (int64_t)B * t_pad * H;
    hipLaunchKernelGGL(k_extract_fwd<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, feats, T, taus, node_off, out, flags, B, t_pad, H, M);
  }
  return gcm_launch_status();
}

extern "C" int gcm_sparse_extract_bwd(const float* g_out, const int64_t* T, const int64_t* taus,
                                      const int64_t* node_off, float* g_feats, int B, int t_pad,
                                      int H, int64_t M, gcm_stream_t stream) {
  GCM_REQUIRE(g_out && T && taus && node_off && B > 0 && t_pad > 0 && H > 0 && M >= 0);
  if (M == 0) return GCM_OK;
  GCM_REQUIRE(g_feats);
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(g_feats, 0, (size_t)M * H * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  if (H % 4 == 0 && aligned16(g_out, g_feats)) {
    const int64_t total = (int64_t)B * t_pad * (H / 4);
    hipLaunchKernelGGL(k_extract_bwd<float4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       (const float4*)g_out, T, taus, node_off, (float4*)g_feats, B, t_pad, H / 4, M);
  } else {
    const int64_t total = (int64_t)B * t_pad * H;
    hipLaunchKernelGGL(k_extract_bwd<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g_out,
                       T, taus, node_off, g_feats, B, t_pad, H, M);
  }
  return gcm_launch_status();
}

extern "C" int gcm_pack_hidden(const int64_t* coo, const float* values, const int64_t* batch_ptr,
                               int64_t* dense_edges, float* dense_weights, uint32_t* flags,
                               int64_t E, int B, int max_edges, gcm_stream_t stream) {
  GCM_REQUIRE(batch_ptr && dense_edges && dense_weights && flags && E >= 0 && B > 0 &&
              max_edges > 0);
  if (E == 0) return GCM_OK;
  GCM_REQUIRE(coo && values);
  hipLaunchKernelGGL(k_pack_hidden, dim3((unsigned)((E + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, coo, values, batch_ptr, dense_edges, dense_weights, flags,
                     E, B, max_edges);
  return gcm_launch_status();
}
