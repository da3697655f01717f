// Shared pieces of the fused DenseGCM step kernels (fused_fwd.hip / fused_bwd.hip).
//
//     h1 = act1( (adj @ x) W_rel1^T + b1 + x W_root1^T )                      [N, H1]
//     mx = act2( (adj[cur,:] @ h1) W_rel2^T + b2 + h1[cur] W_root2^T )        [H2]   (gcm.py:314)
//
// One workgroup (4 waves) owns one graph: the whole adjacency (<= 128x128 fp32 = 64 KB), x and
// h1 live in LDS, adj is read from HBM exactly once for both layers, and the second layer is
// evaluated only on the row DenseGCM keeps.  Each wave owns a 32-row strip.
//
// Rules the code follows (each one was measured with in-kernel stamps, tools/kstamp.py):
//   * every global load of a phase is issued before the first dependent LDS store, from
//     unconditional (clamped) addresses - a predicated load makes hipcc branch around it and
//     wait for it alone;
//   * 32-bit offsets from a per-graph base pointer; an EXACT specialisation (all dims equal to
//     their padded sizes) compiles every bounds check away: at one wave per SIMD the kernel is
//     VALU-issue bound, not MFMA bound, once the loads are batched;
//   * all-zero 32x32 adjacency tiles (the common case for temporal / threshold graphs) skip
//     their MFMAs - exact, decided per wave with a ballot;
//   * LDS images use row strides that are odd in dwords => conflict-free ds_read_b32 fragments.
#pragma once
#include "gcm_common.h"

// Diagnostic build only (make stamps -> libgcm_hip_stamps.so, used by tools/kstamp.py): cycle
// stamps of workgroup 0 / lane 0 at phase boundaries.  Never defined in the product library.
#ifdef GCM_STAMPS
extern __device__ unsigned long long g_stamps[32];
#define STAMP(i)                                                                     \
  do {                                                                               \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                       \
      unsigned long long t_;                                                         \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");    \
      g_stamps[i] = t_;                                                              \
    }                                                                                \
  } while (0)
#else
#define STAMP(i)
#endif

namespace gcm_fused {

__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// acc(32x32) += A(32xK) * B(Kx32), operands in LDS: A(i,k)=a[i*ais+k*aks], B(k,j)=b[k*bks+j*bjs]
__device__ __forceinline__ void mma32(f32x16& acc, const float* a, int ais, int aks,
                                      const float* b, int bks, int bjs, int K, int li, int lh) {
  const float* ap = a + li * ais + lh * aks;
  const float* bp = b + lh * bks + li * bjs;
#pragma unroll 8
  for (int k = 0; k < K; k += 2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k * aks], bp[k * bks], acc, 0, 0, 0);
}

// 16x16 output block += A(16xK) * B(Kx16), operands in LDS (v_mfma_f32_16x16x4_f32: lane l holds
// A[l&15][4s + (l>>4)] and B[4s + (l>>4)][l&15]; acc[r] = C[4*(l>>4) + r][l&15]).
template <int K>
__device__ __forceinline__ void mma16(f32x4& acc, const float* a, int ais, const float* b, int bks,
                                      int m, int kq) {
  const float* ap = a + m * ais + kq;
  const float* bp = b + kq * bks + m;
  float av[K / 4], bv[K / 4];   // every LDS read in flight before the first MFMA
#pragma unroll
  for (int s = 0; s < K / 4; ++s) {
    av[s] = ap[4 * s];
    bv[s] = bp[4 * s * bks];
  }
  __builtin_amdgcn_sched_barrier(0);   // keep the reads batched: one LDS round trip, not K/8
#pragma unroll
  for (int s = 0; s < K / 4; ++s)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
}

// general strides: A(i,k) = a[i*ais + k*aks], B(k,j) = b[k*bks + j]
template <int K>
__device__ __forceinline__ void mma16g(f32x4& acc, const float* a, int ais, int aks, const float* b,
                                       int bks, int m, int kq) {
  const float* ap = a + m * ais + kq * aks;
  const float* bp = b + kq * bks + m;
  float av[K / 4], bv[K / 4];
#pragma unroll
  for (int s = 0; s < K / 4; ++s) {
    av[s] = ap[4 * s * aks];
    bv[s] = bp[4 * s * bks];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < K / 4; ++s)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
}

// mma32 with every LDS operand read in flight before the first MFMA (K compile-time, <= 64)
template <int K>
__device__ __forceinline__ void mma32b(f32x16& acc, const float* a, int ais, int aks, const float* b,
                                       int bks, int bjs, int li, int lh) {
  const float* ap = a + li * ais + lh * aks;
  const float* bp = b + lh * bks + li * bjs;
  float av[K / 2], bv[K / 2];
#pragma unroll
  for (int s = 0; s < K / 2; ++s) {
    av[s] = ap[2 * s * aks];
    bv[s] = bp[2 * s * bks];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < K / 2; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc, 0, 0, 0);
}

struct Gnn2 {
  const float *w_rel1, *b_rel1, *w_root1;  // [H1,F], [H1], [H1,F]
  const float *w_rel2, *b_rel2, *w_root2;  // [H2,H1], [H2], [H2,H1]
  int act1, act2;
};

// Register-staged cooperative copy of a [R x C] matrix (leading dimension ld) into an LDS image
// padded to [RP x CP]: load() issues every global load, store() writes the image
// (dst[r*S + c], or dst[c*S + r] when TR).  256 threads.  EXACT: R == RP and C == CP.
template <int RP, int CP, bool TR, bool EXACT>
struct Stage {
  static constexpr int TOT = RP * CP, PER = (TOT + 255) / 256;
  float v[PER];
  __device__ __forceinline__ void load(const float* __restrict__ src, int R, int C, int ld,
                                       int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 256 * i, r = e / CP, c = e % CP;
      if (EXACT) {
        v[i] = src[r * CP + c];
      } else {
        const int rc = r < R ? r : R - 1, cc = c < C ? c : C - 1;
        const float t = src[rc * ld + cc];
        v[i] = (e < TOT && r < R && c < C) ? t : 0.f;
      }
    }
  }
  __device__ __forceinline__ void store(float* dst, int S, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 256 * i, r = e / CP, c = e % CP;
      if (TOT % 256 == 0 || e < TOT) dst[TR ? c * S + r : r * S + c] = v[i];
    }
  }
};

// index-writing selectors folded into the step kernel (temporal hops + dense), by value
struct Edits {
  int n_hops;        // temporal back edges (temporal.py:72-88)
  int hops[16];
  int dir[16];       // GCM_DIR_* per hop entry (several TemporalBackedge selectors may be chained)
  int dense;         // DenseEdge (dense.py:16-21)
};


// This wave's 32 adjacency rows as 16-byte loads: tile t (32 columns), 8 rows per instruction.
template <int NT, bool EXACT>
struct AdjRows {
  float4 buf[NT * 4];
  __device__ __forceinline__ void load(const float* __restrict__ ag, int N, int r_base, int lane) {
    const bool vec = EXACT || (N & 3) == 0;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = r_base + (lane >> 3) + 8 * q, c = t * 32 + (lane & 7) * 4;
        float4 v;
        if (EXACT) {
          v = *reinterpret_cast<const float4*>(ag + r * (32 * NT) + c);
        } else {
          const int rc = r < N ? r : N - 1;
          const float* p = ag + rc * N;
          if (vec) {  // wave-uniform
            const int cc = c < N ? c : N - 4;
            v = *reinterpret_cast<const float4*>(p + cc);
            if (c >= N) v = make_float4(0.f, 0.f, 0.f, 0.f);
          } else {
            const int c0 = c < N ? c : N - 1, c1 = c + 1 < N ? c + 1 : N - 1;
            const int c2 = c + 2 < N ? c + 2 : N - 1, c3 = c + 3 < N ? c + 3 : N - 1;
            v.x = p[c0]; v.y = p[c1]; v.z = p[c2]; v.w = p[c3];
            if (c >= N) v.x = 0.f;
            if (c + 1 >= N) v.y = 0.f;
            if (c + 2 >= N) v.z = 0.f;
            if (c + 3 >= N) v.w = 0.f;
          }
          if (r >= N) v = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        buf[t * 4 + q] = v;
      }
  }
  // Same rows taken from the PREVIOUS state with the overflow roll applied (gcm.py:323-355):
  // out[r][c] = wrap ? in[r+1][c+1] (last row / column zero) : in[r][c].  vec path only (N%4==0).
  __device__ __forceinline__ void load_advanced(const float* __restrict__ ag, int N, int r_base,
                                                int lane, bool wrap) {
    const int sh = wrap ? 1 : 0;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = r_base + (lane >> 3) + 8 * q, c = t * 32 + (lane & 7) * 4;
        const int rs = r + sh < N ? r + sh : N - 1;
        const int cc = c < N ? c : N - 4;
        // last 4 columns of a wrapped row: load in place and shift in registers (in[.][N] does
        // not exist); elsewhere read one element to the right (dword-aligned 16-byte load)
        const bool tail = wrap && cc + 4 >= N;
        const float* p = ag + rs * N + cc + (tail ? 0 : sh);
        float4 v;   // dword-aligned 16-byte load
        __builtin_memcpy(&v, p, sizeof(float4));
        if (tail) v = make_float4(v.y, v.z, v.w, 0.f);
        if (r + sh >= N || c >= N || r >= N) v = make_float4(0.f, 0.f, 0.f, 0.f);
        buf[t * 4 + q] = v;
      }
  }
  // selector writes on the freshly advanced adjacency, in registers (cur = new node's row).
  // lane_hop / lane_dir: lane i < E.n_hops holds hops[i] / dir[i] (read ONCE from the kernel
  // arguments by the caller): indexing E.hops[i] inside the tile loops made hipcc re-load it from
  // the kernarg segment for every (tile, row group, hop) - ~100 scalar loads with their waits,
  // 6.5 us of a 15 us kernel.
  __device__ __forceinline__ void apply_edits(const Edits& E, int lane_hop, int lane_dir, int cur,
                                              int r_base, int lane) {
    const int n_hops = E.n_hops;
    // wave-uniform early out: does any edit touch this wave's 32 rows?
    bool touched = (cur >> 5) == (r_base >> 5);
    for (int i = 0; i < n_hops; ++i) {
      const int h = __builtin_amdgcn_readlane(lane_hop, i), d = __builtin_amdgcn_readlane(lane_dir, i);
      touched |= (d & GCM_DIR_BACKWARD) && cur >= h && ((cur - h) >> 5) == (r_base >> 5);
    }
    touched |= E.dense && (r_base < cur);
    if (!touched) return;
    for (int i = 0; i < n_hops; ++i) {
      const int h = __builtin_amdgcn_readlane(lane_hop, i), d = __builtin_amdgcn_readlane(lane_dir, i);
      if (h < 0 || cur < h) continue;   // uniform
      const int past = cur - h;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = r_base + (lane >> 3) + 8 * q, c = t * 32 + (lane & 7) * 4;
          float4 v = buf[t * 4 + q];
          int k = -1;   // component of v that becomes 1 (selects, no runtime-indexed registers)
          if ((d & GCM_DIR_FORWARD) && r == cur) k = past - c;
          if ((d & GCM_DIR_BACKWARD) && r == past) k = cur - c;
          v.x = k == 0 ? 1.f : v.x;
          v.y = k == 1 ? 1.f : v.y;
          v.z = k == 2 ? 1.f : v.z;
          v.w = k == 3 ? 1.f : v.w;
          buf[t * 4 + q] = v;
        }
    }
    if (E.dense) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = r_base + (lane >> 3) + 8 * q, c = t * 32 + (lane & 7) * 4;
          float4 v = buf[t * 4 + q];
          if (r == cur) {   // row cur: columns 0..cur (self edge included)
            v.x = c <= cur ? 1.f : v.x;
            v.y = c + 1 <= cur ? 1.f : v.y;
            v.z = c + 2 <= cur ? 1.f : v.z;
            v.w = c + 3 <= cur ? 1.f : v.w;
          } else if (r < cur) {   // column cur: rows 0..cur-1
            const int k = cur - c;
            v.x = k == 0 ? 1.f : v.x;
            v.y = k == 1 ? 1.f : v.y;
            v.z = k == 2 ? 1.f : v.z;
            v.w = k == 3 ? 1.f : v.w;
          }
          buf[t * 4 + q] = v;
        }
    }
  }
  // write this wave's rows to the new adjacency in HBM (16-byte stores)
  __device__ __forceinline__ void store_global(float* __restrict__ ag, int N, int r_base,
                                               int lane) const {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = r_base + (lane >> 3) + 8 * q, c = t * 32 + (lane & 7) * 4;
        if (EXACT || (r < N && c < N)) *reinterpret_cast<float4*>(ag + r * N + c) = buf[t * 4 + q];
      }
  }
  // true when any element of this wave's tile t is non-zero (wave-uniform)
  __device__ __forceinline__ bool tile_nonzero(int t) const {
    bool nz = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = buf[t * 4 + q];
      nz |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
    }
    return __any(nz);
  }
  // image: [col tile][row][33]
  template <int NP>
  __device__ __forceinline__ void store_tile(float* sAdj, int t, int r_base, int lane) const {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = r_base + (lane >> 3) + 8 * q;
      float* d = sAdj + (t * NP + r) * 33 + (lane & 7) * 4;
      const float4 v = buf[t * 4 + q];
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  }
};

template <int NP>
__device__ __forceinline__ int adj_at(int r, int c) {
  return ((c >> 5) * NP + r) * 33 + (c & 31);
}

// LDS carve-up (floats).  N2T = ceil(H2/32).
template <int NT, int NCT, int NHT, int N2T>
struct Lds {
  static constexpr int NP = 32 * NT, FP = 32 * NCT, HP = 32 * NHT, H2P = 32 * N2T;
  static constexpr int FS = FP + 1, HS = HP + 1;
  static constexpr int AS = FS > HS ? FS : HS;     // common stride of the agg / h1 image
  static constexpr int W2S = 2 * HP + 1;           // [o][rel k | root k]
  static constexpr int ADJ = NT * NP * 33;
  static constexpr int X = NP * FS;
  static constexpr int AH = NP * AS;
  static constexpr int W1F = 2 * FP * HS;          // forward: w_rel1^T | w_root1^T   [f][h]
  static constexpr int W1B = 2 * HP * FS;          // backward: w_rel1 | w_root1       [h][f]
  static constexpr int W2 = H2P * W2S;
  static constexpr int SV = 256 + 4 * HP + H2P + 64;       // partials | v | d2 | u | flags
  // forward: the layer-2 weights take over the x image once layer 1 is done (when they fit)
  static constexpr bool W2_IN_X = W2 <= X;
  static constexpr int FWD = ADJ + X + AH + W1F + SV + (W2_IN_X ? 0 : W2);
  static constexpr int BWD = ADJ + NP * HS + NP * FS + W1B + W2 + 4 * 1024 + SV;
};

inline void lds_need(int NT, int NCT, int NHT, int N2T, size_t* fwd, size_t* bwd) {
  const size_t NP = 32 * NT, FP = 32 * NCT, HP = 32 * NHT, H2P = 32 * N2T;
  const size_t FS = FP + 1, HS = HP + 1, AS = FS > HS ? FS : HS, W2S = 2 * HP + 1;
  const size_t ADJ = NT * NP * 33, X = NP * FS, AH = NP * AS, SV = 256 + 4 * HP + H2P + 64;
  const size_t W2 = H2P * W2S;
  *fwd = sizeof(float) * (ADJ + X + AH + 2 * FP * HS + SV + (W2 <= X ? 0 : W2));
  *bwd = sizeof(float) * (ADJ + NP * HS + NP * FS + 2 * HP * FS + W2 + 4 * 1024 + SV);
}

}  // namespace gcm_fused

// instantiated shapes (NT, NCT, NHT, N2T); one X-macro keeps forward and backward dispatch identical
#define GCM_SHAPES_N(X, a) \
  X(a, 1, 1, 1) X(a, 1, 1, 2) X(a, 1, 2, 1) X(a, 1, 2, 2) X(a, 2, 1, 1) X(a, 2, 1, 2) X(a, 2, 2, 1) X(a, 2, 2, 2)
#define GCM_SHAPES(X) GCM_SHAPES_N(X, 1) GCM_SHAPES_N(X, 2) GCM_SHAPES_N(X, 3) GCM_SHAPES_N(X, 4)
