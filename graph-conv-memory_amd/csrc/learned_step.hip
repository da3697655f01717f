// DenseGCM + LearnedEdge as fused per-graph kernels (edge_selectors/learned.py:38-113 with the
// default edge network, DenseGraphConv x 2 as the GNN; observations without gradient).
//
// The default edge network scores a candidate pair (cur, j) from cat(x[cur], x[j]):
//     Linear(2F, F) - ReLU - LayerNorm - Linear(F, F) - ReLU - LayerNorm - Linear(F, 1)
// Its first layer splits into a per-graph bias and ONE [N x F] x [F x F] product,
//     P0[j] = W0b x[j] + (W0a x[cur] + b0),
// so the whole network is two 128 x 32 x 32 products on the fp32 MFMA plus row-wise ReLU-LayerNorms:
// one workgroup per graph keeps the node matrix and both hidden layers in LDS.
//
//   k_learned_select   forward: edge network on all N candidate rows, gumbel-softmax over j < cur,
//                      straight-through threshold, adjacency row cur written in place; <MODE> folds the state
//                      advance in (functional / donated), <TAIL> the GNN on row cur over the chain's caches.
//   k_learned_roll_logits / _pick / _l2   the forward of a whole rollout from empty graphs in three launches
//                      (DenseGCM.rollout), the edge network per 32-row block with a candidate row.
//   k_bptt_rows<.., 2> (rows_bptt.hip) + k_learned_bptt_sel + k_learned_bptt_mlp   the time-parallel backward of
//                      a chain of steps (see "TIME-PARALLEL backward" below): what every fused chain runs.
//   k_learned_step_bwd backward of ONE step (round 2's sequential form, kept behind gcm_learned_step_bwd for the
//                      C ABI's single-step callers), everything in one launch:
//     * GNN adjoint on the live rows (rows j with adj[cur, j] != 0, and cur: gcm.py:314 keeps one
//       row of the last layer) -> parameter-gradient slab;
//     * the gradient w.r.t. the ADJACENCY, which is what trains the edge network.  Its chain through
//       time is dense in the reference (a [B,N,N] tensor per step); here it is carried in compact
//       time is a dense [B,N,N] tensor PER STEP in the reference (every step's node receives and
//       returns one); here ONE chain buffer GA [B,N,N] lives across the steps and is updated
//       sparsely: a step adds dAgg1[l] . x[k] to the rows layer 1 aggregated into (the live rows),
//       reads row cur - the only entries differentiated at this step,
//           g_sel_t[j] = dagg2_t . h1_t[j] + GA[cur_t][j]      (j < cur_t),
//       and undoes the state advance (zero row cur; on overflow, gcm.py:323-355, one shift of the
//       buffer) for the step before;
//     * selection adjoint (softmax; both straight-through estimators are identities);
//     * edge-network adjoint with the forward recomputed in LDS (nothing but `soft` is saved).
//
// Shapes: N <= 128, F <= 32, H1 <= 32, H2 <= 32 (BASELINE cfg5: 128 / 32 / 32 / 32).
#include "fused_common.h"
#include <cstdlib>
#include <utility>
#include <vector>

#include "rows_common.h"
#include "state_copy.h"

#ifdef GCM_STAMPS   // diagnostic build only (make stamps7, tools/kstamp_learned.py)
__device__ unsigned long long g_stamps[32];
extern "C" int gcm_debug_read_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
#define LSTAMP(i) STAMP(i)
// the same from the tail wave of k_learned_select8 (wave 8, lane 0)
#define LSTAMPT(i)                                                                   \
  do {                                                                               \
    if (blockIdx.x == 0 && threadIdx.x == 512) {                                     \
      unsigned long long t_;                                                         \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");    \
      g_stamps[i] = t_;                                                              \
    }                                                                                \
  } while (0)
// first / last s_memrealtime (100 MHz, one counter for the device) of EVERY workgroup of the last k_learned_select launch
__device__ unsigned long long g_span[2048][2];
extern "C" int gcm_debug_read_spans(unsigned long long* out, int n_wg) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_span), sizeof(unsigned long long) * 2 * n_wg);
}
#define LSPAN(i)                                                                   \
  do {                                                                             \
    if (threadIdx.x == 0 && blockIdx.x < 2048) {                                   \
      unsigned long long t_;                                                       \
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");  \
      g_span[blockIdx.x][i] = t_;                                                  \
    }                                                                              \
  } while (0)
#ifdef GCM_STAMPS_B2
#define BSTAMP(i) STAMP(i)
#else
#define BSTAMP(i)
#endif
#else
#define LSTAMP(i)
#define LSTAMPT(i)
#define BSTAMP(i)
#define LSPAN(i)
#endif

namespace gcm_learned {

using gcm_fused::mma32;

constexpr int NP = 128, FP = 32, FS = 33;
constexpr int GS = 36;   // row stride of the GNN weights in the cached step's tail: 16-byte aligned rows

struct Mlp {   // pointers into the packed edge-network parameter vector (gcm_learned_mlp_layout)
  const float *w0, *b0, *g0, *be0, *w1, *b1, *g1, *be1, *w2, *b2;
};
__host__ __device__ inline Mlp unpack_mlp(const float* p, int F) {
  Mlp m;
  m.w0 = p; m.b0 = m.w0 + 2 * F * F; m.g0 = m.b0 + F; m.be0 = m.g0 + F;
  m.w1 = m.be0 + F; m.b1 = m.w1 + F * F; m.g1 = m.b1 + F; m.be1 = m.g1 + F;
  m.w2 = m.be1 + F; m.b2 = m.w2 + F;
  return m;
}

// [rows x F] matrix with leading dimension ld -> LDS image [RP][FS], zero padded; 256 threads
template <int RP>
__device__ __forceinline__ void stage(const float* __restrict__ src, float* dst, int R, int C, int ld, int tid) {
  constexpr int PER = RP * FP / 256;
  float v[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = tid + 256 * i, r = e / FP, c = e % FP;
    v[i] = src[(r < R ? r : R - 1) * ld + (c < C ? c : C - 1)];
  }
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = tid + 256 * i, r = e / FP, c = e % FP;
    dst[r * FS + c] = (r < R && c < C) ? v[i] : 0.f;
  }
}

// out rows [32 wave, 32 wave + 32) = A[rows, :] (K = 32) @ Bt^T, Bt stored [n][k] (stride FS);
// acc(r) -> row 32 wave + acc_row(r, lh), column li
__device__ __forceinline__ f32x16 gemm_rows(const float* sA, const float* sBt, int wave, int li, int lh) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  mma32(acc, sA + 32 * wave * FS, FS, 1, sBt, 1, FS, 32, li, lh);
  return acc;
}

// ReLU + LayerNorm of the rows of sP (in place -> gamma * xhat + beta), TWO adjacent threads per row
// (row = tid / 2, each owns 16 of the 32 columns; the row statistics meet through one lane shuffle):
// all 256 threads work on the 128 rows.  Statistics (mean, rstd) optionally stored.  write = false
// leaves sP untouched (statistics only).
// FULL (F == FP, uniform): the same arithmetic without the column guards - a third of the instructions of a pass
// that the block-granular kernels are bound by (they issue ~2 k VALU / LDS instructions per 32-row block).
// REG: sg / sb are this thread's 16 entries (registers, index k) instead of the LDS vectors (index f).
// VEC: sg / sb are 16-byte aligned LDS vectors, read in 16-byte pieces.
// ofs (optional): this thread's 16 entries of a per-column offset added to the row before the ReLU.
template <bool FULL, bool REG = false, bool VEC = false>
__device__ __forceinline__ void relu_ln_rows_t(float* sP, int tid, int F, const float* sg, const float* sb,
                                               float eps, float* mu_out, float* rs_out, bool write,
                                               const float* ofs = nullptr) {
  const int row = tid >> 1, f0 = (tid & 1) * (FP / 2);
  float a[FP / 2];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < FP / 2; ++k) {
    const int f = f0 + k;
    const float v = ofs ? sP[row * FS + f] + ofs[k] : sP[row * FS + f];
    a[k] = ((FULL || f < F) && v > 0.f) ? v : 0.f;
    s += a[k];
  }
  s += gcm_lane_xor1(s);
  const float mean = s / (float)F;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < FP / 2; ++k) {
    const float d = (FULL || f0 + k < F) ? a[k] - mean : 0.f;
    q = fmaf(d, d, q);
  }
  q += gcm_lane_xor1(q);
  const float rstd = rsqrtf(q / (float)F + eps);
  if (write && VEC) {
#pragma unroll
    for (int q4 = 0; q4 < FP / 8; ++q4) {
      const float4 tg = reinterpret_cast<const float4*>(sg + f0)[q4], tb = reinterpret_cast<const float4*>(sb + f0)[q4];
      const float gv[4] = {tg.x, tg.y, tg.z, tg.w}, bv[4] = {tb.x, tb.y, tb.z, tb.w};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int k = 4 * q4 + kk, f = f0 + k;
        sP[row * FS + f] = (FULL || f < F) ? fmaf((a[k] - mean) * rstd, gv[kk], bv[kk]) : 0.f;
      }
    }
  } else if (write) {
#pragma unroll
    for (int k = 0; k < FP / 2; ++k) {
      const int f = f0 + k;
      sP[row * FS + f] = (FULL || f < F) ? fmaf((a[k] - mean) * rstd, sg[REG ? k : f], sb[REG ? k : f]) : 0.f;
    }
  }
  if (mu_out && (tid & 1) == 0) { mu_out[row] = mean; rs_out[row] = rstd; }
}
// the F -> 1 layer on the rows of an [.][FS] image: two adjacent lanes per row, each half of the dot product in
// ascending f, the halves added, + b2 - ONE function for the per-step and the time-parallel kernel (same logits,
// bit for bit: the sampled edges of the two paths agree)
__device__ __forceinline__ float logit_row2(const float* row, const float* w2, int half, float b2) {
  float p = 0.f;
#pragma unroll
  for (int k = 0; k < FP / 2; ++k) p = fmaf(w2[(FP / 2) * half + k], row[(FP / 2) * half + k], p);
  p += gcm_lane_xor1(p);
  return p + b2;
}
__device__ __forceinline__ void relu_ln_rows(float* sP, int tid, int F, const float* sg, const float* sb,
                                             float eps, float* mu_out, float* rs_out, bool write = true) {
  if (F == FP) relu_ln_rows_t<true>(sP, tid, F, sg, sb, eps, mu_out, rs_out, write);
  else relu_ln_rows_t<false>(sP, tid, F, sg, sb, eps, mu_out, rs_out, write);
}

// LayerNorm + ReLU adjoint of the rows of sP (pre-activation values, overwritten by the gradient w.r.t.
// them): gx[f] = the gradient w.r.t. the normalised value times gamma; two threads per row as above.
template <bool FULL, typename GX>
__device__ __forceinline__ void relu_ln_rows_bwd_t(float* sP, int tid, int F, const float* sMu, const float* sRs,
                                                   GX gx_of) {
  const int row = tid >> 1, f0 = (tid & 1) * (FP / 2);
  const float mean = sMu[row], rstd = sRs[row];
  float xh[FP / 2], gx[FP / 2], m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int k = 0; k < FP / 2; ++k) {
    const int f = f0 + k;
    const float v = sP[row * FS + f];
    xh[k] = (FULL || f < F) ? ((v > 0.f ? v : 0.f) - mean) * rstd : 0.f;
    gx[k] = (FULL || f < F) ? gx_of(row, f) : 0.f;
    m1 += gx[k];
    m2 = fmaf(gx[k], xh[k], m2);
  }
  m1 += gcm_lane_xor1(m1);
  m2 += gcm_lane_xor1(m2);
  m1 /= (float)F;
  m2 /= (float)F;
#pragma unroll
  for (int k = 0; k < FP / 2; ++k) {
    const int f = f0 + k;
    const float v = sP[row * FS + f];
    const float da = rstd * (gx[k] - m1 - xh[k] * m2);
    sP[row * FS + f] = ((FULL || f < F) && v > 0.f) ? da : 0.f;
  }
}
template <typename GX>
__device__ __forceinline__ void relu_ln_rows_bwd(float* sP, int tid, int F, const float* sMu, const float* sRs,
                                                 GX gx_of) {
  if (F == FP) relu_ln_rows_bwd_t<true>(sP, tid, F, sMu, sRs, gx_of);
  else relu_ln_rows_bwd_t<false>(sP, tid, F, sMu, sRs, gx_of);
}
// the same adjoint with the incoming gradient of this thread's 16 columns handed over in registers (gx[k]: the
// gradient w.r.t. the normalised value times gamma, column f0 + k) and the pre-activations read ONCE: a third of
// the LDS reads of the form above (the block-granular backward is bound by the instructions it issues)
template <bool FULL>
__device__ __forceinline__ void relu_ln_rows_bwd_v(float* sP, int tid, int F, const float* sMu, const float* sRs,
                                                   const float (&gxv)[FP / 2]) {
  const int row = tid >> 1, f0 = (tid & 1) * (FP / 2);
  const float mean = sMu[row], rstd = sRs[row];
  float xh[FP / 2], gx[FP / 2], m1 = 0.f, m2 = 0.f;
  unsigned pos = 0;
#pragma unroll
  for (int k = 0; k < FP / 2; ++k) {
    const int f = f0 + k;
    const float v = sP[row * FS + f];
    pos |= (v > 0.f ? 1u : 0u) << k;
    xh[k] = (FULL || f < F) ? ((v > 0.f ? v : 0.f) - mean) * rstd : 0.f;
    gx[k] = (FULL || f < F) ? gxv[k] : 0.f;
    m1 += gx[k];
    m2 = fmaf(gx[k], xh[k], m2);
  }
  m1 += gcm_lane_xor1(m1);
  m2 += gcm_lane_xor1(m2);
  m1 /= (float)F;
  m2 /= (float)F;
#pragma unroll
  for (int k = 0; k < FP / 2; ++k) {
    const int f = f0 + k;
    const float da = rstd * (gx[k] - m1 - xh[k] * m2);
    sP[row * FS + f] = ((FULL || f < F) && ((pos >> k) & 1u)) ? da : 0.f;
  }
}

// c0[o] = b0[o] + W0a[o, :] . x_cur: both vectors in registers before the first use (a load -> fma loop
// exposed up to F memory round trips on the 32 threads that run it), summed in ascending f
__device__ __forceinline__ float c0_dot(const float* __restrict__ w_row, const float* __restrict__ xcur,
                                        float b0, int F) {
  float wv[FP], xv[FP];
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    wv[f] = w_row[f < F ? f : F - 1];
    xv[f] = xcur[f < F ? f : F - 1];
  }
  asm volatile("" ::: "memory");
  float c0 = b0;
#pragma unroll
  for (int f = 0; f < FP; ++f)
    if (f < F) c0 = fmaf(wv[f], xv[f], c0);
  return c0;
}

__device__ __forceinline__ float wave_max(float v) { return gcm_wave_max(v); }
__device__ __forceinline__ float wave_sum(float v) { return gcm_wave_sum(v); }

// ---------------------------------------------------------------------------------------------
// forward: logits of all candidate rows, gumbel-softmax, threshold, adjacency row (learned.py:53-113)
// ---------------------------------------------------------------------------------------------
// ADVANCE: the state advance (gcm.py:262-278, overflow roll :323-355) is done by this kernel too -
// `nodes` / `adj` are then the OUTPUT state, read from nodes_in / adj_in through registers (the copy's
// loads ride with the edge network's, its stores drain under the two GEMMs), the observation goes into
// row cur, cur / count come out; N % 4 == 0 and F % 4 == 0.  Otherwise nodes / adj hold the advanced
// state already (gcm_state_advance_fwd ran) and cur_idx is read.
// MODE 2 (round 3, donated state): as ADVANCE, but `nodes_out` / `adj` ARE `nodes_in` / `adj_in` - the state is
// advanced in place.  Without overflow nothing is copied: the node rows are read once for the edge
// network's image, the observation lands in row cur, the sampled entries in row cur of the adjacency; on
// overflow the roll happens in place (every load lands before the first store).  What the time-parallel
// backward needs of the state goes into the step's record: the node matrix after the insert (`snap`, 4 MB
// at cfg5 instead of the 21 MB state copy) and row cur of the adjacency (`row_out`).
// TAIL (round 3, "cached" steps): the GNN of the step rides on the selection.  Rows of h1 never change once
// written while a graph has not overflowed (row j's adjacency entries and the rows it aggregates are final
// after step j), so a chain that starts from EMPTY graphs keeps h1 / agg1 / x of every node in per-chain caches
// [B,N,.] and a step only computes row cur: agg1[cur] from the selected rows of the node image (already in LDS),
// h1[cur], agg2 from the selected rows of the h1 cache (preloaded to LDS next to the node image), the belief -
// four 32 x 32 matrix-vector products on wave 0 behind the selection instead of a second kernel over the
// whole graph (k_gnn2_row_fwd).  The backward passes read the caches (gcm_learned_bptt_cached).
struct GnnTail {
  const float* gnn;                  // packed GNN parameters (gcm_dense_gnn2_param_count layout)
  int act1, act2, has_bias, H1, H2;
  float *cH, *cA, *cX;               // caches: h1 [B,N,H1], agg1 [B,N,F], nodes [B,N,F]
  float* cU;                         // TAIL = 1 at the exact shapes: [B,N,F] the edge network's first product of every stored row,
                                     // U[j] = W0[:, F:] x_j (below)
  float *mx_out, *agg2_out;          // this step: [B,H2], [B,H1]
  float *h1_out, *agg1_out;          // TAIL = 2: layer 1 of every row of the graph, [B,N,H1] / [B,N,F] (the step's record)
  const float *h1_prev, *agg1_prev;  // TAIL = 2: layer 1 of the state BEFORE the step, [B,N,H1] / [B,N,F] (the previous step's record, or
                                     // the caches of the cached steps before the first steady one)
  uint32_t* abits;                   // TAIL = 2: the adjacency as bits [B][N][4] (bit j & 31 of word j >> 5 of row r: adj[r][j] != 0),
                                     // the state BEFORE the step on entry, after it on exit (gcm_adj_bits builds the first one)
};

// 16 bytes from global memory as ONE global_load_dwordx4: through the compiler's own 4-vector (a load of HIP's float4
// struct is split by field use - dwordx3 + dword, dwordx2 pairs - and the first phase of the step kernels is bound by
// the number of instructions in front of the first wait)
__device__ __forceinline__ float4 ld4(const float* p) {
  const f32x4 t = *reinterpret_cast<const f32x4*>(p);
  return make_float4(t[0], t[1], t[2], t[3]);
}

// one wave of work on a 32 x 32 block of an LDS image shared with nobody: LDS writes -> reads of other lanes
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// half of a 32-wide matrix-vector product on a half wave: the 32 products of a weight row (16-byte aligned, LDS) with
// a vector (LDS, 16-byte broadcast reads) in two chains; the other half wave's sum added (lanes l and l + 32 form a row)
__device__ __forceinline__ float half_dot(const float* wrow, const float* u) {
  float4 wv[8], uv[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    wv[q] = reinterpret_cast<const float4*>(wrow)[q];
    uv[q] = reinterpret_cast<const float4*>(u)[q];
  }
  float pa = 0.f, pb = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    pa = fmaf(wv[q].x, uv[q].x, pa);
    pb = fmaf(wv[q].y, uv[q].y, pb);
    pa = fmaf(wv[q].z, uv[q].z, pa);
    pb = fmaf(wv[q].w, uv[q].w, pb);
  }
  const float p = pa + pb;
  return gcm_xor32_add(p);
}

// gcm_fused::Stage<RP, CP, false, false> with a form for exact shapes (EX: no clamps, no masks, 16-byte global loads -
// a thread then holds four consecutive columns per piece - and 16-byte LDS stores where the row stride S allows);
// any leading dimension that is a multiple of 4 floats when EX.  256 threads.
template <int RP, int CP>
struct StageL {
  static constexpr int PER = RP * CP / 256;
  static_assert(PER % 4 == 0 && CP % 4 == 0, "whole 16-byte pieces per thread");
  float v[PER];
  // rsh: image row r <- source row r + rsh (clamped to the last one)
  template <bool EX>
  __device__ __forceinline__ void load(const float* __restrict__ src, int R, int C, int ld, int tid, int rsh = 0) {
    if (EX) {
#pragma unroll
      for (int i = 0; i < PER / 4; ++i) {
        const int e4 = tid + 256 * i, r = min(e4 / (CP / 4) + rsh, RP - 1), c = (e4 % (CP / 4)) * 4;
        const float4 t = ld4(src + (size_t)r * ld + c);
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int e = tid + 256 * i, r = e / CP, c = e % CP;
        const float t = src[(r + rsh < R ? r + rsh : R - 1) * ld + (c < C ? c : C - 1)];
        v[i] = (r < R && c < C) ? t : 0.f;
      }
    }
  }
  template <bool EX, int S>
  __device__ __forceinline__ void store(float* dst, int tid) const {
    if (EX) {
#pragma unroll
      for (int i = 0; i < PER / 4; ++i) {
        const int e4 = tid + 256 * i, r = e4 / (CP / 4), c = (e4 % (CP / 4)) * 4;
        if (S % 4 == 0) {
          *reinterpret_cast<float4*>(dst + r * S + c) = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) dst[r * S + c + k] = v[4 * i + k];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int e = tid + 256 * i, r = e / CP, c = e % CP;
        dst[r * S + c] = v[i];
      }
    }
  }
};

// TAIL: 0 - selection only; 1 - the cached step (row cur over the chain's caches, above); 2 - the STEADY-STATE step
// (round 5): every graph is full, the step rolls the donated state in place and the rows of layer 1 are no longer final
// (a row loses the sources the roll drops), so layer 1 is re-evaluated for ALL rows right here - the adjacency as a bit
// image in LDS, agg1 = Adj X and h1 on the matrix cores from the node image already staged for the edge network, then
// row cur's layer 2 on wave 0.  The adjacency's roll does not READ the fp32 matrix: the chain keeps its bit image
// (GnnTail::abits, 2 KB a graph), the step shifts that by one row and one column, writes from it the 16-byte pieces of
// the fp32 rows whose bits CHANGE (no load -> barrier -> store of the whole matrix through 64 registers a thread; the
// sampled adjacency is sparse, most pieces of the shifted matrix equal what is already there) and leaves the image of
// the new state behind.  (Stamps of the first form, which rolled through registers: the roll's loads were 40 % of the
// launch; of the second, which wrote every piece: 25 MB of stores leaving 256 CUs in the same phase, 5 us of stalls
// wherever the stores were put.)  It writes what k_gnn2_row_fwd wrote into the step's record (h1, agg1, agg2, mx: gcm_learned_step_layout,
// compact = 1), so the chain's backward reads it unchanged - ONE launch instead of two, and no second pass over the
// 16.8 MB adjacency.
// EX: the exact shapes N = NP, F = H1 = H2 = FP as compile-time constants (cfg5; the launchers check): every bound,
// clamp and row / column index of the staging code folds - the first phase of the kernel was ~1000 VALU instructions of
// address arithmetic and forty conditional blocks around its fifty loads, a quarter of the cached step.
template <int MODE, int TAIL, bool EX = false>
__global__ __launch_bounds__(256) void k_learned_select(
    const float* __restrict__ nodes_c, float* adj, const int64_t* __restrict__ cur_idx_c,
    const float* __restrict__ noise, int noise_is_exp, const float* __restrict__ mlp, float eps0,
    float eps1, float cutoff, float* __restrict__ soft, int N_, int F_, const float* __restrict__ obs,
    const float* nodes_in, const float* adj_in, const int64_t* count_in, float* nodes_out,
    int64_t* __restrict__ cur_out, int64_t* count_out, uint32_t* __restrict__ flags,
    float* __restrict__ snap, float* __restrict__ row_out, GnnTail gt_, int cur_host) {   // (MODE 2: adj == adj_in, nodes_out == nodes_in)
  constexpr bool ADVANCE = MODE != 0, DONATE = MODE == 2;
  const int N = EX ? NP : N_, F = EX ? FP : F_;
  GnnTail gt = gt_;
  if (EX) { gt.H1 = FP; gt.H2 = FP; }
  static_assert(!TAIL || ADVANCE, "the cached step advances the state itself");
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  LSTAMP(14);
  LSPAN(0);
  int cur;
  bool wrap = false;
  int64_t n_chk = 0;   // cur_host: the count as stored, compared at the END of the kernel (nothing waits for it)
  if (ADVANCE && cur_host >= 0) {
    // the host knows the row (a chain from empty graphs that has not overflowed: every graph holds as many nodes as
    // the chain has made steps): no load in front of every address below
    cur = cur_host < N ? cur_host : N - 1;
    if (tid == 0) {
      n_chk = count_in[b];
      cur_out[b] = cur;
      if (!DONATE) count_out[b] = cur + 1;
    }
  } else if (TAIL == 2) {
    // steady state: every graph is full (the chain's guarantee - checked at the END of the kernel, like cur_host):
    // no load in front of every address below
    cur = N - 1;
    wrap = true;
    if (tid == 0) {
      n_chk = count_in[b];
      cur_out[b] = cur;
      // every graph drops a node: ONE workgroup says so, at the start (256 atomics on one word at the END of the launch
      // were ~2 us between the last workgroup and the launch's completion)
      if (b == 0) atomicOr(flags, GCM_FLAG_WRAPPED);
    }
  } else if (ADVANCE) {
    const int64_t n_in = count_in[b];
    wrap = n_in + 1 > N;
    const int64_t c64 = wrap ? n_in - 1 : n_in;
    cur = c64 < 0 ? 0 : (c64 > N - 1 ? N - 1 : (int)c64);
    if (tid == 0) {
      cur_out[b] = cur;
      if (!DONATE) count_out[b] = cur + 1;   // (donated: count_out IS count_in, which every wave is reading - below)
      const uint32_t f = (wrap ? GCM_FLAG_WRAPPED : 0u) | ((n_in < 0 || n_in > N) ? GCM_FLAG_BAD_COUNT : 0u);
      if (f) atomicOr(flags, f);
    }
  } else {
    const int64_t c64 = cur_idx_c[b];
    cur = c64 < 0 ? 0 : (c64 > N - 1 ? N - 1 : (int)c64);
  }
  const Mlp M = unpack_mlp(mlp, F);
  const float* xg = ADVANCE ? nullptr : nodes_c + (size_t)b * N * F;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sX = smem;                 // [NP][FS]
  float* sA = sX + NP * FS;         // P0 -> H0
  float* sB = sA + NP * FS;         // P1 -> H1
  float* sW0b = sB + NP * FS;       // [o][f] = W0[o][F + f]
  float* sW1 = sW0b + FP * FS;      // [o][f]
  float* sVec = sW1 + FP * FS;      // c0 | b1 | g0 | be0 | g1 | be1 | w2   (7 x 32)
  float* sLogit = sVec + 7 * FP;    // [NP]
  float* sW0a = sLogit + NP;        // [o][f] = W0[o][f], rows at stride GS (16-byte aligned)
  float* sHc = sW0a + FP * GS;      // TAIL: [NP][FS] h1 of this graph's rows (TAIL = 1: the chain's cache; 2: the previous step's, one row up)
  float* sWg = sHc + NP * FS;       // TAIL: [4][FP][GS] W_rel1 | W_root1 | W_rel2 | W_root2, row o at stride GS (16-byte aligned rows)
  uint32_t* sBits = reinterpret_cast<uint32_t*>(sWg + 4 * FP * GS);   // TAIL = 2: [NP][4] the advanced adjacency as bits
  uint32_t* sAff = sBits + NP * 4;                                    // | [4] the rows that lost a source, as bits (below)

  // EVERY load of the kernel is requested here, in one round trip (in-kernel stamps of round 3 / 4: a load issued
  // at its point of use - the observation patched into row cur, W0a and x_cur for c0, the seven vectors - is a
  // whole memory round trip on the critical path of a kernel that runs one wave per SIMD), the farthest first: what
  // the caller or the previous step's kernel wrote (observation, gumbel draws, node rows, adjacency row, h1 cache),
  // then the parameters (L2 resident).  `ex`: exact shapes (uniform) - the same loads without clamps and masks, a
  // few hundred VALU instructions less in front of the first wait.
  const bool ex = EX || (F == FP && N == NP && (!TAIL || (gt.H1 == FP && gt.H2 == FP)));
  // UC (cached steps at the exact shapes): the first layer of the edge network on candidate row j is W0b x_j + (W0a x_cur
  // + b0), and x_j never changes once stored - the chain keeps U[j] = W0b x_j [B,N,F] and a step stages it as the image
  // the product P0 = X W0b^T used to fill (16 MFMAs a wave and their operand reads: 2.4 k of the step's 15 k cycles),
  // adds c0 on the way into the LayerNorm, and leaves U[cur] behind for the steps to come (wave 1, under wave 0's tail).
  constexpr bool UC = TAIL == 1 && EX;
  float pf_noise[2], pf_old[2], pf_b1 = 0.f, pf_b2 = 0.f;
  {
    const int pl = tid & 63;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int j = pl + 64 * c < N ? pl + 64 * c : N - 1;
      pf_noise[c] = noise[(size_t)b * N + j];
      pf_old[c] = ADVANCE && TAIL != 2 ? adj_in[((size_t)b * N + cur) * N + j] : 0.f;   // (steady state: the row is new)
    }
  }
  constexpr int ADJ_PER = 16, NODE_PER = (NP * FP / 4 + 255) / 256;
  const int N4 = N >> 2, F4 = F >> 2;
  // piece index -> row: a shift at the exact shapes (a division by a run-time value is ~40 instructions, and a thread
  // does one per piece of the state it moves; the asm keeps the compiler from evaluating both sides)
  auto div_f4 = [&](const int e4) {
    int r;
    if (ex) { r = e4 / (FP / 4); } else { r = e4 / F4; asm volatile("" : "+v"(r)); }
    return r;
  };
  const int lim_n = N * F4;
  float4 ca[ADJ_PER], cn[NODE_PER], ob[NODE_PER];
  uint4 wbits = make_uint4(0u, 0u, 0u, 0u), obits = make_uint4(0u, 0u, 0u, 0u);
  if (ADVANCE) {
    const float* ag_in = adj_in + (size_t)b * N * N;
    const float* ng_in = nodes_in + (size_t)b * N * F;
#pragma unroll
    for (int i = 0; i < NODE_PER; ++i) {   // this thread's pieces of the observation, whichever row turns out to be cur
      const int e4 = min(tid + 256 * i, lim_n - 1);
      ob[i] = ld4(obs + (size_t)b * F + (e4 - div_f4(e4) * F4) * 4);
    }
    if (TAIL == 2) {   // bit row tid + 1 (-> row tid), node rows one down; the fp32 adjacency is not read
      if (wrap && tid + 1 < N) wbits = *reinterpret_cast<const uint4*>(gt.abits + ((size_t)b * N + tid + 1) * 4);
      if (wrap && tid < N) obits = *reinterpret_cast<const uint4*>(gt.abits + ((size_t)b * N + tid) * 4);
#pragma unroll
      for (int i = 0; i < NODE_PER; ++i) {
        const int e4 = min(tid + 256 * i, lim_n - 1), r = div_f4(e4), c = (e4 - r * F4) * 4;
        cn[i] = ld4(ng_in + (wrap ? min(r + 1, N - 1) : r) * F + c);
      }
    } else if (wrap) {
      gcm_state::load_copy<ADJ_PER, NODE_PER, true>(ca, cn, ag_in, ng_in, tid, N, N4, F, F4);
    } else if (DONATE) {   // in place, no overflow: only the node rows are read (for the image)
#pragma unroll
      for (int i = 0; i < NODE_PER; ++i) {
        const int e4 = min(tid + 256 * i, lim_n - 1);
        cn[i] = ld4(ng_in + 4 * e4);
      }
    } else {
      gcm_state::load_copy<ADJ_PER, NODE_PER, false>(ca, cn, ag_in, ng_in, tid, N, N4, F, F4);
    }
  }
  StageL<NP, FP> st_hc, st_ha;
  StageL<FP, FP> st_w0, st_w0a, st_w1, st_g[4];
  if (ex) {
    if (TAIL == 1) st_hc.load<true>(gt.cH + (size_t)b * N * FP, N, FP, FP, tid);
    if (UC) st_ha.load<true>(gt.cU + (size_t)b * N * FP, N, FP, FP, tid);

    st_w0.load<true>(M.w0 + F, F, F, 2 * F, tid);
    st_w0a.load<true>(M.w0, F, F, 2 * F, tid);
    st_w1.load<true>(M.w1, F, F, F, tid);
    if (TAIL) {
      st_g[0].load<true>(gt.gnn, FP, FP, FP, tid);
      st_g[1].load<true>(gt.gnn + FP * FP, FP, FP, FP, tid);
      st_g[2].load<true>(gt.gnn + 2 * FP * FP + FP, FP, FP, FP, tid);
      st_g[3].load<true>(gt.gnn + 3 * FP * FP + FP, FP, FP, FP, tid);
    }
  } else {
    const int H1 = gt.H1, H2 = gt.H2;
    if (TAIL == 1) st_hc.load<false>(gt.cH + (size_t)b * N * H1, N, H1, H1, tid);

    st_w0.load<false>(M.w0 + F, F, F, 2 * F, tid);
    st_w0a.load<false>(M.w0, F, F, 2 * F, tid);
    st_w1.load<false>(M.w1, F, F, F, tid);
    if (TAIL) {
      st_g[0].load<false>(gt.gnn, H1, F, F, tid);
      st_g[1].load<false>(gt.gnn + (size_t)H1 * F, H1, F, F, tid);
      st_g[2].load<false>(gt.gnn + 2 * (size_t)H1 * F + H1, H2, H1, H1, tid);
      st_g[3].load<false>(gt.gnn + 2 * (size_t)H1 * F + H1 + (size_t)H2 * H1, H2, H1, H1, tid);
    }
  }
  float pf_vec[7];                  // b0 | b1 | g0 | be0 | g1 | be1 | w2, entry lane & 31
  {
    const int o = (tid & 31) < F ? (tid & 31) : F - 1;
    pf_vec[0] = M.b0[o]; pf_vec[1] = M.b1[o]; pf_vec[2] = M.g0[o]; pf_vec[3] = M.be0[o];
    pf_vec[4] = M.g1[o]; pf_vec[5] = M.be1[o]; pf_vec[6] = M.w2[o];
  }
  const float pf_b2e = M.b2[0];
  if (TAIL) {
    const int pl = tid & 31, H1 = gt.H1, H2 = gt.H2;   // (both halves of a wave: lane & 31)
    pf_b1 = gt.gnn[2 * (size_t)H1 * F + (pl < H1 ? pl : H1 - 1)];
    pf_b2 = gt.gnn[2 * (size_t)H1 * F + H1 + 2 * (size_t)H2 * H1 + (pl < H2 ? pl : H2 - 1)];
  }
  // TAIL 2: the roll's stores - the node rows and their copy for the record (row cur = N - 1 holds the observation by then), and
  // layer 1 of the rows that keep their sources, one row up (parts 3 and 4) - in parts, one behind the barrier that
  // follows the loads and one in front of each of the next phases of the edge network (every workgroup is in the same
  // phase: issued in one piece the stores of 256 CUs meet in the same microsecond)
  // rows re-evaluated by this step (below) rather than copied: the rows that lost a source, one at a time - or, when they
  // are many, every row of a 32-row tile that holds one, on the matrix cores
  bool by_tile = false;
  auto re_evaluated = [&](const int r) {
    const uint32_t w = sAff[r >> 5];
    return by_tile ? w != 0u : ((w >> (r & 31)) & 1u) != 0u;
  };
  auto steady_stores = [&](const int k) {
    if (TAIL == 2 && wrap) {
      float* ngo = nodes_out + (size_t)b * N * F;
      float* sn = snap + (size_t)b * N * F;
      float* h1g = gt.h1_out + (size_t)b * N * gt.H1;
      float* a1g = gt.agg1_out + (size_t)b * N * F;
#ifndef GCM_LS_PLAIN
      const __amdgpu_buffer_rsrc_t r_ng = __builtin_amdgcn_make_buffer_rsrc(ngo, 0, N * F * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t r_sn = __builtin_amdgcn_make_buffer_rsrc(sn, 0, N * F * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t r_h1 = __builtin_amdgcn_make_buffer_rsrc(h1g, 0, N * gt.H1 * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t r_a1 = __builtin_amdgcn_make_buffer_rsrc(a1g, 0, N * F * 4, 0x00020000);
#endif
#pragma unroll
      for (int i = 0; i < NODE_PER; ++i) {
        if (i * 4 / NODE_PER != k) continue;
        const int e4 = tid + 256 * i;
#if defined(GCM_LS_EXP) && (GCM_LS_EXP & 2)
        if (e4 < lim_n && cn[i].x == 12345.f) {
#else
        if (e4 < lim_n) {
#endif
#ifndef GCM_LS_PLAIN
          wt_store4(r_ng, e4 * 16, cn[i].x, cn[i].y, cn[i].z, cn[i].w);
          wt_store4(r_sn, e4 * 16, cn[i].x, cn[i].y, cn[i].z, cn[i].w);
#else
          *reinterpret_cast<float4*>(ngo + e4 * 4) = cn[i];
          *reinterpret_cast<float4*>(sn + e4 * 4) = cn[i];
#endif
        }
      }
      // layer 1 of row r: that of row r + 1 of the previous state (in st_hc / st_ha) unless the row is re-evaluated
      // below; row cur is this step's
      if (k == 3) {   // the h1 image row cur's layer 2 gathers from (the re-evaluated rows are overwritten below)
        if (ex) st_hc.store<true, FS>(sHc, tid);
        else st_hc.store<false, FS>(sHc, tid);
      }
      if (ex) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i / 2 != k - 3) continue;
          const int e4 = tid + 256 * i, r = e4 / (FP / 4), c = (e4 % (FP / 4)) * 4;
#if defined(GCM_LS_EXP) && (GCM_LS_EXP & 8)
          if (r < cur && !re_evaluated(r) && st_hc.v[4 * i] == 12345.f) {
#else
          if (r < cur && !re_evaluated(r)) {
#endif
#ifndef GCM_LS_PLAIN
            wt_store4(r_h1, (r * FP + c) * 4, st_hc.v[4 * i], st_hc.v[4 * i + 1], st_hc.v[4 * i + 2], st_hc.v[4 * i + 3]);
            wt_store4(r_a1, (r * FP + c) * 4, st_ha.v[4 * i], st_ha.v[4 * i + 1], st_ha.v[4 * i + 2], st_ha.v[4 * i + 3]);
#else
            *reinterpret_cast<float4*>(h1g + r * FP + c) = make_float4(st_hc.v[4 * i], st_hc.v[4 * i + 1], st_hc.v[4 * i + 2], st_hc.v[4 * i + 3]);
            *reinterpret_cast<float4*>(a1g + r * FP + c) = make_float4(st_ha.v[4 * i], st_ha.v[4 * i + 1], st_ha.v[4 * i + 2], st_ha.v[4 * i + 3]);
#endif

          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i / 8 != k - 3) continue;
          const int e = tid + 256 * i, r = e / FP, c = e % FP;
          if (r < cur && !re_evaluated(r)) {
            if (c < gt.H1) h1g[r * gt.H1 + c] = st_hc.v[i];
            if (c < F) a1g[r * F + c] = st_ha.v[i];
          }
        }
      }
    }
  };
  if (ADVANCE) {
    // the state copy through registers (roll folded in), the node image for the edge network from the
    // same registers, the observation patched into row cur
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < NODE_PER; ++i) {
      const int e4 = tid + 256 * i, r = div_f4(e4), c = (e4 - r * F4) * 4;
      if (e4 < lim_n) {
        float4 v = cn[i];
        if (wrap && r + 1 >= N) v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r == cur) {
          v = ob[i];
          *reinterpret_cast<float4*>(sVec + c) = v;   // x_cur, 16-byte aligned (c0 reads it in pieces)
        }
        cn[i] = v;
        sX[r * FS + c] = v.x; sX[r * FS + c + 1] = v.y; sX[r * FS + c + 2] = v.z; sX[r * FS + c + 3] = v.w;
      }
    }
    // zero padding of the image (rows >= N, columns >= F)
    if (!ex)
      for (int e = tid; e < NP * FP; e += 256) {
        const int r = e / FP, c = e % FP;
        if (r >= N || c >= F) sX[r * FS + c] = 0.f;
      }
    float* ng_out = nodes_out + (size_t)b * N * F;
    if (DONATE) {
      if (TAIL == 0) {
        float* sn = snap + (size_t)b * N * F;   // the node matrix after the insert, for the record
#pragma unroll
        for (int i = 0; i < NODE_PER; ++i) {
          const int e4 = tid + 256 * i;
          if (e4 < lim_n) *reinterpret_cast<float4*>(sn + e4 * 4) = cn[i];
        }
      }
      if (TAIL == 2) {
        // the rolled bit image: row r <- row r + 1 one column down (out[r][j] = in[r + 1][j + 1]); row N - 1 empty
        // until this step's selection fills it (below)
        if (tid < NP) {
          uint4 o;
          o.x = (wbits.x >> 1) | (wbits.y << 31); o.y = (wbits.y >> 1) | (wbits.z << 31);
          o.z = (wbits.z >> 1) | (wbits.w << 31); o.w = wbits.w >> 1;
          *reinterpret_cast<uint4*>(sBits + tid * 4) = o;
          // ... and where it differs from the image in place: only those 16-byte pieces of the fp32 matrix are written,
          // by the row's thread (nothing in this kernel reads the fp32 matrix: no need to wait for the barrier; a
          // sparse adjacency changes a piece or two a row)
          if (wrap && tid < N) {
            const uint32_t ow[4] = {o.x, o.y, o.z, o.w};
            const uint32_t dw[4] = {o.x ^ obits.x, o.y ^ obits.y, o.z ^ obits.z, o.w ^ obits.w};
            float* arow = adj + ((size_t)b * N + tid) * N;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              uint32_t m = (dw[w] | (dw[w] >> 1) | (dw[w] >> 2) | (dw[w] >> 3)) & 0x11111111u;   // pieces that change
#if defined(GCM_LS_EXP) && (GCM_LS_EXP & 65)
              m = 0u;
#endif
              while (m) {
                const int sh = __builtin_ctz(m);
                m &= m - 1u;
                const uint32_t u = ow[w] >> sh;
                *reinterpret_cast<float4*>(arow + 32 * w + sh) = make_float4((float)(u & 1u), (float)((u >> 1) & 1u),
                                                                             (float)((u >> 2) & 1u), (float)((u >> 3) & 1u));
              }
            }
          }
          // a row whose sources included the dropped node (column 0 of its old row) is re-evaluated (sAff: those rows)
          const unsigned long long ma = __ballot(tid + 1 < N && (wbits.x & 1u));
          if (lane == 0) { sAff[2 * wave] = (uint32_t)ma; sAff[2 * wave + 1] = (uint32_t)(ma >> 32); }
        }
        LSTAMP(28);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the node rows' source and destination alias: every load
        LSTAMP(29);
        __syncthreads();                                   // lands before the first store
        LSTAMP(30);
        // rows r + 1 of the previous state's layer 1 - what row r of the advanced graph holds unless it lost a source:
        // requested here, behind the wait every other load of the step had to pass (nothing needs them before the
        // edge network's second product: 8.4 MB that arrive under its first)
#if defined(GCM_LS_EXP) && (GCM_LS_EXP & 32)
        if (false) {
#else
        if (ex) {
#endif
          st_hc.load<true>(gt.h1_prev + (size_t)b * N * FP, N, FP, FP, tid, 1);
          st_ha.load<true>(gt.agg1_prev + (size_t)b * N * FP, N, FP, FP, tid, 1);
        } else {
          st_hc.load<false>(gt.h1_prev + (size_t)b * N * gt.H1, N, gt.H1, gt.H1, tid, 1);
          st_ha.load<false>(gt.agg1_prev + (size_t)b * N * F, N, F, F, tid, 1);
        }
        by_tile = __popc(sAff[0]) + __popc(sAff[1]) + __popc(sAff[2]) + __popc(sAff[3]) > 12;
        LSTAMP(31);
        steady_stores(0);
      } else if (wrap) {   // source and destination alias: every load lands before the first store
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        gcm_state::store_copy<ADJ_PER, NODE_PER>(ca, cn, adj + (size_t)b * N * N, ng_out, tid, N, N4, F4, true);
      }
    } else {
      gcm_state::store_copy<ADJ_PER, NODE_PER>(ca, cn, adj + (size_t)b * N * N, ng_out, tid, N, N4, F4, wrap);
    }
    // the inserted node: store_copy's roll fix-up zeroes the last row - the thread that owns a piece of
    // row cur writes the observation behind its own copy store (same thread, same address: ordered)
#pragma unroll
    for (int i = 0; i < NODE_PER; ++i) {
      const int e4 = tid + 256 * i, r = div_f4(e4);
      if (e4 < lim_n && r == cur) *reinterpret_cast<float4*>(ng_out + e4 * 4) = cn[i];
    }
  } else {
    stage<NP>(xg, sX, N, F, F, tid);
  }
  LSTAMP(15);
  if (ex) {
    st_w0.store<true, FS>(sW0b, tid);
    st_w0a.store<true, GS>(sW0a, tid);
    st_w1.store<true, FS>(sW1, tid);
    if (TAIL) {
      if (TAIL == 1) st_hc.store<true, FS>(sHc, tid);
      if (UC) st_ha.store<true, FS>(sA, tid);
#pragma unroll
      for (int q = 0; q < 4; ++q) st_g[q].store<true, GS>(sWg + q * FP * GS, tid);
    }
  } else {
    st_w0.store<false, FS>(sW0b, tid);
    st_w0a.store<false, GS>(sW0a, tid);
    st_w1.store<false, FS>(sW1, tid);
    if (TAIL) {
      if (TAIL == 1) st_hc.store<false, FS>(sHc, tid);
#pragma unroll
      for (int q = 0; q < 4; ++q) st_g[q].store<false, GS>(sWg + q * FP * GS, tid);
    }
  }
  if (tid < FP) {
    const bool ok = tid < F;
#pragma unroll
    for (int q = 1; q < 7; ++q) sVec[q * FP + tid] = ok ? pf_vec[q] : 0.f;
  }
  __syncthreads();
  LSTAMP(16);
  steady_stores(1);
  float gr0[FP / 2], br0[FP / 2], gr1[FP / 2], br1[FP / 2];   // LayerNorm scale / shift of this thread's 16 columns
  {
    const int f0 = (tid & 1) * (FP / 2);
    auto take16 = [&](const float* src, float* dst) {   // (16-byte pieces: sVec and f0 are 16-byte aligned)
#pragma unroll
      for (int q = 0; q < FP / 8; ++q) {
        const float4 t = reinterpret_cast<const float4*>(src + f0)[q];
        dst[4 * q] = t.x; dst[4 * q + 1] = t.y; dst[4 * q + 2] = t.z; dst[4 * q + 3] = t.w;
      }
    };
    take16(sVec + 2 * FP, gr0); take16(sVec + 3 * FP, br0);
    take16(sVec + 4 * FP, gr1); take16(sVec + 5 * FP, br1);
  }
  {
    // c0[o] = b0[o] + W0a[o, :] . x_cur (ascending f), every lane its column's - from the images, not from memory
    // (the product's 16 dependent MFMAs leave the VALU idle for ~1 k cycles: c0's LDS reads and its chain of products
    //  are scheduled into them - one MFMA, four reads, two products per group)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (!UC) gcm_fused::mma32b<32>(acc, sX + 32 * wave * FS, FS, 1, sW0b, 1, FS, li, lh);   // (= gemm_rows: same order)
    float c0 = pf_vec[0];
    {
      const float* w = sW0a + li * GS;
      const float* x = sX + cur * FS;
      float wv[FP], xv[FP];
      if (ADVANCE && ex) {   // 16-byte pieces: the weight row (aligned stride) and x_cur's aligned copy in sVec
#pragma unroll
        for (int q = 0; q < FP / 4; ++q) {
          const float4 tw = reinterpret_cast<const float4*>(w)[q], tx = reinterpret_cast<const float4*>(sVec)[q];
          wv[4 * q] = tw.x; wv[4 * q + 1] = tw.y; wv[4 * q + 2] = tw.z; wv[4 * q + 3] = tw.w;
          xv[4 * q] = tx.x; xv[4 * q + 1] = tx.y; xv[4 * q + 2] = tx.z; xv[4 * q + 3] = tx.w;
        }
      } else {
#pragma unroll
        for (int f = 0; f < FP; ++f) { wv[f] = w[f]; xv[f] = x[f]; }
      }
      if (F == FP) {
#pragma unroll
        for (int f = 0; f < FP; ++f) c0 = fmaf(wv[f], xv[f], c0);
        if (!UC) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          }
        }
      } else {
#pragma unroll
        for (int f = 0; f < FP; ++f)
          if (f < F) c0 = fmaf(wv[f], xv[f], c0);
        c0 = li < F ? c0 : 0.f;
      }
    }
    if (UC) {   // the image holds U already: c0 goes to the LayerNorm's threads (this wave's rows are its own: a wave-local exchange)
      if (lh == 0) sLogit[32 * wave + li] = c0;
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) sA[(32 * wave + gcm_fused::acc_row(r, lh)) * FS + li] = acc[r] + c0;
    }
  }
  __syncthreads();
  LSTAMP(17);
  steady_stores(2);
  if (UC) {
    float c0v[FP / 2];
    const float* src = sLogit + 32 * wave + (tid & 1) * (FP / 2);
#pragma unroll
    for (int q = 0; q < FP / 8; ++q) {
      const float4 t = reinterpret_cast<const float4*>(src)[q];
      c0v[4 * q] = t.x; c0v[4 * q + 1] = t.y; c0v[4 * q + 2] = t.z; c0v[4 * q + 3] = t.w;
    }
    relu_ln_rows_t<true, true>(sA, tid, F, gr0, br0, eps0, nullptr, nullptr, true, c0v);
  } else if (F == FP) relu_ln_rows_t<true, true>(sA, tid, F, gr0, br0, eps0, nullptr, nullptr, true);
  else relu_ln_rows_t<false, true>(sA, tid, F, gr0, br0, eps0, nullptr, nullptr, true);
  __syncthreads();
  LSTAMP(18);
  steady_stores(3);
  {
    const f32x16 acc = gemm_rows(sA, sW1, wave, li, lh);
#pragma unroll
    for (int r = 0; r < 16; ++r) sB[(32 * wave + gcm_fused::acc_row(r, lh)) * FS + li] = acc[r] + sVec[FP + li];
  }
  __syncthreads();
  LSTAMP(19);
  steady_stores(4);
  if (F == FP) relu_ln_rows_t<true, true>(sB, tid, F, gr1, br1, eps1, nullptr, nullptr, true);
  else relu_ln_rows_t<false, true>(sB, tid, F, gr1, br1, eps1, nullptr, nullptr, true);
  __syncthreads();
  LSTAMP(20);
  {
    const float lg = logit_row2(sB + (tid >> 1) * FS, sVec + 6 * FP, tid & 1, pf_b2e);
    if ((tid & 1) == 0) sLogit[tid >> 1] = lg;
  }
  // (ADVANCE) the copy's stores of row cur - other threads', issued long ago - must have landed before
  // wave 0 writes the sampled entries of that row below: released here, ahead of the barrier
  if (ADVANCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (waits for this wave's stores: cheap - they were issued two GEMMs ago; an agent-scope release would write the L2 back)
  __syncthreads();
  LSTAMP(21);
  if (DONATE && tid == 255) count_out[b] = cur + 1;   // every wave read count_in long ago
  if (UC && wave == 1) {   // U[cur] = W0b x_cur for the steps to come: lanes o and o + 32 take half of the row each
    const int o = lane & 31;
    const float* w = sW0b + o * FS + 16 * lh;
    const float* x = sVec + 16 * lh;           // (x_cur's copy)
    float p = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) p = fmaf(w[k], x[k], p);
    p = gcm_xor32_add(p);
    if (lane < FP) gt.cU[((size_t)b * N + cur) * FP + lane] = p;
  }
  float z[2] = {0.f, 0.f};   // wave 0: the softmax terms, then this step's entries of row cur
  if (wave == 0) {   // gumbel-softmax over j < cur (learned.py:88-95), N <= 128: two entries per lane
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int j = lane + 64 * c;
      float nz = 0.f;
      if (j < cur) {
        const float t = pf_noise[c];
        nz = noise_is_exp ? -logf(t) : t;
      }
      z[c] = j < cur ? sLogit[j] + nz : -INFINITY;
      m = fmaxf(m, z[c]);
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      z[c] = (lane + 64 * c < cur) ? expf(z[c] - m) : 0.f;
      s += z[c];
    }
    s = wave_sum(s);
    const float inv = s > 0.f ? 1.f / s : 0.f;
    float* row = adj + ((size_t)b * N + cur) * N;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int j = lane + 64 * c;
      if (j < N) {
        const float p = z[c] * inv;
        soft[(size_t)b * N + j] = p;
        if (j < cur) {
          const float edge = (p - cutoff > 0.f) ? 1.f : 0.f;     // STE forward (util.py:12)
          float old;
          if (ADVANCE) old = wrap ? 0.f : pf_old[c];
          else old = row[j];
          const float nv = (edge + old > 0.f) ? 1.f : 0.f;       // learned.py:108-110
          row[j] = nv;
          if (DONATE) row_out[(size_t)b * N + j] = nv;
          if (TAIL) z[c] = nv;
        } else {
          if (DONATE) row_out[(size_t)b * N + j] = 0.f;
          if (TAIL) z[c] = 0.f;
        }
      } else if (TAIL) {
        z[c] = 0.f;
      }
    }
    LSTAMP(22);
    if (TAIL == 2) {   // row cur of the bit image: this step's entries (the rolled image's last row is empty)
      const unsigned long long m0 = __ballot(z[0] != 0.f), m1 = __ballot(z[1] != 0.f);
      if (lane == 0) {
        sBits[cur * 4] = (uint32_t)m0; sBits[cur * 4 + 1] = (uint32_t)(m0 >> 32);
        sBits[cur * 4 + 2] = (uint32_t)m1; sBits[cur * 4 + 3] = (uint32_t)(m1 >> 32);
      }
    }
  }
  if (TAIL == 2) {
    // ---- layer 1 of the rows that LOST a source (sAff, per 32-row tile; rare: the sampled adjacency is sparse), while
    // wave 0 is in the softmax: waves 1 - 3, tile by tile - agg1 = Adj X from the rolled bit image and the node image on
    // the matrix cores, h1 = act1([agg1 | x] W^T + b1); the rows go into the h1 image (row cur's agg2 reads it below)
    // and into the record.  Row cur itself is wave 0's, below (its bits are being written: whatever this computes for
    // it is dropped).
    const int H1 = gt.H1;
    if (wave != 0 && !by_tile) {
      // one row at a time, round robin over waves 1 - 3, as wave 0 does row cur below: the row's sources from the bit
      // image as a rank-compacted list, agg1 gathered from the node image, h1 as two half products (scratch: this
      // wave's corner of sA, which the edge network is done with)
      int* sIdxW = reinterpret_cast<int*>(sA + wave * 256);
      float* sUW = sA + wave * 256 + 128;        // [agg1 | x_t], 32 each
      float* a1g = gt.agg1_out + (size_t)b * N * F;
      float* h1g = gt.h1_out + (size_t)b * N * H1;
      const int fl_ = lane < FP ? lane : FP - 1, o = lane & 31;
      int idx = 0;
#pragma unroll 1
      for (int wq = 0; wq < 4; ++wq) {
        uint32_t mw = __builtin_amdgcn_readfirstlane(sAff[wq]);
#pragma unroll 1
        while (mw) {
          const int t = 32 * wq + __builtin_ctz(mw);
          mw &= mw - 1u;
          const bool mine = idx % 3 == wave - 1;
          ++idx;
          if (!mine) continue;
          const uint32_t c0 = sBits[t * 4], c1 = sBits[t * 4 + 1], c2 = sBits[t * 4 + 2], c3 = sBits[t * 4 + 3];
          const unsigned long long m0 = ((unsigned long long)c1 << 32) | c0, m1 = ((unsigned long long)c3 << 32) | c2;
          const int n0 = __popcll(m0), n_sel = __builtin_amdgcn_readfirstlane(n0 + __popcll(m1));
          {
            const unsigned long long below = (1ull << lane) - 1ull;
            if ((m0 >> lane) & 1ull) sIdxW[__popcll(m0 & below)] = lane;
            if ((m1 >> lane) & 1ull) sIdxW[n0 + __popcll(m1 & below)] = lane + 64;
          }
          wsync();
          float agg1 = 0.f;
#pragma unroll 1
          for (int q0 = 0; q0 < n_sel; q0 += 8) {
            const int4 ia = *reinterpret_cast<const int4*>(sIdxW + q0), ib = *reinterpret_cast<const int4*>(sIdxW + q0 + 4);
            const int js[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
            float xa[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) xa[q] = sX[(q0 + q < n_sel ? js[q] : 0) * FS + fl_];
#pragma unroll
            for (int q = 0; q < 8; ++q)
              if (q0 + q < n_sel) agg1 += xa[q];
          }
          agg1 = lane < F ? agg1 : 0.f;
          const float xt = sX[t * FS + fl_];
          if (lh == 0) { sUW[o] = agg1; sUW[32 + o] = xt; }
          wsync();
          float p1 = half_dot(sWg + (lh ? FP * GS : 0) + o * GS, sUW + 32 * lh);
          p1 += (gt.has_bias & 1) && o < H1 ? pf_b1 : 0.f;
          float h = gcm_act(p1, gt.act1);
          h = o < H1 ? h : 0.f;
          if (lh == 0) sHc[t * FS + o] = h;
          if (lane < H1) h1g[t * H1 + lane] = h;
          if (lane < F) a1g[t * F + lane] = agg1;
          wsync();
        }
      }
    }
    if (wave != 0 && by_tile) {
#pragma unroll 1
      for (int T = (wave == 1 ? 0 : wave); T < (wave == 1 ? 2 : wave + 1); ++T) {
        if (!sAff[T]) continue;
        const int trow = 32 * T + li;            // A-operand row of this lane
        uint32_t wl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) wl[q] = sBits[trow * 4 + q] >> lh;   // bit (2 q' + lh) of word q -> bit 2 q' here
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 1
        for (int kt = 0; kt <= T; ++kt) {        // sources are older rows: blocks kt <= T only
          const float* bp = sX + (kt * 32 + lh) * FS + li;
          const uint32_t wk = wl[0];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float a = (float)((wk >> (2 * q)) & 1u);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[2 * q * FS], acc, 0, 0, 0);
          }
          wl[0] = wl[1]; wl[1] = wl[2]; wl[2] = wl[3];
        }
        float* a1g = gt.agg1_out + (size_t)b * N * F;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int t = 32 * T + gcm_fused::acc_row(r, lh);
          sA[t * FS + li] = acc[r];
#if defined(GCM_LS_EXP) && (GCM_LS_EXP & 4)
          if (t < cur && li < F && acc[r] == 12345.f) a1g[t * F + li] = acc[r];
#else
          if (t < cur && li < F) a1g[t * F + li] = acc[r];
#endif
        }
        wsync();
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        gcm_fused::mma32b<FP>(acc, sA + 32 * T * FS, FS, 1, sWg, 1, GS, li, lh);            // B(k, o) = W[o][k] (rows at stride GS)
        gcm_fused::mma32b<FP>(acc, sX + 32 * T * FS, FS, 1, sWg + FP * GS, 1, GS, li, lh);
        const float bias1 = (gt.has_bias & 1) && li < H1 ? pf_b1 : 0.f;
        float* h1g = gt.h1_out + (size_t)b * N * H1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int t = 32 * T + gcm_fused::acc_row(r, lh);
          const float h = (t < N && li < H1) ? gcm_act(acc[r] + bias1, gt.act1) : 0.f;
          if (t < cur) {
            sHc[t * FS + li] = h;
#if defined(GCM_LS_EXP) && (GCM_LS_EXP & 4)
            if (li < H1 && h == 12345.f) h1g[t * H1 + li] = h;
#else
            if (li < H1) h1g[t * H1 + li] = h;
#endif
          }
        }
      }
    }
    __syncthreads();   // row cur's bits (wave 0) and the re-evaluated rows of the h1 image are in LDS
    LSTAMP(27);
#if defined(GCM_LS_EXP) && (GCM_LS_EXP & 16)
    if (tid < N && sBits[tid * 4] == 0x12345u)
#else
    if (tid < N)
#endif
      *reinterpret_cast<uint4*>(gt.abits + ((size_t)b * N + tid) * 4) = *reinterpret_cast<const uint4*>(sBits + tid * 4);
  }
  if (wave == 0) {
    if (TAIL) {
      // ---- the GNN on row cur (see GnnTail): the selected rows S = { j < cur : row[j] = 1 }, ascending -------
      const int H1 = gt.H1, H2 = gt.H2;
      const unsigned long long m0 = __ballot(z[0] != 0.f), m1 = __ballot(z[1] != 0.f);
      const int fl_ = lane < FP ? lane : FP - 1;
      float agg1 = 0.f, agg2 = 0.f;          // lane f: agg1[f]; lane h: agg2[h]
      // the selected rows as a compact ascending list: every selected lane writes its row at its rank (a scalar
      // find-first-bit chain over the two masks cost 1.8 k cycles here), then eight rows per trip, their LDS reads
      // in flight together, added in ascending order
      const int n0 = __popcll(m0), n_sel = n0 + __popcll(m1);
      int* sIdx = reinterpret_cast<int*>(sLogit);   // (the logits are consumed; sU takes this place below)
      {
        const unsigned long long below = (1ull << lane) - 1ull;
        if (z[0] != 0.f) sIdx[__popcll(m0 & below)] = lane;
        if (z[1] != 0.f) sIdx[n0 + __popcll(m1 & below)] = lane + 64;
      }
      wsync();
#pragma unroll 1
      for (int q0 = 0; q0 < n_sel; q0 += 8) {
        const int4 ia = *reinterpret_cast<const int4*>(sIdx + q0), ib = *reinterpret_cast<const int4*>(sIdx + q0 + 4);
        const int js[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
        float xa[8], ha[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int j = q0 + q < n_sel ? js[q] : 0;
          xa[q] = sX[j * FS + fl_];
          ha[q] = sHc[j * FS + fl_];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (q0 + q < n_sel) { agg1 += xa[q]; agg2 += ha[q]; }
      }
      wsync();
      const float xc = sX[cur * FS + fl_];    // lane f: x[cur][f]
      agg1 = lane < F ? agg1 : 0.f;
      agg2 = lane < H1 ? agg2 : 0.f;
      // h1[cur][h] = act1(b1[h] + sum_f W_rel1[h][f] agg1[f] + W_root1[h][f] x[cur][f]): lane h takes the W_rel
      // half, lane 32 + h the W_root half; the operand vectors go through LDS (16-byte broadcast reads) and the
      // weight rows sit at a 16-byte aligned stride (GS), so a half is 8 + 8 LDS reads and 32 products in two
      // chains.  (The first form - 64 + 64 v_readlane broadcasts and dependent fmas on one lane - was a fifth of
      // the kernel.)
      const int o = lane & 31;
      float* sU = sLogit;                    // [agg1 | x_cur | agg2 | h1_cur], 32 each (the logits are consumed)
      if (lh == 0) { sU[o] = agg1; sU[32 + o] = xc; sU[64 + o] = agg2; }
      wsync();
      LSTAMP(24);
      float p1 = half_dot(sWg + (lh ? FP * GS : 0) + o * GS, sU + 32 * lh);
      p1 += (gt.has_bias & 1) && o < H1 ? pf_b1 : 0.f;
      float h1c = gcm_act(p1, gt.act1);
      h1c = o < H1 ? h1c : 0.f;              // (both halves hold h1[cur][o])
      if (lh == 0) sU[96 + o] = h1c;
      wsync();
      // mx[o] = act2(b2[o] + sum_h W_rel2[o][h] agg2[h] + W_root2[o][h] h1[cur][h]), the same way
      LSTAMP(25);
      float p2 = half_dot(sWg + (lh ? 3 * FP * GS : 2 * FP * GS) + o * GS, sU + 64 + 32 * lh);
      p2 += (gt.has_bias & 2) && o < H2 ? pf_b2 : 0.f;
      const float v = gcm_act(p2, gt.act2);
      LSTAMP(26);
      const size_t rc = (size_t)b * N + cur;
      if (lane < H1) {
        (TAIL == 1 ? gt.cH : gt.h1_out)[rc * H1 + lane] = h1c;
        gt.agg2_out[(size_t)b * H1 + lane] = agg2;
      }
      if (lane < F) {
        (TAIL == 1 ? gt.cA : gt.agg1_out)[rc * F + lane] = agg1;
        if (TAIL == 1) gt.cX[rc * F + lane] = xc;
      }
      if (lane < H2) gt.mx_out[(size_t)b * H2 + lane] = v;
      const bool bad = __any(lane < H2 && !isfinite(v));
      if (bad && lane == 0) atomicOr(flags, GCM_FLAG_NONFINITE);
    }
  }
  if (ADVANCE && cur_host >= 0 && tid == 0 && n_chk != (int64_t)cur_host) atomicOr(flags, GCM_FLAG_BAD_COUNT);
  if (TAIL == 2 && tid == 0 && n_chk != (int64_t)N) atomicOr(flags, GCM_FLAG_BAD_COUNT);
  LSTAMP(23);
  LSPAN(1);
}

// ---------------------------------------------------------------------------------------------
// k_learned_select<2, 1, true> re-cut for EIGHT tile waves + a tail wave per graph (round 6): the cached step of a chain
// from empty graphs on a donated state at the exact shapes N = 128, F = H1 = H2 = 32, the host's cur (cur_host >= 0).
//
// Why: the four-wave kernel is one wave per SIMD walking nine phases through ~100 KB of LDS images (node rows and h1 rows
// staged only so that the tail can gather a handful of them, the edge network's two 32-column images written and read
// back between every pair of phases, five workgroup barriers) - a wave alone on its SIMD issues a dependent instruction
// about every eight cycles (§3.12 of DESIGN.md), so its 5.2 k-cycle edge network and 4 k-cycle softmax + tail are mostly
// issue bubbles.  Here (7.0 -> 4.9 us per step in situ, cfg5 21.6 -> 26 M belief states/s from this kernel alone):
//   * the edge network lives in REGISTERS, 16 candidate rows per wave, two waves per SIMD: lane (m, g) holds features
//     [8 g, 8 g + 8) of row 16 w + m - its piece of U[j] = W0b x_j straight from the chain's cache (two 16-byte loads),
//     + c0, ReLU, LayerNorm (row statistics: eight values in the lane, then v_permlane16_swap / v_permlane32_swap - VALU
//     instructions where `__shfl_xor` is an LDS round trip), and those eight values ARE an operand of
//     v_mfma_f32_16x16x4_f32 (k index = g: instruction s pairs H0[m][8 g + s] with W1[col][8 g + s]).  W1's rows are the
//     A operand and H0 the B operand, so the accumulators hold the TRANSPOSED product: row m at columns 16 ct + 4 g + i -
//     eight columns of ONE row per lane again, and the second LayerNorm and the F -> 1 layer reduce exactly like the first
//     (the other way round a lane held four rows at two columns: twelve 16-lane DPP reductions, 1.1 k cycles against 0.7 k);
//   * what barrier 1 waits for is one far-memory round trip and nothing else: c0 = b0 + W0a x_cur and U[cur] = W0b x_cur
//     on ONE wave (lower / upper half), the observation through the SCALAR path (two s_load_dwordx16: as vector loads it
//     was four more 1 KB requests in front of the products); W1 and the six vectors come through LDS once per graph (waves
//     5 / 6; as eight waves' own loads they were a third of the bytes the CU's one address path had to process first);
//     every kernel argument is in scalar registers before the first load (one s_load batch: left alone the compiler loads
//     each pointer in the block that first uses it - six dependent kernarg round trips);
//   * a NINTH wave owns everything behind barrier 2 - gumbel-softmax (the draws transformed while the tiles work),
//     threshold, adjacency row, the GNN on row cur: its sixteen 1 KB weight-row loads are issued behind barrier 1, where
//     they no longer sit in the address path ahead of the loads that barrier waits for (in-kernel stamps: barrier 1 released
//     at 4.4 k cycles with them in front, 3.0 k behind), and the selected rows (at most a handful: entries above
//     1 / (1 + num_edge_samples)) are gathered from LDS images of the node and h1 rows that the tile waves fill from the
//     rows they hold anyway (from global memory the gather was a 2 k-cycle round trip on the critical path).
// Same record, caches and state as k_learned_select<2, 1, true> (gcm_learned_bptt_cached reads them unchanged); the
// sums of a row are associated differently (eight per lane group), so logits agree to rounding, not bit for bit.
// GCM_STEP_FOUR_WAVES in has_bias selects the four-wave kernel (the A/B: tools/ab_cfg5.sh).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float row16_sum(float v) {   // sum over the 16 lanes of a DPP row, in every lane of it
  v += GCM_DPP_F(v, 0xB1, 0xF, 0.f);    // quad_perm [1,0,3,2]
  v += GCM_DPP_F(v, 0x4E, 0xF, 0.f);    // quad_perm [2,3,0,1]
  v += GCM_DPP_F(v, 0x141, 0xF, 0.f);   // row_half_mirror
  v += GCM_DPP_F(v, 0x140, 0xF, 0.f);   // row_mirror
  return v;
}

__global__ __launch_bounds__(576) void k_learned_select8(
    const float* __restrict__ obs, float* __restrict__ nodes, float* __restrict__ adj, const int64_t* count_in,
    int64_t* count_out, int64_t* __restrict__ cur_out, const float* __restrict__ noise, int noise_is_exp,
    const float* __restrict__ mlp, float eps0, float eps1, float cutoff, float* __restrict__ soft,
    float* __restrict__ row_out, uint32_t* __restrict__ flags, GnnTail gt, int cur) {
  constexpr int N = NP, F = FP;
  constexpr int TW = 8;   // the tail wave: softmax, selection and the GNN on row cur; it has no tile (16 TW >= N)
#ifdef GCM_STAMPS   // every wave's first instruction and its arrival at barrier 1, kept in registers and written at the end (a
                    // stamp's store would sit in front of the wave's next vmcnt wait)
  unsigned long long t_start, t_arrive;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_start)::"memory");
#endif
  __shared__ __attribute__((aligned(16))) float sC0[FP];
  __shared__ __attribute__((aligned(16))) float sLogit8[NP];
  __shared__ __attribute__((aligned(16))) float sU8[4 * FP];
  __shared__ int sIdx8[NP];
  // the candidate rows' x and h1 for the tail's gather (a handful of them, known only behind the softmax: fetched from
  // global memory there the gather was a 2 k-cycle round trip on the tail wave's critical path; every wave brings its 16 rows
  // along with its other loads instead)
  // W1 and the edge network's six vectors, once per graph (waves 6 / 5 bring them): as eight
  // waves' own global loads they were a third of the bytes the CU's address path had to process before anything ran
  constexpr int WS8 = FP + 4;
  __shared__ __attribute__((aligned(16))) float sW18[FP * WS8];
  __shared__ __attribute__((aligned(16))) float sVec8[6 * FP];   // g0 | be0 | b1 | g1 | be1 | w2
  constexpr int XS8 = FP + 4;
  __shared__ __attribute__((aligned(16))) float sX8[NP * XS8];
  __shared__ __attribute__((aligned(16))) float sH8[NP * XS8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, m = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const Mlp M = unpack_mlp(mlp, F);
  const size_t gb = (size_t)b;
  const bool tile_on = 16 * wave < cur;   // (uniform) this wave's rows hold a candidate (j < cur)
  // every kernel argument in scalar registers HERE, one s_load batch and one wait: left alone the compiler loads each
  // pointer in the block that first uses it, and the load phase below was six dependent kernarg round trips long
  asm volatile("" ::"s"(obs), "s"(nodes), "s"(adj), "s"(count_in), "s"(count_out), "s"(cur_out), "s"(noise), "s"(mlp), "s"(soft),
               "s"(row_out), "s"(flags), "s"(gt.gnn), "s"(gt.cH), "s"(gt.cA), "s"(gt.cX), "s"(gt.cU), "s"(gt.mx_out), "s"(gt.agg2_out), "s"(cur));

  LSTAMP(0);
  // ---- loads, everything up front -----------------------------------------------------------------------------------
  int64_t n_chk = 0;
  if (tid == 64 * TW) n_chk = count_in[b];
  // the edge network's operands of this wave's 16 rows
  const int r = 16 * wave + m;
  f32x4 u4[2], w1[2][2], xg4[2], hg4[2];
  float g0v[8], be0v[8];
  if (tile_on) {
#pragma unroll
    for (int q = 0; q < 2; ++q) u4[q] = *reinterpret_cast<const f32x4*>(gt.cU + (gb * N + r) * FP + 8 * g + 4 * q);
  }
  const float b2e = M.b2[0];
  // wave 7: c0 = W0a x_cur + b0 in front of barrier 1 - lane (o, half) takes sixteen k of row o -, U[cur] = W0b x_cur behind
  // it (nothing in this step reads it: its weight rows stay out of the burst that c0's operands travel in).  The
  // observation comes through the SCALAR path (one row for the whole wave; as vector loads it was four more 1 KB requests
  // in front of the products - and the observation comes from far memory)
  f32x4 wq[4];
  typedef float f32x16s __attribute__((ext_vector_type(16)));
  f32x16s xsa, xsb;   // (s_load_dwordx16 x 2 by hand: the compiler keeps a uniform load behind a branch on the vector path)
  {   // (every wave: a scalar value defined under a branch becomes a vector register)
    const float* xrow = obs + gb * FP;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(xsa), "=&s"(xsb) : "s"(xrow));
  }
  float b0o = 0.f;
  if (wave == 7) {
    const float* wrow = M.w0 + li * 2 * FP + 16 * lh;   // w0 [F][2F]: W0a | W0b
#pragma unroll
    for (int q = 0; q < 4; ++q) wq[q] = *reinterpret_cast<const f32x4*>(wrow + 4 * q);
    b0o = M.b0[li];
  }
  f32x4 stg[4];   // wave 6: W1 (row lane >> 1, sixteen floats);  wave 5: the six vectors (lanes 0 - 47, four floats each)
  if (wave == 6) {
#pragma unroll
    for (int q = 0; q < 4; ++q) stg[q] = *reinterpret_cast<const f32x4*>(M.w1 + (lane >> 1) * FP + 16 * (lane & 1) + 4 * q);
  } else if (wave == 5 && lane < 48) {
    const float* vsrc = lane < 8 ? M.g0 + 4 * lane : (lane < 16 ? M.be0 + 4 * (lane - 8) : (lane < 24 ? M.b1 + 4 * (lane - 16) :
                        (lane < 32 ? M.g1 + 4 * (lane - 24) : (lane < 40 ? M.be1 + 4 * (lane - 32) : M.w2 + 4 * (lane - 40)))));
    stg[0] = *reinterpret_cast<const f32x4*>(vsrc);
  }
  float xo5 = 0.f;
  if (wave == 5 && lane < FP) xo5 = obs[gb * FP + lane];
  // the tail wave: the gumbel draws and the old row (softmax / selection) now, the GNN's weight rows behind barrier 1 (they
  // are wanted after barrier 2; in front of barrier 1 their sixteen 1 KB requests sat in the CU's one address path ahead of
  // the loads that barrier waits for)
  float pf_noise[2] = {0.f, 0.f}, pf_old[2] = {0.f, 0.f}, xc = 0.f, gb1 = 0.f, gb2 = 0.f;
  f32x4 wg1[8], wg2[8];
  if (wave == TW) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      pf_noise[c] = noise[gb * N + lane + 64 * c];
      pf_old[c] = adj[(gb * N + cur) * N + lane + 64 * c];
    }
    xc = obs[gb * FP + li];
  }
  asm volatile("" ::: "memory");
  LSTAMP(1);

  // ---- c0, U[cur]; the observation into the state and the node cache (wave 5) ---------------------------------------
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(xsa), "+s"(xsb));   // (the barrier below waits for the same row through c0)
  // this half's sixteen products with the observation in scalar registers (both halves' chains, one select), the other
  // half's sum added
  auto half_dot_s = [&](const f32x4 (&w)[4]) {
    float pa = 0.f, pb = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pa = fmaf(w[q][e], xsa[4 * q + e], pa);
        pb = fmaf(w[q][e], xsb[4 * q + e], pb);
      }
    const float p = lh ? pb : pa;
    return gcm_xor32_add(p);
  };
  if (wave == 7) {
    const float p = half_dot_s(wq);
    if (lh == 0) sC0[li] = p + b0o;
  }
  if (wave == 6) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(sW18 + (lane >> 1) * WS8 + 16 * (lane & 1) + 4 * q) = stg[q];
  } else if (wave == 5) {
    if (lane < 48) *reinterpret_cast<f32x4*>(sVec8 + 4 * lane) = stg[0];
    if (lane < FP) {
      nodes[(gb * N + cur) * FP + lane] = xo5;     // gcm.py:274
      gt.cX[(gb * N + cur) * FP + lane] = xo5;
    }
  }
  LSTAMP(2);
#ifdef GCM_STAMPS
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_arrive)::"memory");
#endif
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // #1: c0
  LSTAMP(3);
  if (wave == TW) {
    const float* r1 = gt.gnn + (lh ? FP * FP : 0) + li * FP;                       // W_rel1 / W_root1 row o
    const float* r2 = gt.gnn + 2 * FP * FP + FP + (lh ? FP * FP : 0) + li * FP;    // W_rel2 / W_root2 row o
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      wg1[q] = *reinterpret_cast<const f32x4*>(r1 + 4 * q);
      wg2[q] = *reinterpret_cast<const f32x4*>(r2 + 4 * q);
    }
    gb1 = gt.gnn[2 * FP * FP + li];
    gb2 = gt.gnn[2 * FP * FP + FP + 2 * FP * FP + li];
    // the gumbel draws from the exponential ones (learned.py:88-89), while the tiles work
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float t = pf_noise[c];
      pf_noise[c] = (lane + 64 * c < cur) ? (noise_is_exp ? -logf(t) : t) : 0.f;
    }
  }

  // ---- the edge network on this wave's 16 rows, in registers (learned.py:38-51) -----------------------------------
  if (tile_on) {
    // this wave's rows of the tail's gather images, requested now: in front of barrier 1 they were a third of the bytes
    // of the burst (256 CUs at once) that c0's operands travel in; they are wanted at the end of this block
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      xg4[q] = *reinterpret_cast<const f32x4*>(nodes + (gb * N + r) * FP + 8 * g + 4 * q);
      hg4[q] = *reinterpret_cast<const f32x4*>(gt.cH + (gb * N + r) * FP + 8 * g + 4 * q);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 2; ++q) w1[ct][q] = *reinterpret_cast<const f32x4*>(sW18 + (16 * ct + m) * WS8 + 8 * g + 4 * q);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const f32x4 tg = *reinterpret_cast<const f32x4*>(sVec8 + 8 * g + 4 * q);
      const f32x4 tb = *reinterpret_cast<const f32x4*>(sVec8 + FP + 8 * g + 4 * q);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) { g0v[4 * q + kk] = tg[kk]; be0v[4 * q + kk] = tb[kk]; }
    }
    f32x4 b1v[2], g1v[2], be1v[2], w2v[2];   // columns 16 ct + 4 g + (0 .. 3): the transposed product's (below)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      b1v[ct] = *reinterpret_cast<const f32x4*>(sVec8 + 2 * FP + 16 * ct + 4 * g);
      g1v[ct] = *reinterpret_cast<const f32x4*>(sVec8 + 3 * FP + 16 * ct + 4 * g);
      be1v[ct] = *reinterpret_cast<const f32x4*>(sVec8 + 4 * FP + 16 * ct + 4 * g);
      w2v[ct] = *reinterpret_cast<const f32x4*>(sVec8 + 5 * FP + 16 * ct + 4 * g);
    }
    float a[8];
    {
      const f32x4 c0a = *reinterpret_cast<const f32x4*>(sC0 + 8 * g), c0b = *reinterpret_cast<const f32x4*>(sC0 + 8 * g + 4);
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float v = (k < 4 ? u4[0][k] : u4[1][k - 4]) + (k < 4 ? c0a[k] : c0b[k - 4]);
        a[k] = v > 0.f ? v : 0.f;
        s += a[k];
      }
      s = gcm_xor16_add(s);
      s = gcm_xor32_add(s);
      const float mean = s / (float)FP;
      float qv = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float d = a[k] - mean; qv = fmaf(d, d, qv); }
      qv = gcm_xor16_add(qv);
      qv = gcm_xor32_add(qv);
      const float rstd = rsqrtf(qv / (float)FP + eps0);
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] = fmaf((a[k] - mean) * rstd, g0v[k], be0v[k]);
    }
    LSTAMP(4);
    f32x4 acc[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const f32x4 w = w1[ct][s >> 2];
        const float wv = (s & 3) == 0 ? w.x : ((s & 3) == 1 ? w.y : ((s & 3) == 2 ? w.z : w.w));
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, a[s], acc[ct], 0, 0, 0);
      }
    // W1 as the A operand, H0 as B: the accumulators hold the TRANSPOSED product, P1[row m][16 ct + 4 g + i] - eight
    // columns of ONE row per lane, like the first LayerNorm's operands (the other way round a lane held four rows at two
    // columns: twelve 16-lane DPP reductions; here three sums of eight in the lane and two row / half swaps each)
    LSTAMP(5);
    {
      float v[8];
      float sm = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float t = acc[k >> 2][k & 3] + b1v[k >> 2][k & 3];
        v[k] = t > 0.f ? t : 0.f;
        sm += v[k];
      }
      sm = gcm_xor16_add(sm);
      sm = gcm_xor32_add(sm);
      const float mean = sm / (float)FP;
      float qv = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) { v[k] -= mean; qv = fmaf(v[k], v[k], qv); }
      qv = gcm_xor16_add(qv);
      qv = gcm_xor32_add(qv);
      const float rstd = rsqrtf(qv / (float)FP + eps1);
      float lgp = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) lgp = fmaf(w2v[k >> 2][k & 3], fmaf(v[k] * rstd, g1v[k >> 2][k & 3], be1v[k >> 2][k & 3]), lgp);
      lgp = gcm_xor16_add(lgp);
      lgp = gcm_xor32_add(lgp);
      if (g == 0) sLogit8[16 * wave + m] = lgp + b2e;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {   // this wave's rows of the tail's gather images
      *reinterpret_cast<f32x4*>(sX8 + r * XS8 + 8 * g + 4 * q) = xg4[q];
      *reinterpret_cast<f32x4*>(sH8 + r * XS8 + 8 * g + 4 * q) = hg4[q];
    }
  }
  if (wave == 7) {   // U[cur] = W0b x_cur into the chain's cache (later steps' candidates)
    const float* wrow = M.w0 + li * 2 * FP + FP + 16 * lh;
    f32x4 wu[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) wu[q] = *reinterpret_cast<const f32x4*>(wrow + 4 * q);
    const float p = half_dot_s(wu);
    if (lh == 0) gt.cU[(gb * N + cur) * FP + li] = p;
  }
  LSTAMP(6);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // #2: the logits
#ifdef GCM_STAMPS
  if (b == 0 && lane == 0 && wave < 8) { g_stamps[16 + wave] = t_arrive; g_stamps[24 + wave] = t_start; }
#endif
  if (wave != TW) return;
  LSTAMPT(7);

  // ---- the tail wave: gumbel-softmax over j < cur (learned.py:88-95), threshold, row cur of the adjacency -------------------
  float z[2];
  {
    float mx_ = -INFINITY;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int j = lane + 64 * c;
      z[c] = j < cur ? sLogit8[j] + pf_noise[c] : -INFINITY;
      mx_ = fmaxf(mx_, z[c]);
    }
    mx_ = wave_max(mx_);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      z[c] = (lane + 64 * c < cur) ? expf(z[c] - mx_) : 0.f;
      s += z[c];
    }
    s = wave_sum(s);
    const float inv = s > 0.f ? 1.f / s : 0.f;
    float* row = adj + (gb * N + cur) * N;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int j = lane + 64 * c;
      const float p = z[c] * inv;
      soft[gb * N + j] = p;
      if (j < cur) {
        const float edge = (p - cutoff > 0.f) ? 1.f : 0.f;     // STE forward (util.py:12)
        const float nv = (edge + pf_old[c] > 0.f) ? 1.f : 0.f;   // learned.py:108-110
        row[j] = nv;
        row_out[gb * N + j] = nv;
        z[c] = nv;
      } else {
        row_out[gb * N + j] = 0.f;
        z[c] = 0.f;
      }
    }
  }
  LSTAMPT(8);
  // ---- the GNN on row cur (GnnTail): the selected rows, ascending, gathered from the node matrix and the h1 cache -----
  {
    const unsigned long long m0 = __ballot(z[0] != 0.f), m1 = __ballot(z[1] != 0.f);
    const int n0 = __popcll(m0), n_sel = n0 + __popcll(m1);
    {
      const unsigned long long below = (1ull << lane) - 1ull;
      if (z[0] != 0.f) sIdx8[__popcll(m0 & below)] = lane;
      if (z[1] != 0.f) sIdx8[n0 + __popcll(m1 & below)] = lane + 64;
    }
    wsync();
    float agg1 = 0.f, agg2 = 0.f;
#pragma unroll 1
    for (int q0 = 0; q0 < n_sel; q0 += 8) {
      const int4 ia = *reinterpret_cast<const int4*>(sIdx8 + q0), ib = *reinterpret_cast<const int4*>(sIdx8 + q0 + 4);
      const int js[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
      float xa[8], ha[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = q0 + q < n_sel ? js[q] : 0;
        xa[q] = sX8[j * XS8 + li];
        ha[q] = sH8[j * XS8 + li];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (q0 + q < n_sel) { agg1 += xa[q]; agg2 += ha[q]; }
    }
    LSTAMPT(9);
    float* sU = sU8;   // [agg1 | x_cur | agg2 | h1_cur], 32 each
    if (lh == 0) { sU[li] = agg1; sU[32 + li] = xc; sU[64 + li] = agg2; }
    wsync();
    // half_dot (the four-wave kernel's): this half's 32 products in two chains, the other half's sum added
    auto hdot = [&](const f32x4 (&w)[8], const float* u) {
      float pa = 0.f, pb = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 uv = reinterpret_cast<const float4*>(u)[q];
        pa = fmaf(w[q].x, uv.x, pa);
        pb = fmaf(w[q].y, uv.y, pb);
        pa = fmaf(w[q].z, uv.z, pa);
        pb = fmaf(w[q].w, uv.w, pb);
      }
      const float p = pa + pb;
      return gcm_xor32_add(p);
    };
    float p1 = hdot(wg1, sU + 32 * lh);
    p1 += (gt.has_bias & 1) ? gb1 : 0.f;
    const float h1c = gcm_act(p1, gt.act1);
    if (lh == 0) sU[96 + li] = h1c;
    wsync();
    float p2 = hdot(wg2, sU + 64 + 32 * lh);
    p2 += (gt.has_bias & 2) ? gb2 : 0.f;
    const float v = gcm_act(p2, gt.act2);
    LSTAMPT(10);
    const size_t rc = gb * N + cur;
    if (lane < FP) {
      gt.cH[rc * FP + lane] = h1c;
      gt.agg2_out[gb * FP + lane] = agg2;
      gt.cA[rc * FP + lane] = agg1;
      gt.mx_out[gb * FP + lane] = v;
    }
    const bool bad = __any(lane < FP && !isfinite(v));
    if (lane == 0) {
      cur_out[b] = cur;
      count_out[b] = cur + 1;
      const uint32_t f = (bad ? GCM_FLAG_NONFINITE : 0u) | (n_chk != (int64_t)cur ? GCM_FLAG_BAD_COUNT : 0u);
      if (f) atomicOr(flags, f);
    }
    LSTAMPT(11);
  }
}

// ---------------------------------------------------------------------------------------------
// Time-parallel forward of a whole rollout (round 4; DenseGCM.rollout with LearnedEdge from EMPTY graphs, T <= N
// steps, observations without gradient).  Nothing in the selection of step t depends on another step's RESULT: the
// edge network scores pairs of raw observations (learned.py:53-87), the gumbel draws are given, and with empty
// graphs to start from node j is observation j.  So every (graph, step) is independent work of THREE launches
// instead of T latency-bound ones: k_learned_roll_logits (the edge network of k_learned_select<2, 1> with
// cur = t and the node image from the observation tensor [T, B, F], per 32-row block with a candidate row),
// k_learned_roll_pick (softmax, threshold, adjacency / node row, layer 1 of the GNN on row cur - h1 / agg1 / x into
// the chain's caches), and, once every h1 row exists, k_learned_roll_l2 (layer 2 on row cur -> the belief states),
// the last two one wave per (graph, step).  (The first version ran launches 1 + 2 as one workgroup per (graph,
// step) with all 128 rows staged and the weights loaded per item: 303 us at cfg5.)  Same arithmetic in the same order as the
// cached per-step kernel: same sampled edges, same beliefs.  The records (gcm_learned_step_layout, compact = 2: row
// cur of the adjacency, soft, cur, agg2, mx), one per step at a fixed stride, and the caches are what
// gcm_learned_bptt_cached reads.
// ---------------------------------------------------------------------------------------------
struct RollRec {          // the T step records: record t at rec0 + t * stride (floats)
  float* rec0;
  size_t stride, o_row, o_mx, o_agg2, o_idx, o_soft;
};

// Launch 1 of 3: the logits of every candidate row of every (graph, step).  The unit of work is a 32-ROW BLOCK of
// one item that holds a candidate row (32 k < cur = t), one block per WAVE of a persistent 16-wave workgroup per CU
// (weights staged once; no workgroup barrier in the loop; blocks without candidates are never touched - a 64-step
// rollout has 1.5 live blocks per item, not 4).  Per block what k_learned_select does for its rows, in its order:
// c0 = b0 + W0a x_cur (ascending f), P0 = X W0b^T + c0, ReLU + LayerNorm, P1, ReLU + LayerNorm, the F -> 1 layer.
// The logits land in the `soft` section of the step's record; k_learned_roll_pick turns them into probabilities.
constexpr int RL_WAVES = 16;
constexpr int RL_WSZ = 33 * FS + 32 * FS + 32 + 3;   // floats per wave: X (row 32 = x_cur; later P1) | P0 | c0
constexpr size_t lds_roll_logits() { return sizeof(float) * (3 * FP * FS + 7 * FP + RL_WAVES * RL_WSZ); }

__global__ __launch_bounds__(64 * RL_WAVES) void k_learned_roll_logits(const float* __restrict__ obs,
                                                                       const float* __restrict__ mlp, float eps0,
                                                                       float eps1, RollRec R, int B, int T, int N,
                                                                       int F) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const Mlp M = unpack_mlp(mlp, F);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW0b = smem;                  // [o][f] = W0[o][F + f]
  float* sW0a = sW0b + FP * FS;        // [o][f] = W0[o][f]
  float* sW1 = sW0a + FP * FS;
  float* sVec = sW1 + FP * FS;         // b0 | b1 | g0 | be0 | g1 | be1 | w2
  float* sX_ = sVec + 7 * FP + wave * RL_WSZ;
  if (tid < 256) {
    gcm_fused::Stage<FP, FP, false, false> st_a, st_b, st_1;
    st_a.load(M.w0, F, F, 2 * F, tid);
    st_b.load(M.w0 + F, F, F, 2 * F, tid);
    st_1.load(M.w1, F, F, F, tid);
    st_a.store(sW0a, FS, tid);
    st_b.store(sW0b, FS, tid);
    st_1.store(sW1, FS, tid);
    if (tid < FP) {
      const int o = tid < F ? tid : F - 1;
      const bool ok = tid < F;
      sVec[tid] = ok ? M.b0[o] : 0.f;
      sVec[FP + tid] = ok ? M.b1[o] : 0.f;
      sVec[2 * FP + tid] = ok ? M.g0[o] : 0.f;
      sVec[3 * FP + tid] = ok ? M.be0[o] : 0.f;
      sVec[4 * FP + tid] = ok ? M.g1[o] : 0.f;
      sVec[5 * FP + tid] = ok ? M.be1[o] : 0.f;
      sVec[6 * FP + tid] = ok ? M.w2[o] : 0.f;
    }
  }
  for (int e = lane; e < RL_WSZ; e += 64) sX_[e] = 0.f;
  const float b2 = M.b2[0];
  __syncthreads();
  const long items = (long)T * B;
  const long units = items * ((N + 31) / 32);   // block-major
#pragma unroll 1
  for (long u = (long)blockIdx.x * RL_WAVES + wave; u < units; u += (long)gridDim.x * RL_WAVES) {
    const int k = (int)(u / items);
    const long item = u - (long)k * items;
    const int t = (int)(item / B), b = (int)(item - (long)t * B);
    const int cur = t;                                   // empty graphs at the start: node j is observation j
    const int j0 = 32 * k;
    if (j0 >= cur) continue;                             // no candidate row in this block (wave-uniform)
    int zv = 0;
    asm volatile("" : "+v"(zv));
    float* const sX = sX_ + zv;
    float* const sA = sX + 33 * FS;
    float* const sC0 = sA + 32 * FS;
    {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int e = lane + 64 * i, r = e >> 5, c = e & 31;
        const int j = j0 + r <= cur ? j0 + r : cur;
        v[i] = obs[((size_t)j * B + b) * F + (c < F ? c : F - 1)];
      }
      const float xc = obs[((size_t)t * B + b) * F + (li < F ? li : F - 1)];
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int e = lane + 64 * i, r = e >> 5, c = e & 31;
        sX[r * FS + c] = (j0 + r <= cur && c < F) ? v[i] : 0.f;
      }
      if (lh == 0) sX[32 * FS + li] = li < F ? xc : 0.f;
    }
    wsync();
    if (lh == 0) {   // c0[o] = b0[o] + W0a[o, :] . x_cur, ascending f (c0_dot's order)
      float c0 = sVec[li];
      const float* w = sW0a + li * FS;
      const float* x = sX + 32 * FS;
#pragma unroll
      for (int f = 0; f < FP; ++f)
        if (f < F) c0 = fmaf(w[f], x[f], c0);
      sC0[li] = li < F ? c0 : 0.f;
    }
    wsync();
    {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      gcm_fused::mma32b<32>(acc, sX, FS, 1, sW0b, 1, FS, li, lh);
#pragma unroll
      for (int r = 0; r < 16; ++r) sA[gcm_fused::acc_row(r, lh) * FS + li] = acc[r] + sC0[li];
    }
    wsync();
    if (F == FP) relu_ln_rows_t<true, false, true>(sA, lane, F, sVec + 2 * FP, sVec + 3 * FP, eps0, nullptr, nullptr, true);
    else relu_ln_rows_t<false, false, true>(sA, lane, F, sVec + 2 * FP, sVec + 3 * FP, eps0, nullptr, nullptr, true);
    wsync();
    {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      gcm_fused::mma32b<32>(acc, sA, FS, 1, sW1, 1, FS, li, lh);
#pragma unroll
      for (int r = 0; r < 16; ++r) sX[gcm_fused::acc_row(r, lh) * FS + li] = acc[r] + sVec[FP + li];
    }
    wsync();
    if (F == FP) relu_ln_rows_t<true, false, true>(sX, lane, F, sVec + 4 * FP, sVec + 5 * FP, eps1, nullptr, nullptr, true);
    else relu_ln_rows_t<false, false, true>(sX, lane, F, sVec + 4 * FP, sVec + 5 * FP, eps1, nullptr, nullptr, true);
    wsync();
    {
      const float lg = logit_row2(sX + (lane >> 1) * FS, sVec + 6 * FP, lane & 1, b2);
      const int j = j0 + (lane >> 1);
      if ((lane & 1) == 0 && j < cur) (R.rec0 + (size_t)t * R.stride + R.o_soft)[(size_t)b * N + j] = lg;
    }
    wsync();   // the images are rewritten by the wave's next block
  }
}

// Launch 2 of 3: one wave per (graph, step) - gumbel-softmax over the candidates' logits, threshold, row cur of the
// adjacency and of the node matrix, and layer 1 of the GNN on row cur (h1 / agg1 / x into the chain's caches).
__global__ __launch_bounds__(256) void k_learned_roll_pick(
    const float* __restrict__ obs, const float* __restrict__ noise, int noise_is_exp, float cutoff,
    const float* __restrict__ gnn, int act1, int has_bias, int H1, float* __restrict__ nodes,
    float* __restrict__ adj, int64_t* __restrict__ count, RollRec R, float* __restrict__ cH, float* __restrict__ cA,
    float* __restrict__ cX, int B, int T, int N, int F) {
  __shared__ float sWg[2 * FP * FS];   // W_rel1 | W_root1, row o at stride FS
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  {
    gcm_fused::Stage<FP, FP, false, false> st_g0, st_g1;
    st_g0.load(gnn, H1, F, F, tid);
    st_g1.load(gnn + (size_t)H1 * F, H1, F, F, tid);
    st_g0.store(sWg, FS, tid);
    st_g1.store(sWg + FP * FS, FS, tid);
  }
  __syncthreads();
  const long item = (long)blockIdx.x * 4 + wave;
  if (item >= (long)T * B) return;
  const int t = (int)(item / B), b = (int)(item - (long)t * B);
  const int cur = t;
  float* rec = R.rec0 + (size_t)t * R.stride;
  float* soft = rec + R.o_soft;
  float* row_out = rec + R.o_row;
  float pf_noise[2], pf_lg[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int j = lane + 64 * c < N ? lane + 64 * c : N - 1;
    pf_noise[c] = noise[((size_t)t * B + b) * N + j];
    pf_lg[c] = soft[(size_t)b * N + j];   // (written by k_learned_roll_logits for j < cur; masked below)
  }
  const float pf_b1 = gnn[2 * (size_t)H1 * F + (lane < H1 ? lane : H1 - 1)];
  const int fo = lane < F ? lane : F - 1;
  const float xc_raw = obs[((size_t)t * B + b) * F + fo];
  asm volatile("" ::: "memory");
  // the state: the inserted node, the count behind the last step
  if (lane < F) nodes[((size_t)b * N + cur) * F + lane] = xc_raw;
  if (lane == 0) {
    reinterpret_cast<int64_t*>(rec + R.o_idx)[b] = cur;
    if (t == T - 1) count[b] = T;
  }
  // gumbel-softmax over j < cur (learned.py:88-95), N <= 128: two entries per lane
  float z[2], m = -INFINITY;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int j = lane + 64 * c;
    float nz = 0.f;
    if (j < cur) {
      const float tt = pf_noise[c];
      nz = noise_is_exp ? -logf(tt) : tt;
    }
    z[c] = j < cur ? pf_lg[c] + nz : -INFINITY;
    m = fmaxf(m, z[c]);
  }
  m = wave_max(m);
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    z[c] = (lane + 64 * c < cur) ? expf(z[c] - m) : 0.f;
    s += z[c];
  }
  s = wave_sum(s);
  const float inv = s > 0.f ? 1.f / s : 0.f;
  float* row = adj + ((size_t)b * N + cur) * N;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int j = lane + 64 * c;
    if (j < N) {
      const float p = z[c] * inv;
      soft[(size_t)b * N + j] = p;
      float nv = 0.f;
      if (j < cur) nv = (p - cutoff > 0.f) ? 1.f : 0.f;     // STE forward (util.py:12); the incoming row is empty
      if (j < cur) row[j] = nv;                              // (the rest of the row stays zero)
      row_out[(size_t)b * N + j] = nv;
      z[c] = nv;
    } else {
      z[c] = 0.f;
    }
  }
  // ---- layer 1 of the GNN on row cur: the selected rows S = { j < cur : row[j] = 1 }, ascending -------------
  unsigned long long m0 = __ballot(z[0] != 0.f), m1 = __ballot(z[1] != 0.f);
  const int fl_ = lane < FP ? lane : FP - 1;
  float agg1 = 0.f;
  while (m0 | m1) {
    const int j = m0 ? __builtin_ctzll(m0) : 64 + __builtin_ctzll(m1);
    if (m0) m0 &= m0 - 1; else m1 &= m1 - 1;
    agg1 += obs[((size_t)j * B + b) * F + fo];
  }
  const float xc = lane < F ? xc_raw : 0.f;
  agg1 = lane < F ? agg1 : 0.f;
  const float* wr1 = sWg + fl_ * FS;
  const float* wt1 = sWg + FP * FS + fl_ * FS;
  float p1 = (has_bias & 1) && lane < H1 ? pf_b1 : 0.f;
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(agg1), f));
    const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xc), f));
    p1 = fmaf(wr1[f], a, p1);
    p1 = fmaf(wt1[f], x, p1);
  }
  float h1c = gcm_act(p1, act1);
  h1c = lane < H1 ? h1c : 0.f;
  const size_t rc = (size_t)b * N + cur;
  if (lane < H1) cH[rc * H1 + lane] = h1c;
  if (lane < F) {
    cA[rc * F + lane] = agg1;
    cX[rc * F + lane] = xc;
  }
}

// layer 2 on row cur of every (graph, step): one wave each, four per workgroup
__global__ __launch_bounds__(256) void k_learned_roll_l2(const float* __restrict__ gnn, int act2, int has_bias,
                                                         int H1, int H2, RollRec R, const float* __restrict__ cH,
                                                         float* __restrict__ mx_all, uint32_t* __restrict__ flags,
                                                         int B, int T, int N, int F) {
  __shared__ float sW[2 * FP * FS];   // W_rel2 | W_root2, row o at stride FS
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  {
    gcm_fused::Stage<FP, FP, false, false> a, c;
    a.load(gnn + 2 * (size_t)H1 * F + H1, H2, H1, H1, tid);
    c.load(gnn + 2 * (size_t)H1 * F + H1 + (size_t)H2 * H1, H2, H1, H1, tid);
    a.store(sW, FS, tid);
    c.store(sW + FP * FS, FS, tid);
  }
  __syncthreads();
  const long item = (long)blockIdx.x * 4 + wave;
  if (item >= (long)T * B) return;
  const int t = (int)(item / B), b = (int)(item - (long)t * B);
  float* rec = R.rec0 + (size_t)t * R.stride;
  const float* row = rec + R.o_row + (size_t)b * N;
  const float z0 = row[lane < N ? lane : N - 1], z1 = row[lane + 64 < N ? lane + 64 : N - 1];
  const float pf_b2 = gnn[2 * (size_t)H1 * F + H1 + 2 * (size_t)H2 * H1 + (lane < H2 ? lane : H2 - 1)];
  unsigned long long m0 = __ballot(lane < N && z0 != 0.f), m1 = __ballot(lane + 64 < N && z1 != 0.f);
  const int hl = lane < H1 ? lane : H1 - 1;
  float agg2 = 0.f;
  while (m0 | m1) {
    const int j = m0 ? __builtin_ctzll(m0) : 64 + __builtin_ctzll(m1);
    if (m0) m0 &= m0 - 1; else m1 &= m1 - 1;
    agg2 += cH[((size_t)b * N + j) * H1 + hl];
  }
  float h1c = cH[((size_t)b * N + t) * H1 + hl];
  agg2 = lane < H1 ? agg2 : 0.f;
  h1c = lane < H1 ? h1c : 0.f;
  const int fl_ = lane < FP ? lane : FP - 1;
  const float* wr2 = sW + fl_ * FS;
  const float* wt2 = sW + FP * FS + fl_ * FS;
  float p2 = (has_bias & 2) && lane < H2 ? pf_b2 : 0.f;
#pragma unroll
  for (int h = 0; h < FP; ++h) {
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(agg2), h));
    const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h1c), h));
    p2 = fmaf(wr2[h], a, p2);
    p2 = fmaf(wt2[h], x, p2);
  }
  const float v = gcm_act(p2, act2);
  if (lane < H1) rec[R.o_agg2 + (size_t)b * H1 + lane] = agg2;
  if (lane < H2) {
    rec[R.o_mx + (size_t)b * H2 + lane] = v;
    mx_all[((size_t)t * B + b) * H2 + lane] = v;
  }
  const bool bad = __any(lane < H2 && !isfinite(v));
  if (bad && lane == 0) atomicOr(flags, GCM_FLAG_NONFINITE);
}

// ---------------------------------------------------------------------------------------------
// backward of one step (see the header comment)
// ---------------------------------------------------------------------------------------------
// slab per graph (floats): GNN part as the packed GNN vector (dW_rel1 | dW_root1 | db1 | dW_rel2 |
// dW_root2 | db2), then the edge network as its packed vector.
__global__ __launch_bounds__(256) void k_learned_step_bwd(
    const float* __restrict__ g_mx, const float* __restrict__ nodes, const float* __restrict__ adj,
    const int64_t* __restrict__ cur_idx, const int64_t* __restrict__ count_in,
    const float* __restrict__ gnn, int act1, int act2, const float* __restrict__ mx,
    const float* __restrict__ h1, const float* __restrict__ agg1, const float* __restrict__ agg2,
    const float* __restrict__ soft, const float* __restrict__ mlp, float eps0, float eps1,
    float* __restrict__ GA, float* __restrict__ slabs, int accumulate, int N, int F,
    int H1, int H2) {
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  int64_t c64 = cur_idx[b];
  const int cur = c64 < 0 ? 0 : (c64 > N - 1 ? N - 1 : (int)c64);
  const bool wrapped = count_in[b] + 1 > N;
  const Mlp M = unpack_mlp(mlp, F);
  const int Pg = 2 * H1 * F + H1 + 2 * H2 * H1 + H2, Pm = 3 * F * F + 7 * F + 1;
  const float* w_rel1 = gnn;
  const float* w_rel2 = gnn + 2 * H1 * F + H1;
  const float* w_root2 = w_rel2 + H2 * H1;
  const float* xg = nodes + (size_t)b * N * F;
  const float* ag = adj + (size_t)b * N * N;
  const float* h1g = h1 + (size_t)b * N * H1;
  const float* a1g = agg1 + (size_t)b * N * F;
  float* GAg = GA + (size_t)b * N * N;   // this graph's chain buffer (see the header comment)
  float* slab_g = slabs + (size_t)b * (Pg + Pm);

  extern __shared__ float smem[];
  float* sX = smem;                    // [NP][FS]
  float* sP0 = sX + NP * FS;           // P0 -> gP0
  float* sH0 = sP0 + NP * FS;          // H0
  float* sP1 = sH0 + NP * FS;          // P1 -> gP1
  float* sG = sP1 + NP * FS;           // h1 image (phase A), then gH0
  float* sW0b = sG + NP * FS;          // [o][f]
  float* sW1 = sW0b + FP * FS;         // [o][f]
  float* sWr1 = sW1 + FP * FS;         // w_rel1 [h][f]
  float* sR = sWr1 + FP * FS;          // [4][1024] cross-wave reduction of the dW tiles
  float* sVec = sR + 4096;             // c0 | b1 | g0 | be0 | g1 | be1 | w2
  float* sMu0 = sVec + 7 * FP;         // per-row LayerNorm statistics [NP] x 4
  float* sRs0 = sMu0 + NP;
  float* sMu1 = sRs0 + NP;
  float* sRs1 = sMu1 + NP;
  float* sGl = sRs1 + NP;              // g_logit [NP]
  float* sSel = sGl + NP;              // g_sel [NP]
  float* sCoef = sSel + NP;            // adj[cur, :] [NP]
  float* sD2 = sCoef + NP;             // d2 [32] | u [64] | v [64] | (32 spare) | colsum scratch [256]
  float* sU = sD2 + 32;
  float* sVv = sU + 64;
  float* sCs = sVv + 64 + 32;
  int* sLive = reinterpret_cast<int*>(sCs + 256);   // [NP] + count
  // The parameter-gradient slab of this graph is built in LDS and written once: every update below is
  // a read-modify-write, and against HBM each of the dozen update sites exposed a memory round trip.
  float* slab = reinterpret_cast<float*>(sLive + NP + 8);   // [Pg + Pm]
  float* sl_m = slab + Pg;

  LSTAMP(0);
  // ---- loads: node matrix, h1, weights, row cur of the adjacency, the kept row's vectors, the slab so
  // far - every global load in flight before the first LDS store (one round trip, not one per matrix) ---
  {
    using gcm_fused::Stage;
    Stage<NP, FP, false, false> st_x, st_h;
    Stage<FP, FP, false, false> st_w0, st_w1, st_wr;
    st_x.load(xg, N, F, F, tid);
    st_h.load(h1g, N, H1, H1, tid);
    st_w0.load(M.w0 + F, F, F, 2 * F, tid);
    st_w1.load(M.w1, F, F, F, tid);
    st_wr.load(w_rel1, H1, F, F, tid);
    constexpr int SLAB_PER = (2 * FP * FP + FP + 2 * FP * FP + FP + 3 * FP * FP + 7 * FP + 1 + 255) / 256;
    float sv[SLAB_PER];
    if (accumulate) {   // (uniform)
#pragma unroll
      for (int i = 0; i < SLAB_PER; ++i) {
        const int e = tid + 256 * i;
        sv[i] = slab_g[e < Pg + Pm ? e : Pg + Pm - 1];
      }
    }
    asm volatile("" ::: "memory");
    st_x.store(sX, FS, tid);
    st_h.store(sG, FS, tid);
    st_w0.store(sW0b, FS, tid);
    st_w1.store(sW1, FS, tid);
    st_wr.store(sWr1, FS, tid);
#pragma unroll
    for (int i = 0; i < SLAB_PER; ++i) {
      const int e = tid + 256 * i;
      if (e < Pg + Pm) slab[e] = accumulate ? sv[i] : 0.f;
    }
  }
  if (tid < NP) sCoef[tid] = tid < N ? ag[cur * N + tid] : 0.f;
  if (tid < FP) {
    const int o = tid < F ? tid : F - 1;
    const float c0 = c0_dot(M.w0 + (size_t)o * 2 * F, xg + (size_t)cur * F, M.b0[o], F);
    const bool ok = tid < F;
    sVec[tid] = ok ? c0 : 0.f;
    sVec[FP + tid] = ok ? M.b1[o] : 0.f;
    sVec[2 * FP + tid] = ok ? M.g0[o] : 0.f;
    sVec[3 * FP + tid] = ok ? M.be0[o] : 0.f;
    sVec[4 * FP + tid] = ok ? M.g1[o] : 0.f;
    sVec[5 * FP + tid] = ok ? M.be1[o] : 0.f;
    sVec[6 * FP + tid] = ok ? M.w2[o] : 0.f;
    const int oc = tid < H2 ? tid : H2 - 1;
    const float gm = g_mx[(size_t)b * H2 + oc], y = mx[(size_t)b * H2 + oc];
    sD2[tid] = tid < H2 ? gm * gcm_act_grad(y, act2) : 0.f;
  }
  if (tid < 64) {   // v = agg2 | h1[cur]
    const int k = tid < 32 ? tid : tid - 32;
    const int kc = k < H1 ? k : H1 - 1;
    const float t = tid < 32 ? agg2[(size_t)b * H1 + kc] : h1g[cur * H1 + kc];
    sVv[tid] = k < H1 ? t : 0.f;
  }
  LSTAMP(1);
  __syncthreads();
  // ---- phase A: layer-2 adjoint ----------------------------------------------------------------------
  if (tid < 64) {   // u[m] = sum_o W2c[o][m] d2[o]   (m < 32: dagg2, else dh1cur)
    const int k = tid < 32 ? tid : tid - 32;
    const float* wc = (tid < 32 ? w_rel2 : w_root2) + (k < H1 ? k : H1 - 1);
    float s = 0.f;
    for (int o = 0; o < H2; ++o) s = fmaf(wc[o * H1], sD2[o], s);
    sU[tid] = k < H1 ? s : 0.f;
  }
  {   // dW2c[o][k] (+)= d2[o] v[k]; db2
    float* sl_rel2 = slab + 2 * H1 * F + H1;
    float* sl_root2 = sl_rel2 + H2 * H1;
    float* sl_b2 = sl_root2 + H2 * H1;
    for (int e = tid; e < 32 * 64; e += 256) {
      const int o = e >> 6, k = e & 63, kk = k & 31;
      if (o < H2 && kk < H1) {
        float* d = (k < 32 ? sl_rel2 : sl_root2) + o * H1 + kk;
        *d = (accumulate ? *d : 0.f) + sD2[o] * sVv[k];
      }
    }
    if (tid < H2) sl_b2[tid] = (accumulate ? sl_b2[tid] : 0.f) + sD2[tid];
  }
  LSTAMP(2);
  // live rows: ballot over row cur (entries of the sampled row), list in LDS
  {
    const bool pred = tid < N && tid < NP && (sCoef[tid & (NP - 1)] != 0.f || tid == cur);
    const unsigned long long bal = __ballot(pred);
    if (lane == 0 && wave < 2) sLive[NP + wave] = __popcll(bal);
    __syncthreads();
    const int pos = (wave ? sLive[NP] : 0) + __popcll(bal & ((1ull << lane) - 1ull));
    if (pred) sLive[pos] = tid;
  }
  __syncthreads();
  const int L = sLive[NP] + sLive[NP + 1];
  // G1[l][h] = (coef dagg2[h] + [j == cur] dh1cur[h]) act1'(h1[j][h]) -> sR[l][h]  (L <= 128 rows x 32)
  for (int e = tid; e < L * 32; e += 256) {
    const int l = e >> 5, h = e & 31, j = sLive[l];
    const float y = sG[j * FS + h];
    const float d = sCoef[j] * sU[h] + (j == cur ? sU[32 + h] : 0.f);
    sR[l * 32 + h] = h < H1 ? d * gcm_act_grad(y, act1) : 0.f;
  }
  __syncthreads();
  LSTAMP(3);
  {   // layer-1 parameter gradients on the live rows: dW1c[h][m] (+)= sum_l G1[l][h] [agg1 | x][j_l][m]
    const int h = tid >> 3, m0 = (tid & 7) * 8;
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    float bsum = 0.f;
    for (int l = 0; l < L; ++l) {
      const int j = sLive[l];
      const float g = sR[l * 32 + h];
      bsum += g;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int m = m0 + q, f = m & 31;
        const float a = f < F ? (m < 32 ? a1g[j * F + f] : sX[j * FS + f]) : 0.f;
        acc[q] = fmaf(g, a, acc[q]);
      }
    }
    if (h < H1) {
      float* sl_rel1 = slab;
      float* sl_root1 = slab + H1 * F;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int m = m0 + q, f = m & 31;
        if (f < F) {
          float* d = (m < 32 ? sl_rel1 : sl_root1) + h * F + f;
          *d = (accumulate ? *d : 0.f) + acc[q];
        }
      }
      if ((tid & 7) == 0) {
        float* d = slab + 2 * H1 * F + h;
        *d = (accumulate ? *d : 0.f) + bsum;
      }
    }
  }
  LSTAMP(4);
  // dAgg1[l][f] = G1[l] . w_rel1[:, f] -> sP0 [l][FS] (free until the edge network is recomputed)
  for (int e = tid; e < L * 32; e += 256) {
    const int l = e >> 5, f = e & 31;
    float s = 0.f;
    if (f < F) {
#pragma unroll
      for (int h = 0; h < 32; ++h) s = fmaf(sR[l * 32 + h], sWr1[h * FS + f], s);
    }
    sP0[l * FS + f] = s;
  }
  __syncthreads();
  // the adjacency gradient of the rows layer 1 aggregated into: GA[j_l][k] += dAgg1[l] . x[k] for every
  // column k (entry (j_l, k) exists from the step that wrote row j_l until node k is dropped: the
  // chain buffer is rolled with the state below, so later contributions land on the right entries).
  // Row cur is consumed right here (g_sel) and is not written back.
  int l_cur = -1;
  for (int l = 0; l < L; ++l) l_cur = sLive[l] == cur ? l : l_cur;   // (row cur is always in the list)
  for (int e = tid; e < L * N; e += 256) {
    const int l = e / N, k = e - l * N, j = sLive[l];
    if (j != cur) {
      float s = 0.f;
#pragma unroll
      for (int f = 0; f < 32; ++f) s = fmaf(sP0[l * FS + f], sX[k * FS + f], s);
      GAg[j * N + k] += s;
    }
  }
  LSTAMP(5);
  // ---- g_sel[j] = dagg2 . h1[j] + GA[cur][j] + dAgg1[cur] . x[j]  (j < cur); selection adjoint ------
  if (tid < NP) {
    float s = 0.f;
    if (tid < cur) {
      s = GAg[cur * N + tid];
#pragma unroll
      for (int h = 0; h < 32; ++h) s = fmaf(sU[h], sG[tid * FS + h], s);
#pragma unroll
      for (int f = 0; f < 32; ++f) s = fmaf(sP0[l_cur * FS + f], sX[tid * FS + f], s);
    }
    sSel[tid] = s;
  }
  __syncthreads();
  if (wave == 0) {
    float p[2], g[2], dot = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int j = lane + 64 * c;
      const bool live = j < cur;
      p[c] = live ? soft[(size_t)b * N + j] : 0.f;
      g[c] = live ? sSel[j] : 0.f;
      dot = fmaf(p[c], g[c], dot);
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int c = 0; c < 2; ++c) sGl[lane + 64 * c] = p[c] * (g[c] - dot);
  }
  LSTAMP(6);
  // ---- the chain buffer for the previous step: undo the state advance (gcm.py:262-278, 323-355).
  // No overflow: row cur did not exist before (zero).  Overflow: every entry moves back by one row
  // and one column, the dropped node's row / column and the new node's row carry nothing.
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // every GA read / write above is done
    if (wrapped) {
      constexpr int PERG = NP * NP / 256;
      float gv[PERG];
#pragma unroll
      for (int i = 0; i < PERG; ++i) {
        const int e = tid + 256 * i, r = e / NP, c = e % NP;      // destination entry (r, c) <- (r-1, c-1)
        const int rs = r - 1, cs = c - 1;
        const bool ok = r < N && c < N && rs >= 0 && cs >= 0 && rs != cur;
        const float t = GAg[(rs >= 0 && rs < N ? rs : 0) * N + (cs >= 0 && cs < N ? cs : 0)];
        gv[i] = ok ? t : 0.f;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int i = 0; i < PERG; ++i) {
        const int e = tid + 256 * i, r = e / NP, c = e % NP;
        if (r < N && c < N) GAg[r * N + c] = gv[i];
      }
    } else if (tid < N) {
      GAg[cur * N + tid] = 0.f;
    }
  }
  __syncthreads();
  LSTAMP(7);
  // ---- edge network: forward recomputed, then its adjoint -------------------------------------------
  {
    const f32x16 acc = gemm_rows(sX, sW0b, wave, li, lh);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * wave + gcm_fused::acc_row(r, lh);
      const float v = acc[r] + sVec[li];
      sP0[row * FS + li] = v;
      sH0[row * FS + li] = v;
    }
  }
  __syncthreads();
  relu_ln_rows(sH0, tid, F, sVec + 2 * FP, sVec + 3 * FP, eps0, sMu0, sRs0);
  __syncthreads();
  {
    const f32x16 acc = gemm_rows(sH0, sW1, wave, li, lh);
#pragma unroll
    for (int r = 0; r < 16; ++r) sP1[(32 * wave + gcm_fused::acc_row(r, lh)) * FS + li] = acc[r] + sVec[FP + li];
  }
  __syncthreads();
  // statistics of layer 1 (the normalised values are rebuilt where needed)
  relu_ln_rows(sP1, tid, F, nullptr, nullptr, eps1, sMu1, sRs1, /*write=*/false);
  __syncthreads();
  LSTAMP(8);
  // column sums over the rows (8 row groups x 32 columns): dw2, dgamma1, dbeta1, db2
  {
    const int f = tid & 31, grp = tid >> 5;
    float s_w2 = 0.f, s_g1 = 0.f, s_b1 = 0.f, s_bb = 0.f;
    const float w2f = sVec[6 * FP + f], g1f = sVec[4 * FP + f], be1f = sVec[5 * FP + f];
    for (int j = grp; j < N; j += 8) {
      const float gl = sGl[j];
      const float v = sP1[j * FS + f];
      const float xh = ((v > 0.f ? v : 0.f) - sMu1[j]) * sRs1[j];
      s_w2 = fmaf(gl, fmaf(xh, g1f, be1f), s_w2);   // dw2[f] += gl * H1m[j][f]
      s_g1 = fmaf(gl * w2f, xh, s_g1);
      s_b1 = fmaf(gl, w2f, s_b1);
      s_bb += gl;
    }
    auto colsum = [&](float v) {
      sCs[tid] = v;
      __syncthreads();
      float t = 0.f;
      if (tid < 32) {
#pragma unroll
        for (int q = 0; q < 8; ++q) t += sCs[q * 32 + tid];
      }
      __syncthreads();
      return t;
    };
    const int o_w1 = 2 * F * F + 3 * F, o_b1 = o_w1 + F * F, o_g1 = o_b1 + F, o_be1 = o_g1 + F;
    const int o_w2 = o_be1 + F, o_b2 = o_w2 + F;
    const float t_w2 = colsum(s_w2), t_g1 = colsum(s_g1), t_b1 = colsum(s_b1), t_bb = colsum(s_bb);
    if (tid < F) {
      sl_m[o_w2 + tid] = (accumulate ? sl_m[o_w2 + tid] : 0.f) + t_w2;
      sl_m[o_g1 + tid] = (accumulate ? sl_m[o_g1 + tid] : 0.f) + t_g1;
      sl_m[o_be1 + tid] = (accumulate ? sl_m[o_be1 + tid] : 0.f) + t_b1;
    }
    if (tid == 0) sl_m[o_b2] = (accumulate ? sl_m[o_b2] : 0.f) + t_bb;
  }
  LSTAMP(9);
  // LayerNorm-1 + ReLU adjoint, row by row: gP1 in place of P1
  relu_ln_rows_bwd(sP1, tid, F, sMu1, sRs1,
                   [&](int j, int f) { return sGl[j] * sVec[6 * FP + f] * sVec[4 * FP + f]; });
  __syncthreads();
  LSTAMP(10);
  // db1' = column sums of gP1; dW1 = gP1^T H0 (K = rows, split over the waves); gH0 = gP1 W1
  const int o_b0 = 2 * F * F, o_g0 = o_b0 + F, o_be0 = o_g0 + F, o_w1 = o_be0 + F, o_b1 = o_w1 + F * F;
  {
    const int f = tid & 31, grp = tid >> 5;
    float s = 0.f;
    for (int j = grp; j < N; j += 8) s += sP1[j * FS + f];
    sCs[tid] = s;
  }
  {
    f32x16 a;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
    // A(i = o, k = row) = gP1[row][o];  B(k = row, j = f) = H0[row][f]
    mma32(a, sP1 + 32 * wave * FS, 1, FS, sH0 + 32 * wave * FS, FS, 1, 32, li, lh);
#pragma unroll
    for (int r = 0; r < 16; ++r) sR[wave * 1024 + gcm_fused::acc_row(r, lh) * 32 + li] = a[r];
  }
  __syncthreads();
  if (tid < F) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += sCs[q * 32 + tid];
    sl_m[o_b1 + tid] = (accumulate ? sl_m[o_b1 + tid] : 0.f) + t;
  }
  for (int e = tid; e < 1024; e += 256) {
    const int o = e >> 5, f = e & 31;
    if (o < F && f < F) {
      float* d = sl_m + o_w1 + o * F + f;
      *d = (accumulate ? *d : 0.f) + ((sR[e] + sR[1024 + e]) + (sR[2048 + e] + sR[3072 + e]));
    }
  }
  {   // gH0[rows] = gP1[rows] @ W1  (B(k = o, j = f) = W1[o][f]) -> sG
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    mma32(acc, sP1 + 32 * wave * FS, FS, 1, sW1, FS, 1, 32, li, lh);
    __syncthreads();   // sR / sCs reads above are done before they are reused; sG (h1) is free
#pragma unroll
    for (int r = 0; r < 16; ++r) sG[(32 * wave + gcm_fused::acc_row(r, lh)) * FS + li] = acc[r];
  }
  __syncthreads();
  LSTAMP(11);
  // dgamma0 / dbeta0 (column sums of gH0 * xhat0, gH0), then LayerNorm-0 + ReLU adjoint: gP0 in place
  {
    const int f = tid & 31, grp = tid >> 5;
    float s_g = 0.f, s_b = 0.f;
    for (int j = grp; j < N; j += 8) {
      const float v = sP0[j * FS + f];
      const float xh = ((v > 0.f ? v : 0.f) - sMu0[j]) * sRs0[j];
      const float gh = sG[j * FS + f];
      s_g = fmaf(gh, xh, s_g);
      s_b += gh;
    }
    sCs[tid] = s_g;
    __syncthreads();
    if (tid < F) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += sCs[q * 32 + tid];
      sl_m[o_g0 + tid] = (accumulate ? sl_m[o_g0 + tid] : 0.f) + t;
    }
    __syncthreads();
    sCs[tid] = s_b;
    __syncthreads();
    if (tid < F) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) t += sCs[q * 32 + tid];
      sl_m[o_be0 + tid] = (accumulate ? sl_m[o_be0 + tid] : 0.f) + t;
    }
  }
  relu_ln_rows_bwd(sP0, tid, F, sMu0, sRs0, [&](int j, int f) { return sG[j * FS + f] * sVec[2 * FP + f]; });
  __syncthreads();
  LSTAMP(12);
  // db0 = column sums of gP0; dW0a = db0 (x) x[cur]; dW0b = gP0^T X
  {
    const int f = tid & 31, grp = tid >> 5;
    float s = 0.f;
    for (int j = grp; j < N; j += 8) s += sP0[j * FS + f];
    sCs[tid] = s;
  }
  {
    f32x16 a;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
    mma32(a, sP0 + 32 * wave * FS, 1, FS, sX + 32 * wave * FS, FS, 1, 32, li, lh);
#pragma unroll
    for (int r = 0; r < 16; ++r) sR[wave * 1024 + gcm_fused::acc_row(r, lh) * 32 + li] = a[r];
  }
  __syncthreads();
  if (tid < FP) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += sCs[q * 32 + tid];
    sVec[tid] = t;   // db0 (c0 is no longer needed)
    if (tid < F) sl_m[o_b0 + tid] = (accumulate ? sl_m[o_b0 + tid] : 0.f) + t;
  }
  __syncthreads();
  for (int e = tid; e < 1024; e += 256) {
    const int o = e >> 5, f = e & 31;
    if (o < F && f < F) {
      float* da = sl_m + o * 2 * F + f;          // W0a half
      float* db = sl_m + o * 2 * F + F + f;      // W0b half
      *da = (accumulate ? *da : 0.f) + sVec[o] * sX[cur * FS + f];
      *db = (accumulate ? *db : 0.f) + ((sR[e] + sR[1024 + e]) + (sR[2048 + e] + sR[3072 + e]));
    }
  }
  __syncthreads();
  for (int e = tid; e < Pg + Pm; e += 256) slab_g[e] = slab[e];
  LSTAMP(13);
}

// ---------------------------------------------------------------------------------------------
// TIME-PARALLEL backward of a whole chain of steps (round 3; k_learned_step_bwd above is one step at a
// time, one workgroup per CU, ~20 barrier phases behind a sequential [B,N,N] chain buffer).
//
// With observations that carry no gradient the only path between steps is the adjacency: the entries
// (j, k) written when node j was inserted are read by layer 1 of every later step in which row j is
// live.  The gradient w.r.t. them is dAgg1_{t'}[j] . x[k] - a rank-one term per (later step, live row) -
// so the whole "chain buffer" of a node collapses to ONE vector
//     D_t = sum over steps t' >= t in which the node inserted at step t is a live row of dAgg1_{t'}[that row]
// and the selection adjoint of step t is   g_sel_t[j] = dagg2_t . h1_t[j] + D_t^(j) . x[j]  (j < cur_t),
// where D_t^(j) stops at the last step that still holds node j (the overflow roll, gcm.py:323-355, drops the
// oldest node and with it the entries that pointed at it): a prefix of the time-ordered sum - row j of the
// running sums over t', picked by the number of rolls since step t.
// dAgg1 / dagg2 depend on g_mx[t'] and step t' alone, so:
//   pass A (rows_bptt.hip, MODE 2): every graph-step independently - GNN parameter gradient, and per item
//          cur, the live rows, dagg2 and dAgg1 per live row, into arrays indexed [path step, b];
//   pass B (here): every graph-step independently - D_t by scanning the <= N later items of its graph (the
//          node inserted at step t sits in row cur_{t'} - (t' - t) of step t'; fixed order, no atomics),
//          selection adjoint, edge network recomputed and differentiated.
// Pass B is persistent: 2 workgroups per CU (69 KB of LDS each: the node image doubles as H0's, h1's as
// P1's), the weight-gradient tiles of the three products stay in each wave's MFMA accumulators and the
// column sums in registers across ALL items of the workgroup; one slab per workgroup at the end.
// ---------------------------------------------------------------------------------------------
struct BpttB {
  gcm_rows::StepTable tab;            // tab.saved[s]: the buffer of step s0 + s (nodes at offset 0)
  size_t o_h1, o_soft;                // float offsets of h1 [B,N,H1] and soft [B,N] inside it
  const float *c_nodes, *c_h1;        // cached steps: the chain's caches instead (NULL: the step's own buffer)
  const float* c_u;                   // ... and, where the chain keeps it, U[j] = W0[:, F:] x_j [B,N,F] (pass B2; NULL: recomputed)
  const int* hdr;                     // pass A: [T, B, 2] cur, L
  const int* live;                    //         [T, B, N]
  const float* da;                    //         [T, B, N, F]
  const float* dagg2;                 //         [T, B, H1]
  int s0, n_steps, T;
  const float* c0;                    // pass B2: c0_t = b0 + W0a x_cur [T, B, F] where pass B1 left it (k_learned_bptt_sel_graph,
                                      // in dagg2's place); NULL: recomputed per tile
};

// ReLU + LayerNorm of the rows of src -> dst (two threads per row), statistics stored
// (sg / sb: 16-byte aligned LDS vectors - this thread's 16 entries come as four 16-byte reads each)
template <bool FULL>
__device__ __forceinline__ void relu_ln_rows_to_t(const float* src, float* dst, int tid, int F, const float* sg,
                                                  const float* sb, float eps, float* mu_out, float* rs_out) {
  const int row = tid >> 1, f0 = (tid & 1) * (FP / 2);
  float a[FP / 2];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < FP / 2; ++k) {
    const int f = f0 + k;
    const float v = src[row * FS + f];
    a[k] = ((FULL || f < F) && v > 0.f) ? v : 0.f;
    s += a[k];
  }
  s += gcm_lane_xor1(s);
  const float mean = s / (float)F;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < FP / 2; ++k) {
    const float d = (FULL || f0 + k < F) ? a[k] - mean : 0.f;
    q = fmaf(d, d, q);
  }
  q += gcm_lane_xor1(q);
  const float rstd = rsqrtf(q / (float)F + eps);
#pragma unroll
  for (int q4 = 0; q4 < FP / 8; ++q4) {   // scale / shift in 16-byte pieces, right where they are used
    const float4 tg = reinterpret_cast<const float4*>(sg + f0)[q4], tb = reinterpret_cast<const float4*>(sb + f0)[q4];
    const float gv[4] = {tg.x, tg.y, tg.z, tg.w}, bv[4] = {tb.x, tb.y, tb.z, tb.w};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = 4 * q4 + kk, f = f0 + k;
      dst[row * FS + f] = (FULL || f < F) ? fmaf((a[k] - mean) * rstd, gv[kk], bv[kk]) : 0.f;
    }
  }
  if ((tid & 1) == 0) { mu_out[row] = mean; rs_out[row] = rstd; }
}
__device__ __forceinline__ void relu_ln_rows_to(const float* src, float* dst, int tid, int F, const float* sg,
                                                const float* sb, float eps, float* mu_out, float* rs_out) {
  if (F == FP) relu_ln_rows_to_t<true>(src, dst, tid, F, sg, sb, eps, mu_out, rs_out);
  else relu_ln_rows_to_t<false>(src, dst, tid, F, sg, sb, eps, mu_out, rs_out);
}

// ---------------------------------------------------------------------------------------------
// Pass B in TWO kernels (round 4).  Round 3's single kernel (one 256-thread workgroup per item, two per CU, all
// 128 rows of the item's images through every phase) was bound by VALU issue, not by latency - in-kernel stamps:
// ~46 k cycles per item, the two workgroups of a CU filling each other's gaps, 3 - 5 k cycles per LayerNorm /
// column-sum / staging phase - and three quarters of that went to rows >= cur (a rollout of T <= N steps from
// empty graphs has cur < 64 on every item).  Measured at cfg5: 712 us -> 238 + 54 us per 64-step chain.
//   B1 (k_learned_bptt_sel): one 128-thread workgroup per item - D_t from the <= min(N, T - t) later items only,
//      selection adjoint, softmax adjoint -> g_logit [T, B, N];
//   B2 (k_learned_bptt_mlp): the edge network recomputed and differentiated per 32-ROW BLOCK, one block per WAVE
//      (no workgroup barrier inside the loop; each wave owns four 32 x 32 images), blocks with no candidate row
//      skipped: the work is proportional to the candidate rows.  One 8-wave workgroup per CU, weight-gradient
//      tiles in the MFMA accumulators across all blocks of a wave, one slab per workgroup.
// ---------------------------------------------------------------------------------------------
// row p[0 .. n) -> out[32], zero padded (16-byte loads when the row is a whole number of them)
__device__ __forceinline__ void load_row32(const float* __restrict__ p, int n, float* out) {
  if ((n & 3) == 0 && (reinterpret_cast<size_t>(p) & 15) == 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 v = 4 * q < n ? reinterpret_cast<const float4*>(p)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int f = 0; f < 32; ++f) {
      const float v = p[f < n ? f : n - 1];
      out[f] = f < n ? v : 0.f;
    }
  }
}

__global__ __launch_bounds__(128) void k_learned_bptt_sel(BpttB a, float* __restrict__ g_logit, int B, int N,
                                                          int F, int H1) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int s = blockIdx.x / B, b = blockIdx.x - s * B, sg = a.s0 + s;
  const size_t it = (size_t)sg * B + b;
  const int cur = __builtin_amdgcn_readfirstlane(a.hdr[2 * it]);
  if (cur <= 0) return;   // no candidate rows (pass B2 skips the item too)
  __shared__ float sP[NP * FS];        // dAgg1 contributions of the later steps, then their running sums
  __shared__ float sD2[32], sCs[4 * 32], sSel[NP];
  __shared__ int sSlot[NP], sWc[NP], sFirst[NP + 2];
  const float* base = a.tab.saved[s];
  const float* xg = (a.c_nodes ? a.c_nodes : base) + (size_t)b * N * F;
  const float* hg = (a.c_h1 ? a.c_h1 : base + a.o_h1) + (size_t)b * N * H1;
  const int n_fut = a.T - sg < NP ? a.T - sg : NP;   // later steps (this one included) that can hold the node
  float xr[32], hr[32];                // this thread's candidate row
  {
    const int j = tid < cur ? tid : cur - 1;
    load_row32(xg + (size_t)j * F, F, xr);
    load_row32(hg + (size_t)j * H1, H1, hr);
  }
  float pf_soft[2];                    // the softmax row, for the adjoint at the end: requested with everything else
  {
    const float* soft = base + a.o_soft + (size_t)b * N;
#pragma unroll
    for (int c = 0; c < 2; ++c) pf_soft[c] = soft[lane + 64 * c < N ? lane + 64 * c : N - 1];
  }
  constexpr int NONE = 1 << 30;
  {   // where does the node inserted at this step sit in the live list of step t + tid?
    int slot = -1, w = NONE;
    if (tid < n_fut) {
      const size_t it2 = it + (size_t)tid * B;
      int e[8];
      __builtin_memcpy(e, a.live + it2 * N, sizeof(e));
      const int cur2 = a.hdr[2 * it2], L2 = a.hdr[2 * it2 + 1];
      asm volatile("" ::: "memory");
      const int r = cur2 - tid;
      w = cur + tid - cur2;
      if (r >= 0) {
#pragma unroll
        for (int l = 0; l < 8; ++l) slot = (l < L2 && e[l] == r) ? l : slot;
        for (int l = 8; l < L2; ++l)
          if (a.live[it2 * N + l] == r) slot = l;
      }
    }
    sSlot[tid] = slot;
    sWc[tid] = w;
    sFirst[tid] = NONE;
    if (tid < 2) sFirst[NP + tid] = NONE;
  }
  if (tid < 32) sD2[tid] = tid < H1 ? a.dagg2[it * H1 + tid] : 0.f;
  __syncthreads();
  {
    const int w = sWc[tid];
    if (w != NONE) {
      if (tid == 0 || sWc[tid - 1] < w) sFirst[w < NP ? w : NP] = tid;
      if (tid == NP - 1 || sWc[tid + 1] == NONE) sFirst[NP + 1] = tid;
    }
  }
  {   // gather: thread (column f, row group q) takes rows q, q + 4, ... of the steps in range, 8 loads in flight
    const int f = tid & 31, q = tid >> 5;
#pragma unroll 1
    for (int r0 = 0; r0 < n_fut; r0 += 32) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = r0 + q + 4 * i, rc = r < n_fut ? r : n_fut - 1;
        const int slot = sSlot[rc];
        v[i] = a.da[((it + (size_t)rc * B) * N + (slot >= 0 ? slot : 0)) * F + (f < F ? f : F - 1)];
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = r0 + q + 4 * i;
        sP[r * FS + f] = (r < n_fut && sSlot[r] >= 0 && f < F) ? v[i] : 0.f;
      }
    }
  }
  __syncthreads();
  {   // running sums over the later steps in time order: 4 segments of 32 steps, then the segments
    const int cf = tid & 31, cg = tid >> 5;
    const bool on = 32 * cg < n_fut;
    float v[32], run = 0.f;
    if (on) {
#pragma unroll
      for (int r = 0; r < 32; ++r) v[r] = sP[(32 * cg + r) * FS + cf];
#pragma unroll
      for (int r = 0; r < 32; ++r) { run += v[r]; v[r] = run; }
    }
    sCs[cg * 32 + cf] = run;
    __syncthreads();
    float off = 0.f;
    for (int q = 0; q < cg; ++q) off += sCs[q * 32 + cf];
    if (on) {
#pragma unroll
      for (int r = 0; r < 32; ++r) sP[(32 * cg + r) * FS + cf] = v[r] + off;
    }
  }
  __syncthreads();
  {   // g_sel[j] = dagg2 . h1[j] + D^(j) . x[j]
    float t = 0.f;
    if (tid < cur) {
      const int fj = sFirst[tid + 1 < NP ? tid + 1 : NP];
      const int istar = fj != NONE ? fj - 1 : sFirst[NP + 1];
      const float* dv = sP + istar * FS;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        t1 = fmaf(sD2[k], hr[k], t1);
        t2 = fmaf(dv[k], xr[k], t2);
      }
      t = t1 + t2;
    }
    sSel[tid] = t;
  }
  __syncthreads();
  if (tid < 64) {   // softmax adjoint (tau = 1); both straight-through estimators are identities
    float p[2], g[2], dot = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int j = lane + 64 * c;
      const bool live = j < cur;
      p[c] = live ? pf_soft[c] : 0.f;
      g[c] = live ? sSel[j] : 0.f;
      dot = fmaf(p[c], g[c], dot);
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int j = lane + 64 * c;
      if (j < N) g_logit[it * N + j] = p[c] * (g[c] - dot);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Pass B1 per GRAPH (round 6): a backward whose steps are all cached steps of one chain (T <= 128, the exact widths).
// k_learned_bptt_sel gives every (step, graph) a workgroup that re-reads the graph's candidate rows of the node and h1
// caches (132 MB through L2 per cfg5 chain for 8 MB of caches) and scans the live lists of all later steps for its node.
// While no graph has rolled, node j sits at row j in every later step, so D_t - the sum of dAgg1 over the later steps
// that aggregate node t - does not depend on the candidate: one vector per node.  One workgroup per graph then
//   * turns the live lists into bit masks (row masks per step, column masks per node - the columns by LDS atomic OR:
//     order-free), and each quarter-row thread walks its node's column mask in step order (the slot of node j in step t'
//     is the number of live rows below j: one popcount) - D in a fixed order, no scan over steps that do not hold it;
//   * keeps the graph's h1 | x rows in REGISTERS, rows j and j + 64 per lane, and evaluates g_sel_t[j] = dagg2_t . h1[j] +
//     D_t . x[j] for all candidates of a step at once (the step's 64-float vector broadcast from LDS), the softmax adjoint
//     behind it - a wave per step, four steps in flight.
// Same g_logit as k_learned_bptt_sel up to the order of two 32-term sums.
// ---------------------------------------------------------------------------------------------
constexpr int SG_TS = FP + 4;
constexpr int SG_TMAX = 128;   // steps of one backward (= GCM_ROWS_MAX_STEPS: one chunk)
constexpr size_t lds_bptt_sel_graph() {
  return sizeof(float) * (2 * NP * SG_TS + SG_TMAX * SG_TS + 2 * FP * SG_TS) + sizeof(unsigned long long) * (SG_TMAX * 2 + NP * 2) +
         sizeof(int) * SG_TMAX * 2;
}

// w_rel1 (may be NULL): a.da holds G1 rows (pass A per graph, rows_bptt.hip: k_bptt_learned_graph) - dAgg1 = W_rel1^T G1, and the
// matrix is applied ONCE per node, to the sum.
__global__ __launch_bounds__(256) void k_learned_bptt_sel_graph(BpttB a, float* __restrict__ g_logit,
                                                                const float* __restrict__ mlp, float* __restrict__ c0_out,
                                                                const float* __restrict__ w_rel1, int B) {
  constexpr int N = NP, F = FP, TS = SG_TS;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int TM = SG_TMAX;
  const int T = a.n_steps;   // <= TM, a.s0 == 0: the whole backward
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sD = smem;                                   // [N][TS]  D of node j
  float* sG2 = sD + N * TS;                           // [TM][TS] dagg2 of step t
  unsigned long long* sRow = reinterpret_cast<unsigned long long*>(sG2 + TM * TS);   // [TM][2] live rows of step t
  unsigned long long* sCol = sRow + TM * 2;           // [N][2] the steps that aggregate node j
  int* sHdr = reinterpret_cast<int*>(sCol + N * 2);   // [TM][2] cur, L
  float* sW0a = reinterpret_cast<float*>(sHdr + TM * 2);   // [F][TS]  W0[o][f], f < F (c0 below)
  float* sW1r = sW0a + F * TS;                        // [H1][TS] W_rel1[h][f]  (w_rel1)
  float* sS = sW1r + F * TS;                          // [N][TS]  the sums of G1 rows, before W_rel1^T  (w_rel1)

  // ---- this lane's rows j = lane and lane + 64 of the caches: h1 | x, 64 floats each -----------------------------------
  // (rows 64 .. 127 only where the chain got that far: the counts grow by one a step, the last step's is the largest)
  const int cur_last = __builtin_amdgcn_readfirstlane(a.hdr[2 * ((size_t)(T - 1) * B + b)]);
  f32x4 hx[2][16];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const size_t rj = (size_t)b * N + lane + 64 * r;
    if (r == 0 || cur_last > 64) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        hx[r][q] = *reinterpret_cast<const f32x4*>(a.c_h1 + rj * F + 4 * q);
        hx[r][8 + q] = *reinterpret_cast<const f32x4*>(a.c_nodes + rj * F + 4 * q);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) hx[r][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  if (tid < N) { sCol[2 * tid] = 0ull; sCol[2 * tid + 1] = 0ull; }
  if (tid < TM) {
    const bool on = tid < T;
    const size_t it = (size_t)(on ? tid : 0) * B + b;
    sHdr[2 * tid] = on ? a.hdr[2 * it] : 0;
    sHdr[2 * tid + 1] = on ? a.hdr[2 * it + 1] : 0;
  }
  for (int e = tid; e < TM * 8; e += 256) {   // dagg2 rows
    const int t = e >> 3, q = e & 7;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t < T) v = *reinterpret_cast<const f32x4*>(a.dagg2 + ((size_t)t * B + b) * F + 4 * q);
    *reinterpret_cast<f32x4*>(sG2 + t * TS + 4 * q) = v;
  }
  const Mlp M = unpack_mlp(mlp, F);
  *reinterpret_cast<f32x4*>(sW0a + (tid >> 3) * TS + 4 * (tid & 7)) =
      *reinterpret_cast<const f32x4*>(M.w0 + (tid >> 3) * 2 * F + 4 * (tid & 7));
  if (w_rel1)
    *reinterpret_cast<f32x4*>(sW1r + (tid >> 3) * TS + 4 * (tid & 7)) =
        *reinterpret_cast<const f32x4*>(w_rel1 + (tid >> 3) * F + 4 * (tid & 7));
  __syncthreads();
  // ---- c0_t = b0 + W0a x_cur for pass B2 (one vector per step here, 16 matrix instructions per 16-row tile there), written
  //      where dagg2_t was (staged above): quarter-row threads, the step's node row in registers -------------------------
  for (int tq = tid; tq < 4 * T; tq += 256) {
    const int t = tq >> 2, q = tq & 3;
    {
      const int cur = sHdr[2 * t];
      const float* xc = a.c_nodes + ((size_t)b * N + (cur < N && cur >= 0 ? cur : 0)) * F;
      f32x4 xv[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) xv[c] = *reinterpret_cast<const f32x4*>(xc + 4 * c);
      f32x4 r[2];
#pragma unroll
      for (int oo = 0; oo < 8; ++oo) {
        const int o = 8 * q + oo;
        float pa = M.b0[o], pb = 0.f;
#pragma unroll
        for (int c = 0; c < 8; c += 2) {
          const f32x4 wa = *reinterpret_cast<const f32x4*>(sW0a + o * TS + 4 * c), wb = *reinterpret_cast<const f32x4*>(sW0a + o * TS + 4 * c + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) { pa = fmaf(wa[e], xv[c][e], pa); pb = fmaf(wb[e], xv[c + 1][e], pb); }
        }
        r[oo >> 2][oo & 3] = pa + pb;
      }
      float* dst = c0_out + ((size_t)t * B + b) * F + 8 * q;
      *reinterpret_cast<f32x4*>(dst) = r[0];
      *reinterpret_cast<f32x4*>(dst + 4) = r[1];
    }
  }
  // ---- masks: thread t builds the row mask of its step and sets its bit in the columns of the rows it holds -----------
  if (tid < TM) {
    unsigned long long m0 = 0, m1 = 0;
    if (tid < T) {
      const int L = sHdr[2 * tid + 1];
      const int* lv = a.live + ((size_t)tid * B + b) * N;
      // (the first eight entries in one round trip - a handful of live rows is the rule -, the rest one by one: a load per
      //  trip of this loop was a dependent round trip each)
      int e8[8];
      __builtin_memcpy(e8, lv, sizeof(e8));
      asm volatile("" ::: "memory");
#pragma unroll
      for (int l = 0; l < 8; ++l) {
        if (l < L) {
          const int j = e8[l] & (N - 1);
          if (j < 64) m0 |= 1ull << j; else m1 |= 1ull << (j - 64);
          atomicOr(reinterpret_cast<unsigned int*>(sCol + 2 * j) + (tid >> 5), 1u << (tid & 31));   // (four words: 128 steps)
        }
      }
      for (int l = 8; l < L; ++l) {
        const int j = lv[l] & (N - 1);
        if (j < 64) m0 |= 1ull << j; else m1 |= 1ull << (j - 64);
        atomicOr(reinterpret_cast<unsigned int*>(sCol + 2 * j) + (tid >> 5), 1u << (tid & 31));
      }
    }
    sRow[2 * tid] = m0;
    sRow[2 * tid + 1] = m1;
  }
  __syncthreads();
  // ---- D[j] = sum over the steps t' that aggregate node j, in step order, of dAgg1_{t'}[slot of j] -----------------------
  // (only the nodes this backward inserted are asked for below: rows cur of its first step .. cur of its last)
  const int j_lo = min(max(sHdr[0], 0), N - 1), j_hi = min(max(cur_last, j_lo), N - 1);
  for (int jq = 4 * j_lo + tid; jq < 4 * (j_hi + 1); jq += 256) {
    const int j = jq >> 2, q = jq & 3;   // eight floats of row j
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int cw = 0; cw < 2; ++cw) {   // steps 0 - 63, then 64 - 127
      unsigned long long cm = sCol[2 * j + cw];
      while (cm) {
        f32x4 v0[4], v1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v0[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          v1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (cm) {
            const int t = 64 * cw + __builtin_ctzll(cm);
            cm &= cm - 1;
            const unsigned long long r0 = sRow[2 * t], r1 = sRow[2 * t + 1];
            const int slot = j < 64 ? __popcll(r0 & ((1ull << j) - 1ull))
                                    : __popcll(r0) + __popcll(r1 & ((1ull << (j - 64)) - 1ull));
            const float* p = a.da + (((size_t)t * B + b) * N + slot) * F + 8 * q;
            v0[u] = *reinterpret_cast<const f32x4*>(p);
            v1[u] = *reinterpret_cast<const f32x4*>(p + 4);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc0 += v0[u]; acc1 += v1[u]; }
      }
    }
    float* dst = (w_rel1 ? sS : sD) + j * TS + 8 * q;
    *reinterpret_cast<f32x4*>(dst) = acc0;
    *reinterpret_cast<f32x4*>(dst + 4) = acc1;
  }
  __syncthreads();
  if (w_rel1) {   // D_j = W_rel1^T S_j: eight outputs per quarter-row thread
    for (int jq = 4 * j_lo + tid; jq < 4 * (j_hi + 1); jq += 256) {
      const int j = jq >> 2, q = jq & 3;
      f32x4 o0 = f32x4{0.f, 0.f, 0.f, 0.f}, o1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h4 = 0; h4 < 8; ++h4) {
        const f32x4 sv = *reinterpret_cast<const f32x4*>(sS + j * TS + 4 * h4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* wr = sW1r + (4 * h4 + e) * TS + 8 * q;
          o0 += sv[e] * *reinterpret_cast<const f32x4*>(wr);
          o1 += sv[e] * *reinterpret_cast<const f32x4*>(wr + 4);
        }
      }
      *reinterpret_cast<f32x4*>(sD + j * TS + 8 * q) = o0;
      *reinterpret_cast<f32x4*>(sD + j * TS + 8 * q + 4) = o1;
    }
    __syncthreads();
  }
  // ---- a wave per step: g_sel for every candidate, the softmax adjoint ------------------------------------------------
  for (int t = wave; t < T; t += 4) {
    const int cur = __builtin_amdgcn_readfirstlane(sHdr[2 * t]);
    if (cur <= 0) continue;   // no candidate rows (pass B2 skips the item too)
    const float* soft = a.tab.saved[t] + a.o_soft + (size_t)b * N;
    const float p0 = soft[lane], p1 = soft[lane + 64];
    const float* dg = sG2 + t * TS;                       // dagg2_t
    const float* dd = sD + (cur < N ? cur : N - 1) * TS;  // D of the node this step inserted
    float g0 = 0.f, g1 = 0.f, e0 = 0.f, e1 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(dg + 4 * q), v = *reinterpret_cast<const f32x4*>(dd + 4 * q);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        g0 = fmaf(u[k], hx[0][q][k], g0);
        e0 = fmaf(v[k], hx[0][8 + q][k], e0);
        g1 = fmaf(u[k], hx[1][q][k], g1);
        e1 = fmaf(v[k], hx[1][8 + q][k], e1);
      }
    }
    const bool l0 = lane < cur, l1 = lane + 64 < cur;
    const float s0 = l0 ? g0 + e0 : 0.f, s1 = l1 ? g1 + e1 : 0.f;
    const float q0 = l0 ? p0 : 0.f, q1 = l1 ? p1 : 0.f;
    const float dot = wave_sum(fmaf(q0, s0, q1 * s1));
    float* out = g_logit + ((size_t)t * B + b) * N;
    out[lane] = q0 * (s0 - dot);
    out[lane + 64] = q1 * (s1 - dot);
  }
}

constexpr int MLP_WAVES = 8;
constexpr int MLP_WSZ = 33 * FS + 3 * 32 * FS + 5 * 32 + 3;   // floats per wave: X (+ x_cur row) | P0 | H0 | P1 | 5 vectors
constexpr size_t lds_bptt_mlp() {
  return sizeof(float) * (3 * FP * FS + 7 * FP + 2 * MLP_WAVES * 32 + MLP_WAVES * MLP_WSZ);
}

__global__ __launch_bounds__(64 * MLP_WAVES) void k_learned_bptt_mlp(BpttB a, const float* __restrict__ g_logit,
                                                                     const float* __restrict__ mlp, float eps0,
                                                                     float eps1, float* __restrict__ slabs, int B,
                                                                     int N, int F) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int cf = lane & 31, cg = lane >> 5;   // column sums: column cf, rows cg, cg + 2, ... of the block
  const Mlp M = unpack_mlp(mlp, F);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW0b = smem;                  // [o][f] = W0[o][F + f]
  float* sW0a = sW0b + FP * FS;        // [o][f] = W0[o][f]
  float* sW1 = sW0a + FP * FS;
  float* sVec = sW1 + FP * FS;         // b0 | b1 | g0 | be0 | g1 | be1 | w2
  float* sCs = sVec + 7 * FP;          // [2 MLP_WAVES][32] partial column sums (epilogue)
  float* sWave = sCs + 2 * MLP_WAVES * 32;
  float* sX_ = sWave + wave * MLP_WSZ;  // [33][FS]: the block's node rows, row 32 = x_cur
  if (tid < 256) {
    gcm_fused::Stage<FP, FP, false, false> st_a, st_b, st_1;
    st_a.load(M.w0, F, F, 2 * F, tid);
    st_b.load(M.w0 + F, F, F, 2 * F, tid);
    st_1.load(M.w1, F, F, F, tid);
    st_a.store(sW0a, FS, tid);
    st_b.store(sW0b, FS, tid);
    st_1.store(sW1, FS, tid);
    if (tid < FP) {
      const int o = tid < F ? tid : F - 1;
      const bool ok = tid < F;
      sVec[tid] = ok ? M.b0[o] : 0.f;
      sVec[FP + tid] = ok ? M.b1[o] : 0.f;
      sVec[2 * FP + tid] = ok ? M.g0[o] : 0.f;
      sVec[3 * FP + tid] = ok ? M.be0[o] : 0.f;
      sVec[4 * FP + tid] = ok ? M.g1[o] : 0.f;
      sVec[5 * FP + tid] = ok ? M.be1[o] : 0.f;
      sVec[6 * FP + tid] = ok ? M.w2[o] : 0.f;
    }
  }
  for (int e = lane; e < MLP_WSZ; e += 64) sX_[e] = 0.f;
  f32x16 aW1, aW0b, aW0a;   // weight-gradient tiles [o = acc_row][f = li], summed over this wave's blocks
#pragma unroll
  for (int r = 0; r < 16; ++r) { aW1[r] = 0.f; aW0b[r] = 0.f; aW0a[r] = 0.f; }
  float c_w2 = 0.f, c_g1 = 0.f, c_be1 = 0.f, c_b2 = 0.f, c_b1 = 0.f, c_g0 = 0.f, c_be0 = 0.f, c_b0 = 0.f;
  __syncthreads();

  const long items = (long)a.n_steps * B;
  const long units = items * ((N + 31) / 32);   // block-major: the always-live first blocks of all items come first
#pragma unroll 1
  for (long u = (long)blockIdx.x * MLP_WAVES + wave; u < units; u += (long)gridDim.x * MLP_WAVES) {
    const int k = (int)(u / items);
    const long item = u - (long)k * items;
    const int s = (int)(item / B), b = (int)(item - (long)s * B), sg = a.s0 + s;
    const size_t it = (size_t)sg * B + b;
    const int cur = __builtin_amdgcn_readfirstlane(a.hdr[2 * it]);
    if (32 * k >= cur) continue;       // no candidate row in this block (wave-uniform)
    BSTAMP(0);
    int zv = 0;
    asm volatile("" : "+v"(zv));        // loop-variant LDS bases: keeps hipcc's LICM from parking every phase's
                                         // address arithmetic in registers in front of the loop
    float* const sX = sX_ + zv;
    float* const sP0 = sX + 33 * FS;
    float* const sH = sP0 + 32 * FS;
    float* const sP1 = sH + 32 * FS;
    float* const sGl = sP1 + 32 * FS;
    float* const sMu0 = sGl + 32;
    float* const sRs0 = sMu0 + 32;
    float* const sMu1 = sRs0 + 32;
    float* const sRs1 = sMu1 + 32;
    const float* xg = (a.c_nodes ? a.c_nodes : a.tab.saved[s]) + (size_t)b * N * F;
    const int j0 = 32 * k;
    const int jn = cur - j0 < 32 ? cur - j0 : 32;   // candidate rows of the block
    const bool have_u = a.c_u != nullptr;   // (uniform)
    float uv[16];                           // this lane's sixteen entries of the block's U rows: column li of rows r
    {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int e = lane + 64 * i, r = e >> 5, c = e & 31;
        const int j = j0 + r < N ? j0 + r : N - 1;
        v[i] = xg[(size_t)j * F + (c < F ? c : F - 1)];
        uv[i] = 0.f;
      }
      if (have_u) {
        const float* ug = a.c_u + (size_t)b * N * F;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int e = lane + 64 * i, r = e >> 5, c = e & 31;
          const int j = j0 + r < N ? j0 + r : N - 1;
          uv[i] = ug[(size_t)j * F + (c < F ? c : F - 1)];
        }
      }
      const float xc = xg[(size_t)cur * F + (li < F ? li : F - 1)];
      const float glv = g_logit[it * N + (j0 + li < N ? j0 + li : N - 1)];
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int e = lane + 64 * i, r = e >> 5, c = e & 31;
        // (rows behind the candidates carry exact-zero gradients below, but 0 x whatever a cache row that was never
        //  written holds is not 0 when that is a NaN: they enter the products as zeros)
        sX[r * FS + c] = (r < jn && c < F) ? v[i] : 0.f;
      }
      if (lh == 0) {
        sX[32 * FS + li] = li < F ? xc : 0.f;
        sGl[li] = li < jn ? glv : 0.f;
      }
    }
    wsync();
    BSTAMP(1);
    {
      // c0[o] = b0[o] + W0a[o, :] . x_cur, the same for every row of the block: a vector, not a 32 x 32 x 32 product with
      // x_cur repeated in every row (which it was: 16 of the block's MFMAs) - the half-waves split f, one cross-half add
      const float* w = sW0a + li * FS + 16 * lh;
      const float* x = sX + 32 * FS + 16 * lh;
      float p = 0.f;
#pragma unroll
      for (int f = 0; f < 16; ++f) p = fmaf(w[f], x[f], p);
      const float c0 = gcm_xor32_add(p) + sVec[li];
      if (have_u) {
        // P0[j] = U[j] + c0 from the chain's cache of the first-layer product (the forward kept U[j] = W0b x_j of
        // every stored row: gcm_learned_step_cached, cache_u): no matrix product at all, and the P0 the forward normalised
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int r = (lane + 64 * i) >> 5;
          sP0[r * FS + li] = (r < jn ? uv[i] : 0.f) + c0;   // (rows behind the candidates: U is not written there)
        }
      } else {   // P0 = X W0b^T + c0
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        gcm_fused::mma32b<32>(acc, sX, FS, 1, sW0b, 1, FS, li, lh);
#pragma unroll
        for (int r = 0; r < 16; ++r) sP0[gcm_fused::acc_row(r, lh) * FS + li] = acc[r] + c0;
      }
    }
    wsync();
    BSTAMP(2);
    relu_ln_rows_to(sP0, sH, lane, F, sVec + 2 * FP, sVec + 3 * FP, eps0, sMu0, sRs0);
    wsync();
    BSTAMP(3);
    {   // P1 = H0 W1^T + b1
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      gcm_fused::mma32b<32>(acc, sH, FS, 1, sW1, 1, FS, li, lh);
#pragma unroll
      for (int r = 0; r < 16; ++r) sP1[gcm_fused::acc_row(r, lh) * FS + li] = acc[r] + sVec[FP + li];
    }
    wsync();
    BSTAMP(4);
    relu_ln_rows(sP1, lane, F, nullptr, nullptr, eps1, sMu1, sRs1, /*write=*/false);
    wsync();
    BSTAMP(5);
    {   // dw2, dgamma1, dbeta1, db2
      // (all 16 rows of this lane's group, no trip count: rows beyond the candidates carry g_logit = 0 and exact
      //  zeros behind it - a loop over `jn` rows waited for its four LDS reads at every trip)
      const float w2f = sVec[6 * FP + cf], g1f = sVec[4 * FP + cf], be1f = sVec[5 * FP + cf];
#pragma unroll 4
      for (int i = 0; i < 16; ++i) {
        const int j = cg + 2 * i;
        const float gl = sGl[j];
        const float v = sP1[j * FS + cf];
        const float xh = ((v > 0.f ? v : 0.f) - sMu1[j]) * sRs1[j];
        c_w2 = fmaf(gl, fmaf(xh, g1f, be1f), c_w2);
        c_g1 = fmaf(gl * w2f, xh, c_g1);
        c_be1 = fmaf(gl, w2f, c_be1);
        if (cf == 0) c_b2 += gl;
      }
    }
    wsync();
    BSTAMP(6);
    {   // gP1: the LayerNorm-1 adjoint of gx[f] = g_logit[j] w2[f] gamma1[f]
      const int f0 = (lane & 1) * (FP / 2);
      float gx[FP / 2];
      const float gl = sGl[lane >> 1];
#pragma unroll
      for (int q = 0; q < FP / 8; ++q) {
        const float4 w2 = reinterpret_cast<const float4*>(sVec + 6 * FP + f0)[q];
        const float4 g1 = reinterpret_cast<const float4*>(sVec + 4 * FP + f0)[q];
        gx[4 * q] = gl * w2.x * g1.x; gx[4 * q + 1] = gl * w2.y * g1.y;
        gx[4 * q + 2] = gl * w2.z * g1.z; gx[4 * q + 3] = gl * w2.w * g1.w;
      }
      if (F == FP) relu_ln_rows_bwd_v<true>(sP1, lane, F, sMu1, sRs1, gx);
      else relu_ln_rows_bwd_v<false>(sP1, lane, F, sMu1, sRs1, gx);
    }
    wsync();
    BSTAMP(7);
#pragma unroll 4
    for (int i = 0; i < 16; ++i) c_b1 += sP1[(cg + 2 * i) * FS + cf];
    {   // dW1 += gP1^T H0;  gH0 = gP1 W1 (over P1's place)
      f32x16 gh;
#pragma unroll
      for (int r = 0; r < 16; ++r) gh[r] = 0.f;
      gcm_fused::mma32b<32>(aW1, sP1, 1, FS, sH, FS, 1, li, lh);
      gcm_fused::mma32b<32>(gh, sP1, FS, 1, sW1, FS, 1, li, lh);
      wsync();
#pragma unroll
      for (int r = 0; r < 16; ++r) sP1[gcm_fused::acc_row(r, lh) * FS + li] = gh[r];
    }
    wsync();
    BSTAMP(8);
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {   // dgamma0, dbeta0
      const int j = cg + 2 * i;
      const float v = sP0[j * FS + cf];
      const float xh = ((v > 0.f ? v : 0.f) - sMu0[j]) * sRs0[j];
      const float ghv = sP1[j * FS + cf];
      c_g0 = fmaf(ghv, xh, c_g0);
      c_be0 += ghv;
    }
    wsync();
    BSTAMP(9);
    {   // gP0: the LayerNorm-0 adjoint of gx[f] = gH0[j][f] gamma0[f]
      const int f0 = (lane & 1) * (FP / 2);
      float gx[FP / 2];
      const float* gh = sP1 + (lane >> 1) * FS + f0;
#pragma unroll
      for (int q = 0; q < FP / 8; ++q) {
        const float4 g0 = reinterpret_cast<const float4*>(sVec + 2 * FP + f0)[q];
        gx[4 * q] = gh[4 * q] * g0.x; gx[4 * q + 1] = gh[4 * q + 1] * g0.y;
        gx[4 * q + 2] = gh[4 * q + 2] * g0.z; gx[4 * q + 3] = gh[4 * q + 3] * g0.w;
      }
      if (F == FP) relu_ln_rows_bwd_v<true>(sP0, lane, F, sMu0, sRs0, gx);
      else relu_ln_rows_bwd_v<false>(sP0, lane, F, sMu0, sRs0, gx);
    }
    wsync();
    BSTAMP(10);
    float cb = 0.f;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) cb += sP0[(cg + 2 * i) * FS + cf];
    c_b0 += cb;
    gcm_fused::mma32b<32>(aW0b, sP0, 1, FS, sX, FS, 1, li, lh);
    {
      // dW0a += gP0^T (x_cur for every row) = (column sums of gP0) x_cur^T: an outer product of two vectors the wave
      // holds already - sixteen fmas a lane instead of sixteen MFMAs a block
      sGl[cf] = gcm_xor32_add(cb);       // (g_logit is consumed; both half-waves write the same value)
      wsync();
      const float xcf = sX[32 * FS + li];
#pragma unroll
      for (int r = 0; r < 16; ++r) aW0a[r] = fmaf(sGl[gcm_fused::acc_row(r, lh)], xcf, aW0a[r]);
    }
    wsync();   // the images are rewritten by the wave's next block
    BSTAMP(11);
  }

  // ---- one slab per workgroup (packed edge-network layout), waves and row groups summed in fixed order ---
  const int Pm = 3 * F * F + 7 * F + 1;
  float* slab = slabs + (size_t)blockIdx.x * Pm;
  const int o_b0 = 2 * F * F, o_g0 = o_b0 + F, o_be0 = o_g0 + F, o_w1 = o_be0 + F, o_b1 = o_w1 + F * F;
  const int o_g1 = o_b1 + F, o_be1 = o_g1 + F, o_w2 = o_be1 + F, o_b2 = o_w2 + F;
  float* sR = sWave;   // [MLP_WAVES][1024]
  static_assert(MLP_WSZ >= 1024, "the epilogue's tiles live in the waves' images");
  auto tile_out = [&](const f32x16& acc, int row_stride, int col0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sR[wave * MLP_WSZ + gcm_fused::acc_row(r, lh) * 32 + li] = acc[r];
    __syncthreads();
    for (int e = tid; e < 1024; e += 64 * MLP_WAVES) {
      const int o = e >> 5, f = e & 31;
      if (o < F && f < F) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < MLP_WAVES; ++w) t += sR[w * MLP_WSZ + e];
        slab[col0 + o * row_stride + f] = t;
      }
    }
    __syncthreads();
  };
  __syncthreads();
  tile_out(aW0a, 2 * F, 0);
  tile_out(aW0b, 2 * F, F);
  tile_out(aW1, F, o_w1);
  auto col_out = [&](float v, int off, int n) {
    sCs[(2 * wave + cg) * 32 + cf] = v;
    __syncthreads();
    if (tid < n) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 2 * MLP_WAVES; ++q) t += sCs[q * 32 + tid];
      slab[off + tid] = t;
    }
    __syncthreads();
  };
  col_out(c_b0, o_b0, F);
  col_out(c_g0, o_g0, F);
  col_out(c_be0, o_be0, F);
  col_out(c_b1, o_b1, F);
  col_out(c_g1, o_g1, F);
  col_out(c_be1, o_be1, F);
  col_out(c_w2, o_w2, F);
  col_out(c_b2, o_b2, 1);
}

// ---------------------------------------------------------------------------------------------
// Pass B2 at the exact shapes on the chain's U cache, in REGISTERS (round 6): k_learned_bptt_mlp walks every 32-row block
// through eleven phases of LDS images (19 KB per wave: eight waves fill the CU's LDS, two per SIMD) at ~32 k cycles a
// block.  Here a wave owns a 16-ROW tile in the layout of k_learned_select8: lane (m, g) holds row 16 k + m at the eight
// features 16 ct + 4 g + i (ct < 2, i < 4) - the first LayerNorm's operands straight from the U cache, every product on
// v_mfma_f32_16x16x4_f32 with THAT register as the B operand and a weight row from LDS as A, so that each product's
// accumulators are again row m at the lane's own eight features (the transposed product, as in the forward).  Both
// LayerNorms and both adjoints reduce in the lane plus two row / half swaps.  Only the weight-gradient products contract
// over ROWS: their operands go through two 16 x 32 LDS tiles per wave (written as rows, read as columns).  4.6 KB of LDS a
// wave, <= 168 registers: twelve waves per CU, three per SIMD.
//   c0 = b0 + W0a x_cur comes from the same instruction (x_cur as every column of B: the accumulators hold c0 at the lane's
//   features); dW0a = (column sums of gP0) x_cur^T takes the sum over the tile's four row groups before ONE instruction per
//   16 x 16 tile; db1 / db0 are the same column sums.
// Same slab layout and slab count as k_learned_bptt_mlp (one per workgroup, summed by gcm_sum_slabs_acc).
// ---------------------------------------------------------------------------------------------
constexpr int M16_WAVES = 12;
constexpr int M16_TS = FP + 4;
constexpr size_t lds_bptt_mlp16() { return sizeof(float) * (4 * FP * M16_TS + 8 * FP + M16_WAVES * 3 * 16 * M16_TS); }

// CACHED: the steps of the chunk are cached steps (node rows and U from the chain's caches, every graph of a step at the same
// count); else each step's own record holds its node matrix, U is one more product (W0b rows as A, the row's x as B) and a
// graph's count is whatever its header says (tiles without a candidate row are skipped behind the header read).
template <bool CACHED>
__global__ __launch_bounds__(64 * M16_WAVES) void k_learned_bptt_mlp16(BpttB a, const float* __restrict__ g_logit,
                                                                       const float* __restrict__ mlp, float eps0,
                                                                       float eps1, float* __restrict__ slabs, int B) {
  constexpr int N = NP, F = FP, TS = M16_TS;
  static_assert(3 * 16 * M16_TS >= 32 * 32, "the epilogue's 32 x 32 tiles live in the waves' transposition tiles");
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Mlp M = unpack_mlp(mlp, F);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW1 = smem;                 // [o][f]
  float* sW1T = sW1 + FP * TS;       // [f][o]
  float* sW0a = sW1T + FP * TS;      // [o][f] = W0[o][f]
  float* sW0b = sW0a + FP * TS;      // [o][f] = W0[o][F + f]  (!CACHED)
  float* sVec = sW0b + FP * TS;      // b0 | b1 | g0 | be0 | g1 | be1 | w2
  float* sTiles = sVec + 8 * FP;
  float* sTa = sTiles + wave * (3 * 16 * TS);   // this wave's [16][TS] tiles: the gradient rows (gP1, then gP0) | H0 | X
  float* sTb = sTa + 16 * TS;
  float* sTc = sTb + 16 * TS;
  for (int e = tid; e < FP * FP; e += 64 * M16_WAVES) {
    const int o = e >> 5, f = e & 31;
    const float w1 = M.w1[e];
    sW1[o * TS + f] = w1;
    sW1T[f * TS + o] = w1;
    sW0a[o * TS + f] = M.w0[o * 2 * F + f];
    if (!CACHED) sW0b[o * TS + f] = M.w0[o * 2 * F + F + f];
  }
  if (tid < FP) {
    sVec[tid] = M.b0[tid];
    sVec[FP + tid] = M.b1[tid];
    sVec[2 * FP + tid] = M.g0[tid];
    sVec[3 * FP + tid] = M.be0[tid];
    sVec[4 * FP + tid] = M.g1[tid];
    sVec[5 * FP + tid] = M.be1[tid];
    sVec[6 * FP + tid] = M.w2[tid];
  }
  // weight-gradient tiles: D[o = 16 ot + 4 g + r][f = 16 ft + m], summed over this wave's tiles
  f32x4 aW1[2][2], aW0b[2][2], aW0a[2][2];
#pragma unroll
  for (int ot = 0; ot < 2; ++ot)
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
      aW1[ot][ft] = f32x4{0.f, 0.f, 0.f, 0.f};
      aW0b[ot][ft] = f32x4{0.f, 0.f, 0.f, 0.f};
      aW0a[ot][ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  // vector gradients: at the lane's eight features (a partial sum over the rows m), or at column 16 ot + m (a partial
  // sum over the row groups g) where the transposed tile gives them
  // (dw2 and dgamma1 both come from c_s1[f] = sum over rows of g_logit xhat1[f]: dw2 = gamma1 c_s1 + beta1 sum(g_logit),
  //  dgamma1 = w2 c_s1)
  float c_s1[8], c_g0[8], c_be0[8], c_b1[2] = {0.f, 0.f}, c_b0[2] = {0.f, 0.f}, c_gl = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) { c_s1[k] = 0.f; c_g0[k] = 0.f; c_be0[k] = 0.f; }
  __syncthreads();

  // The units: (tile k, step, graph) with a candidate row in the tile.  At the cached steps of a chain every graph holds
  // the same number of nodes, one more each step (gcm_learned_step_cached: a graph whose count differs is left untouched
  // and flagged): cur = cur_first + s at the chunk's step s, so tile k is live at the steps cur_first + s > 16 k - an
  // analytic list, tile-major, no unit is fetched to be skipped (eight tiles x T x B units walked, five of eight of them
  // empty at T = 64, each behind a dependent header read: 12 us of the pass).  `live` below still comes from each item's
  // recorded header.
  const int cur_first = CACHED ? __builtin_amdgcn_readfirstlane(a.hdr[2 * (size_t)a.s0 * B]) : N;
  long pre[N / 16 + 1];
  int first[N / 16];
  pre[0] = 0;
#pragma unroll
  for (int k = 0; k < N / 16; ++k) {
    const int fs = 16 * k + 1 - cur_first > 0 ? 16 * k + 1 - cur_first : 0;   // (!CACHED: every tile at every step)
    first[k] = fs;
    pre[k + 1] = pre[k] + (fs < a.n_steps ? (long)(a.n_steps - fs) * B : 0);
  }
  const long units = pre[N / 16];
#pragma unroll 1
  for (long u = (long)blockIdx.x * M16_WAVES + wave; u < units; u += (long)gridDim.x * M16_WAVES) {
    int k = 0, fs = first[0];
    long base = 0;
#pragma unroll
    for (int q = 1; q < N / 16; ++q)
      if (u >= pre[q]) { k = q; fs = first[q]; base = pre[q]; }
    const long item = u - base;
    const int s = fs + (int)(item / B), b = (int)(item % B), sg = a.s0 + s;
    const size_t it = (size_t)sg * B + b;
    int cur_rec = a.hdr[2 * it];   // (the recorded header: the rows' masks; CACHED: the addresses below do not wait for it)
    int cur = cur_first + s;
    if (!CACHED) {
      cur_rec = __builtin_amdgcn_readfirstlane(cur_rec);
      cur = cur_rec;
      if (16 * k >= cur) continue;   // no candidate row in this tile (wave-uniform)
    }
    const float* xg = CACHED ? a.c_nodes : a.tab.saved[s];
    const int r = 16 * k + m;
    const bool live = r < cur_rec;
    const size_t gb = (size_t)b * N;
    // ---- loads: this row's U and x at the lane's features, x_cur both ways, the logit's gradient -----------------------
    f32x4 u4[2], xc4[2];
    {
      f32x4 x4[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        xc4[ct] = *reinterpret_cast<const f32x4*>(xg + (gb + cur) * F + 16 * ct + 4 * g);
        if (CACHED) u4[ct] = *reinterpret_cast<const f32x4*>(a.c_u + (gb + r) * F + 16 * ct + 4 * g);
        x4[ct] = *reinterpret_cast<const f32x4*>(xg + (gb + r) * F + 16 * ct + 4 * g);
      }
      if (!CACHED) {   // U = X W0b^T: W0b rows as A, this row's x (zeros behind the candidates) as B
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) u4[ot] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int ot = 0; ot < 2; ++ot) {
            const f32x4 wa = *reinterpret_cast<const f32x4*>(sW0b + (16 * ot + m) * TS + 16 * ct + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i)
              u4[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], live ? x4[ct][i] : 0.f, u4[ot], 0, 0, 0);
          }
      }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {   // the X tile, for dW0b at the end (a cache row that was never written may hold anything)
        f32x4 xv = x4[ct];
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[i] = live ? xv[i] : 0.f;
        *reinterpret_cast<f32x4*>(sTc + m * TS + 16 * ct + 4 * g) = xv;
      }
    }
    const float xcn0 = xg[(gb + cur) * F + m], xcn1 = xg[(gb + cur) * F + 16 + m];
    float gl = g_logit[it * N + r];
    gl = live ? gl : 0.f;
    // ---- c0 at the lane's features: W0a (A) x x_cur (B, the same in every column) ------------------------------------
    f32x4 p0[2];
    if (CACHED && a.c0) {   // (uniform) pass B1 left the step's c0
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) p0[ot] = *reinterpret_cast<const f32x4*>(a.c0 + it * F + 16 * ot + 4 * g);
    } else {
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) p0[ot] = *reinterpret_cast<const f32x4*>(sVec + 16 * ot + 4 * g);   // b0
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
          const f32x4 wa = *reinterpret_cast<const f32x4*>(sW0a + (16 * ot + m) * TS + 16 * ct + 4 * g);
#pragma unroll
          for (int i = 0; i < 4; ++i) p0[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], xc4[ct][i], p0[ot], 0, 0, 0);
        }
    }
    // ---- P0 = U + c0, ReLU, LayerNorm 0 (rows behind the candidates: U is not written there - they enter as zeros) ----
    float xh0[8], h0[8];
    unsigned pos0 = 0;
    float rstd0;
    {
      float sm = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float v = (live ? u4[q >> 2][q & 3] : 0.f) + p0[q >> 2][q & 3];
        pos0 |= (v > 0.f ? 1u : 0u) << q;
        xh0[q] = v > 0.f ? v : 0.f;
        sm += xh0[q];
      }
      sm = gcm_xor16_add(sm);
      sm = gcm_xor32_add(sm);
      const float mean = sm / (float)F;
      float qv = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) { xh0[q] -= mean; qv = fmaf(xh0[q], xh0[q], qv); }
      qv = gcm_xor16_add(qv);
      qv = gcm_xor32_add(qv);
      rstd0 = rsqrtf(qv / (float)F + eps0);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(sVec + 2 * FP + 16 * ct + 4 * g);
        const f32x4 be0 = *reinterpret_cast<const f32x4*>(sVec + 3 * FP + 16 * ct + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xh0[4 * ct + i] *= rstd0;
          h0[4 * ct + i] = fmaf(xh0[4 * ct + i], g0[i], be0[i]);
        }
        *reinterpret_cast<f32x4*>(sTb + m * TS + 16 * ct + 4 * g) = f32x4{h0[4 * ct], h0[4 * ct + 1], h0[4 * ct + 2], h0[4 * ct + 3]};
      }
    }
    // ---- P1 = H0 W1^T + b1 (transposed: W1 rows as A), ReLU, LayerNorm 1's statistics -----------------------------------
    float xh1[8];
    unsigned pos1 = 0;
    float rstd1;
    {
      f32x4 acc[2];
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) acc[ot] = *reinterpret_cast<const f32x4*>(sVec + FP + 16 * ot + 4 * g);   // b1
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
          const f32x4 wa = *reinterpret_cast<const f32x4*>(sW1 + (16 * ot + m) * TS + 16 * ct + 4 * g);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], h0[4 * ct + i], acc[ot], 0, 0, 0);
        }
      float sm = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float v = acc[q >> 2][q & 3];
        pos1 |= (v > 0.f ? 1u : 0u) << q;
        xh1[q] = v > 0.f ? v : 0.f;
        sm += xh1[q];
      }
      sm = gcm_xor16_add(sm);
      sm = gcm_xor32_add(sm);
      const float mean = sm / (float)F;
      float qv = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) { xh1[q] -= mean; qv = fmaf(xh1[q], xh1[q], qv); }
      qv = gcm_xor16_add(qv);
      qv = gcm_xor32_add(qv);
      rstd1 = rsqrtf(qv / (float)F + eps1);
#pragma unroll
      for (int q = 0; q < 8; ++q) xh1[q] *= rstd1;
    }
    // ---- dw2, dgamma1, db2 / dbeta1 (through the sum of g_logit); gP1: LayerNorm 1's adjoint of g_logit w2 gamma1 ----------
    float gp1[8];
    {
      float gx[8], m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const f32x4 w2 = *reinterpret_cast<const f32x4*>(sVec + 6 * FP + 16 * ct + 4 * g);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(sVec + 4 * FP + 16 * ct + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int q = 4 * ct + i;
          c_s1[q] = fmaf(gl, xh1[q], c_s1[q]);
          gx[q] = gl * w2[i] * g1[i];
          m1 += gx[q];
          m2 = fmaf(gx[q], xh1[q], m2);
        }
      }
      c_gl += gl;
      m1 = gcm_xor16_add(m1); m1 = gcm_xor32_add(m1);
      m2 = gcm_xor16_add(m2); m2 = gcm_xor32_add(m2);
      m1 /= (float)F;
      m2 /= (float)F;
#pragma unroll
      for (int q = 0; q < 8; ++q) gp1[q] = ((pos1 >> q) & 1u) ? rstd1 * (gx[q] - m1 - xh1[q] * m2) : 0.f;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
        *reinterpret_cast<f32x4*>(sTa + m * TS + 16 * ct + 4 * g) = f32x4{gp1[4 * ct], gp1[4 * ct + 1], gp1[4 * ct + 2], gp1[4 * ct + 3]};
    }
    // ---- gH0 = gP1 W1 (W1's columns as A); dgamma0, dbeta0; gP0: LayerNorm 0's adjoint of gH0 gamma0 ----------------------
    float gp0[8];
    {
      f32x4 acc[2];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) acc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
          const f32x4 wa = *reinterpret_cast<const f32x4*>(sW1T + (16 * ft + m) * TS + 16 * ct + 4 * g);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[i], gp1[4 * ct + i], acc[ft], 0, 0, 0);
        }
      // dW1 += gP1^T H0 now (the weight gradients contract over the tile's ROWS: rows written above, columns read), so that
      // gP1 is dead before LayerNorm 0's adjoint
      wsync();
      {
        float sa[2] = {0.f, 0.f};
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) {   // instruction sI contracts rows 4 g + sI
          const float a0 = sTa[(4 * g + sI) * TS + m], a1 = sTa[(4 * g + sI) * TS + 16 + m];
          const float b0 = sTb[(4 * g + sI) * TS + m], b1 = sTb[(4 * g + sI) * TS + 16 + m];
          sa[0] += a0;
          sa[1] += a1;
          aW1[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, aW1[0][0], 0, 0, 0);
          aW1[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, aW1[0][1], 0, 0, 0);
          aW1[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, aW1[1][0], 0, 0, 0);
          aW1[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, aW1[1][1], 0, 0, 0);
        }
        c_b1[0] += sa[0];
        c_b1[1] += sa[1];
      }
      float gx[8], m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(sVec + 2 * FP + 16 * ct + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int q = 4 * ct + i;
          const float gh = acc[ct][i];
          c_g0[q] = fmaf(gh, xh0[q], c_g0[q]);
          c_be0[q] += gh;
          gx[q] = gh * g0[i];
          m1 += gx[q];
          m2 = fmaf(gx[q], xh0[q], m2);
        }
      }
      m1 = gcm_xor16_add(m1); m1 = gcm_xor32_add(m1);
      m2 = gcm_xor16_add(m2); m2 = gcm_xor32_add(m2);
      m1 /= (float)F;
      m2 /= (float)F;
#pragma unroll
      for (int q = 0; q < 8; ++q) gp0[q] = ((pos0 >> q) & 1u) ? rstd0 * (gx[q] - m1 - xh0[q] * m2) : 0.f;
    }
    // dW0b += gP0^T X;  dW0a += (column sums of gP0) x_cur^T
    wsync();   // (dW1's reads of the gradient tile are done)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
      *reinterpret_cast<f32x4*>(sTa + m * TS + 16 * ct + 4 * g) = f32x4{gp0[4 * ct], gp0[4 * ct + 1], gp0[4 * ct + 2], gp0[4 * ct + 3]};
    wsync();
    {
      float sa[2] = {0.f, 0.f};
#pragma unroll
      for (int sI = 0; sI < 4; ++sI) {
        const float a0 = sTa[(4 * g + sI) * TS + m], a1 = sTa[(4 * g + sI) * TS + 16 + m];
        const float b0 = sTc[(4 * g + sI) * TS + m], b1 = sTc[(4 * g + sI) * TS + 16 + m];
        sa[0] += a0;
        sa[1] += a1;
        aW0b[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, aW0b[0][0], 0, 0, 0);
        aW0b[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, aW0b[0][1], 0, 0, 0);
        aW0b[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, aW0b[1][0], 0, 0, 0);
        aW0b[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, aW0b[1][1], 0, 0, 0);
      }
      c_b0[0] += sa[0];
      c_b0[1] += sa[1];
      aW0a[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[0], xcn0, aW0a[0][0], 0, 0, 0);
      aW0a[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[0], xcn1, aW0a[0][1], 0, 0, 0);
      aW0a[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[1], xcn0, aW0a[1][0], 0, 0, 0);
      aW0a[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[1], xcn1, aW0a[1][1], 0, 0, 0);
    }
    wsync();   // the tiles are rewritten by the wave's next unit
  }

  // ---- one slab per workgroup (packed edge-network layout), waves summed in fixed order ---------------------------------
  const int Pm = 3 * F * F + 7 * F + 1;
  float* slab = slabs + (size_t)blockIdx.x * Pm;
  const int o_b0 = 2 * F * F, o_g0 = o_b0 + F, o_be0 = o_g0 + F, o_w1 = o_be0 + F, o_b1 = o_w1 + F * F;
  const int o_g1 = o_b1 + F, o_be1 = o_g1 + F, o_w2 = o_be1 + F, o_b2 = o_w2 + F;
  float* sR = sTiles;   // [M16_WAVES][3 * 16 * TS >= 1024]
  constexpr int RS = 3 * 16 * M16_TS;
  auto tile_out = [&](const f32x4 (&acc)[2][2], int row_stride, int col0) {
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int i = 0; i < 4; ++i) sR[wave * RS + (16 * ot + 4 * g + i) * 32 + 16 * ft + m] = acc[ot][ft][i];
    __syncthreads();
    for (int e = tid; e < 1024; e += 64 * M16_WAVES) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < M16_WAVES; ++w) t += sR[w * RS + e];
      slab[col0 + (e >> 5) * row_stride + (e & 31)] = t;
    }
    __syncthreads();
  };
  __syncthreads();
  tile_out(aW0a, 2 * F, 0);
  tile_out(aW0b, 2 * F, F);
  tile_out(aW1, F, o_w1);
  // vectors at the lane's features: the rows m of the tile summed (a DPP row), one value per (wave, feature)
  auto feat_out = [&](const float (&v)[8], int off) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float t = row16_sum(v[q]);
      if (m == 0) sR[wave * RS + 16 * (q >> 2) + 4 * g + (q & 3)] = t;
    }
    __syncthreads();
    if (tid < FP) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < M16_WAVES; ++w) t += sR[w * RS + tid];
      slab[off + tid] = t;
    }
    __syncthreads();
  };
  // vectors at column 16 ot + m: the row groups g summed
  auto col_out = [&](const float (&v)[2], int off) {
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
      float t = gcm_xor16_add(v[ot]);
      t = gcm_xor32_add(t);
      if (g == 0) sR[wave * RS + 16 * ot + m] = t;
    }
    __syncthreads();
    if (tid < FP) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < M16_WAVES; ++w) t += sR[w * RS + tid];
      slab[off + tid] = t;
    }
    __syncthreads();
  };
  col_out(c_b0, o_b0);
  feat_out(c_g0, o_g0);
  feat_out(c_be0, o_be0);
  col_out(c_b1, o_b1);
  {   // dw2 = gamma1 S1 + beta1 sum(g_logit), dgamma1 = w2 S1, dbeta1 = w2 sum(g_logit), db2 = sum(g_logit)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float t = row16_sum(c_s1[q]);
      if (m == 0) sR[wave * RS + 16 * (q >> 2) + 4 * g + (q & 3)] = t;
    }
    const float t = wave_sum(c_gl);
    if (lane == 0) sR[wave * RS + FP] = t;
    __syncthreads();
    if (tid <= FP) {
      float sgl = 0.f, s1 = 0.f;
#pragma unroll
      for (int w = 0; w < M16_WAVES; ++w) {
        sgl += sR[w * RS + FP];
        s1 += sR[w * RS + (tid < FP ? tid : 0)];
      }
      if (tid < FP) {
        slab[o_w2 + tid] = fmaf(sVec[4 * FP + tid], s1, sVec[5 * FP + tid] * sgl);
        slab[o_g1 + tid] = sVec[6 * FP + tid] * s1;
        slab[o_be1 + tid] = sVec[6 * FP + tid] * sgl;
      } else {
        slab[o_b2] = sgl;
      }
    }
  }
}

constexpr size_t lds_select() { return sizeof(float) * (3 * NP * FS + 2 * FP * FS + FP * GS + 7 * FP + NP); }
constexpr size_t lds_select_tail() { return lds_select() + sizeof(float) * (NP * FS + 4 * FP * GS); }
constexpr size_t lds_select_steady() { return lds_select_tail() + sizeof(uint32_t) * (NP * 4 + 4); }
constexpr size_t lds_bwd() {
  return sizeof(float) * (5 * NP * FS + 3 * FP * FS + 4096 + 7 * FP + 7 * NP + 32 + 64 + 64 + 32 + 256 + NP + 8 +
                          (2 * FP * FP + FP + 2 * FP * FP + FP) + (3 * FP * FP + 7 * FP + 1) + 3);
}

}  // namespace gcm_learned

extern "C" int gcm_learned_step_supported(int N, int F, int H1, int H2) {
  return (N > 0 && N <= 128 && F > 0 && F <= 32 && H1 > 0 && H1 <= 32 && H2 > 0 && H2 <= 32) ? 1 : 0;
}

extern "C" size_t gcm_learned_mlp_param_count(int F) { return 3 * (size_t)F * F + 7 * (size_t)F + 1; }

extern "C" int gcm_learned_select_fused(const float* nodes, float* adj, const int64_t* cur_idx,
                                        const float* noise, int noise_is_exp, const float* mlp_params,
                                        float eps0, float eps1, float cutoff, float* soft, int B, int N,
                                        int F, gcm_stream_t stream) {
  GCM_REQUIRE(nodes && adj && cur_idx && noise && mlp_params && soft && B > 0);
  if (!gcm_learned_step_supported(N, F, 1, 1)) return GCM_EUNSUPPORTED;
  constexpr size_t lds = gcm_learned::lds_select();
  auto kern = gcm_learned::k_learned_select<0, 0>;
  gcm_allow_dynamic_lds((const void*)kern, lds);
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, (hipStream_t)stream, nodes, adj, cur_idx, noise,
                     noise_is_exp, mlp_params, eps0, eps1, cutoff, soft, N, F, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const int64_t*)nullptr, (float*)nullptr,
                     (int64_t*)nullptr, (int64_t*)nullptr, (uint32_t*)nullptr, (float*)nullptr, (float*)nullptr,
                     gcm_learned::GnnTail{}, -1);
  return gcm_launch_status();
}

/* gcm_state_advance_fwd + gcm_learned_select_fused in ONE kernel: the state copy (overflow roll folded in)
 * travels through the registers of the workgroup that runs the edge network on the same node rows.
 * Needs N % 4 == 0 and F % 4 == 0 (GCM_EUNSUPPORTED otherwise: call the two entry points instead). */
extern "C" int gcm_learned_advance_select_fused(const float* obs, const float* nodes_in, const float* adj_in,
                                                const int64_t* count_in, const float* noise,
                                                int noise_is_exp, const float* mlp_params, float eps0,
                                                float eps1, float cutoff, float* nodes_out, float* adj_out,
                                                int64_t* cur_out, int64_t* count_out, float* soft,
                                                uint32_t* flags, int B, int N, int F, gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes_in && adj_in && count_in && noise && mlp_params && nodes_out && adj_out && cur_out &&
              count_out && soft && flags && B > 0);
  GCM_REQUIRE(nodes_out != nodes_in && adj_out != adj_in);
  if (!gcm_learned_step_supported(N, F, 1, 1) || (N & 3) || (F & 3)) return GCM_EUNSUPPORTED;
  constexpr size_t lds = gcm_learned::lds_select();
  auto kern = gcm_learned::k_learned_select<1, 0>;
  gcm_allow_dynamic_lds((const void*)kern, lds);
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, (hipStream_t)stream, (const float*)nullptr, adj_out,
                     (const int64_t*)nullptr, noise, noise_is_exp, mlp_params, eps0, eps1, cutoff, soft, N, F,
                     obs, nodes_in, adj_in, count_in, nodes_out, cur_out, count_out, flags, (float*)nullptr,
                     (float*)nullptr, gcm_learned::GnnTail{}, -1);
  return gcm_launch_status();
}

/* gcm_learned_advance_select_fused on a DONATED state: nodes / adj / count are advanced in place (count_out may
 * alias count_in); nodes_snap [B,N,F] receives the node matrix after the insert and adj_row [B,N] row cur of the
 * adjacency after the selection - what gcm_learned_bptt reads of the state (gcm_learned_step_layout, compact). */
extern "C" int gcm_learned_advance_select_inplace(const float* obs, float* nodes, float* adj, const int64_t* count_in,
                                                  const float* noise, int noise_is_exp, const float* mlp_params,
                                                  float eps0, float eps1, float cutoff, int64_t* cur_out,
                                                  int64_t* count_out, float* soft, float* nodes_snap,
                                                  float* adj_row, uint32_t* flags, int B, int N, int F,
                                                  gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes && adj && count_in && noise && mlp_params && cur_out && count_out && soft && nodes_snap &&
              adj_row && flags && B > 0);
  if (!gcm_learned_step_supported(N, F, 1, 1) || (N & 3) || (F & 3)) return GCM_EUNSUPPORTED;
  constexpr size_t lds = gcm_learned::lds_select();
  auto kern = gcm_learned::k_learned_select<2, 0>;
  gcm_allow_dynamic_lds((const void*)kern, lds);
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, (hipStream_t)stream, (const float*)nullptr, adj,
                     (const int64_t*)nullptr, noise, noise_is_exp, mlp_params, eps0, eps1, cutoff, soft, N, F, obs,
                     (const float*)nodes, (const float*)adj, count_in, nodes, cur_out, count_out, flags, nodes_snap,
                     adj_row, gcm_learned::GnnTail{}, -1);
  return gcm_launch_status();
}

/* One whole forward step of a chain that started from EMPTY graphs and has not overflowed (fewer than N steps so
 * far), on a DONATED state: gcm_learned_advance_select_inplace with the step's GNN behind the selection (see
 * GnnTail) - one launch.  cache_h1 [B,N,H1], cache_agg1 [B,N,F], cache_nodes [B,N,F]: the chain's caches (row cur is
 * written; rows < cur were written by the earlier steps of the chain; any contents at the chain's head - a row is
 * read only behind the step that wrote it).  cur_host >= 0: the row every graph's new node lands in, when the host knows it (the chain's step count: no
 * load in front of the kernel's addresses; count_in is compared at the end - GCM_FLAG_BAD_COUNT); -1: read count_in.  The step's record (gcm_learned_step_layout, compact = 2): adj_row [B,N], mx [B,H2], agg2 [B,H1], cur /
 * count, soft [B,N]. */
extern "C" int gcm_learned_step_cached(const float* obs, float* nodes, float* adj, const int64_t* count_in,
                                       const float* noise, int noise_is_exp, const float* params, int has_bias,
                                       int act1, int act2, float eps0, float eps1, float cutoff, int64_t* cur_out,
                                       int64_t* count_out, float* soft, float* adj_row, float* mx, float* agg2,
                                       float* cache_h1, float* cache_agg1, float* cache_nodes, float* cache_u,
                                       uint32_t* flags, int B, int N, int F, int H1, int H2, int cur_host,
                                       gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes && adj && count_in && noise && params && cur_out && count_out && soft && adj_row && mx &&
              agg2 && cache_h1 && cache_agg1 && cache_nodes && cache_u && flags && B > 0);
  if (!gcm_learned_step_supported(N, F, H1, H2) || (N & 3) || (F & 3)) return GCM_EUNSUPPORTED;
  const size_t Pg = 2 * (size_t)H1 * F + H1 + 2 * (size_t)H2 * H1 + H2;
  constexpr size_t lds = gcm_learned::lds_select_tail();
  const bool exact = N == gcm_learned::NP && F == gcm_learned::FP && H1 == gcm_learned::FP && H2 == gcm_learned::FP;
  if (exact && cur_host >= 0 && cur_host < N && !(has_bias & GCM_STEP_FOUR_WAVES)) {   // eight waves per graph (round 6)
    gcm_learned::GnnTail gt8{params, act1, act2, has_bias, H1, H2, cache_h1, cache_agg1, cache_nodes, cache_u, mx, agg2, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(gcm_learned::k_learned_select8, dim3(B), dim3(576), 0, (hipStream_t)stream, obs, nodes, adj,
                       count_in, count_out, cur_out, noise, noise_is_exp, params + Pg, eps0, eps1, cutoff, soft, adj_row,
                       flags, gt8, cur_host);
    return gcm_launch_status();
  }
  auto kern = exact ? gcm_learned::k_learned_select<2, 1, true> : gcm_learned::k_learned_select<2, 1, false>;
  gcm_allow_dynamic_lds((const void*)kern, lds);
  gcm_learned::GnnTail gt{params, act1, act2, has_bias, H1, H2, cache_h1, cache_agg1, cache_nodes, cache_u, mx, agg2, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, (hipStream_t)stream, (const float*)nullptr, adj,
                     (const int64_t*)nullptr, noise, noise_is_exp, params + Pg, eps0, eps1, cutoff, soft, N, F, obs,
                     (const float*)nodes, (const float*)adj, count_in, nodes, cur_out, count_out, flags,
                     (float*)nullptr, adj_row, gt, cur_host >= 0 ? cur_host : -1);
  return gcm_launch_status();
}

namespace gcm_learned {
// the adjacency [B,N,N] as bits [B][N][4]: bit j & 31 of word j >> 5 of row r = (adj[r][j] != 0); a thread a word
__global__ __launch_bounds__(256) void k_adj_bits(const float* __restrict__ adj, uint32_t* __restrict__ bits, int N,
                                                  size_t n_words) {
  const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (w >= n_words) return;
  const size_t row = w >> 2;
  const int c0 = (int)(w & 3) * 32;
  const float* src = adj + row * N + c0;
  uint32_t u = 0u;
  if (c0 + 32 <= N) {
    f32x4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = reinterpret_cast<const f32x4*>(src)[q];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k) u |= (v[q][k] != 0.f ? 1u : 0u) << (4 * q + k);
  } else {
    for (int k = 0; c0 + k < N; ++k) u |= (src[k] != 0.f ? 1u : 0u) << k;
  }
  bits[w] = u;
}
}  // namespace gcm_learned

/* The bit image a steady-state chain keeps of its adjacency (gcm_learned_step_steady): adj [B,N,N] -> bits [B][N][4]
 * u32, bit (j & 31) of word (j >> 5) of row r set where adj[b][r][j] != 0.  N <= 128, N % 4 == 0.  Run once when a
 * chain enters the steady state; every steady step then carries the image along. */
extern "C" int gcm_adj_bits(const float* adj, uint32_t* bits, int B, int N, gcm_stream_t stream) {
  GCM_REQUIRE(adj && bits && B > 0);
  if (N <= 0 || N > 128 || (N & 3)) return GCM_EUNSUPPORTED;
  const size_t n_words = (size_t)B * N * 4;
  hipLaunchKernelGGL(gcm_learned::k_adj_bits, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     adj, bits, N, n_words);
  return gcm_launch_status();
}

/* The whole forward step of such a chain PAST graph_size steps (round 5) - the steady state: every graph holds N nodes
 * (count_in[b] == N: the caller's guarantee - a chain from empty graphs that has made >= N steps; GCM_FLAG_BAD_COUNT
 * otherwise), each step drops the oldest one (gcm.py:263-271, 323-355).  gcm_learned_advance_select_inplace with the
 * GNN behind the selection in the same launch (k_learned_select<2, 2>): layer 1 of every row re-evaluated from the
 * node image staged for the edge network and a bit image of the advanced adjacency, row cur's layer 2.  Writes what
 * gcm_learned_advance_select_inplace + gcm_dense_gnn2_row_fwd write into the step's record
 * (gcm_learned_step_layout, compact = 1: nodes_snap, adj_row, mx, h1, agg1, agg2, cur, soft), which gcm_learned_bptt
 * reads unchanged.  N % 4 == 0, F % 4 == 0, F, H1, H2 <= 32. */
extern "C" int gcm_learned_step_steady(const float* obs, float* nodes, float* adj, const int64_t* count_in,
                                       const float* noise, int noise_is_exp, const float* params, int has_bias,
                                       int act1, int act2, float eps0, float eps1, float cutoff, int64_t* cur_out,
                                       int64_t* count_out, float* soft, float* nodes_snap, float* adj_row, float* mx,
                                       float* h1, float* agg1, float* agg2, const float* h1_prev,
                                       const float* agg1_prev, uint32_t* adj_bits, uint32_t* flags, int B, int N, int F,
                                       int H1, int H2, gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes && adj && count_in && noise && params && cur_out && count_out && soft && nodes_snap &&
              adj_row && mx && h1 && agg1 && agg2 && h1_prev && agg1_prev && adj_bits && flags && B > 0);
  if (!gcm_learned_step_supported(N, F, H1, H2) || (N & 3) || (F & 3)) return GCM_EUNSUPPORTED;
  const size_t Pg = 2 * (size_t)H1 * F + H1 + 2 * (size_t)H2 * H1 + H2;
  constexpr size_t lds = gcm_learned::lds_select_steady();
  const bool exact = N == gcm_learned::NP && F == gcm_learned::FP && H1 == gcm_learned::FP && H2 == gcm_learned::FP;
  auto kern = exact ? gcm_learned::k_learned_select<2, 2, true> : gcm_learned::k_learned_select<2, 2, false>;
  gcm_allow_dynamic_lds((const void*)kern, lds);
  gcm_learned::GnnTail gt{params, act1, act2, has_bias, H1, H2, nullptr, nullptr, nullptr, nullptr, mx, agg2, h1, agg1, h1_prev, agg1_prev, adj_bits};
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, (hipStream_t)stream, (const float*)nullptr, adj,
                     (const int64_t*)nullptr, noise, noise_is_exp, params + Pg, eps0, eps1, cutoff, soft, N, F, obs,
                     (const float*)nodes, (const float*)adj, count_in, nodes, cur_out, count_out, flags, nodes_snap,
                     adj_row, gt, -1);
  return gcm_launch_status();
}

/* gcm_learned_step_cached on a FUNCTIONAL state: the new state goes to nodes_out / adj_out (the `nodes` / `adj`
 * sections of the record, gcm_learned_step_layout compact = 3), the inputs are left alone. */
extern "C" int gcm_learned_step_cached_functional(
    const float* obs, const float* nodes_in, const float* adj_in, const int64_t* count_in, const float* noise,
    int noise_is_exp, const float* params, int has_bias, int act1, int act2, float eps0, float eps1, float cutoff,
    float* nodes_out, float* adj_out, int64_t* cur_out, int64_t* count_out, float* soft, float* mx, float* agg2,
    float* cache_h1, float* cache_agg1, float* cache_nodes, float* cache_u, uint32_t* flags, int B, int N, int F,
    int H1, int H2, int cur_host, gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes_in && adj_in && count_in && noise && params && nodes_out && adj_out && cur_out &&
              count_out && soft && mx && agg2 && cache_h1 && cache_agg1 && cache_nodes && cache_u && flags && B > 0);
  GCM_REQUIRE(nodes_out != nodes_in && adj_out != adj_in);
  if (!gcm_learned_step_supported(N, F, H1, H2) || (N & 3) || (F & 3)) return GCM_EUNSUPPORTED;
  const size_t Pg = 2 * (size_t)H1 * F + H1 + 2 * (size_t)H2 * H1 + H2;
  constexpr size_t lds = gcm_learned::lds_select_tail();
  const bool exact = N == gcm_learned::NP && F == gcm_learned::FP && H1 == gcm_learned::FP && H2 == gcm_learned::FP;
  auto kern = exact ? gcm_learned::k_learned_select<1, 1, true> : gcm_learned::k_learned_select<1, 1, false>;
  gcm_allow_dynamic_lds((const void*)kern, lds);
  gcm_learned::GnnTail gt{params, act1, act2, has_bias, H1, H2, cache_h1, cache_agg1, cache_nodes, cache_u, mx, agg2, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, (hipStream_t)stream, (const float*)nullptr, adj_out,
                     (const int64_t*)nullptr, noise, noise_is_exp, params + Pg, eps0, eps1, cutoff, soft, N, F, obs,
                     nodes_in, adj_in, count_in, nodes_out, cur_out, count_out, flags, (float*)nullptr,
                     (float*)nullptr, gt, cur_host >= 0 ? cur_host : -1);
  return gcm_launch_status();
}

/* DenseGCM.rollout with LearnedEdge, the whole forward of T <= N steps from EMPTY graphs in three launches (see
 * k_learned_roll_logits).  obs [T,B,F], noise [T,B,N]; nodes [B,N,F] / adj [B,N,N]: the state AFTER the rollout, both
 * ZERO on entry (rows >= T stay zero), count [B] <- T; records: T step records of gcm_learned_step_layout(compact = 2)
 * at `rec_stride` floats from each other (>= that layout's total); caches [B,N,.] (rows < T written); mx_all [T,B,H2].
 * What gcm_learned_bptt_cached (n_cached = T, cached_layout = 2) reads. */
extern "C" int gcm_learned_rollout_fwd(const float* obs, const float* noise, int noise_is_exp, const float* params,
                                       int has_bias, int act1, int act2, float eps0, float eps1, float cutoff,
                                       float* nodes, float* adj, int64_t* count, float* records, size_t rec_stride,
                                       float* cache_h1, float* cache_agg1, float* cache_nodes, float* mx_all,
                                       uint32_t* flags, int T, int B, int N, int F, int H1, int H2,
                                       gcm_stream_t stream) {
  GCM_REQUIRE(obs && noise && params && nodes && adj && count && records && cache_h1 && cache_agg1 && cache_nodes &&
              mx_all && flags && T > 0 && B > 0);
  if (!gcm_learned_step_supported(N, F, H1, H2) || T > N || B > 65535 || T > 65535) return GCM_EUNSUPPORTED;
  size_t lay[8];
  gcm_learned_step_layout(B, N, F, H1, H2, 2, lay);
  GCM_REQUIRE(rec_stride >= lay[0]);
  const size_t Pg = 2 * (size_t)H1 * F + H1 + 2 * (size_t)H2 * H1 + H2;
  gcm_learned::RollRec R{records, rec_stride, lay[1], lay[2], lay[5], lay[6], lay[7]};
  constexpr size_t lds = gcm_learned::lds_roll_logits();
  static_assert(lds <= 160 * 1024, "one 16-wave workgroup per CU");
  gcm_allow_dynamic_lds((const void*)gcm_learned::k_learned_roll_logits, lds);
  const long items = (long)T * B;
  {
    const long wgs = (items + gcm_learned::RL_WAVES - 1) / gcm_learned::RL_WAVES;
    hipLaunchKernelGGL(gcm_learned::k_learned_roll_logits, dim3((unsigned)(wgs < 256 ? wgs : 256)),
                       dim3(64 * gcm_learned::RL_WAVES), lds, (hipStream_t)stream, obs, params + Pg, eps0, eps1, R, B, T,
                       N, F);
  }
  int rc = gcm_launch_status();
  if (rc) return rc;
  hipLaunchKernelGGL(gcm_learned::k_learned_roll_pick, dim3((unsigned)((items + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, obs, noise, noise_is_exp, cutoff, params, act1, has_bias, H1, nodes, adj,
                     count, R, cache_h1, cache_agg1, cache_nodes, B, T, N, F);
  rc = gcm_launch_status();
  if (rc) return rc;
  hipLaunchKernelGGL(gcm_learned::k_learned_roll_l2, dim3((unsigned)((items + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, params, act2, has_bias, H1, H2, R, cache_h1, mx_all, flags, B, T, N, F);
  return gcm_launch_status();
}

extern "C" int gcm_learned_step_bwd(const float* g_mx, const float* nodes, const float* adj,
                                    const int64_t* cur_idx, const int64_t* count_in,
                                    const float* gnn_params, int act1, int act2, const float* mx,
                                    const float* h1, const float* agg1, const float* agg2,
                                    const float* soft, const float* mlp_params, float eps0, float eps1,
                                    float* GA, float* slabs, int accumulate, int B, int N, int F, int H1,
                                    int H2, gcm_stream_t stream) {
  GCM_REQUIRE(g_mx && nodes && adj && cur_idx && count_in && gnn_params && mx && h1 && agg1 && agg2 && soft &&
              mlp_params && GA && slabs && B > 0);
  if (!gcm_learned_step_supported(N, F, H1, H2)) return GCM_EUNSUPPORTED;
  constexpr size_t lds = gcm_learned::lds_bwd();
  static_assert(lds <= 160 * 1024, "LDS budget");
  gcm_allow_dynamic_lds((const void*)gcm_learned::k_learned_step_bwd, lds);
  hipLaunchKernelGGL(gcm_learned::k_learned_step_bwd, dim3(B), dim3(256), lds, (hipStream_t)stream, g_mx,
                     nodes, adj, cur_idx, count_in, gnn_params, act1, act2, mx, h1, agg1, agg2, soft,
                     mlp_params, eps0, eps1, GA, slabs, accumulate, N, F, H1, H2);
  return gcm_launch_status();
}

/* ---- time-parallel backward of a chain of fused LearnedEdge steps (see k_learned_bptt_sel / _mlp) ------ */
static inline size_t lrn_pad64(size_t n) { return (n + 63) & ~(size_t)63; }

extern "C" int gcm_learned_step_layout(int B, int N, int F, int H1, int H2, int compact, size_t* out8) {
  GCM_REQUIRE(out8 && B > 0 && N > 0 && F > 0 && H1 > 0 && H2 > 0);
  // nodes | adj | mx | h1 | agg1 | agg2 | cur, count_out (2 B int64) | soft      (64-float aligned sections)
  // compact (donated state): `adj` holds row cur only, [B, N]
  // compact = 2 (cached step on a donated state, gcm_learned_step_cached): no nodes / h1 / agg1 sections at all -
  // they live in the chain's caches
  // compact = 3: a cached step on a FUNCTIONAL state (the record holds the new state: nodes | full adj)
  const bool cached = compact >= 2, row_only = compact == 1 || compact == 2;
  const size_t n_nodes = compact == 2 ? 0 : lrn_pad64((size_t)B * N * F),
               n_adj = lrn_pad64((size_t)B * N * (row_only ? 1 : N));
  const size_t n_mx = lrn_pad64((size_t)B * H2), n_h1 = cached ? 0 : lrn_pad64((size_t)B * N * H1),
               n_agg1 = cached ? 0 : lrn_pad64((size_t)B * N * F), n_agg2 = lrn_pad64((size_t)B * H1);
  out8[1] = n_nodes;                 // o_adj
  out8[2] = out8[1] + n_adj;         // o_mx
  out8[3] = out8[2] + n_mx;          // o_h1
  out8[4] = out8[3] + n_h1;          // o_agg1
  out8[5] = out8[4] + n_agg1;        // o_agg2
  out8[6] = out8[5] + n_agg2;        // o_idx
  out8[7] = out8[6] + lrn_pad64(4 * (size_t)B);   // o_soft
  out8[0] = out8[7] + lrn_pad64((size_t)B * N);   // total floats
  return GCM_OK;
}

// pass-A workgroups (= slabs) per launch: what gcm_dense_rows_bptt_slabs gives a full chunk
static inline int lrn_per_a(int n_steps, int B) {
  return gcm_dense_rows_bptt_slabs(GCM_ROWS_MAX_STEPS < n_steps ? GCM_ROWS_MAX_STEPS : n_steps, B);
}

extern "C" size_t gcm_learned_bptt_workspace_bytes(int n_steps, int B, int N, int F, int H1, int H2) {
  if (n_steps <= 0 || B <= 0) return 0;
  const size_t Pg = 2 * (size_t)H1 * F + H1 + 2 * (size_t)H2 * H1 + H2, Pm = 3 * (size_t)F * F + 7 * F + 1;
  // (one launch more than the step count needs: a chain's cached prefix and the steps behind it do not share one)
  const size_t chunks = (n_steps + GCM_ROWS_MAX_STEPS - 1) / GCM_ROWS_MAX_STEPS + 1;
  const size_t TB = (size_t)n_steps * B;
  size_t fl = lrn_pad64(Pg * (size_t)lrn_per_a(n_steps, B) * chunks) + lrn_pad64(Pm * 256 * chunks) +
              lrn_pad64(TB * 2) + lrn_pad64(TB * N) + lrn_pad64(TB * N * F) + lrn_pad64(TB * H1) +
              lrn_pad64(TB * N);   // ... | g_logit
  return sizeof(float) * fl;
}

/* saved_host: HOST array of the n_steps step buffers of ONE chain of hidden states, in step order
 * (gcm_learned_step_layout); gmx_host[t]: that step's g_mx [B, H2] (element strides as given) or NULL
 * (zero gradient: a step whose own gradient another call accounts for).  params: GNN | edge network,
 * packed.  g_params [Pg + Pm] = g_params_prev (NULL = 0) + the gradient. */
extern "C" int gcm_learned_bptt(const float* const* saved_host, const float* const* gmx_host, int n_steps,
                                long gmx_stride_b, long gmx_stride_h, const float* params, int act1, int act2,
                                float eps0, float eps1, int compact, const float* g_params_prev, float* g_params,
                                void* workspace, size_t workspace_bytes, int B, int N, int F, int H1, int H2,
                                gcm_stream_t stream) {
  return gcm_learned_bptt_cached(saved_host, gmx_host, n_steps, 0, 2, nullptr, nullptr, nullptr, nullptr, gmx_stride_b,
                                 gmx_stride_h, params, act1, act2, eps0, eps1, compact, g_params_prev, g_params,
                                 workspace, workspace_bytes, B, N, F, H1, H2, stream);
}

/* gcm_learned_bptt for a chain whose first n_cached steps are cached ones (gcm_learned_step_cached: records in the
 * compact = 2 layout, node matrix / h1 / agg1 of every node in the chain's caches); the steps behind them have
 * the `compact` layout. */
extern "C" int gcm_learned_bptt_cached(const float* const* saved_host, const float* const* gmx_host, int n_steps,
                                       int n_cached, int cached_layout, const float* cache_nodes, const float* cache_h1,
                                       const float* cache_agg1, const float* cache_u, long gmx_stride_b,
                                       long gmx_stride_h,
                                       const float* params, int act1, int act2, float eps0, float eps1, int compact,
                                       const float* g_params_prev, float* g_params, void* workspace,
                                       size_t workspace_bytes, int B, int N, int F, int H1, int H2,
                                       gcm_stream_t stream) {
  GCM_REQUIRE(saved_host && gmx_host && params && g_params && workspace && n_steps > 0 && B > 0);
  GCM_REQUIRE(n_cached >= 0 && n_cached <= n_steps && (n_cached == 0 || (cache_nodes && cache_h1 && cache_agg1)));
  const bool blocks_only = (cached_layout & GCM_BPTT_MLP_BLOCKS) != 0;   // (the A/B of pass B2: the 32-row-block kernel at every shape)
  cached_layout &= ~GCM_BPTT_MLP_BLOCKS;
  GCM_REQUIRE(cached_layout == 2 || cached_layout == 3);
  if (!gcm_learned_step_supported(N, F, H1, H2)) return GCM_EUNSUPPORTED;
  if (workspace_bytes < gcm_learned_bptt_workspace_bytes(n_steps, B, N, F, H1, H2)) return GCM_EWORKSPACE;
  const size_t Pg = 2 * (size_t)H1 * F + H1 + 2 * (size_t)H2 * H1 + H2, Pm = 3 * (size_t)F * F + 7 * F + 1;
  // chunks of <= GCM_ROWS_MAX_STEPS steps of ONE kind (cached | not): {first step, count}
  std::vector<std::pair<int, int>> chunk_list;
  for (int lo = 0, hi = n_cached; lo < n_steps; lo = hi, hi = n_steps)
    for (int s0 = lo; s0 < hi; s0 += GCM_ROWS_MAX_STEPS)
      chunk_list.push_back({s0, hi - s0 < GCM_ROWS_MAX_STEPS ? hi - s0 : GCM_ROWS_MAX_STEPS});
  const int chunks = (int)chunk_list.size();
  const int per_a = lrn_per_a(n_steps, B);
  int total_a = per_a * chunks;
  const size_t TB = (size_t)n_steps * B;
  float* ws = (float*)workspace;
  float* slabs_a = ws;
  float* slabs_b = slabs_a + lrn_pad64(Pg * (size_t)total_a);
  int* hdr = (int*)(slabs_b + lrn_pad64(Pm * 256 * (size_t)chunks));
  int* live = hdr + lrn_pad64(TB * 2);
  float* da = (float*)(live + lrn_pad64(TB * N));
  float* dagg2 = da + lrn_pad64(TB * N * F);
  size_t lay_full[8], lay_c[8];
  gcm_learned_step_layout(B, N, F, H1, H2, compact, lay_full);
  gcm_learned_step_layout(B, N, F, H1, H2, cached_layout, lay_c);
  const float* w_rel2 = params + 2 * (size_t)H1 * F + H1;
  const float* w_root2 = w_rel2 + (size_t)H2 * H1;
  // pass A: every step, every graph (the arrays pass B scans must be complete before it starts)
  bool graph_passes = false;
  for (int c = 0; c < chunks; ++c) {
    const int s0 = chunk_list[c].first, ns = chunk_list[c].second;
    const bool cached = s0 < n_cached;
    const size_t* lay = cached ? lay_c : lay_full;
    gcm_rows::StepTable tab{};
    for (int i = 0; i < ns; ++i) {
      GCM_REQUIRE(saved_host[s0 + i]);
      tab.saved[i] = saved_host[s0 + i];
      tab.gmx[i] = gmx_host[s0 + i];
    }
    gcm_rows::LearnedSrc src{lay[1], lay[2], lay[3], lay[4], lay[5], lay[6], params, hdr, live, da, dagg2, s0,
                             cached ? (cached_layout == 2 ? 1 : 0) : compact, cached ? cache_nodes : nullptr,
                             cached ? cache_h1 : nullptr,
                             cached ? cache_agg1 : nullptr};
    // every step of the backward a cached step of one chain (one chunk), the exact widths: pass A and pass B1 per GRAPH
    // (A hands G1 rows to B1, which applies W_rel1^T once per node; B slabs)
    graph_passes = cached && chunks == 1 && s0 == 0 && ns == n_steps && ns <= gcm_learned::SG_TMAX && cache_nodes && cache_h1 &&
                   cache_agg1 && F == gcm_learned::FP && H1 == gcm_learned::FP && H2 <= 32 && N == gcm_learned::NP &&
                   B <= per_a && cached_layout == 2 && !blocks_only;   // (layout 2: the donated steps' records - the adjacency row compact)
    int rc;
    if (graph_passes) {
      rc = gcm_rows::launch_bptt_learned_graph(stream, tab, ns, gmx_stride_b, gmx_stride_h, w_rel2, w_root2, act1, act2,
                                               slabs_a, src, B, H2);
      total_a = B;
    } else {
      rc = gcm_rows::launch_bptt_learned(stream, per_a, tab, ns, gmx_stride_b, gmx_stride_h, w_rel2, w_root2, act1, act2,
                                         slabs_a + (size_t)c * per_a * Pg, src, B, N, F, H1, H2);
    }
    if (rc) return rc;
  }
  // pass B: B1 (selection + softmax adjoint of every item -> g_logit), then B2 (edge network, per 32-row block)
  float* g_logit = dagg2 + lrn_pad64(TB * H1);
  constexpr size_t lds = gcm_learned::lds_bptt_mlp();
  static_assert(lds <= 160 * 1024, "one 8-wave workgroup per CU");
  gcm_allow_dynamic_lds((const void*)gcm_learned::k_learned_bptt_mlp, lds);
  gcm_allow_dynamic_lds((const void*)gcm_learned::k_learned_bptt_sel_graph, gcm_learned::lds_bptt_sel_graph());
  gcm_allow_dynamic_lds((const void*)gcm_learned::k_learned_bptt_mlp16<true>, gcm_learned::lds_bptt_mlp16());
  gcm_allow_dynamic_lds((const void*)gcm_learned::k_learned_bptt_mlp16<false>, gcm_learned::lds_bptt_mlp16());
  int total_b = 0;
  for (int pass = 0; pass < 2; ++pass)
    for (int c = 0; c < chunks; ++c) {
      const int s0 = chunk_list[c].first, ns = chunk_list[c].second;
      const bool cached = s0 < n_cached;
      const size_t* lay = cached ? lay_c : lay_full;
      gcm_learned::BpttB a{};
      for (int i = 0; i < ns; ++i) a.tab.saved[i] = saved_host[s0 + i];
      a.o_h1 = lay[3];
      a.o_soft = lay[7];
      a.c_nodes = cached ? cache_nodes : nullptr;
      a.c_h1 = cached ? cache_h1 : nullptr;
      a.c_u = cached && F == gcm_learned::FP ? cache_u : nullptr;
      a.hdr = hdr; a.live = live; a.da = da; a.dagg2 = dagg2;
      a.s0 = s0; a.n_steps = ns; a.T = n_steps;
      const bool per_graph = graph_passes || (cached && chunks == 1 && s0 == 0 && ns == n_steps && ns <= gcm_learned::SG_TMAX &&
                                              a.c_nodes && a.c_h1 && F == gcm_learned::FP && H1 == gcm_learned::FP &&
                                              N == gcm_learned::NP && !blocks_only);
      if (pass == 1 && per_graph) a.c0 = dagg2;   // (B1 wrote c0_t over dagg2_t)
      if (pass == 0) {
        // (every step of the backward a cached step of one chain, T <= 64, the exact widths: per graph)
        if (per_graph)
          hipLaunchKernelGGL(gcm_learned::k_learned_bptt_sel_graph, dim3(B), dim3(256), gcm_learned::lds_bptt_sel_graph(),
                             (hipStream_t)stream, a, g_logit, params + Pg, dagg2, graph_passes ? params : (const float*)nullptr, B);
        else
          hipLaunchKernelGGL(gcm_learned::k_learned_bptt_sel, dim3(ns * B), dim3(128), 0, (hipStream_t)stream, a,
                             g_logit, B, N, F, H1);
      } else {
        const long units8 = ((long)ns * B + gcm_learned::MLP_WAVES - 1) / gcm_learned::MLP_WAVES;
        const int grid = (int)(units8 < 256 ? units8 : 256);
        if (a.c_u && a.c_nodes && F == gcm_learned::FP && N == gcm_learned::NP && !blocks_only)
          hipLaunchKernelGGL(gcm_learned::k_learned_bptt_mlp16<true>, dim3(grid), dim3(64 * gcm_learned::M16_WAVES),
                             gcm_learned::lds_bptt_mlp16(), (hipStream_t)stream, a, (const float*)g_logit, params + Pg,
                             eps0, eps1, slabs_b + (size_t)total_b * Pm, B);
        else if (!cached && F == gcm_learned::FP && N == gcm_learned::NP && !blocks_only)
          hipLaunchKernelGGL(gcm_learned::k_learned_bptt_mlp16<false>, dim3(grid), dim3(64 * gcm_learned::M16_WAVES),
                             gcm_learned::lds_bptt_mlp16(), (hipStream_t)stream, a, (const float*)g_logit, params + Pg,
                             eps0, eps1, slabs_b + (size_t)total_b * Pm, B);
        else
          hipLaunchKernelGGL(gcm_learned::k_learned_bptt_mlp, dim3(grid), dim3(64 * gcm_learned::MLP_WAVES), lds,
                             (hipStream_t)stream, a, (const float*)g_logit, params + Pg, eps0, eps1,
                             slabs_b + (size_t)total_b * Pm, B, N, F);
        total_b += grid;
      }
      const int rc = gcm_launch_status();
      if (rc) return rc;
    }
  int rc = gcm_sum_slabs_acc(slabs_a, total_a, (int)Pg, g_params_prev, g_params, stream);
  if (rc) return rc;
  return gcm_sum_slabs_acc(slabs_b, total_b, (int)Pm, g_params_prev ? g_params_prev + Pg : nullptr, g_params + Pg,
                           stream);
}
