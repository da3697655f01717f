// Time-parallel BPTT of the fused DenseGCM step: the GNN adjoint of step t needs g_mx[t] only,
// so all T*B graph-steps ("items") of a rollout are independent.  A persistent grid walks them:
//
//   * the weights are staged in LDS once per workgroup, and the parameter gradients accumulate in
//     registers across items - one slab per WORKGROUP (a few hundred), not one per graph-step;
//   * only the row tiles that can carry gradient are touched.  G1 = dL/dh1 is non-zero only on rows
//     j with adj[cur][j] != 0 and on row cur (gcm.py:314 keeps one row of the last layer), so the
//     kernel reads row `cur` of the adjacency first and then loads just those 32-row tiles of
//     adj / h1 / agg1 / x.  For temporal graphs that is 1-2 tiles of 4;
//   * the adjacency goes HBM -> registers -> MFMA A operand directly (A(i,k) = adj[k][i] is a
//     coalesced 128-byte row segment per half wave): no LDS image, 45 KB of LDS per workgroup at
//     F = H = 32, three workgroups per CU.
//
// Outputs per item: Q = U_t(dX_t) [N,F] (state-advance adjoint applied, see k_gnodes_scan) and
// pobs = dX_t[cur].  EXACT shapes only; everything else uses k_gnn2_row_bwd over T*B.
#include "fused_common.h"

#ifdef GCM_STAMPS   // diagnostic build only (make stamps3): phase stamps of one item of workgroup 0
__device__ unsigned long long g_stamps[32];
extern "C" int gcm_debug_read_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
#define BSTAMP(i) do { if (item == GCM_STAMP_ITEM) STAMP(i); } while (0)
#else
#define BSTAMP(i)
#endif

namespace gcm_fused {

template <int NT, int NCT, int NHT, int N2T>
struct LdsBptt {
  using L = Lds<NT, NCT, NHT, N2T>;
  static constexpr int G = L::NP * L::HS, D = L::NP * L::FS;
  static constexpr int MISC = L::NP + 256 + 4 * L::HP + L::H2P;
  static constexpr int TOTAL = G + D + L::W1B + L::W2 + MISC;
  // waves per SIMD the LDS footprint allows (a workgroup puts one wave on each SIMD)
  static constexpr int WAVES = (TOTAL * 4 <= 80 * 1024 && NCT * NHT * N2T == 1) ? 2 : 1;
  static_assert(G + D + L::W1B >= 4096, "the end-of-kernel reduction buffer aliases sG | sD | sW1");
};

template <int NT, int NCT, int NHT, int N2T>
__global__ __launch_bounds__(256, (LdsBptt<NT, NCT, NHT, N2T>::WAVES)) void k_bptt_batched(
    const float* __restrict__ g_mx, const float* __restrict__ x, const float* __restrict__ adj,
    const int64_t* __restrict__ cur_idx, const int64_t* __restrict__ num_nodes_in, Gnn2 P,
    const float* __restrict__ mx, const float* __restrict__ h1, const float* __restrict__ agg1,
    const float* __restrict__ agg2, float* __restrict__ Q, float* __restrict__ pobs,
    float* __restrict__ slabs, int items) {
  using L = Lds<NT, NCT, NHT, N2T>;
  using LB = LdsBptt<NT, NCT, NHT, N2T>;
  constexpr int N = L::NP, F = L::FP, H1 = L::HP, H2 = L::H2P;
  constexpr int NP = N, FP = F, HP = H1, H2P = H2;
  constexpr int FS = L::FS, HS = L::HS, W2S = L::W2S;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;

  extern __shared__ float smem[];
  float* sG = smem;                     // [NP][HS]  G1 = dh1 * act1'(h1), live row tiles only
  float* sD = sG + LB::G;               // [NP][FS]  dAgg1, live row tiles only
  float* sW1 = sD + LB::D;              // w_rel1 [h][FS] | w_root1 [h][FS]
  float* sW2 = sW1 + L::W1B;            // [o][rel k | root k], stride W2S
  float* sRow = sW2 + L::W2;            // adj[cur][:]
  float* sV = sRow + NP;                // [256] partials
  float* sVv = sV + 256;                // v = agg2 | h1[cur]   [2*HP]
  float* sD2 = sVv + 2 * HP;            // d2                   [H2P]
  float* sU = sD2 + H2P;                // u = dagg2 | dh1cur   [2*HP]
  float* sR = smem;                     // end of kernel: [4][1024] cross-wave reduction (over sG..sW1)

  {  // weights: once per workgroup
    Stage<HP, FP, false, true> st_wr, st_wo;
    Stage<H2P, HP, false, true> st_w2r, st_w2o;
    st_wr.load(P.w_rel1, H1, F, F, tid);
    st_wo.load(P.w_root1, H1, F, F, tid);
    st_w2r.load(P.w_rel2, H2, H1, H1, tid);
    st_w2o.load(P.w_root2, H2, H1, H1, tid);
    st_wr.store(sW1, FS, tid);
    st_wo.store(sW1 + HP * FS, FS, tid);
    st_w2r.store(sW2, W2S, tid);
    st_w2o.store(sW2 + HP, W2S, tid);
  }
  // parameter-gradient accumulators, live across items
  f32x16 accW[2][NHT][NCT];
#pragma unroll
  for (int w = 0; w < 2; ++w)
#pragma unroll
    for (int a = 0; a < NHT; ++a)
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[w][a][c][r] = 0.f;
  constexpr int PER2 = H2P * 2 * HP / 256;
  float dw2[PER2];
#pragma unroll
  for (int i = 0; i < PER2; ++i) dw2[i] = 0.f;
  float db1 = 0.f, db2 = 0.f;
  __syncthreads();

  const int r_base = wave * 32;
  const bool wave_rows = wave < NT;   // this wave owns output rows [r_base, r_base + 32)

#pragma unroll 1
  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const float* xg = x + (size_t)item * N * F;
    const float* ag = adj + (size_t)item * N * N;
    const float* h1g = h1 + (size_t)item * N * H1;
    const float* a1g = agg1 + (size_t)item * N * F;
    float* gin = Q + (size_t)item * N * F;
    BSTAMP(0);
    int64_t cur64 = cur_idx[item];
    const int cur = cur64 < 0 ? 0 : (cur64 > N - 1 ? N - 1 : (int)cur64);
    const bool wrap = num_nodes_in[item] + 1 > N;

    // ---- phase 0: the kept row ---------------------------------------------------------------
    {
      const int o = tid < H2 ? tid : H2 - 1;
      const float gm = g_mx[(size_t)item * H2 + o];
      const float mv = mx[(size_t)item * H2 + o];
      const int k = tid < HP ? tid : (tid < 2 * HP ? tid - HP : 0);
      const float a2 = agg2[(size_t)item * H1 + k];
      const float hc = h1g[cur * H1 + k];
      const float ar = ag[cur * N + (tid < N ? tid : N - 1)];
      if (tid < H2P) sD2[tid] = gm * gcm_act_grad(mv, P.act2);
      if (tid < 2 * HP) sVv[tid] = tid < HP ? a2 : hc;
      if (tid < N) sRow[tid] = ar;
    }
    BSTAMP(1);
    __syncthreads();
    BSTAMP(2);
    {  // u[m] = sum_o W2c[o][m] * d2[o]
      constexpr int G = 256 / (2 * HP), OC = H2P / G;
      const int g = tid / (2 * HP), m = tid - g * (2 * HP);
      float s = 0.f;
#pragma unroll
      for (int o = g * OC; o < (g + 1) * OC; ++o) s = fmaf(sW2[o * W2S + m], sD2[o], s);
      sV[tid] = s;
    }
#pragma unroll
    for (int i = 0; i < PER2; ++i) {   // layer-2 parameter gradients d2[o] * v[k]
      const int e = tid + 256 * i, o = e / (2 * HP), k = e % (2 * HP);
      dw2[i] = fmaf(sD2[o], sVv[k], dw2[i]);
    }
    if (tid < H2) db2 += sD2[tid];
    unsigned live = 1u << (cur >> 5);   // bit t: row tile t can carry gradient (wave-uniform)
#pragma unroll
    for (int t = 0; t < NT; ++t) live |= (__any(sRow[t * 32 + li] != 0.f) ? 1u : 0u) << t;
    constexpr int PERG = 32 * HP / 256;
    BSTAMP(3);
    __syncthreads();
    if (tid < 2 * HP) {
      constexpr int G = 256 / (2 * HP);
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < G; ++q) t += sV[q * 2 * HP + tid];
      sU[tid] = t;
    }
    __syncthreads();
    BSTAMP(4);
    // ---- G1[j][h] = (adj[cur][j] * dagg2[h] + [j==cur] dh1cur[h]) * act1'(h1[j][h]) --------------
#pragma unroll 1
    for (int t = 0; t < NT; ++t)
      if ((live >> t) & 1u) {
        float hv[PERG];   // h1 of this live row tile
#pragma unroll
        for (int i = 0; i < PERG; ++i) hv[i] = h1g[t * 32 * H1 + tid + 256 * i];
#pragma unroll
        for (int i = 0; i < PERG; ++i) {
          const int e = tid + 256 * i, j = t * 32 + e / HP, h = e % HP;
          const float d = sRow[j] * sU[h] + (j == cur ? sU[HP + h] : 0.f);
          float v = d * gcm_act_grad(hv[i], P.act1);
          if (d == 0.f) v = 0.f;
          sG[j * HS + h] = v;
          db1 += v;   // 256 % HP == 0: a thread always sees the same h
        }
      }
    __syncthreads();
    BSTAMP(5);
    // ---- layer-1 parameter gradients: G1^T (H1 x live rows) @ {agg1, x}, jobs dealt to the 4 waves
#pragma unroll
    for (int which = 0; which < 2; ++which)
#pragma unroll
      for (int ht = 0; ht < NHT; ++ht)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const int jobidx = which + 2 * (ht * NCT + ct);
          const float* src = which ? xg : a1g;
#pragma unroll 1
          for (int t = 0; t < NT; ++t) {
            if (((live >> t) & 1u) && ((t + jobidx) & 3) == wave) {
              float bq[16];
#pragma unroll
              for (int s = 0; s < 16; ++s) bq[s] = src[(t * 32 + 2 * s + lh) * F + ct * 32 + li];
              const float* ap = sG + (t * 32 + lh) * HS + ht * 32 + li;   // A(i=h, k=row)
#pragma unroll
              for (int s = 0; s < 16; ++s)
                accW[which][ht][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                    ap[2 * s * HS], bq[s], accW[which][ht][ct], 0, 0, 0);
            }
          }
        }
    BSTAMP(6);
    // ---- dAgg1 = G1 @ W_rel1 -> LDS ;  acc = G1 @ W_root1 (root part of dX), live tiles only -----
    const bool my_rows_live = (live >> wave) & 1u;
    f32x16 acc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
      if (wave_rows && my_rows_live) {
        f32x16 d;
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = 0.f;
        mma32(d, sG + r_base * HS, HS, 1, sW1 + c * 32, FS, 1, HP, li, lh);
        mma32(acc[c], sG + r_base * HS, HS, 1, sW1 + HP * FS + c * 32, FS, 1, HP, li, lh);
#pragma unroll
        for (int r = 0; r < 16; ++r) sD[(r_base + acc_row(r, lh)) * FS + c * 32 + li] = d[r];
      }
    }
    __syncthreads();
    BSTAMP(7);
    // ---- dX[i] += sum_k adj[k][i] * dAgg1[k], k over the live row tiles --------------------------
    if (wave_rows) {
#pragma unroll 1
      for (int t = 0; t < NT; ++t) {
        if ((live >> t) & 1u) {
          // this wave's column strip of adjacency row tile t: the A operands, straight from HBM
          float av[16];
#pragma unroll
          for (int s = 0; s < 16; ++s) av[s] = ag[(t * 32 + lh + 2 * s) * N + r_base + li];
          bool nz = false;
#pragma unroll
          for (int s = 0; s < 16; ++s) nz |= av[s] != 0.f;
          if (__any(nz)) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
              const float* bp = sD + (t * 32 + lh) * FS + c * 32 + li;
#pragma unroll
              for (int s = 0; s < 16; ++s)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bp[2 * s * FS], acc[c], 0, 0, 0);
            }
          }
        }
      }
      BSTAMP(8);
      // epilogue: undo insert + roll (gcm.py:262-278)
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r_base + acc_row(r, lh), col = c * 32 + li;
          const float v = acc[c][r];
          if (row == cur) {
            pobs[(size_t)item * F + col] = v;
            if (!wrap) gin[row * F + col] = 0.f;
          } else if (!wrap) {
            gin[row * F + col] = v;
          } else {
            gin[(row + 1) * F + col] = v;
          }
        }
    }
    if (wrap && tid < F) gin[tid] = 0.f;
    BSTAMP(9);
    __syncthreads();
    BSTAMP(10);   // sG / sD / sRow / sV are rewritten by the next item
  }

  // ---- one slab per workgroup ------------------------------------------------------------------
  float* slab = slabs + (size_t)blockIdx.x * (2 * (size_t)H1 * F + H1 + 2 * (size_t)H2 * H1 + H2);
  float* sl_rel1 = slab;
  float* sl_root1 = sl_rel1 + H1 * F;
  float* sl_b1 = sl_root1 + H1 * F;
  float* sl_rel2 = sl_b1 + H1;
  float* sl_root2 = sl_rel2 + H2 * H1;
  float* sl_b2 = sl_root2 + H2 * H1;
#pragma unroll
  for (int which = 0; which < 2; ++which)
#pragma unroll
    for (int ht = 0; ht < NHT; ++ht)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sR[wave * 1024 + acc_row(r, lh) * 32 + li] = accW[which][ht][ct][r];
        __syncthreads();
        float* dst = which ? sl_root1 : sl_rel1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = tid + 256 * i, hh = ht * 32 + (e >> 5), ff = ct * 32 + (e & 31);
          dst[hh * F + ff] = (sR[e] + sR[1024 + e]) + (sR[2048 + e] + sR[3072 + e]);
        }
        __syncthreads();
      }
#pragma unroll
  for (int i = 0; i < PER2; ++i) {
    const int e = tid + 256 * i, o = e / (2 * HP), k = e % (2 * HP);
    (k < HP ? sl_rel2 : sl_root2)[o * H1 + (k < HP ? k : k - HP)] = dw2[i];
  }
  if (tid < H2) sl_b2[tid] = db2;
  sV[tid] = db1;
  __syncthreads();
  if (tid < H1) {
    constexpr int G = 256 / HP;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < G; ++q) t += sV[q * HP + tid];
    sl_b1[tid] = t;
  }
}

template <int NT, int NCT, int NHT, int N2T>
int launch_bptt(hipStream_t s, int grid, const float* g_mx, const float* x, const float* adj,
                const int64_t* cur, const int64_t* nn_in, Gnn2 P, const float* mx, const float* h1,
                const float* agg1, const float* agg2, float* Q, float* pobs, float* slabs,
                int items) {
  constexpr size_t lds = sizeof(float) * (size_t)LdsBptt<NT, NCT, NHT, N2T>::TOTAL;
  if (lds > 160 * 1024) return GCM_EUNSUPPORTED;
  auto kern = k_bptt_batched<NT, NCT, NHT, N2T>;
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, g_mx, x, adj, cur, nn_in, P, mx, h1, agg1,
                     agg2, Q, pobs, slabs, items);
  return gcm_launch_status();
}

}  // namespace gcm_fused

#define GCM_BSHAPES_N(X, a) \
  X(a, 1, 1, 1) X(a, 1, 1, 2) X(a, 1, 2, 1) X(a, 1, 2, 2) X(a, 2, 1, 1) X(a, 2, 1, 2) X(a, 2, 2, 1) X(a, 2, 2, 2)
#define GCM_BSHAPES(X) GCM_BSHAPES_N(X, 1) GCM_BSHAPES_N(X, 2) GCM_BSHAPES_N(X, 3) GCM_BSHAPES_N(X, 4)

extern "C" int gcm_dense_bptt_batched_slabs(int items) {
  // persistent grid: enough workgroups for 3 per CU; one parameter-gradient slab each
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
      return 0;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int g = 3 * cus;
  return items < g ? items : g;
}

extern "C" int gcm_dense_bptt_batched(const float* g_mx, const float* x, const float* adj,
                                      const int64_t* cur_idx, const int64_t* num_nodes_in,
                                      const float* w_rel1, const float* b_rel1,
                                      const float* w_root1, int act1, const float* w_rel2,
                                      const float* b_rel2, const float* w_root2, int act2,
                                      const float* mx, const float* h1, const float* agg1,
                                      const float* agg2, float* Q, float* pobs, float* slabs,
                                      int n_slabs, int items, int N, int F, int H1, int H2,
                                      gcm_stream_t stream) {
  GCM_REQUIRE(g_mx && x && adj && cur_idx && num_nodes_in && w_rel1 && w_root1 && w_rel2 &&
              w_root2 && mx && h1 && agg1 && agg2 && Q && pobs && slabs);
  GCM_REQUIRE(items > 0 && n_slabs > 0 && n_slabs <= items);
  if ((N & 31) || (F & 31) || (H1 & 31) || (H2 & 31) || N > 128 || F > 64 || H1 > 64 || H2 > 64)
    return GCM_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  gcm_fused::Gnn2 P{w_rel1, b_rel1, w_root1, w_rel2, b_rel2, w_root2, act1, act2};
  const int NT = N / 32, NCT = F / 32, NHT = H1 / 32, N2T = H2 / 32;
#define GCM_B(a, b_, c, d)                                                                      \
  if (NT == a && NCT == b_ && NHT == c && N2T == d)                                             \
    return gcm_fused::launch_bptt<a, b_, c, d>(s, n_slabs, g_mx, x, adj, cur_idx, num_nodes_in, \
                                               P, mx, h1, agg1, agg2, Q, pobs, slabs, items);
  GCM_BSHAPES(GCM_B)
#undef GCM_B
  return GCM_EUNSUPPORTED;
}
