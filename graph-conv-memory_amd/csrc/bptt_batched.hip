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
  static constexpr int TOTAL = G + 2 * D + L::W1B + L::W2 + MISC;
  // waves per SIMD the LDS footprint allows (a workgroup puts one wave on each SIMD)
  static constexpr int WAVES = (TOTAL * 4 <= 80 * 1024 && NCT * NHT * N2T == 1) ? 2 : 1;
};

template <int NT, int NCT, int NHT, int N2T>
__global__ __launch_bounds__(256, (LdsBptt<NT, NCT, NHT, N2T>::WAVES)) void k_bptt_batched(
    const float* __restrict__ g_mx, const float* __restrict__ g_nodes_out,
    const float* __restrict__ x, const float* __restrict__ adj,
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
  const int m16 = lane & 15, kq = lane >> 4;   // 16x16x4 MFMA lane coordinates

  extern __shared__ float smem[];
  float* sG = smem;                     // [NP][HS]  G1 = dh1 * act1'(h1), live row tiles only
  float* sD = sG + LB::G;               // [NP][FS]  dAgg1 = G1 @ W_rel1, live row tiles only
  float* sRt = sD + LB::D;              // [NP][FS]  G1 @ W_root1 (root part of dX), live tiles only
  float* sW1 = sRt + LB::D;             // w_rel1 [h][FS] | w_root1 [h][FS]
  float* sW2 = sW1 + L::W1B;            // [o][rel k | root k], stride W2S
  float* sRow = sW2 + L::W2;            // adj[cur][:]
  float* sV = sRow + NP;                // [256] partials
  float* sVv = sV + 256;                // v = agg2 | h1[cur]   [2*HP]
  float* sD2 = sVv + 2 * HP;            // d2                   [H2P]
  float* sU = sD2 + H2P;                // u = dagg2 | dh1cur   [2*HP]

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
  // parameter-gradient accumulators, live across items.  Layer 1: [H1 x F] as 16x16 blocks, block
  // id = wave + 4*bi -> (h0, c0); one accumulator per block and operand (agg1 / x).
  constexpr int NBW = NHT * NCT;        // blocks per wave
  constexpr int CB = FP / 16;           // column blocks
  f32x4 accW[2][NBW];
#pragma unroll
  for (int w = 0; w < 2; ++w)
#pragma unroll
    for (int bi = 0; bi < NBW; ++bi) accW[w][bi] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int PER2 = H2P * 2 * HP / 256;
  float dw2[PER2];
#pragma unroll
  for (int i = 0; i < PER2; ++i) dw2[i] = 0.f;
  float db1 = 0.f, db2 = 0.f;
  const int act1_v = gcm_vgpr(P.act1), act2_v = gcm_vgpr(P.act2);
  __syncthreads();

  const int r_base = wave * 32;
  const bool wave_rows = wave < NT;   // this wave owns output rows [r_base, r_base + 32) of dX

  // cur / wrap of the next item are fetched one item ahead (the row addresses depend on them)
  int curN;
  bool wrapN;
  {
    int64_t c64 = cur_idx[blockIdx.x];
    curN = c64 < 0 ? 0 : (c64 > N - 1 ? N - 1 : (int)c64);
    wrapN = num_nodes_in[blockIdx.x] + 1 > N;
  }

#pragma unroll 1
  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    BSTAMP(0);
    const float* xg = x + (size_t)item * N * F;
    const float* ag = adj + (size_t)item * N * N;
    const float* h1g = h1 + (size_t)item * N * H1;
    const float* a1g = agg1 + (size_t)item * N * F;
    float* gin = Q + (size_t)item * N * F;
    const int cur = curN;
    const bool wrap = wrapN;
    // ---- phase 0: the kept row ---------------------------------------------------------------
    {
      const int o = tid < H2 ? tid : H2 - 1;
      const float gm = g_mx[(size_t)item * H2 + o];
      const float mv = mx[(size_t)item * H2 + o];
      const int k = tid < HP ? tid : (tid < 2 * HP ? tid - HP : 0);
      const float a2 = agg2[(size_t)item * H1 + k];
      const float hc = h1g[cur * H1 + k];
      const float ar = ag[cur * N + (tid < N ? tid : N - 1)];
      {   // next item's cur (scalar path), in flight for the whole item
        const int nxt = item + gridDim.x, ic = nxt < items ? nxt : items - 1;
        int64_t c64 = cur_idx[ic];
        curN = c64 < 0 ? 0 : (c64 > N - 1 ? N - 1 : (int)c64);
        wrapN = num_nodes_in[ic] + 1 > N;
      }
      const float ga = act2_v == GCM_ACT_TANH ? 1.f - mv * mv : (act2_v == GCM_ACT_RELU ? (mv > 0.f ? 1.f : 0.f) : 1.f);
      if (tid < H2P) sD2[tid] = gm * ga;
      if (tid < 2 * HP) sVv[tid] = tid < HP ? a2 : hc;
      if (tid < N) sRow[tid] = ar;
    }
    BSTAMP(1);
    __syncthreads();
    BSTAMP(2);
    unsigned live = 1u << (cur >> 5);   // bit t: row tile t can carry gradient (wave-uniform)
#pragma unroll
    for (int t = 0; t < NT; ++t) live |= (__any(sRow[t * 32 + li] != 0.f) ? 1u : 0u) << t;
    live = __builtin_amdgcn_readfirstlane(live);
    // this wave's column strip of the FIRST live adjacency row tile (the A operands of dX; usually
    // the only one): in flight from here on, further live tiles are fetched when they are used
    const int t0 = __builtin_ctz(live);
    float av0[16];
    if (wave_rows) {
#pragma unroll
      for (int s = 0; s < 16; ++s) av0[s] = ag[(t0 * 32 + lh + 2 * s) * N + r_base + li];
    }
    // h1 of the live row tiles: loads in flight under the layer-2 arithmetic
    constexpr int PERG = 32 * HP / 256;
    float hv[NT][PERG];
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if ((live >> t) & 1u) {
#pragma unroll
        for (int i = 0; i < PERG; ++i) hv[t][i] = h1g[t * 32 * H1 + tid + 256 * i];
      }
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
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if ((live >> t) & 1u) {
#pragma unroll
        for (int i = 0; i < PERG; ++i) {
          const int e = tid + 256 * i, j = t * 32 + e / HP, h = e % HP;
          const float d = sRow[j] * sU[h] + (j == cur ? sU[HP + h] : 0.f);
          const float y = hv[t][i];
          const float ga = act1_v == GCM_ACT_TANH ? 1.f - y * y : (act1_v == GCM_ACT_RELU ? (y > 0.f ? 1.f : 0.f) : 1.f);
          float v = d * ga;
          if (d == 0.f) v = 0.f;
          sG[j * HS + h] = v;
          db1 += v;   // 256 % HP == 0: a thread always sees the same h
        }
      }
    BSTAMP(5);
    __syncthreads();
    BSTAMP(6);
    // ---- layer-1 parameter gradients: dW[h][f] += sum_rows G1[row][h] * {agg1, x}[row][f], 16x16
    // blocks shared by the four waves; B operands straight from HBM (each element used once)
#pragma unroll 1
    for (int t = 0; t < NT; ++t)
      if ((live >> t) & 1u) {
#pragma unroll
        for (int bi = 0; bi < NBW; ++bi) {
          const int blk = wave + 4 * bi, h0 = (blk / CB) * 16, c0 = (blk % CB) * 16;
          float ga[8], b0[8], b1[8];
          const float* ap = sG + (t * 32 + kq) * HS + h0 + m16;        // A(i=h, k=row)
          const float* p0 = a1g + (t * 32 + kq) * F + c0 + m16;        // B(k=row, j=f)
          const float* p1 = xg + (t * 32 + kq) * F + c0 + m16;
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            ga[s] = ap[4 * s * HS];
            b0[s] = p0[4 * s * F];
            b1[s] = p1[4 * s * F];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            accW[0][bi] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], b0[s], accW[0][bi], 0, 0, 0);
            accW[1][bi] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[s], b1[s], accW[1][bi], 0, 0, 0);
          }
        }
      }
    BSTAMP(7);
    // ---- dAgg1 = G1 @ W_rel1, root = G1 @ W_root1 for the live tiles: [32 x F] as 16x16 blocks ----
#pragma unroll 1
    for (int t = 0; t < NT; ++t)
      if ((live >> t) & 1u) {
#pragma unroll
        for (int bi = 0; bi < NCT; ++bi) {
          const int blk = wave + 4 * bi, r0 = t * 32 + (blk & 1) * 16, c0 = (blk >> 1) * 16;
          f32x4 d = {0.f, 0.f, 0.f, 0.f}, rt = {0.f, 0.f, 0.f, 0.f};
          mma16<HP>(d, sG + r0 * HS, HS, sW1 + c0, FS, m16, kq);
          mma16<HP>(rt, sG + r0 * HS, HS, sW1 + HP * FS + c0, FS, m16, kq);
          float* pd = sD + (r0 + 4 * kq) * FS + c0 + m16;
          float* pr = sRt + (r0 + 4 * kq) * FS + c0 + m16;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pd[r * FS] = d[r];
            pr[r * FS] = rt[r];
          }
        }
      }
    BSTAMP(8);
    __syncthreads();
    BSTAMP(9);
    // ---- dX[i] = sum_k adj[k][i] * dAgg1[k] (k over the live row tiles) + root[i] -----------------
    if (wave_rows) {
      f32x16 acc[NCT];
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll 1
      for (int t = 0; t < NT; ++t) {
        if ((live >> t) & 1u) {
          // this wave's column strip of adjacency row tile t: the A operands, HBM -> registers
          float av[16];
          if (t == t0) {
#pragma unroll
            for (int s = 0; s < 16; ++s) av[s] = av0[s];
          } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) av[s] = ag[(t * 32 + lh + 2 * s) * N + r_base + li];
          }
          bool nz = false;
#pragma unroll
          for (int s = 0; s < 16; ++s) nz |= av[s] != 0.f;
          if (__any(nz)) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
              const float* bp = sD + (t * 32 + lh) * FS + c * 32 + li;
              float bv[16];
#pragma unroll
              for (int s = 0; s < 16; ++s) bv[s] = bp[2 * s * FS];
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int s = 0; s < 16; ++s)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc[c], 0, 0, 0);
            }
          }
        }
      }
      // epilogue: root part, incoming gradient, undo insert + roll (gcm.py:262-278).  One base
      // pointer per array and compile-time row offsets (acc_row(r, lh) = 4*lh + const(r)): the
      // per-element 64-bit addresses of the first version spilled, and every scratch reload waits
      // on vmcnt(0), i.e. for the item's own stores.
      const bool mine = (live >> wave) & 1u;
      const int sh = wrap ? 1 : 0;
      const int row0 = r_base + 4 * lh;
      float* gq = gin + (row0 + sh) * F + li;                       // out[r] = in[r+1] on overflow
      float* po = pobs + (size_t)item * F + li;
      const float* gg = g_nodes_out ? g_nodes_out + (size_t)item * N * F + row0 * F + li : nullptr;
      const float* rt = sRt + row0 * FS + li;
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = (r & 3) + 8 * (r >> 2);                    // row - row0, compile time
          const int row = row0 + ro;
          float v = acc[c][r];
          if (gg) v += gg[ro * F + c * 32];   // gradient from later steps (per-step use only)
          if (mine) v += rt[ro * FS + c * 32];
          const bool is_cur = row == cur;
          if (is_cur) po[c * 32] = v;         // the inserted row belongs to the observation
          if (row + sh < N) gq[ro * F + c * 32] = is_cur ? 0.f : v;
        }
    }
    if (wrap && tid < F) gin[tid] = 0.f;   // in[0] was dropped by the roll: no gradient
    BSTAMP(10);
    __syncthreads();   // sG / sD / sRt / sRow / sV are rewritten by the next item
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
  for (int w = 0; w < 2; ++w)
#pragma unroll
    for (int bi = 0; bi < NBW; ++bi) {   // every block has exactly one owner wave: plain stores
      const int blk = wave + 4 * bi, h0 = (blk / CB) * 16, c0 = (blk % CB) * 16;
      float* dst = (w ? sl_root1 : sl_rel1) + (h0 + 4 * kq) * F + c0 + m16;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[r * F] = accW[w][bi][r];
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
int launch_bptt(hipStream_t s, int grid, const float* g_mx, const float* g_no, const float* x,
                const float* adj,
                const int64_t* cur, const int64_t* nn_in, Gnn2 P, const float* mx, const float* h1,
                const float* agg1, const float* agg2, float* Q, float* pobs, float* slabs,
                int items) {
  constexpr size_t lds = sizeof(float) * (size_t)LdsBptt<NT, NCT, NHT, N2T>::TOTAL;
  if (lds > 160 * 1024) return GCM_EUNSUPPORTED;
  auto kern = k_bptt_batched<NT, NCT, NHT, N2T>;
  gcm_allow_dynamic_lds((const void*)kern, lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, g_mx, g_no, x, adj, cur, nn_in, P, mx, h1,
                     agg1, agg2, Q, pobs, slabs, items);
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

extern "C" int gcm_dense_bptt_batched(const float* g_mx, const float* g_nodes_out, const float* x,
                                      const float* adj,
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
    return gcm_fused::launch_bptt<a, b_, c, d>(s, n_slabs, g_mx, g_nodes_out, x, adj, cur_idx,  \
                                               num_nodes_in, P, mx, h1, agg1, agg2, Q, pobs,    \
                                               slabs, items);
  GCM_BSHAPES(GCM_B)
#undef GCM_B
  return GCM_EUNSUPPORTED;
}
