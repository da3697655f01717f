// Time-parallel backward of the live-row DenseGCM step (rows_step.hip) for the parameters of the
// canonical 2-layer GNN.
//
// When neither the observations nor the incoming node matrix need a gradient (the reference's own
// speed test and RL training loops feed plain observations: tests/test_speed.py:44-63,
// ray_gcm.py:200-202), the backward of step t depends on g_mx[t] and on what step t saved - on no
// other step.  So the per-step autograd nodes only RECORD (saved record, g_mx) and the parameter
// gate, which runs after all of them, hands every recorded graph-step ("item") to ONE launch of
// this kernel.  The gradient is as sparse as the forward: G1 = dL/dh1 is non-zero on the live rows
// only, so an item is a few hundred floats:
//
//     d2   = g_mx * act2'(mx)                                  [H2]
//     u    = W2c^T d2          (dagg2 | dh1cur)                [2*H1]
//     dW2c += d2 (x) v,  db2 += d2                             v = agg2 | h1[cur]
//     G1_l = (coef_l * dagg2 + [l == l_cur] dh1cur) * act1'(h1_l)          for the L live rows
//     dW1c += G1_l (x) [agg1_l | x_l],  db1 += G1_l
//
// One WAVE owns one item at a time: lane m holds column m of the outer products (d2 / G1 values are
// broadcast with v_readlane), so an item is ~100 + 64 L VALU instructions and no LDS traffic; the
// gradient accumulates in registers across the items of a wave, the four waves of a workgroup
// meet in LDS at the end, one slab per workgroup, summed in fixed order (deterministic).
#include "gcm_common.h"
#include "fused_common.h"
#include "rows_common.h"

#ifndef GCM_BPTT_PB
#define GCM_BPTT_PB 8   // MODE 4: row pairs a wave fetches ahead (A/B: make ... CXXFLAGS+=-DGCM_BPTT_PB=12)
#endif

namespace gcm_rows {

// History of DenseGCM.rollout's persistent forward kernel (rollout_persist.hip) as the record source:
// per-step arrays [T, B, ...]; adj / nodes point at slot 1 of the [T+1, ...] state arrays (the state
// AFTER step t).  Only the live 32-row tiles of h1 / agg1 / adj / nodes were written - the live ROWS
// read here lie inside them.
struct Hist {
  const float *adj, *nodes, *h1, *agg1, *agg2, *mx, *gmx;
  const int64_t* cur;
  long gmx_st;   // element stride of g_mx along t
};

__device__ __forceinline__ float act_grad_sel(float y, int act_v) {
  float g = 1.f;
  g = act_v == GCM_ACT_TANH ? 1.f - y * y : g;
  g = act_v == GCM_ACT_RELU ? (y > 0.f ? 1.f : 0.f) : g;
  return g;
}

// MODE 2 - the fused DenseGCM + LearnedEdge step (learned_step.hip) as the record source: every step kept
// its FULL layers in one buffer (tab.saved[s] + the float offsets in LrnSrc; per-graph indexing).  The
// GNN parameter gradient is the same sum over the live rows; on top of it this pass hands the
// selection's backward (learned_bptt.hip) what it needs from the GNN: per item (s0 + s, b) the row cur
// and the live-row list, dagg2, and per live row l
//     dAgg1_l = G1_l W_rel1                                       [F]
// - the gradient w.r.t. row j_l of the aggregate adj @ x, i.e. w.r.t. the adjacency entries (j_l, k)
// through dAgg1_l . x[k].  A step whose tab.gmx entry is NULL takes g_mx = 0 (a record another pass owns).
struct LrnSrc {
  size_t o_adj, o_mx, o_h1, o_agg1, o_agg2, o_idx;   // float offsets inside a step's buffer
  const float* w_rel1;                                // [H1, F]
  int* hdr;          // [T, B, 2]  cur, L
  int* live;         // [T, B, N]  rows j_l
  float* da;         // [T, B, N, F]
  float* dagg2;      // [T, B, H1]
  int s0;            // path index of this launch's first step
  int adj_compact;   // the buffers hold row cur of the adjacency only ([B, N])
  const float *c_nodes, *c_h1, *c_agg1;   // cached steps: the chain's caches instead of the step's own sections
};

// FP / HP / H2P: F, H1, H2 rounded up to 32 or 64.  C1 = columns of [agg1 | x] per lane,
// C2 = columns of v per lane (column m = lane + 64 c).  MODE: 0 live-row records, 1 rollout history,
// 2 learned-step buffers.
// X32: F = H1 = 32 exactly and H2 <= 32 as a COMPILE-TIME fact (the launcher checks): the kernel is then the matrix-core
// form alone - as a run-time branch beside the general form the compiler sized the registers for both (236 VGPRs:
// two waves per SIMD on a pass that is bound by the latency of its row gathers)
template <int FP, int HP, int H2P, int MODE, bool X32 = false>
__global__ __launch_bounds__(256) void k_bptt_rows(
    StepTable tab, Hist hs, int n_steps, long gmx_sb, long gmx_sh, const float* __restrict__ w_rel2,
    const float* __restrict__ w_root2, int act1, int act2, SavedLayout lay, float* __restrict__ slabs,
    int B, int N, int F, int H1, int H2, int deg_term, LrnSrc lrn, int wave_regions) {
  constexpr bool HIST = MODE == 1 || MODE == 2;   // (3: records of cached steps, rows in the chain's caches)
  constexpr int C1 = 2 * FP / 64, C2 = 2 * HP / 64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int act1_v = gcm_vgpr(act1), act2_v = gcm_vgpr(act2);
  const int items = n_steps * B;
  const int n_waves = gridDim.x * 4, wid = blockIdx.x * 4 + wave;

  // layer-2 weights, column m of W2c = [w_rel2 | w_root2] per lane: w2c[c][o]
  // (every load issued before the first select: written as "load, then mask" in one loop the compiler sank each
  //  load under its mask's branch and waited for it alone - 32 to 64 dependent round trips, ~20 us of a 50 us launch)
  float w2c[C2][H2P];
#pragma unroll
  for (int c = 0; c < C2; ++c) {
    const int m = lane + 64 * c;
    const int mc = m < 2 * H1 ? m : 2 * H1 - 1;
    const float* src = mc < H1 ? w_rel2 + mc : w_root2 + (mc - H1);
#pragma unroll
    for (int o = 0; o < H2P; ++o) w2c[c][o] = src[(size_t)(o < H2 ? o : H2 - 1) * H1];
  }
  asm volatile("" ::: "memory");
#pragma unroll
  for (int c = 0; c < C2; ++c) {
    const bool ok = lane + 64 * c < 2 * H1;
#pragma unroll
    for (int o = 0; o < H2P; ++o) w2c[c][o] = (ok && o < H2) ? w2c[c][o] : 0.f;
  }
  float wr1c[MODE == 2 ? HP : 1];   // MODE 2: column `lane` of W_rel1
  if (MODE == 2) {
#pragma unroll
    for (int h = 0; h < HP; ++h)
      wr1c[MODE == 2 ? h : 0] = lrn.w_rel1[(size_t)(h < H1 ? h : H1 - 1) * F + (lane < F ? lane : F - 1)];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int h = 0; h < HP; ++h) wr1c[MODE == 2 ? h : 0] = (h < H1 && lane < F) ? wr1c[MODE == 2 ? h : 0] : 0.f;
  }
  float acc1[C1][HP], acc2[C2][H2P];
#pragma unroll
  for (int c = 0; c < C1; ++c)
#pragma unroll
    for (int h = 0; h < HP; ++h) acc1[c][h] = 0.f;
#pragma unroll
  for (int c = 0; c < C2; ++c)
#pragma unroll
    for (int o = 0; o < H2P; ++o) acc2[c][o] = 0.f;
  float db1 = 0.f, db2 = 0.f;
  float dc1 = 0.f;   // deg_term: gradient of the folded preprocessor-bias vector (sum_l deg_l G1_l)

  const int oc = lane < H2 ? lane : H2 - 1;
  // layer 2 of an item: d2 = g act2'(y); dW2 += d2 (x) [agg2 | h1cur]; -> dagg2, dh1cur in lane h
  auto layer2 = [&](float g, float y, const float (&vv)[C2], bool empty, float& dagg2, float& dh1c)
      __attribute__((always_inline)) {
    // (empty: the record of a graph that got no node this step - SparseGCM with taus[b] = 0 - whose zero padded
    //  output row takes no gradient)
    const float d2 = (lane < H2 && !empty) ? g * act_grad_sel(y, act2_v) : 0.f;   // d2 in lane o < H2
    db2 += d2;
    float u[C2];
#pragma unroll
    for (int c = 0; c < C2; ++c) u[c] = 0.f;
#pragma unroll
    for (int o = 0; o < H2P; ++o) {
      const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d2), o));
#pragma unroll
      for (int c = 0; c < C2; ++c) {
        u[c] = fmaf(w2c[c][o], d, u[c]);
        acc2[c][o] = fmaf(d, vv[c], acc2[c][o]);
      }
    }
    // dagg2[h] = u at column h, dh1cur[h] = u at column H1 + h: bring both to lane h
    const int hh = lane < H1 ? lane : 0;
    const int m1c = H1 + hh;
    float t0 = __shfl(u[0], hh & 63), t1 = __shfl(u[0], m1c & 63);
    if (C2 == 2) {
      const float b1 = __shfl(u[C2 - 1], m1c & 63);
      t1 = m1c >= 64 ? b1 : t1;
    }
    dagg2 = t0;
    dh1c = t1;
  };
  // what a live row adds: G1_l = (coef_l dagg2 + [l = cur] dh1cur) * act1'(h1_l);  dW1 += G1_l (x) [agg1_l | x_l]
  auto consume = [&](float cf, bool is_cur, float dagg2, float dh1c, float hv, const float (&ax)[C1], float dg,
                     float& da_out) __attribute__((always_inline)) {
    float g1 = (cf * dagg2 + (is_cur ? dh1c : 0.f)) * act_grad_sel(hv, act1_v);
    g1 = lane < H1 ? g1 : 0.f;
    db1 += g1;
    dc1 = fmaf(dg, g1, dc1);
    float da = 0.f;
#pragma unroll
    for (int h = 0; h < HP; ++h) {
      const float gh = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g1), h));
#pragma unroll
      for (int c = 0; c < C1; ++c) acc1[c][h] = fmaf(gh, ax[c], acc1[c][h]);
      if (MODE == 2) da = fmaf(gh, wr1c[MODE == 2 ? h : 0], da);
    }
    da_out = da;
  };

  // ---- MODE 4 (round 6): general records at F = H1 = 32 whose graphs hold MANY live rows (DenseEdge: every row <= cur,
  // dense.py:16-21) ------------------------------------------------------------------------------------------------
  // The pair form below keeps two row pairs of a wave in flight - enough for the handful of live rows temporal hops
  // leave, but a dense_edge rollout reads 0.8 GB of rows and that depth left the launch latency-bound (233 us per 64
  // steps at B = 256: 1.7 TB/s).  Here a wave fetches SIXTEEN rows (eight pairs: 24 dword loads a lane) ahead of the
  // sixteen it is consuming - the next item's first batch behind an item's last - so ~6 KB per wave are always on
  // their way.  Same arithmetic, same order of the sums per wave as the pair form.
  if (MODE == 4) {
    constexpr int PB = GCM_BPTT_PB;   // row pairs per batch
    const int q = lane & 31, half = lane >> 5;
    const int ocq = q < H2 ? q : H2 - 1;
    struct FrontD {
      const float* sv;
      int b, hdr0, hdr1;
      float g, y, vv, cfa, cfb;
    };
    struct Rows {
      float hv[PB], ag[PB], xx[PB];
    };
    auto front = [&](int item, FrontD& f) __attribute__((always_inline)) {
      const int s = item / B, b = item - s * B;
      const float* sv = tab.saved[s];
      f.sv = sv;
      f.b = b;
      const int* hdr = reinterpret_cast<const int*>(sv + lay.o_hdr) + 4 * b;
      f.hdr0 = hdr[0];
      f.hdr1 = hdr[1];
      f.g = tab.gmx[s][(long)b * gmx_sb + (long)ocq * gmx_sh];   // (both halves: d2 replicated)
      f.y = sv[(size_t)b * H2 + ocq];
      f.vv = sv[lay.o_v + (size_t)b * 64 + lane];                // agg2 [32] | h1cur [32]
      const int e0 = lane < N ? lane : N - 1, e1 = lane + 64 < N ? lane + 64 : N - 1;   // (entries >= L: unused)
      f.cfa = sv[lay.o_coef + (size_t)b * N + e0];
      f.cfb = sv[lay.o_coef + (size_t)b * N + e1];
    };
    auto rl = [&](int v, int l) __attribute__((always_inline)) { return __builtin_amdgcn_readlane(v, l & 63); };
    // rows l0 .. l0 + 15 (L > 0): pair p = rows l0 + 2 p (lanes 0-31) and l0 + 2 p + 1 (lanes 32-63), element q;
    // rows beyond the list are read from its last row and masked when consumed
    auto fetch = [&](const FrontD& f, int l0, int L, Rows& R) __attribute__((always_inline)) {
      const float* base = f.sv + lay.o_rows + (size_t)f.b * N * lay.rw + q;
#pragma unroll
      for (int p = 0; p < PB; ++p) {
        const int l = l0 + 2 * p + half;
        const float* row = base + (size_t)(l < L ? l : L - 1) * lay.rw;
        R.hv[p] = row[0];
        R.ag[p] = row[32];
        R.xx[p] = row[64];
      }
    };
    f32x16 aR, aT, a2r, a2t;   // dW_rel1 | dW_root1 | dW_rel2 | dW_root2 tiles: [out = acc row][in = q]
#pragma unroll
    for (int r = 0; r < 16; ++r) { aR[r] = 0.f; aT[r] = 0.f; a2r[r] = 0.f; a2t[r] = 0.f; }
    float db1p = 0.f, db2p = 0.f;
    FrontD f0{}, f1{};
    Rows RC, RN;
#pragma unroll
    for (int p = 0; p < PB; ++p) { RC.hv[p] = RC.ag[p] = RC.xx[p] = 0.f; RN.hv[p] = RN.ag[p] = RN.xx[p] = 0.f; }
    if (wid < items) {
      front(wid, f0);
      const int L0 = min(__builtin_amdgcn_readfirstlane(f0.hdr0), 128);
      if (L0 > 0) fetch(f0, 0, L0, RC);
    }
#pragma unroll 1
    for (int item = wid; item < items; item += n_waves) {
      const bool has_next = item + n_waves < items;
      if (has_next) front(item + n_waves, f1);
      const int L = min(__builtin_amdgcn_readfirstlane(f0.hdr0), 128);
      const int l_cur = __builtin_amdgcn_readfirstlane(f0.hdr1);
      const float d2 = q < H2 ? f0.g * act_grad_sel(f0.y, act2_v) : 0.f;
      db2p += d2;
      float u = 0.f;
#pragma unroll
      for (int o = 0; o < 32; ++o)
        u = fmaf(w2c[0][o], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d2), o)), u);
      a2r = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? 0.f : d2, f0.vv, a2r, 0, 0, 0);   // d2 (x) agg2
      a2t = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? d2 : 0.f, f0.vv, a2t, 0, 0, 0);   // d2 (x) h1cur
      const float dagg2 = __shfl(u, q), dh1c = __shfl(u, 32 + q);   // (in both halves)
      auto fetch_next0 = [&](Rows& R) __attribute__((always_inline)) {
        if (has_next) {
          const int Ln = min(__builtin_amdgcn_readfirstlane(f1.hdr0), 128);
          if (Ln > 0) fetch(f1, 0, Ln, R);
        }
      };
      if (L == 0) fetch_next0(RC);
#pragma unroll 1
      for (int l0 = 0; l0 < L; l0 += 2 * PB) {
        if (l0 + 2 * PB < L) fetch(f0, l0 + 2 * PB, L, RN);
        else fetch_next0(RN);
#pragma unroll
        for (int p = 0; p < PB; ++p) {
          const int l = l0 + 2 * p;
          if (l >= L) break;   // uniform
          const int l1 = l + 1;
          const float c0 = __int_as_float(l < 64 ? rl(__float_as_int(f0.cfa), l) : rl(__float_as_int(f0.cfb), l));
          const float c1 = __int_as_float(l1 < 64 ? rl(__float_as_int(f0.cfa), l1) : rl(__float_as_int(f0.cfb), l1));
          const int lm = l + half;
          float g1 = ((half ? c1 : c0) * dagg2 + (lm == l_cur ? dh1c : 0.f)) * act_grad_sel(RC.hv[p], act1_v);
          g1 = lm < L ? g1 : 0.f;
          db1p += g1;
          aR = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, RC.ag[p], aR, 0, 0, 0);
          aT = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, RC.xx[p], aT, 0, 0, 0);
        }
        RC = RN;
      }
      f0 = f1;
    }
    extern __shared__ float sSlabD[];
    const int Pm = 2 * 32 * 32 + 32 + 2 * H2 * 32 + H2;
    const int m_root1 = 32 * 32, m_b1 = 2 * 32 * 32, m_rel2 = m_b1 + 32, m_root2 = m_rel2 + H2 * 32, m_b2 = m_root2 + H2 * 32;
    float* mine = sSlabD + (size_t)wave * Pm;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * half;   // accumulator row of a 32x32 tile
      mine[i * 32 + q] = aR[r];
      mine[m_root1 + i * 32 + q] = aT[r];
      if (i < H2) {
        mine[m_rel2 + i * 32 + q] = a2r[r];
        mine[m_root2 + i * 32 + q] = a2t[r];
      }
    }
    const float db1m = db1p + __shfl_xor(db1p, 32);
    if (lane < 32) mine[m_b1 + lane] = db1m;
    if (lane < H2) mine[m_b2 + lane] = db2p;
    __syncthreads();
    float* slabm = slabs + (size_t)blockIdx.x * Pm;
    for (int e = tid; e < Pm; e += 256)
      slabm[e] = ((sSlabD[e] + sSlabD[Pm + e]) + sSlabD[2 * Pm + e]) + sSlabD[3 * (size_t)Pm + e];
    return;
  }

  // ---- records at F = H1 = 32 (cfg2 / cfg4 / cfg5's GNN): the rank-1 updates on the matrix cores ----------------
  // dW1 += G1_l (x) [agg1_l | x_l] over the live rows is a [32 x rows] . [rows x 64] product: rows are taken in
  // PAIRS - lanes 0-31 hold row l, lanes 32-63 row l + 1 (k = 0 / 1 of v_mfma_f32_32x32x2_f32) - as two MFMAs per
  // pair (the agg1 block, the x block) instead of 2 x 64 v_readlane + v_fma; layer 2's d2 (x) [agg2 | h1cur] as two
  // MFMAs per item with d2 masked to one half each.  The VALU form executes ~875 instructions per item and was
  // issue-bound at two waves per SIMD.  Same pipelining as below (front of item i + 1 ahead of item i, pair p + 1
  // ahead of pair p, the next item's first pair ahead of this item's last).
  if (X32 || (!HIST && FP == 32 && HP == 32 && H2P == 32 && F == 32 && H1 == 32 && wave_regions == 3)) {
    const int q = lane & 31, half = lane >> 5;
    const int ocq = q < H2 ? q : H2 - 1;
    struct FrontM {
      const float* sv;
      int b, hdr0, hdr1, ja, jb;
      float g, y, vv, cfa, cfb;
    };
    auto front = [&](int item, FrontM& f) __attribute__((always_inline)) {
      const int s = item / B, b = item - s * B;
      const float* sv = tab.saved[s];
      f.sv = sv;
      f.b = b;
      const int* hdr = reinterpret_cast<const int*>(sv + lay.o_hdr) + 4 * b;
      f.hdr0 = hdr[0];
      f.hdr1 = hdr[1];
      f.g = tab.gmx[s][(long)b * gmx_sb + (long)ocq * gmx_sh];   // (both halves: d2 replicated)
      f.y = sv[(size_t)b * H2 + ocq];
      f.vv = sv[lay.o_v + (size_t)b * 64 + lane];                // agg2 [32] | h1cur [32]
      const int e0 = lane < N ? lane : N - 1, e1 = lane + 64 < N ? lane + 64 : N - 1;   // (entries >= L: unused)
      f.cfa = sv[lay.o_coef + (size_t)b * N + e0];
      f.cfb = sv[lay.o_coef + (size_t)b * N + e1];
      f.ja = f.jb = 0;
      if (MODE == 3) {
        const int* live = reinterpret_cast<const int*>(sv + lay.o_live) + (size_t)b * N;
        f.ja = live[e0];
        f.jb = live[e1];
      }
    };
    auto rl = [&](int v, int l) __attribute__((always_inline)) { return __builtin_amdgcn_readlane(v, l & 63); };
    // the pair (l, l + 1): h1 | agg1 | x of row l + half, element q (an absent second row: the first one's, masked)
    auto fetch = [&](const FrontM& f, int l, int L, float& hv, float& ag, float& xx, float& dg)
        __attribute__((always_inline)) {
      const int l1 = l + 1 < L ? l + 1 : l;
      dg = 0.f;
      if (MODE == 3) {
        const int j0 = l < 64 ? rl(f.ja, l) : rl(f.jb, l), j1 = l1 < 64 ? rl(f.ja, l1) : rl(f.jb, l1);
        const size_t rj = ((size_t)f.b * N + (half ? j1 : j0)) * 32 + q;
        hv = lrn.c_h1[rj];
        ag = lrn.c_agg1[rj];
        xx = lrn.c_nodes[rj];
      } else {
        const float* row = f.sv + lay.o_rows + ((size_t)f.b * N + (half ? l1 : l)) * lay.rw;
        if (deg_term) dg = f.sv[lay.o_deg + (size_t)f.b * N + (half ? l1 : l)];
        hv = row[q];
        ag = row[32 + q];
        xx = row[64 + q];
      }
    };
    f32x16 aR, aT, a2r, a2t;   // dW_rel1 | dW_root1 | dW_rel2 | dW_root2 tiles: [out = acc row][in = q]
#pragma unroll
    for (int r = 0; r < 16; ++r) { aR[r] = 0.f; aT[r] = 0.f; a2r[r] = 0.f; a2t[r] = 0.f; }
    float db1p = 0.f, dc1p = 0.f, db2p = 0.f;   // per lane: halves (and for db2 the replica) met at the end
    auto step = [&](int item, FrontM& cur, float& hv0, float& ag0, float& xx0, float& dg0, FrontM& nxt, float& hvN,
                    float& agN, float& xxN, float& dgN) __attribute__((always_inline)) {
      const bool has_next = item + n_waves < items;
      if (has_next) front(item + n_waves, nxt);
      const int L = min(__builtin_amdgcn_readfirstlane(cur.hdr0), 128);
      const int l_cur = __builtin_amdgcn_readfirstlane(cur.hdr1);
      // layer 2: d2 in BOTH halves; u = [w_rel2 | w_root2]^T d2 on the VALU (a matrix-vector product)
      const float d2 = (q < H2 && !(MODE == 3 && L == 0)) ? cur.g * act_grad_sel(cur.y, act2_v) : 0.f;
      db2p += d2;
      float u = 0.f;
#pragma unroll
      for (int o = 0; o < 32; ++o)
        u = fmaf(w2c[0][o], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d2), o)), u);
      a2r = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? 0.f : d2, cur.vv, a2r, 0, 0, 0);   // d2 (x) agg2
      a2t = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? d2 : 0.f, cur.vv, a2t, 0, 0, 0);   // d2 (x) h1cur
      const float dagg2 = __shfl(u, q), dh1c = __shfl(u, 32 + q);   // (in both halves)
      auto fetch_next0 = [&]() __attribute__((always_inline)) {
        if (has_next) {
          const int Ln = min(__builtin_amdgcn_readfirstlane(nxt.hdr0), 128);
          if (Ln > 0) fetch(nxt, 0, Ln, hvN, agN, xxN, dgN);
        }
      };
      auto consume = [&](int l, float hv, float ag, float xx, float dg) __attribute__((always_inline)) {
        const int l1 = l + 1;
        const float c0 = __int_as_float(l < 64 ? rl(__float_as_int(cur.cfa), l) : rl(__float_as_int(cur.cfb), l));
        const float c1 = __int_as_float(l1 < 64 ? rl(__float_as_int(cur.cfa), l1) : rl(__float_as_int(cur.cfb), l1));
        const int lm = l + half;
        float g1 = ((half ? c1 : c0) * dagg2 + (lm == l_cur ? dh1c : 0.f)) * act_grad_sel(hv, act1_v);
        g1 = lm < L ? g1 : 0.f;
        db1p += g1;
        dc1p = fmaf(dg, g1, dc1p);
        aR = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, ag, aR, 0, 0, 0);
        aT = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, xx, aT, 0, 0, 0);
      };
      float hvB = 0.f, agB = 0.f, xxB = 0.f, dgB = 0.f;
      if (L == 0) fetch_next0();
#pragma unroll 1
      for (int l = 0; l < L; l += 4) {
        const bool two = l + 2 < L;
        if (two) fetch(cur, l + 2, L, hvB, agB, xxB, dgB);
        else fetch_next0();
        consume(l, hv0, ag0, xx0, dg0);
        if (two) {
          if (l + 4 < L) fetch(cur, l + 4, L, hv0, ag0, xx0, dg0);
          else fetch_next0();
          consume(l + 2, hvB, agB, xxB, dgB);
        }
      }
    };
    FrontM f0{}, f1{};
    float hv0 = 0.f, ag0 = 0.f, xx0 = 0.f, dg0 = 0.f, hv1 = 0.f, ag1 = 0.f, xx1 = 0.f, dg1 = 0.f;
    if (wid < items) {
      front(wid, f0);
      const int L0 = min(__builtin_amdgcn_readfirstlane(f0.hdr0), 128);
      if (L0 > 0) fetch(f0, 0, L0, hv0, ag0, xx0, dg0);
    }
#pragma unroll 1
    for (int item = wid; item < items; item += 2 * n_waves) {
      step(item, f0, hv0, ag0, xx0, dg0, f1, hv1, ag1, xx1, dg1);
      if (item + n_waves < items) step(item + n_waves, f1, hv1, ag1, xx1, dg1, f0, hv0, ag0, xx0, dg0);
    }
    // ---- the wave's tiles into its LDS region, the four regions summed in fixed order (as below) -------------
    extern __shared__ float sSlabM[];
    const int P0m = 2 * 32 * 32 + 32 + 2 * H2 * 32 + H2;
    const int Pm = P0m + (deg_term ? 32 : 0);
    const int m_root1 = 32 * 32, m_b1 = 2 * 32 * 32, m_rel2 = m_b1 + 32, m_root2 = m_rel2 + H2 * 32, m_b2 = m_root2 + H2 * 32;
    float* mine = sSlabM + (size_t)wave * Pm;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * half;   // accumulator row of a 32x32 tile
      mine[i * 32 + q] = aR[r];
      mine[m_root1 + i * 32 + q] = aT[r];
      if (i < H2) {
        mine[m_rel2 + i * 32 + q] = a2r[r];
        mine[m_root2 + i * 32 + q] = a2t[r];
      }
    }
    const float db1m = db1p + __shfl_xor(db1p, 32), dc1m = dc1p + __shfl_xor(dc1p, 32);
    if (lane < 32) mine[m_b1 + lane] = db1m;
    if (lane < H2) mine[m_b2 + lane] = db2p;
    if (deg_term && lane < 32) mine[P0m + lane] = dc1m;
    __syncthreads();
    float* slabm = slabs + (size_t)blockIdx.x * Pm;
    for (int e = tid; e < Pm; e += 256)
      slabm[e] = ((sSlabM[e] + sSlabM[Pm + e]) + sSlabM[2 * Pm + e]) + sSlabM[3 * (size_t)Pm + e];
    return;
  }

  // ---- the same at F = 64 (cfg3): two 32-column blocks per node row (four MFMAs per row pair), the tiles met in LDS
  //      in two rounds - layer 1, then layer 2: four regions of the larger round fit beside a second workgroup ----
  if (!HIST && FP == 64 && HP == 32 && H2P == 32 && F == FP && H1 == 32 && wave_regions == 2) {
    constexpr int FB = FP / 32;   // 32-column blocks of a node row (F = 32: 1, F = 64: 2)
    const int q = lane & 31, half = lane >> 5;
    const int ocq = q < H2 ? q : H2 - 1;
    struct FrontM {
      const float* sv;
      int b, hdr0, hdr1, ja, jb;
      float g, y, vv, cfa, cfb;
    };
    auto front = [&](int item, FrontM& f) __attribute__((always_inline)) {
      const int s = item / B, b = item - s * B;
      const float* sv = tab.saved[s];
      f.sv = sv;
      f.b = b;
      const int* hdr = reinterpret_cast<const int*>(sv + lay.o_hdr) + 4 * b;
      f.hdr0 = hdr[0];
      f.hdr1 = hdr[1];
      f.g = tab.gmx[s][(long)b * gmx_sb + (long)ocq * gmx_sh];   // (both halves: d2 replicated)
      f.y = sv[(size_t)b * H2 + ocq];
      f.vv = sv[lay.o_v + (size_t)b * 64 + lane];                // agg2 [32] | h1cur [32]
      const int e0 = lane < N ? lane : N - 1, e1 = lane + 64 < N ? lane + 64 : N - 1;   // (entries >= L: unused)
      f.cfa = sv[lay.o_coef + (size_t)b * N + e0];
      f.cfb = sv[lay.o_coef + (size_t)b * N + e1];
      f.ja = f.jb = 0;
      if (MODE == 3) {
        const int* live = reinterpret_cast<const int*>(sv + lay.o_live) + (size_t)b * N;
        f.ja = live[e0];
        f.jb = live[e1];
      }
    };
    auto rl = [&](int v, int l) __attribute__((always_inline)) { return __builtin_amdgcn_readlane(v, l & 63); };
    // the pair (l, l + 1): h1 | agg1 | x of row l + half, element q (an absent second row: the first one's, masked)
    auto fetch = [&](const FrontM& f, int l, int L, float& hv, float (&ag)[FB], float (&xx)[FB], float& dg)
        __attribute__((always_inline)) {
      const int l1 = l + 1 < L ? l + 1 : l;
      dg = 0.f;
      if (MODE == 3) {
        const int j0 = l < 64 ? rl(f.ja, l) : rl(f.jb, l), j1 = l1 < 64 ? rl(f.ja, l1) : rl(f.jb, l1);
        const size_t rr = (size_t)f.b * N + (half ? j1 : j0);
        hv = lrn.c_h1[rr * 32 + q];
#pragma unroll
        for (int c = 0; c < FB; ++c) {
          ag[c] = lrn.c_agg1[rr * FP + 32 * c + q];
          xx[c] = lrn.c_nodes[rr * FP + 32 * c + q];
        }
      } else {
        const float* row = f.sv + lay.o_rows + ((size_t)f.b * N + (half ? l1 : l)) * lay.rw;
        if (deg_term) dg = f.sv[lay.o_deg + (size_t)f.b * N + (half ? l1 : l)];
        hv = row[q];
#pragma unroll
        for (int c = 0; c < FB; ++c) {
          ag[c] = row[32 + 32 * c + q];
          xx[c] = row[32 + FP + 32 * c + q];
        }
      }
    };
    f32x16 aR[FB], aT[FB], a2r, a2t;   // dW_rel1 | dW_root1 (per 32-column block) | dW_rel2 | dW_root2 tiles: [out = acc row][in = q]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      a2r[r] = 0.f; a2t[r] = 0.f;
#pragma unroll
      for (int c = 0; c < FB; ++c) { aR[c][r] = 0.f; aT[c][r] = 0.f; }
    }
    float db1p = 0.f, dc1p = 0.f, db2p = 0.f;   // per lane: halves (and for db2 the replica) met at the end
    auto step = [&](int item, FrontM& cur, float& hv0, float (&ag0)[FB], float (&xx0)[FB], float& dg0, FrontM& nxt,
                    float& hvN, float (&agN)[FB], float (&xxN)[FB], float& dgN) __attribute__((always_inline)) {
      const bool has_next = item + n_waves < items;
      if (has_next) front(item + n_waves, nxt);
      const int L = min(__builtin_amdgcn_readfirstlane(cur.hdr0), 128);
      const int l_cur = __builtin_amdgcn_readfirstlane(cur.hdr1);
      // layer 2: d2 in BOTH halves; u = [w_rel2 | w_root2]^T d2 on the VALU (a matrix-vector product)
      const float d2 = (q < H2 && !(MODE == 3 && L == 0)) ? cur.g * act_grad_sel(cur.y, act2_v) : 0.f;
      db2p += d2;
      float u = 0.f;
#pragma unroll
      for (int o = 0; o < 32; ++o)
        u = fmaf(w2c[0][o], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d2), o)), u);
      a2r = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? 0.f : d2, cur.vv, a2r, 0, 0, 0);   // d2 (x) agg2
      a2t = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? d2 : 0.f, cur.vv, a2t, 0, 0, 0);   // d2 (x) h1cur
      const float dagg2 = __shfl(u, q), dh1c = __shfl(u, 32 + q);   // (in both halves)
      auto fetch_next0 = [&]() __attribute__((always_inline)) {
        if (has_next) {
          const int Ln = min(__builtin_amdgcn_readfirstlane(nxt.hdr0), 128);
          if (Ln > 0) fetch(nxt, 0, Ln, hvN, agN, xxN, dgN);
        }
      };
      auto consume = [&](int l, float hv, const float (&ag)[FB], const float (&xx)[FB], float dg)
          __attribute__((always_inline)) {
        const int l1 = l + 1;
        const float c0 = __int_as_float(l < 64 ? rl(__float_as_int(cur.cfa), l) : rl(__float_as_int(cur.cfb), l));
        const float c1 = __int_as_float(l1 < 64 ? rl(__float_as_int(cur.cfa), l1) : rl(__float_as_int(cur.cfb), l1));
        const int lm = l + half;
        float g1 = ((half ? c1 : c0) * dagg2 + (lm == l_cur ? dh1c : 0.f)) * act_grad_sel(hv, act1_v);
        g1 = lm < L ? g1 : 0.f;
        db1p += g1;
        dc1p = fmaf(dg, g1, dc1p);
#pragma unroll
        for (int c = 0; c < FB; ++c) {
          aR[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, ag[c], aR[c], 0, 0, 0);
          aT[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, xx[c], aT[c], 0, 0, 0);
        }
      };
      float hvB = 0.f, agB[FB], xxB[FB], dgB = 0.f;
#pragma unroll
      for (int c = 0; c < FB; ++c) { agB[c] = 0.f; xxB[c] = 0.f; }
      if (L == 0) fetch_next0();
#pragma unroll 1
      for (int l = 0; l < L; l += 4) {
        const bool two = l + 2 < L;
        if (two) fetch(cur, l + 2, L, hvB, agB, xxB, dgB);
        else fetch_next0();
        consume(l, hv0, ag0, xx0, dg0);
        if (two) {
          if (l + 4 < L) fetch(cur, l + 4, L, hv0, ag0, xx0, dg0);
          else fetch_next0();
          consume(l + 2, hvB, agB, xxB, dgB);
        }
      }
    };
    FrontM f0{}, f1{};
    float hv0 = 0.f, ag0[FB], xx0[FB], dg0 = 0.f, hv1 = 0.f, ag1[FB], xx1[FB], dg1 = 0.f;
#pragma unroll
    for (int c = 0; c < FB; ++c) { ag0[c] = xx0[c] = ag1[c] = xx1[c] = 0.f; }
    if (wid < items) {
      front(wid, f0);
      const int L0 = min(__builtin_amdgcn_readfirstlane(f0.hdr0), 128);
      if (L0 > 0) fetch(f0, 0, L0, hv0, ag0, xx0, dg0);
    }
#pragma unroll 1
    for (int item = wid; item < items; item += 2 * n_waves) {
      step(item, f0, hv0, ag0, xx0, dg0, f1, hv1, ag1, xx1, dg1);
      if (item + n_waves < items) step(item + n_waves, f1, hv1, ag1, xx1, dg1, f0, hv0, ag0, xx0, dg0);
    }
    // ---- the waves' tiles meet in LDS in two rounds (layer 1, then layer 2: four regions of the larger of the two fit
    //      beside a second workgroup at F = 64 too), every element summed ((w0 + w1) + w2) + w3 ----------------------
    extern __shared__ float sSlabM[];
    const int P0m = 2 * 32 * FP + 32 + 2 * H2 * 32 + H2;
    const int Pm = P0m + (deg_term ? 32 : 0);
    const int m_root1 = 32 * FP, m_b1 = 2 * 32 * FP, m_rel2 = m_b1 + 32, m_root2 = m_rel2 + H2 * 32, m_b2 = m_root2 + H2 * 32;
    const int R1 = m_rel2, R2 = Pm - m_rel2;            // floats of a region in round 1 / 2
    const int RS = R1 > R2 ? R1 : R2;
    float* mine = sSlabM + (size_t)wave * RS;
    float* slabm = slabs + (size_t)blockIdx.x * Pm;
    const float db1m = db1p + __shfl_xor(db1p, 32), dc1m = dc1p + __shfl_xor(dc1p, 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * half;   // accumulator row of a 32x32 tile
#pragma unroll
      for (int c = 0; c < FB; ++c) {
        mine[i * FP + 32 * c + q] = aR[c][r];
        mine[m_root1 + i * FP + 32 * c + q] = aT[c][r];
      }
    }
    if (lane < 32) mine[m_b1 + lane] = db1m;
    __syncthreads();
    for (int e = tid; e < R1; e += 256)
      slabm[e] = ((sSlabM[e] + sSlabM[RS + e]) + sSlabM[2 * RS + e]) + sSlabM[3 * (size_t)RS + e];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (i < H2) {
        mine[i * 32 + q] = a2r[r];
        mine[(m_root2 - m_rel2) + i * 32 + q] = a2t[r];
      }
    }
    if (lane < H2) mine[(m_b2 - m_rel2) + lane] = db2p;
    if (deg_term && lane < 32) mine[(P0m - m_rel2) + lane] = dc1m;
    __syncthreads();
    for (int e = tid; e < R2; e += 256)
      slabm[m_rel2 + e] = ((sSlabM[e] + sSlabM[RS + e]) + sSlabM[2 * RS + e]) + sSlabM[3 * (size_t)RS + e];
    return;
  }

  if (!HIST) {
    // Records (MODE 0 / 3), software-pipelined ACROSS items and across the rows of an item.  An item's data hang on
    // a chain of dependent loads - the record pointer, its header / row list, then the rows the list names - which
    // a wave with eight items to do used to pay in full for every item and every row (~7 us per item at four live
    // rows, two waves per SIMD to hide it).  Now: the "front" of item i + 1 (header, incoming gradient, y, v, the row
    // list: coefficients, and for cached steps the row indices, lane l holding entries l and l + 64) is requested
    // before item i is worked on; row l + 1 is requested before row l is consumed; and the first row of item
    // i + 1 before the last row of item i.  Same items, same rows, same order of every sum.
    struct Front {
      const float* sv;
      int b, hdr0, hdr1, ja, jb;
      float g, y, vv[C2], cfa, cfb;
    };
    auto front = [&](int item, Front& f) __attribute__((always_inline)) {
      const int s = item / B, b = item - s * B;
      const float* sv = tab.saved[s];
      f.sv = sv;
      f.b = b;
      const int* hdr = reinterpret_cast<const int*>(sv + lay.o_hdr) + 4 * b;
      f.hdr0 = hdr[0];
      f.hdr1 = hdr[1];
      f.g = tab.gmx[s][(long)b * gmx_sb + (long)oc * gmx_sh];
      f.y = sv[(size_t)b * H2 + oc];
#pragma unroll
      for (int c = 0; c < C2; ++c) {
        const int m = lane + 64 * c;
        const float t = sv[lay.o_v + (size_t)b * 2 * H1 + (m < 2 * H1 ? m : 2 * H1 - 1)];
        f.vv[c] = m < 2 * H1 ? t : 0.f;
      }
      const int e0 = lane < N ? lane : N - 1, e1 = lane + 64 < N ? lane + 64 : N - 1;   // (entries >= L: unused)
      f.cfa = sv[lay.o_coef + (size_t)b * N + e0];
      f.cfb = sv[lay.o_coef + (size_t)b * N + e1];
      f.ja = f.jb = 0;
      if (MODE == 3) {
        const int* live = reinterpret_cast<const int*>(sv + lay.o_live) + (size_t)b * N;
        f.ja = live[e0];
        f.jb = live[e1];
      }
    };
    auto fetch = [&](const Front& f, int l, float& hv, float (&ax)[C1], float& dg) __attribute__((always_inline)) {
      dg = 0.f;
      if (MODE == 3) {   // cached step: the row's h1 | agg1 | x from the chain's caches
        const int j = l < 64 ? __builtin_amdgcn_readlane(f.ja, l & 63) : __builtin_amdgcn_readlane(f.jb, l & 63);
        const size_t rj = (size_t)f.b * N + j;
        hv = lrn.c_h1[rj * H1 + (lane < H1 ? lane : H1 - 1)];
#pragma unroll
        for (int c = 0; c < C1; ++c) {
          const int m = lane + 64 * c;
          const int k = m < F ? m : (m - F < F ? m - F : F - 1);
          const float t = m < F ? lrn.c_agg1[rj * F + k] : lrn.c_nodes[rj * F + k];
          ax[c] = m < 2 * F ? t : 0.f;
        }
      } else {
        const float* row = f.sv + lay.o_rows + ((size_t)f.b * N + l) * lay.rw;
        if (deg_term) dg = f.sv[lay.o_deg + (size_t)f.b * N + l];
        hv = row[lane < H1 ? lane : H1 - 1];
#pragma unroll
        for (int c = 0; c < C1; ++c) {
          const int m = lane + 64 * c;
          const float t = row[H1 + (m < 2 * F ? m : 2 * F - 1)];
          ax[c] = m < 2 * F ? t : 0.f;
        }
      }
    };
    // (the row lists hold at most 128 entries: N <= 128 on the dense path, <= 17 selected rows on the sparse one)
    auto coef_of = [&](const Front& f, int l) __attribute__((always_inline)) {
      return __int_as_float(l < 64 ? __builtin_amdgcn_readlane(__float_as_int(f.cfa), l & 63)
                                   : __builtin_amdgcn_readlane(__float_as_int(f.cfb), l & 63));
    };
    // one item: `cur` and its first row (hv0 / ax0 / dg0) are in registers; leaves `nxt` and ITS first row there.
    // (Two fronts used in turn - the item loop below is unrolled by two with the roles swapped - so that nothing
    //  is copied between them: a copy would wait for the loads it copies.)
    auto step = [&](int item, Front& cur, float& hv0, float (&ax0)[C1], float& dg0, Front& nxt, float& hvN,
                    float (&axN)[C1], float& dgN) __attribute__((always_inline)) {
      const bool has_next = item + n_waves < items;
      if (has_next) front(item + n_waves, nxt);
      const int L = min(__builtin_amdgcn_readfirstlane(cur.hdr0), 128);
      const int l_cur = __builtin_amdgcn_readfirstlane(cur.hdr1);
      float dagg2, dh1c, unused;
      layer2(cur.g, cur.y, cur.vv, MODE == 3 && L == 0, dagg2, dh1c);
      float hvB = 0.f, axB[C1], dgB = 0.f;
#pragma unroll
      for (int c = 0; c < C1; ++c) axB[c] = 0.f;
      // (the next item's first row: requested before this item's last row is consumed)
      auto fetch_next0 = [&]() __attribute__((always_inline)) {
        if (has_next && __builtin_amdgcn_readfirstlane(nxt.hdr0) > 0) fetch(nxt, 0, hvN, axN, dgN);
      };
      if (L == 0) fetch_next0();
#pragma unroll 1
      for (int l = 0; l < L; l += 2) {
        const bool two = l + 1 < L;
        if (two) fetch(cur, l + 1, hvB, axB, dgB);
        else fetch_next0();
        consume(coef_of(cur, l), l == l_cur, dagg2, dh1c, hv0, ax0, dg0, unused);
        if (two) {
          if (l + 2 < L) fetch(cur, l + 2, hv0, ax0, dg0);
          else fetch_next0();
          consume(coef_of(cur, l + 1), l + 1 == l_cur, dagg2, dh1c, hvB, axB, dgB, unused);
        }
      }
    };
    Front f0{}, f1{};
    float hv0 = 0.f, ax0[C1], dg0 = 0.f, hv1 = 0.f, ax1[C1], dg1 = 0.f;
#pragma unroll
    for (int c = 0; c < C1; ++c) ax0[c] = ax1[c] = 0.f;
    if (wid < items) {
      front(wid, f0);
      if (__builtin_amdgcn_readfirstlane(f0.hdr0) > 0) fetch(f0, 0, hv0, ax0, dg0);
    }
#pragma unroll 1
    for (int item = wid; item < items; item += 2 * n_waves) {
      step(item, f0, hv0, ax0, dg0, f1, hv1, ax1, dg1);
      if (item + n_waves < items) step(item + n_waves, f1, hv1, ax1, dg1, f0, hv0, ax0, dg0);
    }
  } else {
#pragma unroll 1
    for (int item = wid; item < items; item += n_waves) {
      const int s = item / B, b = item - s * B;
      Hist src{};        // where the full layers of this item live
      size_t gi = 0;     // and the index of its graph in them
      int n_live = 0;    // MODE 2: live rows handed on so far
      float g, vv[C2];
      if (MODE == 2) {   // this step's own buffer, per-graph indexing
        const float* base = tab.saved[s];
        src.adj = base + lrn.o_adj;
        src.nodes = lrn.c_nodes ? lrn.c_nodes : base;
        src.h1 = lrn.c_h1 ? lrn.c_h1 : base + lrn.o_h1;
        src.agg1 = lrn.c_agg1 ? lrn.c_agg1 : base + lrn.o_agg1;
        src.agg2 = base + lrn.o_agg2; src.mx = base + lrn.o_mx;
        src.cur = reinterpret_cast<const int64_t*>(base + lrn.o_idx);
        gi = (size_t)b;
        const float* gp = tab.gmx[s];
        g = gp ? gp[(long)b * gmx_sb + (long)oc * gmx_sh] : 0.f;
      } else {
        src = hs;
        gi = (size_t)item;   // = s * B + b
        g = hs.gmx[(long)s * hs.gmx_st + (long)b * gmx_sb + (long)oc * gmx_sh];
      }
      const int64_t c64 = src.cur[gi];
      const int cur = __builtin_amdgcn_readfirstlane(c64 < 0 ? 0 : (c64 > N - 1 ? N - 1 : (int)c64));
      const float y = src.mx[gi * H2 + oc];
      const float* arow = (MODE == 2 && lrn.adj_compact) ? src.adj + gi * N : src.adj + (gi * N + cur) * N;
      const float a0 = arow[lane < N ? lane : N - 1];
      const float a1 = arow[lane + 64 < N ? lane + 64 : N - 1];
      unsigned long long m0 = __ballot(lane < N && (a0 != 0.f || lane == cur));       // live rows as two 64-bit masks
      unsigned long long m1 = __ballot(lane + 64 < N && (a1 != 0.f || lane + 64 == cur));
#pragma unroll
      for (int c = 0; c < C2; ++c) {
        const int m = lane + 64 * c;
        const int k = m < H1 ? m : (m - H1 < H1 ? m - H1 : H1 - 1);
        const float t = m < H1 ? src.agg2[gi * H1 + k] : src.h1[(gi * N + cur) * H1 + k];
        vv[c] = m < 2 * H1 ? t : 0.f;
      }
      float dagg2, dh1c;
      layer2(g, y, vv, false, dagg2, dh1c);
#pragma unroll 1
      for (;;) {
        if (!(m0 | m1)) break;
        const int j = m0 ? __builtin_ctzll(m0) : 64 + __builtin_ctzll(m1);
        if (m0) m0 &= m0 - 1; else m1 &= m1 - 1;
        const float cf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(j < 64 ? a0 : a1), j & 63));
        const size_t rj = gi * N + j;
        float ax[C1], da;
        const float hv = src.h1[rj * H1 + (lane < H1 ? lane : H1 - 1)];
#pragma unroll
        for (int c = 0; c < C1; ++c) {
          const int m = lane + 64 * c;
          const int f = m < F ? m : (m - F < F ? m - F : F - 1);
          const float t = m < F ? src.agg1[rj * F + f] : src.nodes[rj * F + f];
          ax[c] = m < 2 * F ? t : 0.f;
        }
        if (MODE == 2 && lane == 0) lrn.live[((size_t)(lrn.s0 + s) * B + b) * N + n_live] = j;
        consume(cf, j == cur, dagg2, dh1c, hv, ax, 0.f, da);
        if (MODE == 2) {   // dAgg1_l, column `lane`
          if (lane < F) lrn.da[(((size_t)(lrn.s0 + s) * B + b) * N + n_live) * F + lane] = da;
          ++n_live;
        }
      }
      if (MODE == 2) {
        const size_t it = (size_t)(lrn.s0 + s) * B + b;
        if (lane == 0) {
          lrn.hdr[2 * it] = cur;
          lrn.hdr[2 * it + 1] = n_live;
        }
        if (lane < H1) lrn.dagg2[it * H1 + lane] = dagg2;
      }
    }
  }

  // ---- one slab per workgroup: the waves add their registers into LDS one after another ----------
  // slab: dW_rel1 [H1*F] | dW_root1 [H1*F] | db1 [H1] | dW_rel2 [H2*H1] | dW_root2 [H2*H1] | db2 [H2]
  //       (| dc1 [H1] with deg_term)
  extern __shared__ float sSlab[];
  const int P0 = 2 * H1 * F + H1 + 2 * H2 * H1 + H2;
  const int P = P0 + (deg_term ? H1 : 0);
  const int o_root1 = H1 * F, o_b1 = 2 * H1 * F, o_rel2 = o_b1 + H1, o_root2 = o_rel2 + H2 * H1;
  const int o_b2 = o_root2 + H2 * H1;
  // wave_regions: each wave lays its registers down in a region of its own (plain stores, nothing to wait for), one
  // barrier, then every thread adds the four regions of its elements in the order the waves used to add themselves,
  // ((w0 + w1) + w2) + w3 - the same sums bit for bit.  Otherwise (shapes whose four regions would not leave room
  // for two workgroups per CU) the waves take turns on one region: 64+ read-modify-writes per lane, four times over.
  float* slab = slabs + (size_t)blockIdx.x * P;
  if (wave_regions) {
    float* mine = sSlab + (size_t)wave * P;
#pragma unroll
    for (int c = 0; c < C1; ++c) {
      const int m = lane + 64 * c;
      if (m < 2 * F) {
        const int base = m < F ? m : o_root1 + (m - F);
#pragma unroll
        for (int h = 0; h < HP; ++h)
          if (h < H1) mine[base + h * F] = acc1[c][h];
      }
    }
#pragma unroll
    for (int c = 0; c < C2; ++c) {
      const int m = lane + 64 * c;
      if (m < 2 * H1) {
        const int base = m < H1 ? o_rel2 + m : o_root2 + (m - H1);
#pragma unroll
        for (int o = 0; o < H2P; ++o)
          if (o < H2) mine[base + o * H1] = acc2[c][o];
      }
    }
    if (lane < H1) mine[o_b1 + lane] = db1;
    if (lane < H2) mine[o_b2 + lane] = db2;
    if (deg_term && lane < H1) mine[P0 + lane] = dc1;
    __syncthreads();
    for (int e = tid; e < P; e += 256)
      slab[e] = ((sSlab[e] + sSlab[P + e]) + sSlab[2 * P + e]) + sSlab[3 * (size_t)P + e];
    return;
  }
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int c = 0; c < C1; ++c) {
        const int m = lane + 64 * c;
        if (m < 2 * F) {
          const int base = m < F ? m : o_root1 + (m - F);
#pragma unroll
          for (int h = 0; h < HP; ++h)
            if (h < H1) {
              float* d = sSlab + base + h * F;
              *d = (w ? *d : 0.f) + acc1[c][h];
            }
        }
      }
#pragma unroll
      for (int c = 0; c < C2; ++c) {
        const int m = lane + 64 * c;
        if (m < 2 * H1) {
          const int base = m < H1 ? o_rel2 + m : o_root2 + (m - H1);
#pragma unroll
          for (int o = 0; o < H2P; ++o)
            if (o < H2) {
              float* d = sSlab + base + o * H1;
              *d = (w ? *d : 0.f) + acc2[c][o];
            }
        }
      }
      if (lane < H1) sSlab[o_b1 + lane] = (w ? sSlab[o_b1 + lane] : 0.f) + db1;
      if (lane < H2) sSlab[o_b2 + lane] = (w ? sSlab[o_b2 + lane] : 0.f) + db2;
      if (deg_term && lane < H1) sSlab[P0 + lane] = (w ? sSlab[P0 + lane] : 0.f) + dc1;
    }
    __syncthreads();
  }
  for (int e = tid; e < P; e += 256) slab[e] = sSlab[e];
}


template <int FP, int HP, int H2P, int MODE>
int launch_bptt(hipStream_t s, int grid, const StepTable& tab, const Hist& hs, int n_steps, long sb, long sh,
                const float* w_rel2, const float* w_root2, int act1, int act2,
                const SavedLayout& lay, float* slabs, int B, int N, int F, int H1, int H2,
                int deg_term = 0, const LrnSrc& lrn = LrnSrc{}) {
  const size_t P = 2 * (size_t)H1 * F + H1 + 2 * (size_t)H2 * H1 + H2 + (deg_term ? H1 : 0);
  // (one LDS region per wave while four of them leave room for two workgroups per CU: see the kernel's tail)
  int wave_regions = sizeof(float) * P * 4 <= 80 * 1024 ? 1 : 0;
  size_t lds = sizeof(float) * P * (wave_regions ? 4 : 1);
  if (MODE == 4) {   // (the launcher's caller checked F == H1 == 32, H2 <= 32, no folded terms)
    wave_regions = 3;
    lds = sizeof(float) * P * 4;
  } else if ((MODE == 0 || MODE == 3) && HP == 32 && H2P == 32 && H1 == 32 && F == FP) {
    // the matrix-core forms (see the kernel).  F = 32: one region of P floats per wave (wave_regions = 3);
    // F = 64: the tiles meet in two rounds, four regions of the larger round (wave_regions = 2)
    if (FP == 32) {
      wave_regions = 3;
      lds = sizeof(float) * P * 4;
    } else {
      const size_t r1 = 2 * (size_t)32 * F + 32, r2 = P - r1;
      wave_regions = 2;
      lds = sizeof(float) * 4 * (r1 > r2 ? r1 : r2);
    }
  }
  if constexpr ((MODE == 0 || MODE == 3) && FP == 32 && HP == 32 && H2P == 32) {
    if (F == 32 && H1 == 32 && wave_regions == 3) {   // the matrix-core form alone (see X32)
      auto kx = k_bptt_rows<FP, HP, H2P, MODE, true>;
      gcm_allow_dynamic_lds((const void*)kx, lds);
      hipLaunchKernelGGL(kx, dim3(grid), dim3(256), lds, s, tab, hs, n_steps, sb, sh, w_rel2, w_root2, act1, act2, lay,
                         slabs, B, N, F, H1, H2, deg_term, lrn, wave_regions);
      return gcm_launch_status();
    }
  }
  auto kern = k_bptt_rows<FP, HP, H2P, MODE>;
  gcm_allow_dynamic_lds((const void*)kern, lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, tab, hs, n_steps, sb, sh, w_rel2, w_root2, act1,
                     act2, lay, slabs, B, N, F, H1, H2, deg_term, lrn, wave_regions);
  return gcm_launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// The backward over records of CACHED steps per graph (round 6; cfg2's / cfg3's backward): F = 32 or 64, H1 = 32, H2 <= 32, N <= 128, up to 128
// steps (one launch).  k_bptt_rows<32, 32, 32, 3, true> walks the (step, graph) items with a persistent grid - every item
// behind its record's header, then its rows' loads from the caches - 40 us per cfg2 rollout at 2.5 us an item and wave.
// One workgroup per graph instead (the form of k_bptt_learned_graph below): the caches' rows in LDS once, eight waves, a wave
// every eighth step with two steps' record loads in flight, the same matrix-core arithmetic.  One slab per graph.
// ---------------------------------------------------------------------------------------------------------
template <int FT>   // F = 32 FT
__global__ __launch_bounds__(512) void k_bptt_cached_graph(StepTable tab, int T, long gmx_sb, long gmx_sh,
                                                           const float* __restrict__ w_rel2,
                                                           const float* __restrict__ w_root2, int act1, int act2,
                                                           SavedLayout lay, float* __restrict__ slabs, int B, int N, int H2,
                                                           LrnSrc lrn) {
  constexpr int NMAX = 128, F = 32 * FT, H1 = 32, RS = 33, AS = F + 1;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, q = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int act1_v = gcm_vgpr(act1), act2_v = gcm_vgpr(act2);
  constexpr int NW = 8;   // waves: two per SIMD (a wave alone on its SIMD issues a dependent instruction every ~8 cycles)
  extern __shared__ float sImg[];   // the graph's h1 [NMAX][RS] / agg1, x [NMAX][AS] rows; the epilogue's tiles later
  float* sH = sImg;
  float* sA = sH + NMAX * RS;
  float* sX = sA + NMAX * AS;
  // ---- every row of the graph's caches (a steady-state chain's are rings: any slot may be live) -----------------------------
  {
    f32x4 vh[2], va[2 * FT], vx[2 * FT];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 512 * i, r = e >> 3, c = (e & 7) * 4;
      vh[i] = *reinterpret_cast<const f32x4*>(lrn.c_h1 + ((size_t)b * N + (r < N ? r : N - 1)) * H1 + c);
    }
#pragma unroll
    for (int i = 0; i < 2 * FT; ++i) {
      const int e = tid + 512 * i, r = e / (F / 4), c = (e % (F / 4)) * 4;
      const size_t rj = ((size_t)b * N + (r < N ? r : N - 1)) * F + c;
      va[i] = *reinterpret_cast<const f32x4*>(lrn.c_agg1 + rj);
      vx[i] = *reinterpret_cast<const f32x4*>(lrn.c_nodes + rj);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 512 * i, r = e >> 3, c = (e & 7) * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) sH[r * RS + c + k] = vh[i][k];
    }
#pragma unroll
    for (int i = 0; i < 2 * FT; ++i) {
      const int e = tid + 512 * i, r = e / (F / 4), c = (e % (F / 4)) * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sA[r * AS + c + k] = va[i][k];
        sX[r * AS + c + k] = vx[i][k];
      }
    }
  }
  // layer 2's weights: lanes 0-31 column q of W_rel2, lanes 32-63 column q of W_root2 (dagg2[q] / dh1cur[q] = column . d2)
  float w2c[32];
  {
    const float* src = (half ? w_root2 : w_rel2) + q;
#pragma unroll
    for (int o = 0; o < 32; ++o) w2c[o] = src[(size_t)(o < H2 ? o : H2 - 1) * H1];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int o = 0; o < 32; ++o) w2c[o] = o < H2 ? w2c[o] : 0.f;
  }
  f32x16 aR[FT], aT[FT], aR2, aT2;   // dW_rel1 [h][32 ft + f], dW_root1, dW_rel2 [o][k], dW_root2 [o][k] of this wave's steps
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    aR2[r] = 0.f; aT2[r] = 0.f;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) { aR[ft][r] = 0.f; aT[ft][r] = 0.f; }
  }
  float db1 = 0.f, db2 = 0.f;
  __syncthreads();

  const int oq = q < H2 ? q : H2 - 1;
  struct Front { int L, l_cur, ja, jb; float cfa, cfb, g, y, vv; };
  auto front = [&](int t, Front& f) __attribute__((always_inline)) {
    const float* sv = tab.saved[t];
    const int* hdr = reinterpret_cast<const int*>(sv + lay.o_hdr) + 4 * b;
    f.L = hdr[0];
    f.l_cur = hdr[1];
    const int e0 = lane < N ? lane : N - 1, e1 = lane + 64 < N ? lane + 64 : N - 1;   // (entries >= L: unused)
    f.cfa = sv[lay.o_coef + (size_t)b * N + e0];
    f.cfb = sv[lay.o_coef + (size_t)b * N + e1];
    const int* live = reinterpret_cast<const int*>(sv + lay.o_live) + (size_t)b * N;
    f.ja = live[e0];
    f.jb = live[e1];
    const float* gp = tab.gmx[t];
    f.g = gp ? gp[(long)b * gmx_sb + (long)oq * gmx_sh] : 0.f;
    f.y = sv[(size_t)b * H2 + oq];
    f.vv = sv[lay.o_v + (size_t)b * 64 + lane];   // agg2 [32] | h1cur [32]
  };
  Front fa{}, fb{}, fc{};
  if (wave < T) front(wave, fa);
  if (wave + NW < T) front(wave + NW, fb);
#pragma unroll 1
  for (int t = wave; t < T; t += NW) {
    if (t + 2 * NW < T) front(t + 2 * NW, fc);   // (two of this wave's steps ahead: a step is shorter than a memory round trip)
    const int L = min(__builtin_amdgcn_readfirstlane(fa.L), NMAX);
    const int l_cur = __builtin_amdgcn_readfirstlane(fa.l_cur);
    // ---- layer 2: d2 = g act2'(y);  dW2 += d2 (x) [agg2 | h1cur];  dagg2 / dh1cur = W2^T d2 ------------------------------
    const float d2 = (q < H2 && L > 0) ? fa.g * gcm_act_grad_sel(fa.y, act2_v) : 0.f;   // (both halves hold d2[q]; L = 0: a
                                                                                       //  graph that got no node this step)
    db2 += half ? 0.f : d2;
    aR2 = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? 0.f : d2, fa.vv, aR2, 0, 0, 0);   // d2 (x) agg2
    aT2 = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? d2 : 0.f, fa.vv, aT2, 0, 0, 0);   // d2 (x) h1cur
    float u = 0.f;
#pragma unroll
    for (int o = 0; o < 32; ++o)
      u = fmaf(w2c[o], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d2), o)), u);
    float dagg2, dh1c;
    {
      const unsigned uu = __float_as_uint(u);
      const gcm_u32x2 r = __builtin_amdgcn_permlane32_swap(uu, uu, false, false);
      dagg2 = __uint_as_float(r[0]);   // the lower half's value (W_rel2 columns) in both halves
      dh1c = __uint_as_float(r[1]);    // the upper half's (W_root2 columns)
    }
    // ---- the live rows (the record's list), two per instruction ---------------------------------------------------------
#pragma unroll 1
    for (int l = 0; l < L; l += 2) {
      const int l1 = l + 1 < L ? l + 1 : l;
      const int j0 = l < 64 ? __builtin_amdgcn_readlane(fa.ja, l) : __builtin_amdgcn_readlane(fa.jb, l & 63);
      const int j1 = l1 < 64 ? __builtin_amdgcn_readlane(fa.ja, l1) : __builtin_amdgcn_readlane(fa.jb, l1 & 63);
      const float c0 = __int_as_float(l < 64 ? __builtin_amdgcn_readlane(__float_as_int(fa.cfa), l)
                                             : __builtin_amdgcn_readlane(__float_as_int(fa.cfb), l & 63));
      const float c1 = __int_as_float(l1 < 64 ? __builtin_amdgcn_readlane(__float_as_int(fa.cfa), l1)
                                              : __builtin_amdgcn_readlane(__float_as_int(fa.cfb), l1 & 63));
      const int j = (half ? j1 : j0) & (NMAX - 1), lm = l + half;
      const float hv = sH[j * RS + q];
      float g1 = ((half ? c1 : c0) * dagg2 + (lm == l_cur ? dh1c : 0.f)) * gcm_act_grad_sel(hv, act1_v);
      g1 = lm < L ? g1 : 0.f;
      db1 += g1;
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) {
        aR[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, sA[j * AS + 32 * ft + q], aR[ft], 0, 0, 0);
        aT[ft] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, sX[j * AS + 32 * ft + q], aT[ft], 0, 0, 0);
      }
    }
    fa = fb;
    fb = fc;
  }

  // ---- one slab per graph: dW_rel1 [H1*F] | dW_root1 [H1*F] | db1 [H1] | dW_rel2 [H2*H1] | dW_root2 [H2*H1] | db2 [H2] ----
  const int Pg = 2 * H1 * F + H1 + 2 * H2 * H1 + H2;
  float* slab = slabs + (size_t)b * Pg;
  float* sR = sImg;   // [NW][1024] (the images are dead: 3 x 4224 floats hold it)
  const int li = lane & 31, lh = lane >> 5;
  // (a 32 x 32 tile of this wave's accumulators -> slab[off + row * ld + column], the eight waves summed in order)
  auto tile_out = [&](const f32x16& acc, int off, int ld, int rows) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) sR[wave * 1024 + gcm_fused::acc_row(r, lh) * 32 + li] = acc[r];
    __syncthreads();
    for (int e = tid; e < 32 * rows; e += 512) {
      float t_ = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t_ += sR[w * 1024 + e];
      slab[off + (e >> 5) * ld + (e & 31)] = t_;
    }
  };
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) {
    tile_out(aR[ft], 32 * ft, F, H1);
    tile_out(aT[ft], H1 * F + 32 * ft, F, H1);
  }
  tile_out(aR2, 2 * H1 * F + H1, H1, H2);
  tile_out(aT2, 2 * H1 * F + H1 + H2 * H1, H1, H2);
  __syncthreads();
  {
    const float s1 = gcm_xor32_add(db1), s2 = gcm_xor32_add(db2);
    if (lane < 32) { sR[wave * 64 + lane] = s1; sR[wave * 64 + 32 + lane] = s2; }
  }
  __syncthreads();
  if (tid < 64) {
    float t_ = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t_ += sR[w * 64 + tid];
    if (tid < 32) slab[2 * H1 * F + tid] = t_;
    else if (tid - 32 < H2) slab[2 * H1 * F + H1 + 2 * H2 * H1 + tid - 32] = t_;
  }
}


int launch_bptt_cached_graph(hipStream_t s, const StepTable& tab, int n_steps, long gmx_sb, long gmx_sh, const float* w_rel2,
                             const float* w_root2, int act1, int act2, const SavedLayout& lay, float* slabs, int B, int N,
                             int F, int H2, const LrnSrc& caches) {
  const size_t lds = sizeof(float) * (128 * 33 + 2 * 128 * (size_t)(F + 1));
  if (F == 32) {
    gcm_allow_dynamic_lds((const void*)k_bptt_cached_graph<1>, lds);
    hipLaunchKernelGGL(k_bptt_cached_graph<1>, dim3(B), dim3(512), lds, s, tab, n_steps, gmx_sb, gmx_sh, w_rel2, w_root2, act1,
                       act2, lay, slabs, B, N, H2, caches);
  } else {
    gcm_allow_dynamic_lds((const void*)k_bptt_cached_graph<2>, lds);
    hipLaunchKernelGGL(k_bptt_cached_graph<2>, dim3(B), dim3(512), lds, s, tab, n_steps, gmx_sb, gmx_sh, w_rel2, w_root2, act1,
                       act2, lay, slabs, B, N, H2, caches);
  }
  return gcm_launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// Pass A of the LearnedEdge backward per GRAPH (round 6): every step of the backward a cached step of one chain (T <= 128),
// F = H1 = 32, H2 <= 32, N = 128.  k_bptt_rows<.., 2> gives every (step, graph) to a wave that starts behind four dependent
// loads and walks its live rows through 32 v_readlane + 64 v_fma each: 46 us per cfg5 chain, bound by VALU issue.  Here one
// workgroup (eight waves) owns a graph: its h1 / agg1 / x rows are staged in LDS once (every step reads the SAME caches), a
// wave takes every eighth step with its next two steps' record loads in flight, and the rank-1 updates run on the matrix cores - two live
// rows per v_mfma_f32_32x32x2_f32 (lanes 0-31 row ja, lanes 32-63 row jb: G1 as A, the rows' agg1 / x as B), layer 2's
// d2 (x) [agg2 | h1cur] as two instructions with d2 in one half.  What it hands to pass B1 per live row is G1 itself, not
// dAgg1 = W_rel1^T G1: the sum over the steps that aggregate a node commutes with the matrix, so k_learned_bptt_sel_graph
// multiplies ONCE per node (da_is_g1).  One slab per graph.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_bptt_learned_graph(StepTable tab, int T, long gmx_sb, long gmx_sh,
                                                            const float* __restrict__ w_rel2,
                                                            const float* __restrict__ w_root2, int act1, int act2,
                                                            float* __restrict__ slabs, int B, int H2, LrnSrc lrn) {
  constexpr int N = 128, F = 32, H1 = 32, RS = 33;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, q = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int act1_v = gcm_vgpr(act1), act2_v = gcm_vgpr(act2);
  constexpr int NW = 8;   // waves: two per SIMD (a wave alone on its SIMD issues a dependent instruction every ~8 cycles)
  __shared__ float sImg[3 * N * RS];   // the graph's h1 / agg1 / x rows; the epilogue's tiles later
  float* sH = sImg;
  float* sA = sH + N * RS;
  float* sX = sA + N * RS;
  // ---- the caches' rows 0 .. cur of the last step (the counts grow by one a step) ----------------------------------------
  int64_t cl = reinterpret_cast<const int64_t*>(tab.saved[T - 1] + lrn.o_idx)[b];
  const int n_rows = (int)(cl < 0 ? 0 : (cl > N - 1 ? N - 1 : cl)) + 1;
  {
    f32x4 vh[2], va[2], vx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 512 * i, r = e >> 3, c = (e & 7) * 4;
      const size_t rj = ((size_t)b * N + (r < n_rows ? r : n_rows - 1)) * F + c;
      vh[i] = *reinterpret_cast<const f32x4*>(lrn.c_h1 + rj);
      va[i] = *reinterpret_cast<const f32x4*>(lrn.c_agg1 + rj);
      vx[i] = *reinterpret_cast<const f32x4*>(lrn.c_nodes + rj);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 512 * i, r = e >> 3, c = (e & 7) * 4;
      const bool on = r < n_rows;   // (rows behind: never written, may hold anything)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        sH[r * RS + c + k] = on ? vh[i][k] : 0.f;
        sA[r * RS + c + k] = on ? va[i][k] : 0.f;
        sX[r * RS + c + k] = on ? vx[i][k] : 0.f;
      }
    }
  }
  // layer 2's weights: lanes 0-31 column q of W_rel2, lanes 32-63 column q of W_root2 (dagg2[q] / dh1cur[q] = column . d2)
  float w2c[32];
  {
    const float* src = (half ? w_root2 : w_rel2) + q;
#pragma unroll
    for (int o = 0; o < 32; ++o) w2c[o] = src[(size_t)(o < H2 ? o : H2 - 1) * H1];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int o = 0; o < 32; ++o) w2c[o] = o < H2 ? w2c[o] : 0.f;
  }
  f32x16 aR, aT, aR2, aT2;   // dW_rel1 [h][f], dW_root1 [h][f], dW_rel2 [o][k], dW_root2 [o][k] of this wave's steps
#pragma unroll
  for (int r = 0; r < 16; ++r) { aR[r] = 0.f; aT[r] = 0.f; aR2[r] = 0.f; aT2[r] = 0.f; }
  float db1 = 0.f, db2 = 0.f;
  __syncthreads();

  const int oq = q < H2 ? q : H2 - 1;
  struct Front { int cur; float a0, a1, g, y, agg2; };
  auto front = [&](int t, Front& f) __attribute__((always_inline)) {
    const float* sv = tab.saved[t];
    const int64_t c64 = reinterpret_cast<const int64_t*>(sv + lrn.o_idx)[b];
    f.cur = (int)(c64 < 0 ? 0 : (c64 > N - 1 ? N - 1 : c64));
    const float* arow = sv + lrn.o_adj + (size_t)b * N;   // (cached steps: the compact row)
    f.a0 = arow[lane];
    f.a1 = arow[lane + 64];
    const float* gp = tab.gmx[t];
    f.g = gp ? gp[(long)b * gmx_sb + (long)oq * gmx_sh] : 0.f;
    f.y = (sv + lrn.o_mx)[(size_t)b * H2 + oq];
    f.agg2 = (sv + lrn.o_agg2)[(size_t)b * H1 + q];
  };
  Front fa{}, fb{}, fc{};
  if (wave < T) front(wave, fa);
  if (wave + NW < T) front(wave + NW, fb);
#pragma unroll 1
  for (int t = wave; t < T; t += NW) {
    if (t + 2 * NW < T) front(t + 2 * NW, fc);   // (two of this wave's steps ahead: a step is shorter than a memory round trip)
    const int cur = __builtin_amdgcn_readfirstlane(fa.cur);
    const size_t it = (size_t)(lrn.s0 + t) * B + b;
    // ---- layer 2: d2 = g act2'(y);  dW2 += d2 (x) [agg2 | h1cur];  dagg2 / dh1cur = W2^T d2 ------------------------------
    const float d2 = q < H2 ? fa.g * gcm_act_grad_sel(fa.y, act2_v) : 0.f;   // (both halves hold d2[q])
    db2 += half ? 0.f : d2;
    const float h1c = sH[cur * RS + q];
    const float d2lo = half ? 0.f : d2;
    aR2 = __builtin_amdgcn_mfma_f32_32x32x2f32(d2lo, fa.agg2, aR2, 0, 0, 0);
    aT2 = __builtin_amdgcn_mfma_f32_32x32x2f32(d2lo, h1c, aT2, 0, 0, 0);
    float u = 0.f;
#pragma unroll
    for (int o = 0; o < 32; ++o)
      u = fmaf(w2c[o], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d2), o)), u);
    float dagg2, dh1c;
    {
      const unsigned uu = __float_as_uint(u);
      const gcm_u32x2 r = __builtin_amdgcn_permlane32_swap(uu, uu, false, false);
      dagg2 = __uint_as_float(r[0]);   // the lower half's value (W_rel2 columns) in both halves
      dh1c = __uint_as_float(r[1]);    // the upper half's (W_root2 columns)
    }
    if (lane < H1) lrn.dagg2[it * H1 + lane] = dagg2;
    // ---- the live rows, two per instruction --------------------------------------------------------------------------
    unsigned long long m0 = __ballot(fa.a0 != 0.f || lane == cur);
    unsigned long long m1 = __ballot(fa.a1 != 0.f || lane + 64 == cur);
    int n_live = 0;
#pragma unroll 1
    while (m0 | m1) {
      const int ja = m0 ? __builtin_ctzll(m0) : 64 + __builtin_ctzll(m1);
      if (m0) m0 &= m0 - 1; else m1 &= m1 - 1;
      const bool two = (m0 | m1) != 0;
      int jb = ja;
      if (two) {
        jb = m0 ? __builtin_ctzll(m0) : 64 + __builtin_ctzll(m1);
        if (m0) m0 &= m0 - 1; else m1 &= m1 - 1;
      }
      const int j = half ? jb : ja;
      const bool valid = !half || two;
      const float hv = sH[j * RS + q], ag = sA[j * RS + q], xx = sX[j * RS + q];
      const float cfa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ja < 64 ? fa.a0 : fa.a1), ja & 63));
      const float cfb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(jb < 64 ? fa.a0 : fa.a1), jb & 63));
      const float cf = half ? cfb : cfa;
      float g1 = (cf * dagg2 + (j == cur ? dh1c : 0.f)) * gcm_act_grad_sel(hv, act1_v);
      g1 = valid ? g1 : 0.f;
      db1 += g1;
      aR = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, ag, aR, 0, 0, 0);
      aT = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, xx, aT, 0, 0, 0);
      if (valid) {
        lrn.da[((it * N) + n_live + half) * F + q] = g1;   // (G1 of the row: da_is_g1)
        if (q == 0) lrn.live[it * N + n_live + half] = j;
      }
      n_live += two ? 2 : 1;
    }
    if (lane == 0) {
      lrn.hdr[2 * it] = cur;
      lrn.hdr[2 * it + 1] = n_live;
    }
    fa = fb;
    fb = fc;
  }

  // ---- one slab per graph: dW_rel1 [H1*F] | dW_root1 [H1*F] | db1 [H1] | dW_rel2 [H2*H1] | dW_root2 [H2*H1] | db2 [H2] ----
  const int Pg = 2 * H1 * F + H1 + 2 * H2 * H1 + H2;
  float* slab = slabs + (size_t)b * Pg;
  float* sR = sImg;   // [NW][1024] (the images are dead: 3 x 4224 floats hold it)
  const int li = lane & 31, lh = lane >> 5;
  auto tile_out = [&](const f32x16& acc, int off, int rows) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) sR[wave * 1024 + gcm_fused::acc_row(r, lh) * 32 + li] = acc[r];
    __syncthreads();
    for (int e = tid; e < 32 * rows; e += 512) {
      float t_ = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t_ += sR[w * 1024 + e];
      slab[off + e] = t_;
    }
  };
  tile_out(aR, 0, H1);
  tile_out(aT, H1 * F, H1);
  tile_out(aR2, 2 * H1 * F + H1, H2);
  tile_out(aT2, 2 * H1 * F + H1 + H2 * H1, H2);
  __syncthreads();
  {
    const float s1 = gcm_xor32_add(db1), s2 = gcm_xor32_add(db2);
    if (lane < 32) { sR[wave * 64 + lane] = s1; sR[wave * 64 + 32 + lane] = s2; }
  }
  __syncthreads();
  if (tid < 64) {
    float t_ = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t_ += sR[w * 64 + tid];
    if (tid < 32) slab[2 * H1 * F + tid] = t_;
    else if (tid - 32 < H2) slab[2 * H1 * F + H1 + 2 * H2 * H1 + tid - 32] = t_;
  }
}

// ... its launch (the caller has checked the shapes and that the chunk is the whole backward): B slabs
int launch_bptt_learned_graph(void* stream, const StepTable& tab, int n_steps, long gmx_sb, long gmx_sh, const float* w_rel2,
                              const float* w_root2, int act1, int act2, float* slabs, const LearnedSrc& src, int B, int H2) {
  LrnSrc l{src.o_adj, src.o_mx, src.o_h1, src.o_agg1, src.o_agg2, src.o_idx, src.w_rel1,
           src.hdr,   src.live, src.da,   src.dagg2,  src.s0, src.adj_compact,
           src.c_nodes, src.c_h1, src.c_agg1};
  hipLaunchKernelGGL(k_bptt_learned_graph, dim3(B), dim3(512), 0, (hipStream_t)stream, tab, n_steps, gmx_sb, gmx_sh, w_rel2,
                     w_root2, act1, act2, slabs, B, H2, l);
  return gcm_launch_status();
}

int launch_bptt_learned(void* stream, int grid, const StepTable& tab, int n_steps, long gmx_sb, long gmx_sh,
                        const float* w_rel2, const float* w_root2, int act1, int act2, float* slabs,
                        const LearnedSrc& src, int B, int N, int F, int H1, int H2) {
  LrnSrc l{src.o_adj, src.o_mx, src.o_h1, src.o_agg1, src.o_agg2, src.o_idx, src.w_rel1,
           src.hdr,   src.live, src.da,   src.dagg2,  src.s0, src.adj_compact,
           src.c_nodes, src.c_h1, src.c_agg1};
  return launch_bptt<32, 32, 32, 2>((hipStream_t)stream, grid, tab, Hist{}, n_steps, gmx_sb, gmx_sh, w_rel2,
                                    w_root2, act1, act2, SavedLayout{}, slabs, B, N, F, H1, H2, 0, l);
}

}  // namespace gcm_rows

extern "C" int gcm_dense_rows_bptt_slabs(int n_steps, int B) {
  if (n_steps <= 0 || B <= 0) return 0;
  const long items = (long)n_steps * B;
  const long chunks = (n_steps + GCM_ROWS_MAX_STEPS - 1) / GCM_ROWS_MAX_STEPS;
  // per launch: up to 512 workgroups of 4 waves (two per CU), at least one item per wave
  long per = (items / chunks + 3) / 4;
  if (per > 512) per = 512;
  if (per < 1) per = 1;
  return (int)(per * chunks);
}

extern "C" size_t gcm_dense_rows_bptt_workspace_bytes(int n_steps, int B, int F, int H1, int H2) {
  const size_t P = 2 * (size_t)H1 * F + 2 * H1 + 2 * (size_t)H2 * H1 + H2;   // with the dc1 section
  return sizeof(float) * P * (size_t)gcm_dense_rows_bptt_slabs(n_steps, B);
}

/* saved_host / gmx_host: HOST arrays of n_steps device pointers (the record each step's forward
 * wrote, and that step's g_mx [B, H2] with element strides gmx_stride_b / gmx_stride_h - an expanded
 * gradient has stride 0).  g_params = g_params_prev (NULL = 0) + the parameter gradient. */
static int rows_bptt_impl(const float* const* saved_host, const float* const* gmx_host, int n_steps,
                          long gmx_stride_b, long gmx_stride_h, const float* params, int has_bias, int act1, int act2,
                          const float* cache_nodes, const float* cache_h1, const float* cache_agg1,
                          const float* g_params_prev, float* g_params, void* workspace, size_t workspace_bytes, int B,
                          int N, int F, int H1, int H2, gcm_stream_t stream);

extern "C" int gcm_dense_rows_bptt(const float* const* saved_host, const float* const* gmx_host,
                                   int n_steps, long gmx_stride_b, long gmx_stride_h,
                                   const float* params, int has_bias, int act1, int act2,
                                   const float* g_params_prev, float* g_params, void* workspace,
                                   size_t workspace_bytes, int B, int N, int F, int H1, int H2,
                                   gcm_stream_t stream) {
  return rows_bptt_impl(saved_host, gmx_host, n_steps, gmx_stride_b, gmx_stride_h, params, has_bias, act1, act2,
                        nullptr, nullptr, nullptr, g_params_prev, g_params, workspace, workspace_bytes, B, N, F, H1,
                        H2, stream);
}

/* The same for the records of CACHED steps (rows_cached.hip: gcm_dense_rows_step_cached): the live rows'
 * h1 | agg1 | x come from the chain's caches. */
extern "C" int gcm_dense_rows_bptt_cached(const float* const* saved_host, const float* const* gmx_host, int n_steps,
                                          long gmx_stride_b, long gmx_stride_h, const float* params, int has_bias,
                                          int act1, int act2, const float* cache_nodes, const float* cache_h1,
                                          const float* cache_agg1, const float* g_params_prev, float* g_params,
                                          void* workspace, size_t workspace_bytes, int B, int N, int F, int H1,
                                          int H2, gcm_stream_t stream) {
  GCM_REQUIRE(cache_nodes && cache_h1 && cache_agg1);
  if (has_bias & (GCM_GNN_HAS_DEG_TERM | GCM_GNN_HAS_PE_TABLE)) return GCM_EUNSUPPORTED;
  return rows_bptt_impl(saved_host, gmx_host, n_steps, gmx_stride_b, gmx_stride_h, params, has_bias, act1, act2,
                        cache_nodes, cache_h1, cache_agg1, g_params_prev, g_params, workspace, workspace_bytes, B, N,
                        F, H1, H2, stream);
}

static int rows_bptt_impl(const float* const* saved_host, const float* const* gmx_host, int n_steps,
                          long gmx_stride_b, long gmx_stride_h, const float* params, int has_bias, int act1, int act2,
                          const float* cache_nodes, const float* cache_h1, const float* cache_agg1,
                          const float* g_params_prev, float* g_params, void* workspace, size_t workspace_bytes, int B,
                          int N, int F, int H1, int H2, gcm_stream_t stream) {
  GCM_REQUIRE(saved_host && gmx_host && params && g_params && workspace);
  GCM_REQUIRE(n_steps > 0 && B > 0);
  // (records of cached steps: the kernel only indexes the caches by row - any graph size; SparseGCM's stepwise use)
  if (cache_h1 ? (N <= 0 || F <= 0 || F > 64 || H1 <= 0 || H1 > 64 || H2 <= 0 || H2 > 64)
               : !gcm_dense_rows_supported(N, F, H1, H2))
    return GCM_EUNSUPPORTED;
  if (workspace_bytes < gcm_dense_rows_bptt_workspace_bytes(n_steps, B, F, H1, H2))
    return GCM_EWORKSPACE;
  const int deg_term = (has_bias & GCM_GNN_HAS_DEG_TERM) ? 1 : 0;
  const size_t P = 2 * (size_t)H1 * F + H1 + 2 * (size_t)H2 * H1 + H2 + (deg_term ? H1 : 0);
  const float* w_rel2 = params + 2 * (size_t)H1 * F + H1;
  const float* w_root2 = w_rel2 + (size_t)H2 * H1;
  gcm_rows::SavedLayout lay = gcm_rows::make_layout(B, N, F, H1, H2);
  gcm_rows::LrnSrc caches{};
  if (cache_h1) {   // cached records: v / hdr / coef at the same offsets, the live list behind coef, no rows section
    lay.o_live = gcm_rows::make_cached_layout(B, N, H1, H2).o_live;
    caches.c_nodes = cache_nodes;
    caches.c_h1 = cache_h1;
    caches.c_agg1 = cache_agg1;
  }
  hipStream_t s = (hipStream_t)stream;
  const int fp = F <= 32 ? 32 : 64, hp = H1 <= 32 ? 32 : 64, h2p = H2 <= 32 ? 32 : 64;
  const int chunks = (n_steps + GCM_ROWS_MAX_STEPS - 1) / GCM_ROWS_MAX_STEPS;
  const int total_slabs = gcm_dense_rows_bptt_slabs(n_steps, B);
  const int per = total_slabs / chunks;
  float* slabs = (float*)workspace;
  // records of cached steps at F = 32 or 64, H1 = 32 on graphs of <= 128 nodes, one launch's worth: per GRAPH (k_bptt_cached_graph: the
  // caches' rows in LDS once; B slabs).  Any other bit in has_bias - GCM_STEP_FOUR_WAVES is the A/B switch - keeps the
  // per-item kernel.
  if (cache_h1 && cache_nodes && cache_agg1 && chunks == 1 && (F == 32 || F == 64) && H1 == 32 && H2 <= 32 && N <= 128 && !deg_term &&
      !(has_bias & ~(3 | GCM_BPTT_MANY_ROWS)) && B <= total_slabs) {
    gcm_rows::StepTable tab{};
    for (int i = 0; i < n_steps; ++i) {
      GCM_REQUIRE(saved_host[i] && gmx_host[i]);
      tab.saved[i] = saved_host[i];
      tab.gmx[i] = gmx_host[i];
    }
    const int rc = gcm_rows::launch_bptt_cached_graph(s, tab, n_steps, gmx_stride_b, gmx_stride_h, w_rel2, w_root2, act1,
                                                      act2, lay, slabs, B, N, F, H2, caches);
    if (rc) return rc;
    return gcm_sum_slabs_acc(slabs, B, (int)P, g_params_prev, g_params, stream);
  }
  for (int c = 0; c < chunks; ++c) {
    const int s0 = c * GCM_ROWS_MAX_STEPS;
    const int ns = n_steps - s0 < GCM_ROWS_MAX_STEPS ? n_steps - s0 : GCM_ROWS_MAX_STEPS;
    gcm_rows::StepTable tab{};
    for (int i = 0; i < ns; ++i) {
      GCM_REQUIRE(saved_host[s0 + i] && gmx_host[s0 + i]);
      tab.saved[i] = saved_host[s0 + i];
      tab.gmx[i] = gmx_host[s0 + i];
    }
    float* sl = slabs + (size_t)c * per * P;
    int rc = GCM_EUNSUPPORTED;
    // GCM_BPTT_MANY_ROWS: the caller knows the records hold many live rows per graph (DenseEdge): the deep-prefetch
    // form of the matrix-core pass, where it exists
    if ((has_bias & GCM_BPTT_MANY_ROWS) && !cache_h1 && !deg_term && F == 32 && H1 == 32 && H2 <= 32) {
      rc = gcm_rows::launch_bptt<32, 32, 32, 4>(s, per, tab, gcm_rows::Hist{}, ns, gmx_stride_b, gmx_stride_h, w_rel2,
                                                w_root2, act1, act2, lay, sl, B, N, F, H1, H2, 0);
      if (rc) return rc;
      continue;
    }
#define GCM_RB(a, b_, cc)                                                                          \
  if (fp == a && hp == b_ && h2p == cc)                                                            \
    rc = cache_h1 ? gcm_rows::launch_bptt<a, b_, cc, 3>(s, per, tab, gcm_rows::Hist{}, ns, gmx_stride_b,   \
                                                        gmx_stride_h, w_rel2, w_root2, act1, act2, lay, sl, B, \
                                                        N, F, H1, H2, 0, caches)                               \
                  : gcm_rows::launch_bptt<a, b_, cc, 0>(s, per, tab, gcm_rows::Hist{}, ns, gmx_stride_b,   \
                                                        gmx_stride_h, w_rel2, w_root2, act1, act2, lay, sl, B, \
                                                        N, F, H1, H2, deg_term);
    GCM_RB(32, 32, 32) GCM_RB(32, 32, 64) GCM_RB(32, 64, 32) GCM_RB(32, 64, 64)
    GCM_RB(64, 32, 32) GCM_RB(64, 32, 64) GCM_RB(64, 64, 32) GCM_RB(64, 64, 64)
#undef GCM_RB
    if (rc) return rc;
  }
  return gcm_sum_slabs_acc(slabs, total_slabs, (int)P, g_params_prev, g_params, stream);
}

/* Parameter gradient of a DenseGCM.rollout from the history its forward kept (gcm_dense_rollout_fwd /
 * gcm_dense_rollout_persistent_fwd), for the case that neither the observations nor the initial node
 * matrix need a gradient: every graph-step is independent, ONE launch over all T*B of them reading
 * only the live ROWS (no reverse scan of the node gradient, no per-step Q array).
 * adj_all [T+1,B,N,N], nodes_all [T+1,B,N,F] (slot t+1 = state after step t), cur_all [T,B],
 * mx_all / h1_all / agg1_all / agg2_all as written by the forward; g_mx_all with element strides. */
extern "C" size_t gcm_dense_rollout_bwd_params_workspace_bytes(int T, int B, int F, int H1, int H2) {
  return gcm_dense_rows_bptt_workspace_bytes(T < GCM_ROWS_MAX_STEPS ? T : GCM_ROWS_MAX_STEPS, B, F, H1, H2);
}

extern "C" int gcm_dense_rollout_bwd_params(const float* g_mx_all, long gmx_stride_t, long gmx_stride_b,
                                            long gmx_stride_h, const float* nodes_all,
                                            const float* adj_all, const int64_t* cur_all,
                                            const float* params, int act1, int act2,
                                            const float* mx_all, const float* h1_all,
                                            const float* agg1_all, const float* agg2_all,
                                            float* g_params, void* workspace, size_t workspace_bytes,
                                            int T, int B, int N, int F, int H1, int H2,
                                            gcm_stream_t stream) {
  GCM_REQUIRE(g_mx_all && nodes_all && adj_all && cur_all && params && mx_all && h1_all && agg1_all &&
              agg2_all && g_params && workspace);
  GCM_REQUIRE(T > 0 && B > 0 && (long)T * B < (1l << 31));
  if (!gcm_dense_rows_supported(N, F, H1, H2)) return GCM_EUNSUPPORTED;
  if (workspace_bytes < gcm_dense_rollout_bwd_params_workspace_bytes(T, B, F, H1, H2)) return GCM_EWORKSPACE;
  const size_t P = 2 * (size_t)H1 * F + H1 + 2 * (size_t)H2 * H1 + H2;
  const float* w_rel2 = params + 2 * (size_t)H1 * F + H1;
  const float* w_root2 = w_rel2 + (size_t)H2 * H1;
  const gcm_rows::SavedLayout lay = gcm_rows::make_layout(B, N, F, H1, H2);
  gcm_rows::Hist hs{adj_all + (size_t)B * N * N, nodes_all + (size_t)B * N * F, h1_all, agg1_all, agg2_all,
                    mx_all, g_mx_all, cur_all, gmx_stride_t};
  // one launch over all T*B items (the step index is item / B): grid as for 64 steps
  const int grid = gcm_dense_rows_bptt_slabs(T < GCM_ROWS_MAX_STEPS ? T : GCM_ROWS_MAX_STEPS, B);
  hipStream_t s = (hipStream_t)stream;
  const int fp = F <= 32 ? 32 : 64, hp = H1 <= 32 ? 32 : 64, h2p = H2 <= 32 ? 32 : 64;
  float* slabs = (float*)workspace;
  int rc = GCM_EUNSUPPORTED;
  const gcm_rows::StepTable none{};
#define GCM_RH(a, b_, cc)                                                                          \
  if (fp == a && hp == b_ && h2p == cc)                                                            \
    rc = gcm_rows::launch_bptt<a, b_, cc, 1>(s, grid, none, hs, T, gmx_stride_b, gmx_stride_h,  \
                                                w_rel2, w_root2, act1, act2, lay, slabs, B, N, F,  \
                                                H1, H2);
  GCM_RH(32, 32, 32) GCM_RH(32, 32, 64) GCM_RH(32, 64, 32) GCM_RH(32, 64, 64)
  GCM_RH(64, 32, 32) GCM_RH(64, 32, 64) GCM_RH(64, 64, 32) GCM_RH(64, 64, 64)
#undef GCM_RH
  if (rc) return rc;
  (void)P;
  return gcm_sum_slabs(slabs, grid, (int)P, g_params, stream);
}

/* ---- gradient w.r.t. the observations / the incoming node matrix of a chain of live-row steps ----------- */
extern "C" int gcm_dense_rows_dx_supported(int N, int F, int H1, int H2) {
  return gcm_dense_rows_supported(N, F, H1, H2) && F <= 64 && H1 <= 32 && H2 <= 32;
}

namespace gcm_rows {

// The observations / incoming nodes of a chain of live-row steps need a gradient too (records written with
// GCM_GNN_RECORD_DX).  The node matrix x enters a step through layer 1 only - as the root term of the live rows
// and through the aggregate of their adjacency rows - so per step the gradient w.r.t. x is sparse in rows:
//     dx[j_l] += G1_l W_root1 ;   dx[k] += adj'[j_l, k] * (G1_l W_rel1)   for the non-zero k of live row l
// (a dozen rows of F floats for TemporalBackedge([1,2,4])).  Row k of step s holds the node that was inserted
// at step s - (cur - k) of the chain (one roll per step once the graph is full) - or, when that is negative,
// an initial node of the state the chain started from.  The backward of a chain runs its steps LAST TO FIRST
// (autograd has one node per step here: an observation's producer is younger than the chain's head, so no
// single node could reach them all), one launch per step: the rows are added straight into the accumulators
// of the nodes they belong to - gx [T, B, F], one slot per step's observation, complete when that step's own
// launch has run - and gn0 [B, N, F] for the initial nodes.  No [B,N,F] gradient tensor per step, no reverse
// scan kernel.  (The parameter gradient of those steps: one time-parallel k_bptt_rows launch at the end.)
// The dx part of the backward of ONE step, as its own small kernel: one wave per graph, every
// load that does not depend on another one issued before the first use (weights, the step's vectors, the first
// eight live rows with their adjacency rows), the rows summed in the wave's LDS tile, one batch of
// read-modify-writes on the accumulators at the end.  H1, H2 <= 32, F <= 64.
// ATOMIC: many steps of a chain run concurrently (k_rows_dx_all) - the accumulators take hardware float adds
// the four weight matrices as the lanes hold them: lane h the columns h of W_rel2 / W_root2 (rows o), lane f the
// columns f of W_rel1 / W_root1 (rows h)
struct DxWeights {
  float w2a[32], w2r[32], wa[32], wr[32];
  __device__ __forceinline__ void load(const float* __restrict__ w_rel1, const float* __restrict__ w_root1,
                                       const float* __restrict__ w_rel2, const float* __restrict__ w_root2, int lane,
                                       int F, int H1, int H2) {
    const int hc = lane < H1 ? lane : H1 - 1, fc = lane < F ? lane : F - 1;
#pragma unroll
    for (int o = 0; o < 32; ++o) {
      const size_t i2 = (size_t)(o < H2 ? o : H2 - 1) * H1 + hc, i1 = (size_t)(o < H1 ? o : H1 - 1) * F + fc;
      const float t0 = w_rel2[i2], t1 = w_root2[i2], t2 = w_rel1[i1], t3 = w_root1[i1];
      const bool k2 = o < H2 && lane < H1, k1 = o < H1 && lane < F;   // (masked once, here)
      w2a[o] = k2 ? t0 : 0.f;
      w2r[o] = k2 ? t1 : 0.f;
      wa[o] = k1 ? t2 : 0.f;
      wr[o] = k1 ? t3 : 0.f;
    }
  }
};

// CACHED: the record of a cached step (rows_cached.hip) - no rows / arows sections: the chain started from empty
// graphs, so row cur IS the chain index s_lin, the live rows are cur - hop (E: forward hops only), their h1 rows
// sit in the chain's cache and their adjacency rows follow from the hop table.
template <bool ATOMIC, bool CACHED = false>
__device__ __forceinline__ void rows_dx_body(
    const float* __restrict__ sv, const float* __restrict__ gmx, long gmx_sb, long gmx_sh,
    const float* __restrict__ gnodes, const DxWeights& W, int act1, int act2, const SavedLayout& lay,
    const int64_t* __restrict__ count0, float* gx, float* gn0, int s_lin, int b, int lane, float* tile_, int B, int N,
    int F, int H1, int H2, const gcm_fused::Edits* Ep = nullptr, const float* __restrict__ cH = nullptr) {
  typedef __attribute__((address_space(3))) float lds_float;
  lds_float* const tile = (lds_float*)tile_;   // (the wave's accumulator tile: ds_ instructions, not flat ones)
  const float (&w2a)[32] = W.w2a;
  const float (&w2r)[32] = W.w2r;
  const float (&wa)[32] = W.wa;
  const float (&wr)[32] = W.wr;
  const int act1_v = gcm_vgpr(act1), act2_v = gcm_vgpr(act2);
  constexpr int LB = 8;   // live rows fetched ahead
  // ---- loads ---------------------------------------------------------------------------------------
  const int oc = lane < H2 ? lane : H2 - 1, hc = lane < H1 ? lane : H1 - 1;
  const float g = gmx ? gmx[(long)b * gmx_sb + (long)oc * gmx_sh] : 0.f;
  const float y = sv[(size_t)b * H2 + oc];
  const int* live = reinterpret_cast<const int*>(sv + lay.o_live) + (size_t)b * N;
  const float* coef = sv + lay.o_coef + (size_t)b * N;
  float cf8[LB], hv8[LB], r0[LB], r1[LB];
  int jl8[LB];
  int L, l_cur, cur, n0;
  unsigned long long lm0 = 0, lm1 = 0;   // CACHED: the live rows beyond the first LB
  bool self = false;
  // adjacency entry (j, k) of a chain whose selectors are the forward hops of E: k = j - hop (hop 0: the self loop)
  auto hop_edge = [&](int j, int k) __attribute__((always_inline)) {
    bool e = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) e = e || (i < Ep->n_hops && Ep->hops[i] >= 0 && k >= 0 && j - Ep->hops[i] == k);
    return e;
  };
  if (CACHED) {
    cur = s_lin;
    n0 = 0;
    unsigned long long m0 = 0, m1 = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int h = Ep->hops[i], j = cur - h;
      const bool use = i < Ep->n_hops && h >= 0 && j >= 0;
      self = self || (use && h == 0);
      const bool edge = use && h > 0;
      m0 |= (edge && j < 64) ? 1ull << (j & 63) : 0ull;
      m1 |= (edge && j >= 64) ? 1ull << ((j - 64) & 63) : 0ull;
    }
    m0 |= cur < 64 ? 1ull << cur : 0ull;
    m1 |= cur >= 64 ? 1ull << (cur - 64) : 0ull;
    L = __popcll(m0) + __popcll(m1);
    l_cur = L - 1;                          // row cur is the largest index of the list
#pragma unroll
    for (int l = 0; l < LB; ++l) {
      const bool any = (m0 | m1) != 0;
      const int j = !any ? 0 : (m0 ? __builtin_ctzll(m0) : 64 + __builtin_ctzll(m1));
      const bool low = m0 != 0;
      m0 &= low ? m0 - 1 : m0;
      m1 &= (low || !any) ? m1 : m1 - 1;
      jl8[l] = j;
      cf8[l] = (j == cur && !self) ? 0.f : 1.f;
      hv8[l] = cH[((size_t)b * N + j) * H1 + hc];
      r0[l] = hop_edge(j, lane) ? 1.f : 0.f;
      r1[l] = hop_edge(j, lane + 64) ? 1.f : 0.f;
    }
    lm0 = m0;
    lm1 = m1;
    asm volatile("" ::: "memory");
  } else {
    const int* hdr = reinterpret_cast<const int*>(sv + lay.o_hdr) + 4 * b;
    const int h_l = hdr[0], h_lc = hdr[1], h_cur = hdr[2];
    const int64_t c0 = count0[b];
#pragma unroll
    for (int l = 0; l < LB; ++l) {
      const float* row = sv + lay.o_rows + ((size_t)b * N + l) * lay.rw;
      const float* ar = sv + lay.o_arows + ((size_t)b * N + l) * N;
      cf8[l] = coef[l];
      jl8[l] = live[l];
      hv8[l] = row[hc];
      r0[l] = ar[lane < N ? lane : N - 1];
      r1[l] = ar[lane + 64 < N ? lane + 64 : N - 1];
    }
    asm volatile("" ::: "memory");
    L = __builtin_amdgcn_readfirstlane(h_l);
    l_cur = __builtin_amdgcn_readfirstlane(h_lc);
    cur = __builtin_amdgcn_readfirstlane(h_cur);
    n0 = __builtin_amdgcn_readfirstlane((int)(c0 < 0 ? 0 : (c0 > N ? N : c0)));
  }
  // ---- layer-2 adjoint: d2 in lane o, dagg2 / dh1cur in lane h ---------------------------------------
  const float d2 = lane < H2 ? g * act_grad_sel(y, act2_v) : 0.f;
  float dagg2 = 0.f, dh1c = 0.f;
#pragma unroll
  for (int o = 0; o < 32; ++o) {
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d2), o));
    dagg2 = fmaf(w2a[o], d, dagg2);
    dh1c = fmaf(w2r[o], d, dh1c);
  }
  // ---- the live rows -----------------------------------------------------------------------------------
  unsigned long long tm0 = 0, tm1 = 0;
  auto add_row = [&](int k, float val) __attribute__((always_inline)) {
    // (value selects throughout: with `k < 64 ? tm0 : tm1` written as branches the compiler keeps both masks
    //  in scratch memory and selects an address)
    const unsigned long long bit = 1ull << (k & 63);
    const unsigned long long m = k < 64 ? tm0 : tm1;
    const bool seen = (m & bit) != 0;
    if (lane < F) {
      lds_float* q = tile + k * F + lane;
      *q = (seen ? *q : 0.f) + val;
    }
    tm0 |= k < 64 ? bit : 0ull;
    tm1 |= k < 64 ? 0ull : bit;
  };
  auto live_row = [&](int l, float cf, float hv, int jl, float a0, float a1) __attribute__((always_inline)) {
    float g1 = (cf * dagg2 + (l == l_cur ? dh1c : 0.f)) * act_grad_sel(hv, act1_v);
    g1 = lane < H1 ? g1 : 0.f;
    float dxa = 0.f, dxr = 0.f;
#pragma unroll
    for (int h = 0; h < 32; ++h) {
      const float gh = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g1), h));
      dxa = fmaf(gh, wa[h], dxa);
      dxr = fmaf(gh, wr[h], dxr);
    }
    add_row(__builtin_amdgcn_readfirstlane(jl), dxr);
    unsigned long long z0 = __ballot(lane < N && a0 != 0.f), z1 = __ballot(lane + 64 < N && a1 != 0.f);
    while (z0 | z1) {
      const int k = z0 ? __builtin_ctzll(z0) : 64 + __builtin_ctzll(z1);
      if (z0) z0 &= z0 - 1; else z1 &= z1 - 1;
      const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(k < 64 ? a0 : a1), k & 63));
      add_row(k, a * dxa);
    }
  };
#pragma unroll
  for (int l = 0; l < LB; ++l)
    if (l < L) live_row(l, cf8[l], hv8[l], jl8[l], r0[l], r1[l]);
  if (CACHED) {
#pragma unroll 1
    for (int l = LB; l < L; ++l) {
      const int j = lm0 ? __builtin_ctzll(lm0) : 64 + __builtin_ctzll(lm1);
      if (lm0) lm0 &= lm0 - 1; else lm1 &= lm1 - 1;
      live_row(l, (j == cur && !self) ? 0.f : 1.f, cH[((size_t)b * N + j) * H1 + hc], j, hop_edge(j, lane) ? 1.f : 0.f,
               hop_edge(j, lane + 64) ? 1.f : 0.f);
    }
  } else {
#pragma unroll 1
    for (int l = LB; l < L; ++l) {
      const float* row = sv + lay.o_rows + ((size_t)b * N + l) * lay.rw;
      const float* ar = sv + lay.o_arows + ((size_t)b * N + l) * N;
      live_row(l, coef[l], row[hc], live[l], ar[lane < N ? lane : N - 1], ar[lane + 64 < N ? lane + 64 : N - 1]);
    }
  }
  if (gnodes) {   // a gradient handed to the node matrix this step returned: every row up to cur
    for (int k = 0; k <= cur; ++k) add_row(k, lane < F ? gnodes[((size_t)b * N + k) * F + lane] : 0.f);
    // ... and the rows beyond, which no step has touched yet: still the rows of the state the chain started from
    if (gn0 && lane < F)
      for (int k = cur + 1; k < N; ++k) {
        float* q = gn0 + ((size_t)b * N + k) * F + lane;
        const float v = gnodes[((size_t)b * N + k) * F + lane];
        if (ATOMIC) unsafeAtomicAdd(q, v);
        else *q += v;
      }
  }
  // ---- row k holds the node inserted at chain step s_lin - (cur - k) (negative: an initial node) -------
  constexpr int NONE = -(1 << 30);
  auto target = [&](int kk) __attribute__((always_inline)) -> float* {
    if (kk >= 0) return gx + ((size_t)kk * B + b) * F + lane;
    if (kk != NONE && gn0 && kk + n0 >= 0) return gn0 + ((size_t)b * N + (kk + n0)) * F + lane;
    return nullptr;
  };
  while (tm0 | tm1) {
    int kk8[8];
    float v8[8], old8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      kk8[i] = NONE;
      v8[i] = 0.f;
      if (tm0 | tm1) {
        const int k = tm0 ? __builtin_ctzll(tm0) : 64 + __builtin_ctzll(tm1 | (1ull << 63));
        const bool low = tm0 != 0;
        tm0 &= low ? tm0 - 1 : tm0;
        tm1 &= low ? tm1 : tm1 - 1;
        kk8[i] = s_lin - (cur - k);
        if (lane < F) v8[i] = tile[k * F + lane];
      }
    }
    if (ATOMIC) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float* q = target(kk8[i]);
        if (q && lane < F) unsafeAtomicAdd(q, v8[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float* q = target(kk8[i]);
        old8[i] = (q && lane < F) ? *q : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float* q = target(kk8[i]);
        if (q && lane < F) *q = old8[i] + v8[i];
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_rows_dx_step(
    const float* __restrict__ sv, const float* __restrict__ gmx, long gmx_sb, long gmx_sh,
    const float* __restrict__ gnodes, const float* __restrict__ w_rel1, const float* __restrict__ w_root1,
    const float* __restrict__ w_rel2, const float* __restrict__ w_root2, int act1, int act2, SavedLayout lay,
    const int64_t* __restrict__ count0, float* gx, float* gn0, int s_lin, int B, int N, int F, int H1, int H2) {
  extern __shared__ float dxs[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  DxWeights W;
  W.load(w_rel1, w_root1, w_rel2, w_root2, lane, F, H1, H2);
  rows_dx_body<false>(sv, gmx, gmx_sb, gmx_sh, gnodes, W, act1, act2, lay, count0, gx, gn0, s_lin, b, lane,
                      dxs + (size_t)wave * N * F, B, N, F, H1, H2);
}

// device pointers of up to GCM_ROWS_MAX_STEPS consecutive steps of a chain (NULL: that step has none)
struct DxTable {
  const float* saved[GCM_ROWS_MAX_STEPS];
  const float* gmx[GCM_ROWS_MAX_STEPS];
  const float* gn[GCM_ROWS_MAX_STEPS];
};

// every (step, graph) of the table, one wave per item, the waves of a grid sized to the machine striding over
// the items with the weights in their registers; steps with neither gradient are skipped
template <bool CACHED>
__global__ __launch_bounds__(256) void k_rows_dx_all(
    DxTable tab, int n_steps, int s0, long gmx_sb, long gmx_sh, const float* __restrict__ w_rel1,
    const float* __restrict__ w_root1, const float* __restrict__ w_rel2, const float* __restrict__ w_root2, int act1,
    int act2, SavedLayout lay, const int64_t* __restrict__ count0, float* gx, float* gn0, int B, int N, int F, int H1,
    int H2, gcm_fused::Edits E, const float* __restrict__ cH) {
  extern __shared__ float dxs[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  DxWeights W;
  W.load(w_rel1, w_root1, w_rel2, w_root2, lane, F, H1, H2);
  const int items = n_steps * B;
#pragma unroll 1
  for (int item = blockIdx.x * 4 + wave; item < items; item += gridDim.x * 4) {
    const int s = item / B, b = item - s * B;
    const float* gm = tab.gmx[s];
    const float* gn = tab.gn[s];
    if (!gm && !gn) continue;
    rows_dx_body<true, CACHED>(tab.saved[s], gm, gmx_sb, gmx_sh, gn, W, act1, act2, lay, count0, gx, gn0, s0 + s, b,
                               lane, dxs + (size_t)wave * N * F, B, N, F, H1, H2, &E, cH);
  }
}

}  // namespace gcm_rows

/* The dx part of the backward of ONE step of a chain (record written with GCM_GNN_RECORD_DX); the steps of a
 * chain are handed over last to first.  g_mx [B, H2] with element strides, or NULL; g_nodes_out: the gradient
 * handed to the node matrix this step returned ([B,N,F] contiguous) or NULL.  gx [T, B, F] (zeroed by the
 * caller before the first launch of the chain's backward): slot t is the gradient w.r.t. the observation of
 * step t, complete once step t itself has been handed over.  gn0 [B, N, F] (zeroed) or NULL: the node matrix
 * the chain started from; count0 [B]: num_nodes entering its first step.  s_lin: index of this step in the
 * chain.  (The parameter gradient of the chain's steps: gcm_dense_rows_bptt over their records - the dx
 * sections sit behind the others.) */
extern "C" int gcm_dense_rows_bptt_dx_step(const float* saved, const float* g_mx, long gmx_stride_b,
                                           long gmx_stride_h, const float* g_nodes_out, const float* params,
                                           int has_bias, int act1, int act2, const int64_t* count0, float* gx,
                                           float* gn0, int s_lin, int B, int N, int F, int H1, int H2,
                                           gcm_stream_t stream) {
  GCM_REQUIRE(saved && params && count0 && gx && B > 0 && s_lin >= 0);
  if (!gcm_dense_rows_dx_supported(N, F, H1, H2) || (has_bias & (GCM_GNN_HAS_DEG_TERM | GCM_GNN_HAS_PE_TABLE)))
    return GCM_EUNSUPPORTED;
  const float* w_rel2 = params + 2 * (size_t)H1 * F + H1;
  const float* w_root2 = w_rel2 + (size_t)H2 * H1;
  const gcm_rows::SavedLayout lay = gcm_rows::make_layout(B, N, F, H1, H2, true);
  const size_t lds = sizeof(float) * 4 * (size_t)N * F;
  gcm_allow_dynamic_lds((const void*)gcm_rows::k_rows_dx_step, lds);
  hipLaunchKernelGGL(gcm_rows::k_rows_dx_step, dim3((B + 3) / 4), dim3(256), lds, (hipStream_t)stream, saved, g_mx,
                     gmx_stride_b, gmx_stride_h, g_nodes_out, params, params + (size_t)H1 * F, w_rel2, w_root2, act1,
                     act2, lay, count0, gx, gn0, s_lin, B, N, F, H1, H2);
  return gcm_launch_status();
}

/* The same for up to GCM_ROWS_MAX_STEPS consecutive steps of a chain in ONE launch (one wave per step and
 * graph): saved / g_mx / g_nodes_out are HOST arrays of n_steps device pointers (g_mx[s], g_nodes_out[s] NULL:
 * none; every g_mx with the same element strides), s0 the chain index of the first.  The steps run
 * concurrently, so gx / gn0 (zeroed by the caller) take hardware float atomic adds: the order of summation -
 * and with it the last bits of the result - is not fixed; gcm_dense_rows_bptt_dx_step, handed the steps last
 * to first, is the ordered form. */
static int rows_dx_all_impl(const float* const* saved, const float* const* g_mx, long gmx_stride_b,
                            long gmx_stride_h, const float* const* g_nodes_out, int n_steps, int s0,
                            const float* params, int has_bias, int act1, int act2, const int64_t* count0, float* gx,
                            float* gn0, const gcm_selector_desc* selectors, int n_selectors, const float* cache_h1,
                            int B, int N, int F, int H1, int H2, gcm_stream_t stream);

extern "C" int gcm_dense_rows_bptt_dx_all(const float* const* saved, const float* const* g_mx, long gmx_stride_b,
                                          long gmx_stride_h, const float* const* g_nodes_out, int n_steps, int s0,
                                          const float* params, int has_bias, int act1, int act2,
                                          const int64_t* count0, float* gx, float* gn0, int B, int N, int F, int H1,
                                          int H2, gcm_stream_t stream) {
  return rows_dx_all_impl(saved, g_mx, gmx_stride_b, gmx_stride_h, g_nodes_out, n_steps, s0, params, has_bias, act1,
                          act2, count0, gx, gn0, nullptr, 0, nullptr, B, N, F, H1, H2, stream);
}

/* The same over the records of CACHED steps (rows_cached.hip) of a chain that started from empty graphs: s0 is then
 * also the row the first step's node landed in; the live rows and their adjacency rows follow from the selectors'
 * forward hops, their h1 rows come from the chain's cache. */
extern "C" int gcm_dense_rows_bptt_dx_all_cached(const float* const* saved, const float* const* g_mx,
                                                 long gmx_stride_b, long gmx_stride_h, int n_steps, int s0,
                                                 const float* params, int has_bias, int act1, int act2,
                                                 const gcm_selector_desc* selectors, int n_selectors,
                                                 const float* cache_h1, float* gx, int B, int N, int F, int H1, int H2,
                                                 gcm_stream_t stream) {
  GCM_REQUIRE(cache_h1 && (selectors || n_selectors == 0));
  if (!gcm_dense_rows_cached_supported(selectors, n_selectors, has_bias, N, F, H1, H2)) return GCM_EUNSUPPORTED;
  static const float* const kNone[GCM_ROWS_MAX_STEPS] = {};
  return rows_dx_all_impl(saved, g_mx, gmx_stride_b, gmx_stride_h, kNone, n_steps, s0, params, has_bias, act1, act2,
                          nullptr, gx, nullptr, selectors, n_selectors, cache_h1, B, N, F, H1, H2, stream);
}

static int rows_dx_all_impl(const float* const* saved, const float* const* g_mx, long gmx_stride_b,
                            long gmx_stride_h, const float* const* g_nodes_out, int n_steps, int s0,
                            const float* params, int has_bias, int act1, int act2, const int64_t* count0, float* gx,
                            float* gn0, const gcm_selector_desc* selectors, int n_selectors, const float* cache_h1,
                            int B, int N, int F, int H1, int H2, gcm_stream_t stream) {
  GCM_REQUIRE(saved && g_mx && g_nodes_out && params && (count0 || cache_h1) && gx && B > 0 && s0 >= 0);
  GCM_REQUIRE(n_steps > 0 && n_steps <= GCM_ROWS_MAX_STEPS);
  if (!gcm_dense_rows_dx_supported(N, F, H1, H2) || (has_bias & (GCM_GNN_HAS_DEG_TERM | GCM_GNN_HAS_PE_TABLE)))
    return GCM_EUNSUPPORTED;
  gcm_rows::DxTable tab;
  for (int s = 0; s < GCM_ROWS_MAX_STEPS; ++s) {
    tab.saved[s] = s < n_steps ? saved[s] : nullptr;
    tab.gmx[s] = s < n_steps ? g_mx[s] : nullptr;
    tab.gn[s] = s < n_steps ? g_nodes_out[s] : nullptr;
    if (s < n_steps) GCM_REQUIRE(saved[s] != nullptr);
  }
  const float* w_rel2 = params + 2 * (size_t)H1 * F + H1;
  const float* w_root2 = w_rel2 + (size_t)H2 * H1;
  gcm_rows::SavedLayout lay = gcm_rows::make_layout(B, N, F, H1, H2, true);
  gcm_fused::Edits E{};
  if (cache_h1) {
    lay.o_live = gcm_rows::make_cached_layout(B, N, H1, H2).o_live;   // (v / hdr / coef sit where the full record has them)
    for (int i = 0; i < n_selectors; ++i)
      for (int k = 0; k < selectors[i].n_hops; ++k) {
        E.hops[E.n_hops] = selectors[i].hops[k];
        E.dir[E.n_hops++] = selectors[i].direction;
      }
  }
  const size_t lds = sizeof(float) * 4 * (size_t)N * F;
  const long items = (long)n_steps * B;
  const long wgs = (items + 3) / 4, resident = 2L * gcm_cu_count();   // (64 KB of LDS per workgroup: two per CU)
  const dim3 grid((unsigned)(wgs < resident ? wgs : resident));
  if (cache_h1) {
    gcm_allow_dynamic_lds((const void*)gcm_rows::k_rows_dx_all<true>, lds);
    hipLaunchKernelGGL(gcm_rows::k_rows_dx_all<true>, grid, dim3(256), lds, (hipStream_t)stream, tab, n_steps, s0,
                       gmx_stride_b, gmx_stride_h, params, params + (size_t)H1 * F, w_rel2, w_root2, act1, act2, lay,
                       count0, gx, gn0, B, N, F, H1, H2, E, cache_h1);
  } else {
    gcm_allow_dynamic_lds((const void*)gcm_rows::k_rows_dx_all<false>, lds);
    hipLaunchKernelGGL(gcm_rows::k_rows_dx_all<false>, grid, dim3(256), lds, (hipStream_t)stream, tab, n_steps, s0,
                       gmx_stride_b, gmx_stride_h, params, params + (size_t)H1 * F, w_rel2, w_root2, act1, act2, lay,
                       count0, gx, gn0, B, N, F, H1, H2, E, cache_h1);
  }
  return gcm_launch_status();
}
