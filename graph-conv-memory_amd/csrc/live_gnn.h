// The two-layer GNN of one DenseGCM step on the LIVE 32-row tiles of a graph whose adjacency image,
// node image and weights are resident in LDS.  Shared by the per-step kernel (k_step_fwd_live,
// fused_fwd.hip) and the persistent rollout kernel (k_rollout_fwd, rollout_persist.hip).
//
// Live = the tiles holding row `cur` and the non-zeros of adj[cur, :]: only their h1 rows reach the
// belief (gcm.py:314 keeps row cur of the last layer) and only they carry gradient in BPTT.  All
// four waves share each live tile as 16x16 output blocks on v_mfma_f32_16x16x4_f32 (block id =
// wave + 4*bi -> row half id & 1, column block id >> 1); all-zero 32x32 adjacency tiles are skipped
// by the tile mask; layer 2 runs on wave 0 alone, without workgroup barriers.
#pragma once
#include "fused_common.h"

namespace gcm_fused {

template <int NT, int NCT, int NHT, int N2T>
struct LiveGnn {
  using L = Lds<NT, NCT, NHT, N2T>;
  static constexpr int NP = L::NP, FP = L::FP, HP = L::HP, H2P = L::H2P;
  static constexpr int FS = L::FS, HS = L::HS, AS = L::AS, W2S = L::W2S;

  float* sAdj;      // [col tile][row][33]
  float* sX;        // [NP][FS]
  float* sAH;       // agg, then h1 (stride AS); only live tiles are ever written / read
  float* sW1;       // w_rel1^T | w_root1^T  [f][HS]
  float* sW2;       // [o][rel k | root k], stride W2S
  float* sVv;       // [2*HP] scratch of wave 0
  const unsigned* sMask;   // word R: non-zero column tiles of row tile R
  float bias1[NHT];        // b_rel1 at this lane's h1 columns
  float bias2;             // b_rel2[lane] (wave 0)
  int act1_v, act2_v;      // activation codes in VGPRs (gcm_act_sel)
  int wave, lane, m16, kq;

  __device__ __forceinline__ void init_lane(const Gnn2& P, int tid) {
    wave = tid >> 6;
    lane = tid & 63;
    m16 = lane & 15;
    kq = lane >> 4;
#pragma unroll
    for (int bi = 0; bi < NHT; ++bi) {
      const int blk = wave + 4 * bi, h0 = (blk >> 1) * 16;
      bias1[bi] = P.b_rel1 ? P.b_rel1[h0 + m16] : 0.f;
    }
    bias2 = P.b_rel2 ? P.b_rel2[lane < H2P ? lane : H2P - 1] : 0.f;
    act1_v = gcm_vgpr(P.act1);
    act2_v = gcm_vgpr(P.act2);
  }

  // tile mask and live flags of this step (call after the barrier that publishes the edits).
  // The flags are separate scalars: hipcc 7.2 inverted the third `(live >> R) & 1` test on one
  // mask for NT >= 3 (regression: the N = 96 / 128 rollout tests).
  __device__ __forceinline__ void flags(int cur, unsigned& nzmask, bool (&lvt)[NT]) const {
    nzmask = 0;
#pragma unroll
    for (int R = 0; R < NT; ++R) nzmask |= sMask[R] << (4 * R);
    nzmask = __builtin_amdgcn_readfirstlane(nzmask);
    unsigned live = 1u << (cur >> 5);
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
      live |= (__any(sAdj[adj_at<NP>(cur, tt * 32 + (lane & 31))] != 0.f) ? 1u : 0u) << tt;
#pragma unroll
    for (int R = 0; R < NT; ++R) lvt[R] = __builtin_amdgcn_readfirstlane((live >> R) & 1u) != 0;
  }

  // a1g / h1g: this graph-step's [N,F] / [N,H1] activation slots in HBM (live tiles are written) or
  // NULL; agg2g [H1] or NULL; mxg [H2].  Three workgroup barriers inside; the caller owns the one
  // that protects the images from the next step's writes.
  __device__ __forceinline__ void run(int cur, unsigned nzmask, const bool (&lvt)[NT], float* a1g,
                                      float* h1g, float* agg2g, float* mxg, uint32_t* flag_word) {
    constexpr int F = FP, H1 = HP, H2 = H2P;
#pragma unroll
    for (int R = 0; R < NT; ++R)
      if (lvt[R]) {   // layer 1 aggregation of this tile's rows, zero tiles skipped
#pragma unroll
        for (int bi = 0; bi < NCT; ++bi) {
          const int blk = wave + 4 * bi, r0 = R * 32 + (blk & 1) * 16, c0 = (blk >> 1) * 16;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int tt = 0; tt < NT; ++tt)
            if ((nzmask >> (R * 4 + tt)) & 1u)
              mma16<32>(acc, sAdj + (tt * NP + r0) * 33, 33, sX + (tt * 32) * FS + c0, FS, m16, kq);
          float* d = sAH + (r0 + 4 * kq) * AS + c0 + m16;
#pragma unroll
          for (int r = 0; r < 4; ++r) d[r * AS] = acc[r];
          if (a1g) {
            float* g = a1g + (r0 + 4 * kq) * F + c0 + m16;
#pragma unroll
            for (int r = 0; r < 4; ++r) g[r * F] = acc[r];
          }
        }
      }
    __syncthreads();   // every wave's agg blocks are in LDS
    f32x4 o[NT][NHT];
#pragma unroll
    for (int R = 0; R < NT; ++R) {
#pragma unroll
      for (int bi = 0; bi < NHT; ++bi) o[R][bi] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (lvt[R]) {
#pragma unroll
        for (int bi = 0; bi < NHT; ++bi) {
          const int blk = wave + 4 * bi, r0 = R * 32 + (blk & 1) * 16, h0 = (blk >> 1) * 16;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          mma16<FP>(acc, sAH + r0 * AS, AS, sW1 + h0, HS, m16, kq);
          mma16<FP>(acc, sX + r0 * FS, FS, sW1 + FP * HS + h0, HS, m16, kq);
          o[R][bi] = acc;
        }
      }
    }
    __syncthreads();   // nobody reads agg any more: h1 takes its place
#pragma unroll
    for (int R = 0; R < NT; ++R)
      if (lvt[R]) {
#pragma unroll
        for (int bi = 0; bi < NHT; ++bi) {
          const int blk = wave + 4 * bi, r0 = R * 32 + (blk & 1) * 16, h0 = (blk >> 1) * 16;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gcm_act_sel(o[R][bi][r] + bias1[bi], act1_v);
          float* d = sAH + (r0 + 4 * kq) * AS + h0 + m16;
#pragma unroll
          for (int r = 0; r < 4; ++r) d[r * AS] = v[r];
          if (h1g) {
            float* g = h1g + (r0 + 4 * kq) * H1 + h0 + m16;
#pragma unroll
            for (int r = 0; r < 4; ++r) g[r * H1] = v[r];
          }
        }
      }
    __syncthreads();
    // ---- layer 2 on row `cur`: wave 0 alone ------------------------------------------------------
    if (wave == 0) {
      constexpr int PA = 64 / HP;            // lanes per h (1 or 2): split of the j range
      constexpr int JN = 32 / PA;
      const int h = lane & (HP - 1), part = lane / HP;
      float s = 0.f;
#pragma unroll
      for (int R = 0; R < NT; ++R)
        if (lvt[R]) {
          const float* arow = sAdj + (R * NP + cur) * 33 + part;    // adj[cur][R*32 + part + PA*i]
          const float* hcol = sAH + (R * 32 + part) * AS + h;       // h1[R*32 + part + PA*i][h]
          float av[JN], hv[JN];   // every LDS read in flight before the first FMA
#pragma unroll
          for (int i = 0; i < JN; ++i) {
            av[i] = arow[PA * i];
            hv[i] = hcol[PA * i * AS];
          }
#pragma unroll
          for (int i = 0; i < JN; ++i) s = fmaf(av[i], hv[i], s);
        }
      if (PA == 2) s += __shfl_xor(s, 32);
      const float hc = sAH[cur * AS + h];
      if (part == 0) {
        sVv[h] = s;              // v[0:HP)   = agg2
        sVv[HP + h] = hc;        // v[HP:2HP) = h1[cur]
        if (agg2g) agg2g[h] = s;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // pre2[o] = b2[o] + sum_k W2c[o][k] * v[k], K = 2*HP
      constexpr int PB = 64 / H2P;           // lanes per output (1 or 2): split of K
      constexpr int KC = 2 * HP / PB;
      const int o2 = lane & (H2P - 1), kp = lane / H2P;
      const float* wrow = sW2 + o2 * W2S + kp * KC;
      const float* vv = sVv + kp * KC;
      float a = 0.f;
#pragma unroll
      for (int k0 = 0; k0 < KC; k0 += 16) {
        float wv[16], xv[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          wv[k] = wrow[k0 + k];
          xv[k] = vv[k0 + k];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) a = fmaf(wv[k], xv[k], a);
      }
      if (PB == 2) a += __shfl_xor(a, 32);
      const float v = gcm_act_sel(a + bias2, act2_v);
      if (lane < H2) mxg[lane] = v;
      const bool any_bad = __any(lane < H2 && !isfinite(v));
      if (any_bad && lane == 0) atomicOr(flag_word, GCM_FLAG_NONFINITE);
    }
  }
};

}  // namespace gcm_fused
