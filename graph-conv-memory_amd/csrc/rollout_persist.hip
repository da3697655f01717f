// Persistent rollout forward: one workgroup per graph runs ALL T DenseGCM steps.
//
// B = 256 graphs is exactly one workgroup per CU, and a graph's whole state (adjacency 64 KB +
// node matrix 16 KB at N=128, F=32) fits the CU's 160 KB LDS.  So the state never leaves LDS
// between steps: per step the kernel inserts the observation row, rolls the images on overflow
// (gcm.py:323-355), applies the TemporalBackedge / DenseEdge writes, runs the two-layer GNN on
// the matrix cores and emits the belief.  HBM traffic per step is the 128-B observation in and
// fire-and-forget stores of what BPTT needs (the step's state and activations) - no loads on the
// critical path (the next observation row is prefetched one step ahead).
//
// Shapes: EXACT only (N, F, H1, H2 multiples of 32); other shapes use the per-step launches.
#include "fused_common.h"

namespace gcm_fused {

template <int NT, int NCT, int NHT, int N2T>
struct LdsRoll {
  using L = Lds<NT, NCT, NHT, N2T>;
  static constexpr int TOTAL = L::ADJ + L::X + L::AH + L::W1F + L::W2 + L::SV;
};

template <int NT, int NCT, int NHT, int N2T>
__global__ __launch_bounds__(256) void k_rollout_fwd(
    const float* __restrict__ obs, float* __restrict__ nodes_all, float* __restrict__ adj_all,
    int64_t* __restrict__ count_all, int64_t* __restrict__ cur_all, Edits E, Gnn2 P,
    float* __restrict__ mx_all, float* __restrict__ h1_all, float* __restrict__ agg1_all,
    float* __restrict__ agg2_all, uint32_t* __restrict__ flags, int T, int B) {
  using L = Lds<NT, NCT, NHT, N2T>;
  constexpr int N = L::NP, F = L::FP, H1 = L::HP, H2 = L::H2P;
  constexpr int NP = N, FP = F, HP = H1, H2P = H2;
  constexpr int FS = L::FS, HS = L::HS, AS = L::AS, W2S = L::W2S;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const size_t nodes_sz = (size_t)B * N * F, adj_sz = (size_t)B * N * N;

  extern __shared__ float smem[];
  float* sAdj = smem;                    // [col tile][row][33]   the graph's adjacency, resident
  float* sX = sAdj + L::ADJ;             // [N][FS]               the graph's nodes, resident
  float* sAH = sX + L::X;                // agg, then h1 (stride AS)
  float* sW1 = sAH + L::AH;              // w_rel1^T | w_root1^T  [f][HS]
  float* sW2 = sW1 + L::W1F;             // [o][rel k | root k], stride W2S
  float* sV = sW2 + L::W2;               // partials | v
  float* sVv = sV + 256;

  const int r_base = wave * 32;
  const bool wave_live = wave < NT;

  // ---- prologue: weights and the incoming state into LDS -----------------------------------
  {
    Stage<NP, FP, false, true> st_x;
    Stage<HP, FP, true, true> st_wr, st_wo;
    Stage<H2P, HP, false, true> st_w2r, st_w2o;
    AdjRows<NT, true> rows;
    st_x.load(nodes_all + (size_t)b * N * F, N, F, F, tid);
    st_wr.load(P.w_rel1, H1, F, F, tid);
    st_wo.load(P.w_root1, H1, F, F, tid);
    st_w2r.load(P.w_rel2, H2, H1, H1, tid);
    st_w2o.load(P.w_root2, H2, H1, H1, tid);
    if (wave_live) rows.load(adj_all + (size_t)b * N * N, N, r_base, lane);
    st_x.store(sX, FS, tid);
    st_wr.store(sW1, HS, tid);
    st_wo.store(sW1 + FP * HS, HS, tid);
    st_w2r.store(sW2, W2S, tid);
    st_w2o.store(sW2 + HP, W2S, tid);
    if (wave_live) {
#pragma unroll
      for (int t = 0; t < NT; ++t) rows.template store_tile<NP>(sAdj, t, r_base, lane);
    }
  }
  int64_t n = count_all[b];
  const float bias1 = (P.b_rel1 && li < H1) ? P.b_rel1[li] : 0.f;   // NHT == 1 fast path below uses it
  float ob_next = obs[(size_t)b * F + (tid % FP)];
  __syncthreads();

#pragma unroll 1
  for (int t = 0; t < T; ++t) {
    const float ob = ob_next;
    if (t + 1 < T) ob_next = obs[((size_t)(t + 1) * B + b) * F + (tid % FP)];   // prefetch
    const bool bad = n < 0 || n > N;
    const bool wrap = n + 1 > N;
    int64_t c64 = wrap ? n - 1 : n;
    const int cur = c64 < 0 ? 0 : (c64 > N - 1 ? N - 1 : (int)c64);

    // ---- overflow: rotate both images one slot towards index 0 (registers as the bounce buffer)
    if (wrap) {   // workgroup-uniform
      float av[NT * 16], xv[NP * FP / 256];
      if (wave_live) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int r = r_base + (lane >> 3) + 8 * q + 1, c = tt * 32 + (lane & 7) * 4 + k + 1;
              av[(tt * 4 + q) * 4 + k] = (r < N && c < N) ? sAdj[adj_at<NP>(r, c)] : 0.f;
            }
      }
#pragma unroll
      for (int i = 0; i < NP * FP / 256; ++i) {
        const int e = tid + 256 * i, r = e / FP + 1, c = e % FP;
        xv[i] = r < N ? sX[r * FS + c] : 0.f;
      }
      __syncthreads();
      if (wave_live) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int r = r_base + (lane >> 3) + 8 * q, c = tt * 32 + (lane & 7) * 4 + k;
              sAdj[adj_at<NP>(r, c)] = av[(tt * 4 + q) * 4 + k];
            }
      }
#pragma unroll
      for (int i = 0; i < NP * FP / 256; ++i) {
        const int e = tid + 256 * i, r = e / FP, c = e % FP;
        sX[r * FS + c] = xv[i];
      }
      __syncthreads();
    }
    // ---- insert the observation, apply the selector writes ---------------------------------
    if (tid < FP) sX[cur * FS + tid] = ob;
    if (tid < E.n_hops) {
      const int h = E.hops[tid];
      if (h >= 0 && cur >= h) {
        if (E.dir[tid] & GCM_DIR_FORWARD) sAdj[adj_at<NP>(cur, cur - h)] = 1.f;
        if (E.dir[tid] & GCM_DIR_BACKWARD) sAdj[adj_at<NP>(cur - h, cur)] = 1.f;
      }
    }
    if (E.dense) {
      for (int j = tid; j <= cur; j += 256) {
        sAdj[adj_at<NP>(cur, j)] = 1.f;
        if (j < cur) sAdj[adj_at<NP>(j, cur)] = 1.f;
      }
    }
    if (tid == 0) {
      cur_all[(size_t)t * B + b] = cur;
      count_all[(size_t)(t + 1) * B + b] = cur + 1;
      const uint32_t f = (wrap ? GCM_FLAG_WRAPPED : 0u) | (bad ? GCM_FLAG_BAD_COUNT : 0u);
      if (f) atomicOr(flags, f);
    }
    __syncthreads();

    // ---- the step's state goes to HBM (functional history for BPTT): fire-and-forget stores ----
    {
      float* no = nodes_all + (size_t)(t + 1) * nodes_sz + (size_t)b * N * F;
#pragma unroll
      for (int i = 0; i < NP * FP / 1024; ++i) {   // one float4 per thread and pass
        const int e4 = tid + 256 * i, r = e4 / (FP / 4), c = (e4 % (FP / 4)) * 4;
        const float* s = sX + r * FS + c;
        *reinterpret_cast<float4*>(no + r * F + c) = make_float4(s[0], s[1], s[2], s[3]);
      }
    }
    f32x16 acc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    if (wave_live) {
      float* ao = adj_all + (size_t)(t + 1) * adj_sz + (size_t)b * N * N;
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        bool nz = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = r_base + (lane >> 3) + 8 * q, c = tt * 32 + (lane & 7) * 4;
          const float* s = sAdj + (tt * NP + r) * 33 + (lane & 7) * 4;
          const float4 v = make_float4(s[0], s[1], s[2], s[3]);
          *reinterpret_cast<float4*>(ao + r * N + c) = v;
          nz |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
        }
        // ---- layer 1 aggregation on the resident image, zero tiles skipped ---------------------
        if (__any(nz)) {
#pragma unroll
          for (int c = 0; c < NCT; ++c)
            mma32(acc[c], sAdj + (tt * NP + r_base) * 33, 33, 1, sX + (tt * 32) * FS + c * 32, FS, 1,
                  32, li, lh);
        }
      }
      float* a1g = agg1_all ? agg1_all + (size_t)t * nodes_sz + (size_t)b * N * F : nullptr;
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r_base + acc_row(r, lh), col = c * 32 + li;
          sAH[row * AS + col] = acc[c][r];
          if (a1g) a1g[row * F + col] = acc[c][r];
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      f32x16 o[NHT];
#pragma unroll
      for (int ht = 0; ht < NHT; ++ht) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ht][r] = 0.f;
        mma32(o[ht], sAH + r_base * AS, AS, 1, sW1 + ht * 32, HS, 1, FP, li, lh);
        mma32(o[ht], sX + r_base * FS, FS, 1, sW1 + FP * HS + ht * 32, HS, 1, FP, li, lh);
      }
      __builtin_amdgcn_wave_barrier();
      float* h1g = h1_all ? h1_all + ((size_t)t * B + b) * N * H1 : nullptr;
#pragma unroll
      for (int ht = 0; ht < NHT; ++ht) {
        const int col = ht * 32 + li;
        const float bias = NHT == 1 ? bias1 : (P.b_rel1 ? P.b_rel1[col] : 0.f);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r_base + acc_row(r, lh);
          const float v = gcm_act(o[ht][r] + bias, P.act1);
          sAH[row * AS + col] = v;
          if (h1g) h1g[row * H1 + col] = v;
        }
      }
    }
    __syncthreads();
    // ---- layer 2 on row `cur` -------------------------------------------------------------------
    {
      constexpr int G = 256 / HP;
      const int g = tid / HP, h = tid - g * HP;
      float s = 0.f;
#pragma unroll 4
      for (int j = g; j < N; j += G) s = fmaf(sAdj[adj_at<NP>(cur, j)], sAH[j * AS + h], s);
      sV[tid] = s;
      __syncthreads();
      if (tid < HP) {
        float a2 = 0.f;
#pragma unroll
        for (int q = 0; q < G; ++q) a2 += sV[q * HP + tid];
        sVv[tid] = a2;
        sVv[HP + tid] = sAH[cur * AS + tid];
        if (agg2_all) agg2_all[((size_t)t * B + b) * H1 + tid] = a2;
      }
    }
    __syncthreads();
    {
      constexpr int G2 = 256 / H2P, KC = (2 * HP) / G2;
      const int g = tid / H2P, o2 = tid - g * H2P;
      float s = 0.f;
      const float* wrow = sW2 + o2 * W2S + g * KC;
      const float* vv = sVv + g * KC;
#pragma unroll
      for (int k = 0; k < KC; ++k) s = fmaf(wrow[k], vv[k], s);
      sV[tid] = s;
      __syncthreads();
      bool nonfinite = false;
      if (tid < H2) {
        float a = P.b_rel2 ? P.b_rel2[tid] : 0.f;
#pragma unroll
        for (int q = 0; q < G2; ++q) a += sV[q * H2P + tid];
        const float v = gcm_act(a, P.act2);
        mx_all[((size_t)t * B + b) * H2 + tid] = v;
        nonfinite = !isfinite(v);
      }
      if (wave == 0) {
        const bool any_bad = __any(nonfinite);
        if (any_bad && lane == 0) atomicOr(flags, GCM_FLAG_NONFINITE);
      }
    }
    n = cur + 1;
    __syncthreads();   // the next step rewrites sX / sAdj / sAH / sV
  }
}

template <int NT, int NCT, int NHT, int N2T>
int launch_rollout(hipStream_t s, const float* obs, float* nodes_all, float* adj_all,
                   int64_t* count_all, int64_t* cur_all, Edits E, Gnn2 P, float* mx_all,
                   float* h1_all, float* agg1_all, float* agg2_all, uint32_t* flags, int T, int B) {
  constexpr size_t lds = sizeof(float) * (size_t)LdsRoll<NT, NCT, NHT, N2T>::TOTAL;
  if (lds > 160 * 1024) return GCM_EUNSUPPORTED;
  auto kern = k_rollout_fwd<NT, NCT, NHT, N2T>;
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, s, obs, nodes_all, adj_all, count_all, cur_all,
                     E, P, mx_all, h1_all, agg1_all, agg2_all, flags, T, B);
  return gcm_launch_status();
}

// ---------------------------------------------------------------------------
// reverse scan of the node-matrix gradient through time (the only sequential part of BPTT here):
//   C_t = U_t(C_{t+1}) + Q_t ,  g_obs[t] = C_{t+1}[cur_t] + p_obs[t]
// C_t = gradient w.r.t. the nodes entering step t; U_t = adjoint of the state advance (zero row
// cur_t; on overflow shift one row away from index 0, gcm.py:262-278 / 323-355).  Q_t = U_t(dX_t)
// and p_obs[t] = dX_t[cur_t] come from the time-batched backward launch (g_nodes_in / g_obs of
// gcm_dense_gnn2_row_bwd with no incoming node gradient).
// One workgroup per graph, the running gradient lives in LDS.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gnodes_scan(
    const float* __restrict__ Q_all, const float* __restrict__ pobs_all,
    const float* __restrict__ g_nodes_T, const int64_t* __restrict__ cur_all,
    const int64_t* __restrict__ count_all, float* __restrict__ g_obs_all,
    float* __restrict__ g_nodes_0, int T, int B, int N, int F) {
  extern __shared__ float sG[];   // [N*F] gradient w.r.t. the nodes after step t (from steps > t)
  const int b = blockIdx.x, tid = threadIdx.x;
  const int NF = N * F;
  const size_t nodes_sz = (size_t)B * NF;
  constexpr int PER = 32;   // N*F <= 128*64
  for (int e = tid; e < NF; e += 256) sG[e] = g_nodes_T ? g_nodes_T[(size_t)b * NF + e] : 0.f;
  float q[PER];
  {
    const float* Qt = Q_all + (size_t)(T - 1) * nodes_sz + (size_t)b * NF;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 256 * i;
      q[i] = Qt[e < NF ? e : NF - 1];
    }
  }
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) {
    int64_t c64 = cur_all[(size_t)t * B + b];
    const int cur = c64 < 0 ? 0 : (c64 > N - 1 ? N - 1 : (int)c64);
    const bool wrap = count_all[(size_t)t * B + b] + 1 > N;
    if (tid < F)
      g_obs_all[((size_t)t * B + b) * F + tid] = sG[cur * F + tid] + pobs_all[((size_t)t * B + b) * F + tid];
    float v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 256 * i;
      float u = 0.f;
      if (e < NF) {
        const int r = e / F;
        if (wrap) u = r >= 1 ? sG[e - F] : 0.f;
        else u = r == cur ? 0.f : sG[e];
      }
      v[i] = u + q[i];
    }
    if (t > 0) {   // next step's Q while this one settles
      const float* Qt = Q_all + (size_t)(t - 1) * nodes_sz + (size_t)b * NF;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int e = tid + 256 * i;
        q[i] = Qt[e < NF ? e : NF - 1];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 256 * i;
      if (e < NF) sG[e] = v[i];
    }
    __syncthreads();
  }
  for (int e = tid; e < NF; e += 256) g_nodes_0[(size_t)b * NF + e] = sG[e];
}

}  // namespace gcm_fused

// instantiated EXACT shapes of the persistent forward
#define GCM_RSHAPES_N(X, a) \
  X(a, 1, 1, 1) X(a, 1, 1, 2) X(a, 1, 2, 1) X(a, 1, 2, 2) X(a, 2, 1, 1) X(a, 2, 1, 2) X(a, 2, 2, 1) X(a, 2, 2, 2)
#define GCM_RSHAPES(X) GCM_RSHAPES_N(X, 1) GCM_RSHAPES_N(X, 2) GCM_RSHAPES_N(X, 3) GCM_RSHAPES_N(X, 4)

extern "C" int gcm_dense_rollout_persistent_fwd(const float* obs, float* nodes_all, float* adj_all,
                               int64_t* count_all, int64_t* cur_all,
                               const gcm_selector_desc* selectors, int n_selectors,
                               const float* w_rel1, const float* b_rel1, const float* w_root1,
                               int act1, const float* w_rel2, const float* b_rel2,
                               const float* w_root2, int act2, float* mx_all, float* h1_all,
                               float* agg1_all, float* agg2_all, uint32_t* flags, int T, int B,
                               int N, int F, int H1, int H2, gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes_all && adj_all && count_all && cur_all && mx_all && flags && w_rel1 &&
              w_root1 && w_rel2 && w_root2);
  GCM_REQUIRE(T > 0 && B > 0 && (selectors || n_selectors == 0));
  hipStream_t s = (hipStream_t)stream;
  if ((N & 31) || (F & 31) || (H1 & 31) || (H2 & 31) || N > 128 || F > 64 || H1 > 64 || H2 > 64)
    return GCM_EUNSUPPORTED;
  gcm_fused::Edits E{};
  for (int i = 0; i < n_selectors; ++i) {
    const gcm_selector_desc& d = selectors[i];
    if (d.kind == GCM_SEL_TEMPORAL) {
      for (int k = 0; k < d.n_hops; ++k) {
        if (E.n_hops >= 16) return GCM_EUNSUPPORTED;
        E.hops[E.n_hops] = d.hops[k];
        E.dir[E.n_hops++] = d.direction;
      }
    } else if (d.kind == GCM_SEL_DENSE) {
      E.dense = 1;
    } else {
      return GCM_EUNSUPPORTED;
    }
  }
  gcm_fused::Gnn2 P{w_rel1, b_rel1, w_root1, w_rel2, b_rel2, w_root2, act1, act2};
  const int NT = N / 32, NCT = F / 32, NHT = H1 / 32, N2T = H2 / 32;
#define GCM_R(a, b_, c, d)                                                                     \
  if (NT == a && NCT == b_ && NHT == c && N2T == d)                                            \
    return gcm_fused::launch_rollout<a, b_, c, d>(s, obs, nodes_all, adj_all, count_all,       \
                                                  cur_all, E, P, mx_all, h1_all, agg1_all,     \
                                                  agg2_all, flags, T, B);
  GCM_RSHAPES(GCM_R)
#undef GCM_R
  return GCM_EUNSUPPORTED;
}

extern "C" int gcm_dense_gnodes_scan(const float* Q_all, const float* pobs_all,
                                     const float* g_nodes_T, const int64_t* cur_all,
                                     const int64_t* count_all, float* g_obs_all,
                                     float* g_nodes_0, int T, int B, int N, int F,
                                     gcm_stream_t stream) {
  GCM_REQUIRE(Q_all && pobs_all && cur_all && count_all && g_obs_all && g_nodes_0);
  GCM_REQUIRE(T > 0 && B > 0 && N > 0 && F > 0);
  if ((size_t)N * F > 8192) return GCM_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = sizeof(float) * (size_t)N * F;
  if (lds > 64 * 1024) return GCM_EUNSUPPORTED;
  hipLaunchKernelGGL(gcm_fused::k_gnodes_scan, dim3(B), dim3(256), lds, s, Q_all, pobs_all,
                     g_nodes_T, cur_all, count_all, g_obs_all, g_nodes_0, T, B, N, F);
  return gcm_launch_status();
}
