// Fused two-layer DenseGraphConv step, backward, with the state-advance adjoint folded in
// (see fused_common.h for the design notes).
//   in : g_mx [B,H2], g_nodes_out [B,N,F] (gradient arriving from later steps, may be NULL)
//   out: g_nodes_in [B,N,F], g_obs [B,F], parameter-gradient slab of this graph
// slab layout (floats): dW_rel1 [H1*F] | dW_root1 [H1*F] | db1 [H1] | dW_rel2 [H2*H1] |
//                       dW_root2 [H2*H1] | db2 [H2]
#include "fused_common.h"

#ifdef GCM_STAMPS   // diagnostic build only (make stamps4)
__device__ unsigned long long g_stamps[32];
extern "C" int gcm_debug_read_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
// stand-alone diagnostic library: the shape check lives in fused_fwd.hip
extern "C" int gcm_dense_gnn2_row_supported(int, int, int, int) { return 1; }
#endif

namespace gcm_fused {

template <int NT, int NCT, int NHT, int N2T, bool EXACT>
__global__ __launch_bounds__(256) void k_gnn2_row_bwd(
    const float* __restrict__ g_mx, const float* __restrict__ g_nodes_out,
    const float* __restrict__ x, const float* __restrict__ adj,
    const int64_t* __restrict__ cur_idx, const int64_t* __restrict__ num_nodes_in, Gnn2 P,
    const float* __restrict__ mx, const float* __restrict__ h1, const float* __restrict__ agg1,
    const float* __restrict__ agg2, float* __restrict__ g_nodes_in, float* __restrict__ g_obs,
    float* __restrict__ slabs, int accumulate, int N_, int F_, int H1_, int H2_) {
  using L = Lds<NT, NCT, NHT, N2T>;
  constexpr int NP = L::NP, FP = L::FP, HP = L::HP, H2P = L::H2P;
  constexpr int FS = L::FS, HS = L::HS, W2S = L::W2S;
  const int N = EXACT ? NP : N_, F = EXACT ? FP : F_, H1 = EXACT ? HP : H1_, H2 = EXACT ? H2P : H2_;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const float* xg = x + (size_t)b * N * F;
  const float* ag = adj + (size_t)b * N * N;
  const float* h1g = h1 + (size_t)b * N * H1;
  const float* a1g = agg1 + (size_t)b * N * F;
  const float* gng = g_nodes_out ? g_nodes_out + (size_t)b * N * F : nullptr;
  float* gin = g_nodes_in + (size_t)b * N * F;
  int64_t cur64 = cur_idx[b];
  const int cur = cur64 < 0 ? 0 : (cur64 > N - 1 ? N - 1 : (int)cur64);
  const bool wrap = num_nodes_in[b] + 1 > N;
  float* slab = slabs + (size_t)b * (2 * (size_t)H1 * F + H1 + 2 * (size_t)H2 * H1 + H2);
  float* sl_rel1 = slab;
  float* sl_root1 = sl_rel1 + H1 * F;
  float* sl_b1 = sl_root1 + H1 * F;
  float* sl_rel2 = sl_b1 + H1;
  float* sl_root2 = sl_rel2 + H2 * H1;
  float* sl_b2 = sl_root2 + H2 * H1;

  extern __shared__ float smem[];
  float* sAdj = smem;                   // [col tile][row][33]
  float* sG = sAdj + L::ADJ;            // [NP][HS]   h1, then G1 = dh1 * act1'(h1)
  float* sD = sG + NP * HS;             // [NP][FS]   dAgg1
  float* sW1 = sD + NP * FS;            // w_rel1 [h][FS] | w_root1 [h][FS]          (B(k=h, j=f))
  float* sW2 = sW1 + L::W1B;            // [o][rel k | root k], stride W2S
  float* sR = sW2 + L::W2;              // [4][1024]  cross-wave reduction of the dW tiles
  float* sV = sR + 4 * 1024;            // [0,256) partials
  float* sVv = sV + 256;                // v = agg2 | h1[cur]            [2*HP]
  float* sD2 = sVv + 2 * HP;            // d2                            [H2P]
  float* sU = sD2 + H2P;                // u = dagg2 | dh1cur            [2*HP]
  int* sFlag = reinterpret_cast<int*>(sU + 2 * HP);   // adj tile (row tile, col tile) non-zero [16]

  STAMP(0);
  // ---- issue every load ----------------------------------------------------------------
  const int r_base = wave * 32;
  const bool wave_live = wave < NT;
  Stage<HP, FP, false, EXACT> st_wr, st_wo;
  Stage<H2P, HP, false, EXACT> st_w2r, st_w2o;
  Stage<NP, HP, false, EXACT> st_h1;
  AdjRows<NT, EXACT> rows;
  st_w2r.load(P.w_rel2, H2, H1, H1, tid);
  st_w2o.load(P.w_root2, H2, H1, H1, tid);
  float gm = 0.f, mv = 0.f, v_in = 0.f;
  {
    const int o = tid < H2 ? tid : H2 - 1;
    gm = g_mx[(size_t)b * H2 + o];
    mv = mx[(size_t)b * H2 + o];
    const int k = tid < HP ? (tid < H1 ? tid : H1 - 1) : (tid - HP < H1 ? tid - HP : H1 - 1);
    const float a2 = agg2[(size_t)b * H1 + (k < 0 ? 0 : k)];
    const float hc = h1g[cur * H1 + (k < 0 ? 0 : k)];
    v_in = tid < HP ? (tid < H1 ? a2 : 0.f) : ((tid < 2 * HP && tid - HP < H1) ? hc : 0.f);
  }
  st_h1.load(h1g, N, H1, H1, tid);
  st_wr.load(P.w_rel1, H1, F, F, tid);
  st_wo.load(P.w_root1, H1, F, F, tid);
  if (wave_live) rows.load(ag, N, r_base, lane);

  STAMP(1);
  st_w2r.store(sW2, W2S, tid);
  st_w2o.store(sW2 + HP, W2S, tid);
  if (tid < H2P) sD2[tid] = tid < H2 ? gm * gcm_act_grad(mv, P.act2) : 0.f;
  if (tid < 2 * HP) sVv[tid] = v_in;
  __syncthreads();
  STAMP(2);
  // ---- u[m] = sum_o W2c[o][m] * d2[o]  (m < HP: dagg2, m >= HP: dh1cur) -------------------
  {
    constexpr int G = 256 / (2 * HP), OC = H2P / G;
    const int g = tid / (2 * HP), m = tid - g * (2 * HP);
    float s = 0.f;
#pragma unroll
    for (int o = g * OC; o < (g + 1) * OC; ++o) s = fmaf(sW2[o * W2S + m], sD2[o], s);
    sV[tid] = s;
  }
  // ---- layer-2 parameter gradients: d2[o] * v[k] ------------------------------------------
  {
    constexpr int PER = (H2P * 2 * HP + 255) / 256;
    float old[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 256 * i, o = e / (2 * HP), k = e % (2 * HP);
      const int kk = k < HP ? k : k - HP;
      old[i] = 0.f;
      if (accumulate) {  // uniform
        const int oc = o < H2 ? o : H2 - 1, kc = kk < H1 ? kk : H1 - 1;
        old[i] = (k < HP ? sl_rel2 : sl_root2)[oc * H1 + kc];
      }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 256 * i, o = e / (2 * HP), k = e % (2 * HP);
      const int kk = k < HP ? k : k - HP;
      if (EXACT || (o < H2 && kk < H1))
        (k < HP ? sl_rel2 : sl_root2)[o * H1 + kk] = old[i] + sD2[o] * sVv[k];
    }
    if (tid < H2) sl_b2[tid] = (accumulate ? sl_b2[tid] : 0.f) + sD2[tid];
  }
  STAMP(3);
  st_h1.store(sG, HS, tid);
  st_wr.store(sW1, FS, tid);
  st_wo.store(sW1 + HP * FS, FS, tid);
  if (wave_live) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      rows.template store_tile<NP>(sAdj, t, r_base, lane);
      const bool nz = rows.tile_nonzero(t);
      if (lane == 0) sFlag[wave * 4 + t] = nz ? 1 : 0;
    }
  }
  __syncthreads();
  if (tid < 2 * HP) {
    constexpr int G = 256 / (2 * HP);
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < G; ++q) t += sV[q * 2 * HP + tid];
    sU[tid] = t;
  }
  STAMP(4);
  // which 32-row tiles can hold a non-zero G1 row: rows j with adj[cur][j] != 0, and row cur
  bool g_live[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float a = sAdj[adj_at<NP>(cur, t * 32 + li)];
    g_live[t] = __any(a != 0.f) || (cur >> 5) == t;
  }
  __syncthreads();
  STAMP(5);
  bool my_rows_live = false;   // g_live[wave] without a runtime-indexed register array
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t == wave) my_rows_live = g_live[t];
  // B(k=row, j=f) operands straight from HBM/L2 (every element is used exactly once); the loads of
  // job j+1 are issued before job j's cross-wave reduction, job 0's before the G1 phase above
  auto load_bq = [&](float (&bq)[16], int job) {
    const int which = job & 1, ct = (job >> 1) % NCT;
    const float* src = which ? xg : a1g;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int row = r_base + 2 * s + lh, f = ct * 32 + li;
      if (EXACT) {
        bq[s] = src[row * F + f];
      } else {
        const float t = src[(row < N ? row : N - 1) * F + (f < F ? f : F - 1)];
        bq[s] = (row < N && f < F) ? t : 0.f;
      }
    }
  };
  float bq[16];
  if (my_rows_live) load_bq(bq, 0);
  // ---- G1[j][h] = (adj[cur][j] * dagg2[h] + [j==cur] dh1cur[h]) * act1'(h1[j][h]), in place --
  {
    // only the live row tiles: the others hold no gradient and nobody reads their sG rows
    constexpr int PT = 32 * HP / 256;   // elements per thread and row tile
    const int act1_v = gcm_vgpr(P.act1);
    float part = 0.f;  // column sums of G1 (db1): a thread always handles the same h
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (g_live[t]) {   // uniform
        float av[PT], hv[PT];
#pragma unroll
        for (int i = 0; i < PT; ++i) {   // every LDS read in flight before the arithmetic
          const int e = tid + 256 * (t * PT + i), j = e / HP, h = e % HP;
          av[i] = sAdj[adj_at<NP>(cur, j)];
          hv[i] = sG[j * HS + h];
        }
#pragma unroll
        for (int i = 0; i < PT; ++i) {
          const int e = tid + 256 * (t * PT + i), j = e / HP, h = e % HP;
          const float d = av[i] * sU[h] + (j == cur ? sU[HP + h] : 0.f);
          const float y = hv[i];
          const float ga = act1_v == GCM_ACT_TANH ? 1.f - y * y
                                                  : (act1_v == GCM_ACT_RELU ? (y > 0.f ? 1.f : 0.f) : 1.f);
          float v = d * ga;
          if (d == 0.f || !(EXACT || (j < N && h < H1))) v = 0.f;   // also keeps 0 * garbage out
          sG[j * HS + h] = v;
          part += v;
        }
      }
    sV[tid] = part;  // 256/HP partial sums per h
  }
  __syncthreads();
  if (tid < H1) {
    constexpr int G = 256 / HP;
    float t = accumulate ? sl_b1[tid] : 0.f;
#pragma unroll
    for (int q = 0; q < G; ++q) t += sV[q * HP + tid];
    sl_b1[tid] = t;
  }

  STAMP(6);
  // ---- layer-1 parameter gradients: [H1 x F] = G1^T (H1 x N) @ {agg1, x} (N x F) ---------
  // every wave contracts over its own 32 rows (skipped when its G1 rows are all zero); the
  // partial tiles meet in LDS.
#pragma unroll 1
  for (int job = 0; job < 2 * NHT * NCT; ++job) {
    const int which = job & 1, ct = (job >> 1) % NCT, ht = (job >> 1) / NCT;
    f32x16 a;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
    if (my_rows_live) {
      const float* ap = sG + (r_base + lh) * HS + ht * 32 + li;   // A(i=h, k=row)
      float aq[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) aq[s] = ap[2 * s * HS];
      __builtin_amdgcn_sched_barrier(0);   // one LDS round trip in front of the MFMA chain
#pragma unroll
      for (int s = 0; s < 16; ++s)
        a = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[s], bq[s], a, 0, 0, 0);
      if (job + 1 < 2 * NHT * NCT) load_bq(bq, job + 1);
    }
    float* dst = which ? sl_root1 : sl_rel1;
    float old[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i, hh = ht * 32 + (e >> 5), ff = ct * 32 + (e & 31);
      old[i] = 0.f;
      if (accumulate) old[i] = dst[(hh < H1 ? hh : H1 - 1) * F + (ff < F ? ff : F - 1)];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sR[wave * 1024 + acc_row(r, lh) * 32 + li] = a[r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i, hh = ht * 32 + (e >> 5), ff = ct * 32 + (e & 31);
      if (EXACT || (hh < H1 && ff < F))
        dst[hh * F + ff] = old[i] + ((sR[e] + sR[1024 + e]) + (sR[2048 + e] + sR[3072 + e]));
    }
    __syncthreads();
  }

  STAMP(7);
  // ---- dAgg1 = G1 @ W_rel1 -> LDS ;  acc = G1 @ W_root1 (root part of dX) -----------------
  f32x16 acc[NCT];
  float gno[NCT][16];
  if (wave_live) {
    // the gradient arriving from later steps: issue the loads now, they land under the MFMAs
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r_base + acc_row(r, lh), col = c * 32 + li;
        float t = 0.f;
        if (gng) t = gng[(EXACT || row < N ? row : N - 1) * F + (EXACT || col < F ? col : F - 1)];
        gno[c][r] = (EXACT || (row < N && col < F)) ? t : 0.f;
      }
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
      if (my_rows_live) {
        f32x16 d;
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = 0.f;
        mma32b<HP>(d, sG + r_base * HS, HS, 1, sW1 + c * 32, FS, 1, li, lh);
        mma32b<HP>(acc[c], sG + r_base * HS, HS, 1, sW1 + HP * FS + c * 32, FS, 1, li, lh);
#pragma unroll
        for (int r = 0; r < 16; ++r) sD[(r_base + acc_row(r, lh)) * FS + c * 32 + li] = d[r];
      }
    }
  }
  __syncthreads();
  STAMP(8);
  // ---- dX[i] += sum_k adj[k][i] * dAgg1[k]   (A read down the columns of the adj image) ----
  // K tile kt contributes only when its dAgg rows can be non-zero and adj tile (kt, wave) is
  if (wave_live) {
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      if (g_live[kt] && sFlag[kt * 4 + wave]) {
#pragma unroll
        for (int c = 0; c < NCT; ++c)
          mma32b<32>(acc[c], sAdj + (wave * NP + kt * 32) * 33, 1, 33, sD + (kt * 32) * FS + c * 32,
                     FS, 1, li, lh);
      }
    }
    STAMP(9);
    // ---- epilogue: add the gradient from later steps, undo insert + roll (gcm.py:262-278);
    // selects instead of a three-way branch per element ------------------------------------------
    const int sh = wrap ? 1 : 0;
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r_base + acc_row(r, lh), col = c * 32 + li;
        if (EXACT || (row < N && col < F)) {
          const float v = acc[c][r] + gno[c][r];
          const bool is_cur = row == cur;
          if (is_cur) g_obs[(size_t)b * F + col] = v;   // the inserted row belongs to the observation
          const int dst = row + sh;                      // out[r] = in[r+1] on overflow
          if (dst < N) gin[dst * F + col] = is_cur ? 0.f : v;
        }
      }
  }
  STAMP(10);
  if (wrap)  // in[0] was cleared before the roll: no gradient
    for (int c = tid; c < F; c += 256) gin[c] = 0.f;
}

// sum the per-graph slabs: out[e] = sum_b slabs[b][e]   (fixed order => deterministic).
// block = 16 elements x 16 slab groups, every thread sums its slabs with the loads in flight
// together, the 16 partials of an element meet in LDS.
__global__ __launch_bounds__(256) void k_sum_slabs(const float* __restrict__ slabs, int n_slabs,
                                                   int len, const float* __restrict__ prev,
                                                   float* __restrict__ out) {
  __shared__ float part[256];
  const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  const int ec = e < len ? e : len - 1;
  float s = 0.f;
  for (int i0 = grp; i0 < n_slabs; i0 += 16 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 16 * u;
      const float t = slabs[(size_t)(i < n_slabs ? i : n_slabs - 1) * len + ec];
      v[u] = i < n_slabs ? t : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  if (grp == 0 && e < len) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += part[g * 16 + el];
    out[e] = prev ? prev[e] + t : t;
  }
}

template <int NT, int NCT, int NHT, int N2T>
int launch_bwd(hipStream_t s, const float* g_mx, const float* g_nodes_out, const float* x,
               const float* adj, const int64_t* cur, const int64_t* nn_in, Gnn2 P, const float* mx,
               const float* h1, const float* agg1, const float* agg2, float* g_nodes_in,
               float* g_obs, float* slabs, int accumulate, int B, int N, int F, int H1, int H2) {
  using L = Lds<NT, NCT, NHT, N2T>;
  constexpr size_t lds = sizeof(float) * (size_t)L::BWD;
  const bool exact = N == L::NP && F == L::FP && H1 == L::HP && H2 == L::H2P;
  auto kern = exact ? k_gnn2_row_bwd<NT, NCT, NHT, N2T, true> : k_gnn2_row_bwd<NT, NCT, NHT, N2T, false>;
  gcm_allow_dynamic_lds((const void*)kern, lds);
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, s, g_mx, g_nodes_out, x, adj, cur, nn_in, P,
                     mx, h1, agg1, agg2, g_nodes_in, g_obs, slabs, accumulate, N, F, H1, H2);
  return gcm_launch_status();
}

}  // namespace gcm_fused

extern "C" int gcm_dense_gnn2_row_bwd(const float* g_mx, const float* g_nodes_out, const float* x,
                                      const float* adj, const int64_t* cur_idx,
                                      const int64_t* num_nodes_in, const float* w_rel1,
                                      const float* b_rel1, const float* w_root1, int act1,
                                      const float* w_rel2, const float* b_rel2,
                                      const float* w_root2, int act2, const float* mx,
                                      const float* h1, const float* agg1, const float* agg2,
                                      float* g_nodes_in, float* g_obs, float* slabs,
                                      int accumulate, int B, int N, int F, int H1, int H2,
                                      gcm_stream_t stream) {
  GCM_REQUIRE(g_mx && x && adj && cur_idx && num_nodes_in && w_rel1 && w_root1 && w_rel2 &&
              w_root2 && mx && h1 && agg1 && agg2 && g_nodes_in && g_obs && slabs);
  GCM_REQUIRE(B > 0);
  if (!gcm_dense_gnn2_row_supported(N, F, H1, H2)) return GCM_EUNSUPPORTED;
  gcm_fused::Gnn2 P{w_rel1, b_rel1, w_root1, w_rel2, b_rel2, w_root2, act1, act2};
  hipStream_t s = (hipStream_t)stream;
  const int NT = (N + 31) / 32, NCT = (F + 31) / 32, NHT = (H1 + 31) / 32, N2T = (H2 + 31) / 32;
#define GCM_B(a, b_, c, d)                                                                     \
  if (NT == a && NCT == b_ && NHT == c && N2T == d)                                            \
    return gcm_fused::launch_bwd<a, b_, c, d>(s, g_mx, g_nodes_out, x, adj, cur_idx,           \
                                              num_nodes_in, P, mx, h1, agg1, agg2, g_nodes_in, \
                                              g_obs, slabs, accumulate, B, N, F, H1, H2);
  GCM_SHAPES(GCM_B)
#undef GCM_B
  return GCM_EUNSUPPORTED;
}

extern "C" int gcm_sum_slabs(const float* slabs, int n_slabs, int len, float* out,
                             gcm_stream_t stream) {
  return gcm_sum_slabs_acc(slabs, n_slabs, len, nullptr, out, stream);
}

extern "C" int gcm_sum_slabs_acc(const float* slabs, int n_slabs, int len, const float* prev,
                                 float* out, gcm_stream_t stream) {
  GCM_REQUIRE(slabs && out && n_slabs > 0 && len > 0);
  hipLaunchKernelGGL(gcm_fused::k_sum_slabs, dim3((len + 15) / 16), dim3(256), 0,
                     (hipStream_t)stream, slabs, n_slabs, len, prev, out);
  return gcm_launch_status();
}
