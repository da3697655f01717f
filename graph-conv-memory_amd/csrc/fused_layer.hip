// One DenseGraphConv layer, forward and backward, for graphs that fit one workgroup
// (N <= 128, Fi <= 64, Fo <= 64) - the layered path's fast kernels, built like the fused step
// (fused_common.h): one workgroup per graph, adjacency rows wave-private in LDS, every load issued
// up front, all-zero 32x32 tiles skipped.  Larger shapes use the tiled kernels in graphconv.hip.
//
//   out = act( (adj @ x) W_rel^T + b + x W_root^T )
//   backward: G = g_out * act'(out);  dW_rel = G^T agg, dW_root = G^T x, db = colsum G (slab per graph)
//             g_x = adj^T (G W_rel) + G W_root;   g_adj = (G W_rel) x^T   (optional)
#include "fused_common.h"

namespace gcm_fused {

template <int NT, int NCT, int NHT>
struct LdsL {
  static constexpr int NP = 32 * NT, FP = 32 * NCT, HP = 32 * NHT;
  static constexpr int FS = FP + 1, HS = HP + 1, AS = FS > HS ? FS : HS;
  static constexpr int ADJ = NT * NP * 33;
  static constexpr int FWD = ADJ + NP * FS + NP * AS + 2 * FP * HS;
  // backward without / with the x image (x is only needed for g_adj)
  static constexpr int BWD = ADJ + NP * HS + NP * FS + 2 * HP * FS + 4 * 1024 + 256 + 32;
  static constexpr int BWD_X = NP * FS;
};

inline void lds_need_layer(int NT, int NCT, int NHT, int want_adj, size_t* fwd, size_t* bwd) {
  const size_t NP = 32 * NT, FP = 32 * NCT, HP = 32 * NHT, FS = FP + 1, HS = HP + 1;
  const size_t AS = FS > HS ? FS : HS, ADJ = NT * NP * 33;
  *fwd = sizeof(float) * (ADJ + NP * FS + NP * AS + 2 * FP * HS);
  *bwd = sizeof(float) * (ADJ + NP * HS + NP * FS + (want_adj ? NP * FS : 0) + 2 * HP * FS +
                          4 * 1024 + 256 + 32);
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <int NT, int NCT, int NHT, bool EXACT>
__global__ __launch_bounds__(256) void k_layer_fwd(
    const float* __restrict__ x, const float* __restrict__ adj, const float* __restrict__ w_rel,
    const float* __restrict__ b_rel, const float* __restrict__ w_root, float* __restrict__ out,
    float* __restrict__ agg_out, int N_, int F_, int H_, int act) {
  using L = LdsL<NT, NCT, NHT>;
  constexpr int NP = L::NP, FP = L::FP, HP = L::HP, FS = L::FS, HS = L::HS, AS = L::AS;
  const int N = EXACT ? NP : N_, F = EXACT ? FP : F_, H = EXACT ? HP : H_;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const float* xg = x + (size_t)b * N * F;
  const float* ag = adj + (size_t)b * N * N;
  float* og = out + (size_t)b * N * H;
  float* a1g = agg_out ? agg_out + (size_t)b * N * F : nullptr;

  extern __shared__ float smem[];
  float* sAdj = smem;
  float* sX = sAdj + L::ADJ;
  float* sA = sX + NP * FS;               // agg (stride AS)
  float* sW = sA + NP * AS;               // w_rel^T [f][HS] | w_root^T [f][HS]

  const int r_base = wave * 32;
  const bool wave_live = wave < NT;
  Stage<NP, FP, false, EXACT> st_x;
  Stage<HP, FP, true, EXACT> st_wr, st_wo;
  AdjRows<NT, EXACT> rows;
  st_x.load(xg, N, F, F, tid);
  st_wr.load(w_rel, H, F, F, tid);
  st_wo.load(w_root, H, F, F, tid);
  if (wave_live) rows.load(ag, N, r_base, lane);
  st_x.store(sX, FS, tid);
  st_wr.store(sW, HS, tid);
  st_wo.store(sW + FP * HS, HS, tid);
  __syncthreads();
  if (!wave_live) return;

  f32x16 acc[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (rows.tile_nonzero(t)) {
      rows.template store_tile<NP>(sAdj, t, r_base, lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < NCT; ++c)
        mma32(acc[c], sAdj + (t * NP + r_base) * 33, 33, 1, sX + (t * 32) * FS + c * 32, FS, 1, 32,
              li, lh);
    }
  }
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = r_base + acc_row(r, lh), col = c * 32 + li;
      sA[row * AS + col] = acc[c][r];
      if (a1g && (EXACT || (row < N && col < F))) a1g[row * F + col] = acc[c][r];
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int t = 0; t < NHT; ++t) {
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    mma32(o, sA + r_base * AS, AS, 1, sW + t * 32, HS, 1, FP, li, lh);
    mma32(o, sX + r_base * FS, FS, 1, sW + FP * HS + t * 32, HS, 1, FP, li, lh);
    const int col = t * 32 + li;
    const float bias = (b_rel && (EXACT || col < H)) ? b_rel[EXACT ? col : min(col, H - 1)] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = r_base + acc_row(r, lh);
      if (EXACT || (row < N && col < H)) og[row * H + col] = gcm_act(o[r] + bias, act);
    }
  }
}

// ---------------------------------------------------------------------------
// backward.  slab layout: dW_rel [H*F] | dW_root [H*F] | db [H]
// ---------------------------------------------------------------------------
template <int NT, int NCT, int NHT, bool EXACT>
__global__ __launch_bounds__(256) void k_layer_bwd(
    const float* __restrict__ g_out, const float* __restrict__ out, const float* __restrict__ x,
    const float* __restrict__ adj, const float* __restrict__ agg, const float* __restrict__ w_rel,
    const float* __restrict__ w_root, float* __restrict__ g_x, float* __restrict__ g_adj,
    float* __restrict__ slabs, int want_w, int N_, int F_, int H_, int act) {
  using L = LdsL<NT, NCT, NHT>;
  constexpr int NP = L::NP, FP = L::FP, HP = L::HP, FS = L::FS, HS = L::HS;
  const int N = EXACT ? NP : N_, F = EXACT ? FP : F_, H = EXACT ? HP : H_;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const float* xg = x + (size_t)b * N * F;
  const float* ag = adj + (size_t)b * N * N;
  const float* gog = g_out + (size_t)b * N * H;
  const float* og = out + (size_t)b * N * H;
  const float* a1g = agg + (size_t)b * N * F;
  float* slab = slabs + (size_t)b * (2 * (size_t)H * F + H);

  extern __shared__ float smem[];
  float* sAdj = smem;                   // [col tile][row][33]
  float* sG = sAdj + L::ADJ;            // [NP][HS]  G
  float* sD = sG + NP * HS;             // [NP][FS]  dAgg
  float* sW = sD + NP * FS;             // w_rel [h][FS] | w_root [h][FS]
  float* sR = sW + 2 * HP * FS;         // [4][1024]
  float* sV = sR + 4 * 1024;            // [256] partials
  int* sFlag = reinterpret_cast<int*>(sV + 256);   // [0,16) adj tile non-zero, [16,20) G row tile live
  float* sX = sV + 256 + 32;            // [NP][FS]  x, present only when g_adj is wanted

  const int r_base = wave * 32;
  const bool wave_live = wave < NT;
  Stage<NP, HP, false, EXACT> st_go, st_o;
  Stage<HP, FP, false, EXACT> st_wr, st_wo;
  Stage<NP, FP, false, EXACT> st_x;
  AdjRows<NT, EXACT> rows;
  st_go.load(gog, N, H, H, tid);
  st_o.load(og, N, H, H, tid);
  st_wr.load(w_rel, H, F, F, tid);
  st_wo.load(w_root, H, F, F, tid);
  if (g_adj) st_x.load(xg, N, F, F, tid);
  if (wave_live) rows.load(ag, N, r_base, lane);

  // G = g_out * act'(out)
  float part = 0.f;
#pragma unroll
  for (int i = 0; i < st_go.PER; ++i) {
    const int e = tid + 256 * i, r = e / HP, c = e % HP;
    const float v = st_go.v[i] * gcm_act_grad(st_o.v[i], act);   // padding: g_out staged as 0
    st_go.v[i] = v;
    part += v;
    (void)r; (void)c;
  }
  st_go.store(sG, HS, tid);
  sV[tid] = part;                       // column sums (db): a thread always holds the same column
  st_wr.store(sW, FS, tid);
  st_wo.store(sW + HP * FS, FS, tid);
  if (g_adj) st_x.store(sX, FS, tid);
  if (wave_live) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      rows.template store_tile<NP>(sAdj, t, r_base, lane);
      const bool nz = rows.tile_nonzero(t);
      if (lane == 0) sFlag[wave * 4 + t] = nz ? 1 : 0;
    }
  }
  __syncthreads();
  // G row tiles that hold anything (wave w checks tile w, shares through LDS)
  if (wave_live) {
    bool nz = false;
    for (int c = lh; c < HP; c += 2) nz |= sG[(r_base + li) * HS + c] != 0.f;
    const bool live = __any(nz);
    if (lane == 0) sFlag[16 + wave] = live ? 1 : 0;
  }
  if (want_w && tid < H) {
    constexpr int G = 256 / HP;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < G; ++q) t += sV[q * HP + tid];
    slab[2 * H * F + tid] = t;
  }
  __syncthreads();
  const bool my_rows_live = wave_live && sFlag[16 + (wave_live ? wave : 0)] != 0;

  // ---- parameter gradients: [H x F] = G^T (H x N) @ {agg, x} (N x F) ------------------------
  if (want_w) {
#pragma unroll 1
    for (int job = 0; job < 2 * NHT * NCT; ++job) {
      const int which = job & 1, ct = (job >> 1) % NCT, ht = (job >> 1) / NCT;
      const float* src = which ? xg : a1g;
      f32x16 a;
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = 0.f;
      if (my_rows_live) {
        float bq[16];
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
        const float* ap = sG + (r_base + lh) * HS + ht * 32 + li;
#pragma unroll
        for (int s = 0; s < 16; ++s)
          a = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s * HS], bq[s], a, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) sR[wave * 1024 + acc_row(r, lh) * 32 + li] = a[r];
      __syncthreads();
      float* dst = slab + (which ? H * F : 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, hh = ht * 32 + (e >> 5), ff = ct * 32 + (e & 31);
        if (EXACT || (hh < H && ff < F))
          dst[hh * F + ff] = (sR[e] + sR[1024 + e]) + (sR[2048 + e] + sR[3072 + e]);
      }
      __syncthreads();
    }
  }

  // ---- dAgg = G @ W_rel -> LDS ;  acc = G @ W_root ------------------------------------------
  f32x16 acc[NCT];
  if (wave_live) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
      if (my_rows_live) {
        f32x16 d;
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = 0.f;
        mma32(d, sG + r_base * HS, HS, 1, sW + c * 32, FS, 1, HP, li, lh);
        mma32(acc[c], sG + r_base * HS, HS, 1, sW + HP * FS + c * 32, FS, 1, HP, li, lh);
#pragma unroll
        for (int r = 0; r < 16; ++r) sD[(r_base + acc_row(r, lh)) * FS + c * 32 + li] = d[r];
      }
    }
    // ---- g_adj[rows, :] = dAgg[rows, :] @ x^T  (zero rows where G is zero) ------------------
    if (g_adj) {
      float* gag = g_adj + (size_t)b * N * N;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        f32x16 g;
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = 0.f;
        if (my_rows_live)   // A(i=row, k=f) = dAgg ; B(k=f, j=node) = x[j][f]
          mma32(g, sD + r_base * FS, FS, 1, sX + (jt * 32) * FS, 1, FS, FP, li, lh);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r_base + acc_row(r, lh), col = jt * 32 + li;
          if (EXACT || (row < N && col < N)) gag[row * N + col] = g[r];
        }
      }
    }
  }
  __syncthreads();
  // ---- g_x[i] = acc + sum_k adj[k][i] * dAgg[k] -----------------------------------------------
  if (wave_live && g_x) {
    float* gxg = g_x + (size_t)b * N * F;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      if (sFlag[16 + kt] && sFlag[kt * 4 + wave]) {
#pragma unroll
        for (int c = 0; c < NCT; ++c)
          mma32(acc[c], sAdj + (wave * NP + kt * 32) * 33, 1, 33, sD + (kt * 32) * FS + c * 32, FS,
                1, 32, li, lh);
      }
    }
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r_base + acc_row(r, lh), col = c * 32 + li;
        if (EXACT || (row < N && col < F)) gxg[row * F + col] = acc[c][r];
      }
  }
}

template <int NT, int NCT, int NHT>
int launch_layer_fwd(hipStream_t s, const float* x, const float* adj, const float* w_rel,
                     const float* b_rel, const float* w_root, float* out, float* agg, int B, int N,
                     int F, int H, int act) {
  using L = LdsL<NT, NCT, NHT>;
  constexpr size_t lds = sizeof(float) * (size_t)L::FWD;
  const bool exact = N == L::NP && F == L::FP && H == L::HP;
  auto kern = exact ? k_layer_fwd<NT, NCT, NHT, true> : k_layer_fwd<NT, NCT, NHT, false>;
  gcm_allow_dynamic_lds((const void*)kern, lds);
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, s, x, adj, w_rel, b_rel, w_root, out, agg, N,
                     F, H, act);
  return gcm_launch_status();
}

template <int NT, int NCT, int NHT>
int launch_layer_bwd(hipStream_t s, const float* g_out, const float* out, const float* x,
                     const float* adj, const float* agg, const float* w_rel, const float* w_root,
                     float* g_x, float* g_adj, float* slabs, int want_w, int B, int N, int F,
                     int H, int act) {
  using L = LdsL<NT, NCT, NHT>;
  const size_t lds = sizeof(float) * ((size_t)L::BWD + (g_adj ? L::BWD_X : 0));
  const bool exact = N == L::NP && F == L::FP && H == L::HP;
  auto kern = exact ? k_layer_bwd<NT, NCT, NHT, true> : k_layer_bwd<NT, NCT, NHT, false>;
  gcm_allow_dynamic_lds((const void*)kern, lds);
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, s, g_out, out, x, adj, agg, w_rel, w_root, g_x,
                     g_adj, slabs, want_w, N, F, H, act);
  return gcm_launch_status();
}

}  // namespace gcm_fused

#define GCM_LSHAPES_N(X, a) X(a, 1, 1) X(a, 1, 2) X(a, 2, 1) X(a, 2, 2)
#define GCM_LSHAPES(X) GCM_LSHAPES_N(X, 1) GCM_LSHAPES_N(X, 2) GCM_LSHAPES_N(X, 3) GCM_LSHAPES_N(X, 4)

// internal entry points used by graphconv.hip's C ABI when the graph fits one workgroup
// which = 0: forward kernel, 1: backward without g_adj, 2: backward with g_adj
int gcm_layer_fits(int N, int Fi, int Fo, int which) {
  if (N <= 0 || N > 128 || Fi <= 0 || Fi > 64 || Fo <= 0 || Fo > 64) return 0;
  size_t f, bw;
  gcm_fused::lds_need_layer((N + 31) / 32, (Fi + 31) / 32, (Fo + 31) / 32, which == 2, &f, &bw);
  return (which == 0 ? f : bw) <= 160 * 1024;
}

int gcm_layer_fwd(const float* x, const float* adj, const float* w_rel, const float* b_rel,
                  const float* w_root, float* out, float* agg, int B, int N, int Fi, int Fo,
                  int act, hipStream_t s) {
  const int NT = (N + 31) / 32, NCT = (Fi + 31) / 32, NHT = (Fo + 31) / 32;
#define GCM_LF(a, b_, c)                      \
  if (NT == a && NCT == b_ && NHT == c)       \
    return gcm_fused::launch_layer_fwd<a, b_, c>(s, x, adj, w_rel, b_rel, w_root, out, agg, B, N, Fi, Fo, act);
  GCM_LSHAPES(GCM_LF)
#undef GCM_LF
  return GCM_EUNSUPPORTED;
}

int gcm_layer_bwd(const float* g_out, const float* out, const float* x, const float* adj,
                  const float* agg, const float* w_rel, const float* w_root, float* g_x,
                  float* g_adj, float* slabs, int want_w, int B, int N, int Fi, int Fo, int act,
                  hipStream_t s) {
  const int NT = (N + 31) / 32, NCT = (Fi + 31) / 32, NHT = (Fo + 31) / 32;
#define GCM_LB(a, b_, c)                      \
  if (NT == a && NCT == b_ && NHT == c)       \
    return gcm_fused::launch_layer_bwd<a, b_, c>(s, g_out, out, x, adj, agg, w_rel, w_root, g_x, g_adj, slabs, want_w, B, N, Fi, Fo, act);
  GCM_LSHAPES(GCM_LB)
#undef GCM_LB
  return GCM_EUNSUPPORTED;
}
