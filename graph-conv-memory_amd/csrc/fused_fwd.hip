// Fused two-layer DenseGraphConv step, forward (see fused_common.h for the design notes).
#include "live_gnn.h"

#ifdef GCM_STAMPS
__device__ unsigned long long g_stamps[32];
extern "C" int gcm_debug_read_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
#endif

namespace gcm_fused {

// state of the previous step + what the advance needs (ADV variant: gcm.py:262-287 folded in)
struct Advance {
  const float* obs;            // [B,F]
  const int64_t* count_in;     // [B]
  float* nodes_out;            // [B,N,F]
  float* adj_out;              // [B,N,N]
  int64_t* cur_out;            // [B]
  int64_t* count_out;          // [B]
  Edits edits;
};

// ADV = false: x / adj are the already advanced state (nodes_out, adj_out), cur_idx given.
// ADV = true : x / adj are the PREVIOUS state; the kernel inserts obs, rolls on overflow, applies
//              the folded selectors, writes nodes_out / adj_out / cur / count and runs the GNN -
//              one kernel per forward step (needs N % 4 == 0 and F % 4 == 0).
template <int NT, int NCT, int NHT, int N2T, bool EXACT, bool ADV>
__device__ __forceinline__ void gnn2_row_fwd_body(
    const float* __restrict__ x, const float* __restrict__ adj, const int64_t* __restrict__ cur_idx,
    const Advance& A, const Gnn2& P, float* __restrict__ mx_out, float* __restrict__ h1_out,
    float* __restrict__ agg1_out, float* __restrict__ agg2_out, uint32_t* __restrict__ flags,
    int N_, int F_, int H1_, int H2_) {
  using L = Lds<NT, NCT, NHT, N2T>;
  constexpr int NP = L::NP, FP = L::FP, HP = L::HP, H2P = L::H2P;
  constexpr int FS = L::FS, HS = L::HS, AS = L::AS, W2S = L::W2S;
  // EXACT: the dims ARE the padded sizes (compile-time constants, every bounds check folds away)
  const int N = EXACT ? NP : N_, F = EXACT ? FP : F_, H1 = EXACT ? HP : H1_, H2 = EXACT ? H2P : H2_;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const float* xg = x + (size_t)b * N * F;
  const float* ag = adj + (size_t)b * N * N;
  float* h1g = h1_out ? h1_out + (size_t)b * N * H1 : nullptr;
  float* a1g = agg1_out ? agg1_out + (size_t)b * N * F : nullptr;
  bool wrap = false;
  int cur;
  if (ADV) {
    const int64_t n_in = A.count_in[b];
    wrap = n_in + 1 > N;
    int64_t c64 = wrap ? n_in - 1 : n_in;
    cur = c64 < 0 ? 0 : (c64 > N - 1 ? N - 1 : (int)c64);
    if (tid == 0) {
      A.cur_out[b] = cur;
      A.count_out[b] = cur + 1;
      const uint32_t f = (wrap ? GCM_FLAG_WRAPPED : 0u) | ((n_in < 0 || n_in > N) ? GCM_FLAG_BAD_COUNT : 0u);
      if (f) atomicOr(flags, f);
    }
  } else {
    const int64_t cur64 = cur_idx[b];
    cur = cur64 < 0 ? 0 : (cur64 > N - 1 ? N - 1 : (int)cur64);
  }

  extern __shared__ float smem[];
  float* sAdj = smem;
  float* sX = sAdj + L::ADJ;             // x; after layer 1: the layer-2 weights [o][rel|root]
  float* sAH = sX + L::X;                // agg, later h1 (stride AS)
  float* sW1 = sAH + L::AH;              // w_rel1^T [f][HS] | w_root1^T [f][HS]    (B(k=f, j=h))
  float* sV = sW1 + L::W1F;              // [0,256) partials | [256, 256+2HP) v = agg2|h1[cur]
  float* sVv = sV + 256;
  float* sW2 = L::W2_IN_X ? sX : sV + L::SV;   // [o][rel k | root k], stride W2S

  STAMP(0);
  // ---- issue every load: x, layer-1 weights, this wave's adjacency rows, layer-2 weights ----
  const int r_base = wave * 32;
  int lane_hop = -1, lane_dir = 0;   // lane i < n_hops: the i-th folded temporal edit
  if (ADV && lane < A.edits.n_hops) {
    lane_hop = A.edits.hops[lane & 15];
    lane_dir = A.edits.dir[lane & 15];
  }
  bool wave_live = wave < NT;
  if (!ADV && wave_live && r_base > cur) {
    // A 32-row block beyond row cur is needed only if row cur aggregates from it (layer 2 reads h1[j] where
    // adj[cur, j] != 0; the backward passes read the rows j < cur and the live rows): look at row cur's entries
    // in this block first - one more round trip for the waves that most probably have nothing to do, and a
    // partly filled graph (LearnedEdge rollouts from empty graphs) no longer streams its empty rows.
    const int j = r_base + (lane & 31);
    const float a = ag[(size_t)cur * N + (j < N ? j : N - 1)];
    wave_live = __ballot(j < N && a != 0.f) != 0ull;
  }
  Stage<NP, FP, false, EXACT> st_x;
  Stage<HP, FP, true, EXACT> st_wr, st_wo;
  Stage<H2P, HP, false, EXACT> st_w2r, st_w2o;
  AdjRows<NT, EXACT> rows;
  if (ADV) {
    // previous nodes with the roll applied; row cur takes the observation (gcm.py:274)
    const float ob = A.obs[(size_t)b * F + min(tid % FP, F - 1)];   // 256 % FP == 0: fixed column
#pragma unroll
    for (int i = 0; i < st_x.PER; ++i) {
      const int e = tid + 256 * i, r = e / FP, c = e % FP;
      const int rs = r + (wrap ? 1 : 0);
      const float t = xg[(rs < N ? rs : N - 1) * F + (c < F ? c : F - 1)];
      float v = (rs < N && r < N && c < F) ? t : 0.f;
      if (r == cur && c < F) v = ob;
      st_x.v[i] = v;
    }
  } else {
    st_x.load(xg, N, F, F, tid);
  }
  st_wr.load(P.w_rel1, H1, F, F, tid);
  st_wo.load(P.w_root1, H1, F, F, tid);
  if (wave_live) {
    if (ADV) rows.load_advanced(ag, N, r_base, lane, wrap);
    else rows.load(ag, N, r_base, lane);
  }
  st_w2r.load(P.w_rel2, H2, H1, H1, tid);
  st_w2o.load(P.w_root2, H2, H1, H1, tid);
  STAMP(1);
  if (ADV) {  // the new node matrix goes back to HBM as it lands in LDS
    float* no = A.nodes_out + (size_t)b * N * F;
#pragma unroll
    for (int i = 0; i < st_x.PER; ++i) {
      const int e = tid + 256 * i, r = e / FP, c = e % FP;
      if (EXACT || (r < N && c < F)) no[r * F + c] = st_x.v[i];
    }
  }
  st_x.store(sX, FS, tid);
  st_wr.store(sW1, HS, tid);
  st_wo.store(sW1 + FP * HS, HS, tid);
  STAMP(2);
  __syncthreads();  // x and W1 are in LDS; the adjacency / W2 loads are still in flight
  STAMP(3);

  if (wave_live) {
    if (ADV) {
      rows.apply_edits(A.edits, lane_hop, lane_dir, cur, r_base, lane);
      rows.store_global(A.adj_out + (size_t)b * N * N, N, r_base, lane);
    }
    // ---- layer 1, aggregation: agg = adj[rows,:] @ x, K tile by K tile ------------------
    f32x16 acc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      rows.template store_tile<NP>(sAdj, t, r_base, lane);   // always: layer 2 / backward read it
      if (rows.tile_nonzero(t)) {                            // exact: a zero tile adds nothing
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < NCT; ++c)
          mma32(acc[c], sAdj + (t * NP + r_base) * 33, 33, 1, sX + (t * 32) * FS + c * 32, FS, 1,
                32, li, lh);
      }
    }
    STAMP(4);
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r_base + acc_row(r, lh), col = c * 32 + li;
        sAH[row * AS + col] = acc[c][r];
        if (a1g && (EXACT || (row < N && col < F))) a1g[row * F + col] = acc[c][r];
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    STAMP(5);
    // ---- layer 1, linears: h1 = act1(agg W_rel1^T + x W_root1^T + b1) --------------------
    f32x16 o[NHT];
#pragma unroll
    for (int t = 0; t < NHT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
      mma32(o[t], sAH + r_base * AS, AS, 1, sW1 + t * 32, HS, 1, FP, li, lh);
      mma32(o[t], sX + r_base * FS, FS, 1, sW1 + FP * HS + t * 32, HS, 1, FP, li, lh);
    }
    __builtin_amdgcn_wave_barrier();  // every lane is done reading this wave's agg rows
    STAMP(6);
#pragma unroll
    for (int t = 0; t < NHT; ++t) {
      const int col = t * 32 + li;
      const float bias = (P.b_rel1 && (EXACT || col < H1)) ? P.b_rel1[EXACT ? col : min(col, H1 - 1)] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r_base + acc_row(r, lh);
        const bool ok = EXACT || (row < N && col < H1);
        const float v = ok ? gcm_act(o[t][r] + bias, P.act1) : 0.f;
        sAH[row * AS + col] = v;
        if (h1g && ok) h1g[row * H1 + col] = v;
      }
    }
  }
  STAMP(7);
  __syncthreads();  // all of h1 and adj are in LDS; nobody reads x any more
  STAMP(8);

  // ---- layer 2 on row `cur` only ---------------------------------------------------------
  st_w2r.store(sW2, W2S, tid);           // [o][k]        (rel half)
  st_w2o.store(sW2 + HP, W2S, tid);      // [o][HP + k]   (root half)
  {
    // agg2[h] = sum_j adj[cur][j] * h1[j][h]: 256/HP partial sums per h
    constexpr int G = 256 / HP;
    const int g = tid / HP, h = tid - g * HP;
    float s = 0.f;
#pragma unroll 4
    for (int j = g; j < N; j += G) {   // (h1 rows of skipped blocks are not in LDS: their coefficient is zero)
      const float a = sAdj[adj_at<NP>(cur, j)];
      s = a != 0.f ? fmaf(a, sAH[j * AS + h], s) : s;
    }
    sV[tid] = s;
    __syncthreads();
    if (tid < HP) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < G; ++q) t += sV[q * HP + tid];
      sVv[tid] = t;                              // v[0:HP)   = agg2
      sVv[HP + tid] = sAH[cur * AS + tid];       // v[HP:2HP) = h1[cur]
      if (agg2_out && tid < H1) agg2_out[(size_t)b * H1 + tid] = t;
    }
  }
  __syncthreads();
  {
    // pre2[o] = b2[o] + sum_k W2c[o][k] * v[k], K = 2*HP split over 256/H2P thread groups
    constexpr int G2 = 256 / H2P, KC = (2 * HP) / G2;
    const int g = tid / H2P, o = tid - g * H2P;
    float s = 0.f;
    const float* wrow = sW2 + o * W2S + g * KC;
    const float* vv = sVv + g * KC;
#pragma unroll
    for (int k = 0; k < KC; ++k) s = fmaf(wrow[k], vv[k], s);
    sV[tid] = s;
    __syncthreads();
    bool nonfinite = false;
    if (tid < H2) {
      float t = P.b_rel2 ? P.b_rel2[tid] : 0.f;
#pragma unroll
      for (int q = 0; q < G2; ++q) t += sV[q * H2P + tid];
      const float v = gcm_act(t, P.act2);
      mx_out[(size_t)b * H2 + tid] = v;
      nonfinite = !isfinite(v);
    }
    STAMP(9);
    if (wave == 0) {   // H2 <= 64: every output lives in wave 0
      const bool any_bad = __any(nonfinite);
      if (flags && any_bad && lane == 0) atomicOr(flags, GCM_FLAG_NONFINITE);
    }
  }
}

template <int NT, int NCT, int NHT, int N2T, bool EXACT>
__global__ __launch_bounds__(256) void k_gnn2_row_fwd(
    const float* __restrict__ x, const float* __restrict__ adj, const int64_t* __restrict__ cur_idx,
    Gnn2 P, float* __restrict__ mx_out, float* __restrict__ h1_out, float* __restrict__ agg1_out,
    float* __restrict__ agg2_out, uint32_t* __restrict__ flags, int N, int F, int H1, int H2) {
  Advance none{};
  gnn2_row_fwd_body<NT, NCT, NHT, N2T, EXACT, false>(x, adj, cur_idx, none, P, mx_out, h1_out,
                                                     agg1_out, agg2_out, flags, N, F, H1, H2);
}

template <int NT, int NCT, int NHT, int N2T, bool EXACT>
__global__ __launch_bounds__(256) void k_step_fwd(
    const float* __restrict__ nodes_in, const float* __restrict__ adj_in, Advance A, Gnn2 P,
    float* __restrict__ mx_out, float* __restrict__ h1_out, float* __restrict__ agg1_out,
    float* __restrict__ agg2_out, uint32_t* __restrict__ flags, int N, int F, int H1, int H2) {
  gnn2_row_fwd_body<NT, NCT, NHT, N2T, EXACT, true>(nodes_in, adj_in, nullptr, A, P, mx_out,
                                                    h1_out, agg1_out, agg2_out, flags, N, F, H1,
                                                    H2);
}


// ---------------------------------------------------------------------------
// One forward step, live-tile variant (tile-exact shapes).  The state copy (gcm.py:262-286 hands
// back fresh nodes / adjacency) streams HBM -> registers -> HBM; the GNN is evaluated only on the
// 32-row tiles that reach the belief (the tiles holding row cur and the non-zeros of adj[cur,:],
// gcm.py:314), shared by all four waves as 16x16 blocks on the 16x16x4 MFMA, and only those tiles
// of h1 / agg1 are saved for the backward (which reads no others).  Layer 2 runs on one wave.
// ---------------------------------------------------------------------------
template <int NT, int NCT, int NHT, int N2T>
struct LdsLive {
  using L = Lds<NT, NCT, NHT, N2T>;
  static constexpr int TOTAL = L::ADJ + L::X + L::AH + L::W1F + L::W2 + L::SV;
};

template <int NT, int NCT, int NHT, int N2T>
__global__ __launch_bounds__(256) void k_step_fwd_live(
    const float* __restrict__ nodes_in, const float* __restrict__ adj_in, Advance A, Gnn2 P,
    float* __restrict__ mx_out, float* __restrict__ h1_out, float* __restrict__ agg1_out,
    float* __restrict__ agg2_out, uint32_t* __restrict__ flags) {
  using L = Lds<NT, NCT, NHT, N2T>;
  constexpr int N = L::NP, F = L::FP, H1 = L::HP, H2 = L::H2P;
  constexpr int NP = N, FP = F, HP = H1, H2P = H2;
  constexpr int FS = L::FS, HS = L::HS, W2S = L::W2S;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* xg = nodes_in + (size_t)b * N * F;
  const float* ag = adj_in + (size_t)b * N * N;
  const int64_t n_in = A.count_in[b];
  const bool wrap = n_in + 1 > N;
  int64_t c64 = wrap ? n_in - 1 : n_in;
  const int cur = c64 < 0 ? 0 : (c64 > N - 1 ? N - 1 : (int)c64);
  if (tid == 0) {
    A.cur_out[b] = cur;
    A.count_out[b] = cur + 1;
    const uint32_t f = (wrap ? GCM_FLAG_WRAPPED : 0u) | ((n_in < 0 || n_in > N) ? GCM_FLAG_BAD_COUNT : 0u);
    if (f) atomicOr(flags, f);
  }

  extern __shared__ float smem[];
  float* sAdj = smem;                    // [col tile][row][33]
  float* sX = sAdj + L::ADJ;             // [N][FS]
  float* sAH = sX + L::X;                // agg, then h1 (stride AS), live tiles only
  float* sW1 = sAH + L::AH;              // w_rel1^T | w_root1^T  [f][HS]
  float* sW2 = sW1 + L::W1F;             // [o][rel k | root k], stride W2S
  float* sV = sW2 + L::W2;
  float* sVv = sV + 256;
  unsigned* sMask = reinterpret_cast<unsigned*>(sV);   // word R: non-zero col tiles of row tile R

  STAMP(0);
  // ---- every load, then the state copy ------------------------------------------------------
  const int r_base = wave * 32;
  int lane_hop = -1, lane_dir = 0;   // lane i < n_hops: the i-th folded temporal edit
  if (lane < A.edits.n_hops) {
    lane_hop = A.edits.hops[lane & 15];
    lane_dir = A.edits.dir[lane & 15];
  }
  const bool wave_rows = wave < NT;
  Stage<NP, FP, false, true> st_x;
  Stage<HP, FP, true, true> st_wr, st_wo;
  Stage<H2P, HP, false, true> st_w2r, st_w2o;
  AdjRows<NT, true> rows;
  {
    const float ob = A.obs[(size_t)b * F + (tid % FP)];   // 256 % FP == 0: fixed column
#pragma unroll
    for (int i = 0; i < st_x.PER; ++i) {
      const int e = tid + 256 * i, r = e / FP, c = e % FP;
      const int rs = r + (wrap ? 1 : 0);
      const float t = xg[(rs < N ? rs : N - 1) * F + c];
      float v = rs < N ? t : 0.f;
      if (r == cur) v = ob;
      st_x.v[i] = v;
    }
  }
  st_wr.load(P.w_rel1, H1, F, F, tid);
  st_wo.load(P.w_root1, H1, F, F, tid);
  if (wave_rows) rows.load_advanced(ag, N, r_base, lane, wrap);
  st_w2r.load(P.w_rel2, H2, H1, H1, tid);
  st_w2o.load(P.w_root2, H2, H1, H1, tid);
  LiveGnn<NT, NCT, NHT, N2T> G;
  G.sAdj = sAdj; G.sX = sX; G.sAH = sAH; G.sW1 = sW1; G.sW2 = sW2; G.sVv = sVv; G.sMask = sMask;
  G.init_lane(P, tid);
  STAMP(1);
  {
    float* no = A.nodes_out + (size_t)b * N * F;
#pragma unroll
    for (int i = 0; i < st_x.PER; ++i) no[tid + 256 * i] = st_x.v[i];
  }
  STAMP(2);
  st_x.store(sX, FS, tid);
  st_wr.store(sW1, HS, tid);
  st_wo.store(sW1 + FP * HS, HS, tid);
  st_w2r.store(sW2, W2S, tid);
  st_w2o.store(sW2 + HP, W2S, tid);
  STAMP(3);
  {
    unsigned bits = 0;
    if (wave_rows) {
      rows.apply_edits(A.edits, lane_hop, lane_dir, cur, r_base, lane);
      rows.store_global(A.adj_out + (size_t)b * N * N, N, r_base, lane);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        rows.template store_tile<NP>(sAdj, t, r_base, lane);
        bits |= (rows.tile_nonzero(t) ? 1u : 0u) << t;
      }
    }
    if (lane == 0) sMask[wave] = bits;
  }
  STAMP(4);
  __syncthreads();

  STAMP(5);
  // ---- the live tiles (live_gnn.h) ------------------------------------------------------------
  unsigned nzmask;
  bool lvt[NT];
  G.flags(cur, nzmask, lvt);
  STAMP(6);
  G.run(cur, nzmask, lvt, agg1_out ? agg1_out + (size_t)b * N * F : nullptr,
        h1_out ? h1_out + (size_t)b * N * H1 : nullptr,
        agg2_out ? agg2_out + (size_t)b * H1 : nullptr, mx_out + (size_t)b * H2, flags);
  STAMP(7);
}

template <int NT, int NCT, int NHT, int N2T>
int launch_step(hipStream_t s, const float* nodes_in, const float* adj_in, Advance A, Gnn2 P,
                float* mx, float* h1, float* agg1, float* agg2, uint32_t* flags, int B, int N,
                int F, int H1, int H2) {
  using L = Lds<NT, NCT, NHT, N2T>;
  constexpr size_t lds = sizeof(float) * (size_t)L::FWD;
  const bool exact = N == L::NP && F == L::FP && H1 == L::HP && H2 == L::H2P;
  constexpr size_t lds_live = sizeof(float) * (size_t)LdsLive<NT, NCT, NHT, N2T>::TOTAL;
  if (exact && lds_live <= 160 * 1024) {   // live-tile kernel
    auto kl = k_step_fwd_live<NT, NCT, NHT, N2T>;
    gcm_allow_dynamic_lds((const void*)kl, lds_live);
    hipLaunchKernelGGL(kl, dim3(B), dim3(256), lds_live, s, nodes_in, adj_in, A, P, mx, h1, agg1,
                       agg2, flags);
    return gcm_launch_status();
  }
  auto kern = exact ? k_step_fwd<NT, NCT, NHT, N2T, true> : k_step_fwd<NT, NCT, NHT, N2T, false>;
  gcm_allow_dynamic_lds((const void*)kern, lds);
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, s, nodes_in, adj_in, A, P, mx, h1, agg1, agg2,
                     flags, N, F, H1, H2);
  return gcm_launch_status();
}

template <int NT, int NCT, int NHT, int N2T>
int launch_fwd(hipStream_t s, const float* x, const float* adj, const int64_t* cur, Gnn2 P,
               float* mx, float* h1, float* agg1, float* agg2, uint32_t* flags, int B, int N,
               int F, int H1, int H2) {
  using L = Lds<NT, NCT, NHT, N2T>;
  constexpr size_t lds = sizeof(float) * (size_t)L::FWD;
  const bool exact = N == L::NP && F == L::FP && H1 == L::HP && H2 == L::H2P;
  auto kern = exact ? k_gnn2_row_fwd<NT, NCT, NHT, N2T, true> : k_gnn2_row_fwd<NT, NCT, NHT, N2T, false>;
  gcm_allow_dynamic_lds((const void*)kern, lds);
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, s, x, adj, cur, P, mx, h1, agg1, agg2, flags,
                     N, F, H1, H2);
  return gcm_launch_status();
}

}  // namespace gcm_fused

extern "C" int gcm_dense_gnn2_row_supported(int N, int F, int H1, int H2) {
  if (N <= 0 || F <= 0 || H1 <= 0 || H2 <= 0) return 0;
  if (N > 128 || F > 64 || H1 > 64 || H2 > 64) return 0;
  const int NT = (N + 31) / 32, NCT = (F + 31) / 32, NHT = (H1 + 31) / 32, N2T = (H2 + 31) / 32;
  size_t fwd, bwd;
  gcm_fused::lds_need(NT, NCT, NHT, N2T, &fwd, &bwd);
  return (fwd <= 160 * 1024 && bwd <= 160 * 1024) ? 1 : 0;
}

extern "C" size_t gcm_dense_gnn2_param_count(int F, int H1, int H2) {
  return 2 * (size_t)H1 * F + H1 + 2 * (size_t)H2 * H1 + H2;
}

extern "C" int gcm_dense_gnn2_row_fwd(const float* x, const float* adj, const int64_t* cur_idx,
                                      const float* w_rel1, const float* b_rel1,
                                      const float* w_root1, int act1, const float* w_rel2,
                                      const float* b_rel2, const float* w_root2, int act2,
                                      float* mx, float* h1, float* agg1, float* agg2,
                                      uint32_t* flags, int B, int N, int F, int H1, int H2,
                                      gcm_stream_t stream) {
  GCM_REQUIRE(x && adj && cur_idx && w_rel1 && w_root1 && w_rel2 && w_root2 && mx);
  GCM_REQUIRE(B > 0);
  if (!gcm_dense_gnn2_row_supported(N, F, H1, H2)) return GCM_EUNSUPPORTED;
  gcm_fused::Gnn2 P{w_rel1, b_rel1, w_root1, w_rel2, b_rel2, w_root2, act1, act2};
  hipStream_t s = (hipStream_t)stream;
  const int NT = (N + 31) / 32, NCT = (F + 31) / 32, NHT = (H1 + 31) / 32, N2T = (H2 + 31) / 32;
#define GCM_F(a, b_, c, d)                                                                   \
  if (NT == a && NCT == b_ && NHT == c && N2T == d)                                          \
    return gcm_fused::launch_fwd<a, b_, c, d>(s, x, adj, cur_idx, P, mx, h1, agg1, agg2,     \
                                              flags, B, N, F, H1, H2);
  GCM_SHAPES(GCM_F)
#undef GCM_F
  return GCM_EUNSUPPORTED;
}

/* One kernel per forward step: state advance (gcm.py:262-278, 323-355) + temporal/dense selector
 * writes + fused GNN.  Only index-writing selectors can be folded (GCM_SEL_TEMPORAL /
 * GCM_SEL_DENSE); returns GCM_EUNSUPPORTED otherwise or when N % 4 or F % 4 != 0 - the caller
 * then uses the three-kernel sequence. */
extern "C" int gcm_dense_step_fused_fwd(const float* obs, const float* nodes_in,
                                        const float* adj_in, const int64_t* count_in,
                                        float* nodes_out, float* adj_out, int64_t* cur_out,
                                        int64_t* count_out, const gcm_selector_desc* selectors,
                                        int n_selectors, const float* w_rel1, const float* b_rel1,
                                        const float* w_root1, int act1, const float* w_rel2,
                                        const float* b_rel2, const float* w_root2, int act2,
                                        float* mx, float* h1, float* agg1, float* agg2,
                                        uint32_t* flags, int B, int N, int F, int H1, int H2,
                                        gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes_in && adj_in && count_in && nodes_out && adj_out && cur_out &&
              count_out && w_rel1 && w_root1 && w_rel2 && w_root2 && mx && flags);
  GCM_REQUIRE(B > 0 && (selectors || n_selectors == 0));
  if (!gcm_dense_gnn2_row_supported(N, F, H1, H2) || (N & 3) || (F & 3)) return GCM_EUNSUPPORTED;
  gcm_fused::Advance A{};
  A.obs = obs; A.count_in = count_in; A.nodes_out = nodes_out; A.adj_out = adj_out;
  A.cur_out = cur_out; A.count_out = count_out;
  for (int i = 0; i < n_selectors; ++i) {
    const gcm_selector_desc& d = selectors[i];
    if (d.kind == GCM_SEL_TEMPORAL) {
      for (int k = 0; k < d.n_hops; ++k) {
        if (A.edits.n_hops >= 16) return GCM_EUNSUPPORTED;
        A.edits.hops[A.edits.n_hops] = d.hops[k];
        A.edits.dir[A.edits.n_hops++] = d.direction;
      }
    } else if (d.kind == GCM_SEL_DENSE) {
      A.edits.dense = 1;
    } else {
      return GCM_EUNSUPPORTED;
    }
  }
  gcm_fused::Gnn2 P{w_rel1, b_rel1, w_root1, w_rel2, b_rel2, w_root2, act1, act2};
  hipStream_t s = (hipStream_t)stream;
  const int NT = (N + 31) / 32, NCT = (F + 31) / 32, NHT = (H1 + 31) / 32, N2T = (H2 + 31) / 32;
#define GCM_S(a, b_, c, d)                                                                    \
  if (NT == a && NCT == b_ && NHT == c && N2T == d)                                           \
    return gcm_fused::launch_step<a, b_, c, d>(s, nodes_in, adj_in, A, P, mx, h1, agg1, agg2, \
                                               flags, B, N, F, H1, H2);
  GCM_SHAPES(GCM_S)
#undef GCM_S
  return GCM_EUNSUPPORTED;
}
