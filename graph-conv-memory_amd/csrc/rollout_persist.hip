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
#include "live_gnn.h"

#ifdef GCM_STAMPS   // diagnostic build only (make stamps2): phase stamps of step GCM_STAMP_STEP
__device__ unsigned long long g_stamps[32];
extern "C" int gcm_debug_read_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
#define RSTAMP(i) do { if (t == GCM_STAMP_STEP) STAMP(i); } while (0)
#else
#define RSTAMP(i)
#endif

namespace gcm_fused {

template <int NT, int NCT, int NHT, int N2T>
struct LdsRoll {
  using L = Lds<NT, NCT, NHT, N2T>;
  static constexpr int BASE = L::ADJ + L::X + L::AH + L::W1F + L::W2 + L::SV;
  // observation ring: two chunks of CH steps.  Global loads share the wave's in-order vmcnt with
  // the history stores, so waiting for a load drains every store issued before it: the ring makes
  // that happen once per CH steps instead of every step.
  static constexpr int ROOM = (160 * 1024 / 4 - BASE) / (2 * L::FP);
  static constexpr int CH = ROOM >= 16 ? 16 : (ROOM >= 8 ? 8 : (ROOM >= 4 ? 4 : (ROOM >= 2 ? 2 : 1)));
  static constexpr int TOTAL = BASE + 2 * CH * L::FP;
};

template <int NT, int NCT, int NHT, int N2T>
__global__ __launch_bounds__(256) void k_rollout_fwd(
    const float* __restrict__ obs, float* __restrict__ nodes_all, float* __restrict__ adj_all,
    int64_t* __restrict__ count_all, int64_t* __restrict__ cur_all, Edits E, Gnn2 P,
    float* __restrict__ mx_all, float* __restrict__ h1_all, float* __restrict__ agg1_all,
    float* __restrict__ agg2_all, uint32_t* __restrict__ flags, int T, int B, int hist) {
  using L = Lds<NT, NCT, NHT, N2T>;
  constexpr int N = L::NP, F = L::FP, H1 = L::HP, H2 = L::H2P;
  constexpr int NP = N, FP = F, HP = H1, H2P = H2;
  constexpr int FS = L::FS, HS = L::HS, W2S = L::W2S;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const size_t nodes_sz = (size_t)B * N * F, adj_sz = (size_t)B * N * N;

  extern __shared__ float smem[];
  float* sAdj = smem;                    // [col tile][row][33]   the graph's adjacency, resident
  float* sX = sAdj + L::ADJ;             // [N][FS]               the graph's nodes, resident
  float* sAH = sX + L::X;                // agg, then h1 (stride AS)
  float* sW1 = sAH + L::AH;              // w_rel1^T | w_root1^T  [f][HS]
  float* sW2 = sW1 + L::W1F;             // [o][rel k | root k], stride W2S
  float* sV = sW2 + L::W2;               // scratch: tile mask | v
  float* sVv = sV + 256;
  // non-zero 32x32 tiles of the resident adjacency: word R holds the bits of row tile R.  Kept up to
  // date by the edits (LDS atomics) and the roll; a superset would only cost MFMAs on zero tiles.
  unsigned* sMask = reinterpret_cast<unsigned*>(sV);
  float* sObs = sV + L::SV;              // [2][CH][FP] observation ring
  constexpr int CH = LdsRoll<NT, NCT, NHT, N2T>::CH;
  constexpr int OPER = (CH * FP + 255) / 256;   // ring-chunk elements per thread

  const int r_base = wave * 32;
  const bool wave_rows = wave < NT;      // rows [r_base, r_base+32) exist (roll / prologue ownership)

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
    if (wave_rows) rows.load(adj_all + (size_t)b * N * N, N, r_base, lane);
    st_x.store(sX, FS, tid);
    st_wr.store(sW1, HS, tid);
    st_wo.store(sW1 + FP * HS, HS, tid);
    st_w2r.store(sW2, W2S, tid);
    st_w2o.store(sW2 + HP, W2S, tid);
    unsigned bits = 0;
    if (wave_rows) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        rows.template store_tile<NP>(sAdj, t, r_base, lane);
        bits |= (rows.tile_nonzero(t) ? 1u : 0u) << t;
      }
    }
    if (lane == 0) sMask[wave] = bits;
  }
  // h1 rows of tiles that are not live in a step keep older (finite) values; they only ever meet
  // adj[cur][j] == 0 in layer 2, so they must start finite too
  for (int e = tid; e < L::AH; e += 256) sAH[e] = 0.f;
  int64_t n = count_all[b];
  // every per-step operand that lives in HBM is read once, here (a global load inside the loop
  // would share the in-order vmcnt with the history stores and wait for all of them)
  LiveGnn<NT, NCT, NHT, N2T> G;
  G.sAdj = sAdj; G.sX = sX; G.sAH = sAH; G.sW1 = sW1; G.sW2 = sW2; G.sVv = sVv; G.sMask = sMask;
  G.init_lane(P, tid);
  const int my_hop = tid < E.n_hops ? E.hops[tid < 16 ? tid : 0] : -1;
  const int my_dir = tid < E.n_hops ? E.dir[tid < 16 ? tid : 0] : 0;
  // observations: chunk 0 into the ring now, chunk 1 into registers
  float ob_regs[OPER];
#pragma unroll
  for (int i = 0; i < OPER; ++i) {
    const int e = tid + 256 * i, st = e / FP, c = e % FP;
    const int tt = st < T ? st : T - 1;
    ob_regs[i] = obs[((size_t)tt * B + b) * F + c];
  }
#pragma unroll
  for (int i = 0; i < OPER; ++i) {
    const int e = tid + 256 * i;
    if (e < CH * FP) sObs[e] = ob_regs[i];
  }
#pragma unroll
  for (int i = 0; i < OPER; ++i) {
    const int e = tid + 256 * i, st = CH + e / FP, c = e % FP;
    const int tt = st < T ? st : T - 1;
    ob_regs[i] = obs[((size_t)tt * B + b) * F + c];
  }
  __syncthreads();
#pragma unroll 1
  for (int t = 0; t < T; ++t) {
    RSTAMP(0);
    if (t % CH == 0 && t > 0) {   // next ring chunk: registers -> LDS, then fetch the one after
      const int k = t / CH;
#pragma unroll
      for (int i = 0; i < OPER; ++i) {
        const int e = tid + 256 * i;
        if (e < CH * FP) sObs[(k & 1) * CH * FP + e] = ob_regs[i];
      }
#pragma unroll
      for (int i = 0; i < OPER; ++i) {
        const int e = tid + 256 * i, st = (k + 1) * CH + e / FP, c = e % FP;
        const int tt = st < T ? st : T - 1;
        ob_regs[i] = obs[((size_t)tt * B + b) * F + c];
      }
      __syncthreads();
    }
    const bool bad = n < 0 || n > N;
    const bool wrap = n + 1 > N;
    int64_t c64 = wrap ? n - 1 : n;
    const int cur = c64 < 0 ? 0 : (c64 > N - 1 ? N - 1 : (int)c64);

    // ---- overflow: rotate both images one slot towards index 0 (registers as the bounce buffer)
    if (wrap) {   // workgroup-uniform
      float av[NT * 16], xv[NP * FP / 256];
      unsigned rbits = 0;
      if (wave_rows) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          bool nz = false;
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int r = r_base + (lane >> 3) + 8 * q + 1, c = tt * 32 + (lane & 7) * 4 + k + 1;
              const float v = (r < N && c < N) ? sAdj[adj_at<NP>(r, c)] : 0.f;
              av[(tt * 4 + q) * 4 + k] = v;
              nz |= v != 0.f;
            }
          rbits |= (__any(nz) ? 1u : 0u) << tt;
        }
      }
#pragma unroll
      for (int i = 0; i < NP * FP / 256; ++i) {
        const int e = tid + 256 * i, r = e / FP + 1, c = e % FP;
        xv[i] = r < N ? sX[r * FS + c] : 0.f;
      }
      __syncthreads();
      if (wave_rows) {
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
      if (lane == 0) sMask[wave] = rbits;
      __syncthreads();
    }
    // ---- insert the observation, apply the selector writes ---------------------------------
    if (tid < FP) sX[cur * FS + tid] = sObs[((t / CH) & 1) * CH * FP + (t % CH) * FP + tid];
    if (my_hop >= 0 && cur >= my_hop) {
      const int past = cur - my_hop;
      if (my_dir & GCM_DIR_FORWARD) {
        sAdj[adj_at<NP>(cur, past)] = 1.f;
        atomicOr(&sMask[cur >> 5], 1u << (past >> 5));
      }
      if (my_dir & GCM_DIR_BACKWARD) {
        sAdj[adj_at<NP>(past, cur)] = 1.f;
        atomicOr(&sMask[past >> 5], 1u << (cur >> 5));
      }
    }
    if (E.dense) {
      for (int j = tid; j <= cur; j += 256) {
        sAdj[adj_at<NP>(cur, j)] = 1.f;
        if (j < cur) sAdj[adj_at<NP>(j, cur)] = 1.f;
      }
      if (tid <= (cur >> 5)) {
        atomicOr(&sMask[cur >> 5], 1u << tid);
        atomicOr(&sMask[tid], 1u << (cur >> 5));
      }
    }
    if (tid == 0) {
      if (hist) {
        cur_all[(size_t)t * B + b] = cur;
        count_all[(size_t)(t + 1) * B + b] = cur + 1;
      } else if (t == T - 1) {
        count_all[(size_t)B + b] = cur + 1;
      }
      const uint32_t f = (wrap ? GCM_FLAG_WRAPPED : 0u) | (bad ? GCM_FLAG_BAD_COUNT : 0u);
      if (f) atomicOr(flags, f);
    }
    __syncthreads();
    RSTAMP(1);
    // ---- the live tiles (live_gnn.h): history stores first (fire-and-forget), then the GNN --------
    unsigned nzmask;
    bool lvt[NT];
    G.flags(cur, nzmask, lvt);
    const bool last = t == T - 1;
    const size_t slot = hist ? (size_t)(t + 1) : 1;
    float* no = nodes_all + slot * nodes_sz + (size_t)b * N * F;
    float* ao = adj_all + slot * adj_sz + (size_t)b * N * N;
#pragma unroll
    for (int R = 0; R < NT; ++R)
      if (last || (hist && lvt[R])) {   // 8 rows of the tile per wave
        const int row8 = R * 32 + wave * 8;
#pragma unroll
        for (int i = 0; i < FP / 32; ++i) {
          const int e4 = lane + 64 * i, r = row8 + e4 / (FP / 4), c = (e4 % (FP / 4)) * 4;
          const float* s = sX + r * FS + c;
          *reinterpret_cast<float4*>(no + r * F + c) = make_float4(s[0], s[1], s[2], s[3]);
        }
        const int r = row8 + (lane >> 3);
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const float* s = sAdj + (tt * NP + r) * 33 + (lane & 7) * 4;
          *reinterpret_cast<float4*>(ao + r * N + tt * 32 + (lane & 7) * 4) =
              make_float4(s[0], s[1], s[2], s[3]);
        }
      }
    RSTAMP(2);
    G.run(cur, nzmask, lvt, agg1_all ? agg1_all + (size_t)t * nodes_sz + (size_t)b * N * F : nullptr,
          h1_all ? h1_all + ((size_t)t * B + b) * N * H1 : nullptr,
          agg2_all ? agg2_all + ((size_t)t * B + b) * H1 : nullptr, mx_all + ((size_t)t * B + b) * H2,
          flags);
    n = cur + 1;
    RSTAMP(8);
    __syncthreads();   // the next step rewrites sX / sAdj / sAH / sV
    RSTAMP(9);
  }
}

template <int NT, int NCT, int NHT, int N2T>
int launch_rollout(hipStream_t s, const float* obs, float* nodes_all, float* adj_all,
                   int64_t* count_all, int64_t* cur_all, Edits E, Gnn2 P, float* mx_all,
                   float* h1_all, float* agg1_all, float* agg2_all, uint32_t* flags, int T, int B,
                   int hist) {
  constexpr size_t lds = sizeof(float) * (size_t)LdsRoll<NT, NCT, NHT, N2T>::TOTAL;
  if (lds > 160 * 1024) return GCM_EUNSUPPORTED;
  auto kern = k_rollout_fwd<NT, NCT, NHT, N2T>;
  gcm_allow_dynamic_lds((const void*)kern, lds);
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, s, obs, nodes_all, adj_all, count_all, cur_all,
                     E, P, mx_all, h1_all, agg1_all, agg2_all, flags, T, B, hist);
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
template <int PER>   // ceil(N*F / 256) for the common sizes, 32 = any N*F <= 8192
__global__ __launch_bounds__(256) void k_gnodes_scan(
    const float* __restrict__ Q_all, const float* __restrict__ pobs_all,
    const float* __restrict__ g_nodes_T, const int64_t* __restrict__ cur_all,
    const int64_t* __restrict__ count_all, float* __restrict__ g_obs_all,
    float* __restrict__ g_nodes_0, int T, int B, int N, int F) {
  // LDS: two copies of the running gradient (ping-pong: one barrier per step) + every step's
  // (cur | wrap << 16), read once up front - a scalar global load per step would sit on the
  // critical path of a loop that has nothing else to wait for
  extern __shared__ float smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int NF = N * F;
  float* sG0 = smem;
  int* sCur = reinterpret_cast<int*>(smem + 2 * NF);
  const size_t nodes_sz = (size_t)B * NF;
  for (int t = tid; t < T; t += 256) {
    int64_t c64 = cur_all[(size_t)t * B + b];
    const int cur = c64 < 0 ? 0 : (c64 > N - 1 ? N - 1 : (int)c64);
    const int wrap = count_all[(size_t)t * B + b] + 1 > N ? 1 : 0;
    sCur[t] = cur | (wrap << 16);
  }
  for (int e = tid; e < NF; e += 256) sG0[e] = g_nodes_T ? g_nodes_T[(size_t)b * NF + e] : 0.f;

  // Q and pobs D steps ahead of their use: every step reads a fresh 16 KB from a new region
  // (HBM + TLB latency is several steps of this loop)
  constexpr int D = PER <= 16 ? 6 : 3;
  float q[D][PER], po[D];
  // (unconditional, clamped: with skipped fetches on some path hipcc must assume the minimum
  // number of younger loads and waits for all but the last fetch - an effective depth of one;
  // pobs first, so that waiting for it never covers the Q loads issued with it)
  auto fetch = [&](float (&q)[PER], float& po_, int t_) {
    const int t = t_ < 0 ? 0 : t_;
    po_ = pobs_all[((size_t)t * B + b) * F + (tid < F ? tid : F - 1)];
    const float* Qt = Q_all + (size_t)t * nodes_sz + (size_t)b * NF;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 256 * i;
      q[i] = Qt[e < NF ? e : NF - 1];
    }
  };
#pragma unroll
  for (int d = 0; d < D; ++d) {
    po[d] = 0.f;
    fetch(q[d], po[d], T - 1 - d);
  }
  __syncthreads();

  // ping-pong by OFFSET into the one LDS array: swapping two pointers makes hipcc fall back to flat
  // loads/stores, whose waits also cover the global prefetches in flight
  int so = 0, dof = NF;
  auto step = [&](float (&q)[PER], float po_, int t) {
    const int cw = sCur[t], cur = cw & 0xffff;
    const bool wrap = (cw >> 16) != 0;
    if (tid < F) g_obs_all[((size_t)t * B + b) * F + tid] = smem[so + cur * F + tid] + po_;
    // C_t[r] = U(C_{t+1})[r] + Q_t[r];  U(G)[r] = wrap ? (r >= 1 ? G[r-1] : 0) : (r == cur ? 0 : G[r])
    // (no division: row r == cur <=> e - cur*F in [0, F); r >= 1 <=> e >= F)
    const int sh = wrap ? F : 0;
    const unsigned c0 = (unsigned)(cur * F);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 256 * i;
      if (e < NF) {
        const int es = e - sh;
        float u = smem[so + (es < 0 ? 0 : es)];
        const bool zero = wrap ? e < F : ((unsigned)e - c0) < (unsigned)F;
        smem[dof + e] = (zero ? 0.f : u) + q[i];
      }
    }
    __syncthreads();
    const int tmp = so;
    so = dof;
    dof = tmp;
  };
  int t0 = T - 1;
  for (; t0 >= D - 1; t0 -= D) {   // full groups of D steps: no conditionals around the loads
#pragma unroll
    for (int d = 0; d < D; ++d) {
      step(q[d], po[d], t0 - d);
      fetch(q[d], po[d], t0 - d - D);
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d)      // the last < D steps (their operands are already in registers)
    if (t0 - d >= 0) step(q[d], po[d], t0 - d);
  for (int e = tid; e < NF; e += 256) g_nodes_0[(size_t)b * NF + e] = smem[so + e];
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
                               float* agg1_all, float* agg2_all, uint32_t* flags, int history,
                               int T, int B, int N, int F, int H1, int H2, gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes_all && adj_all && count_all && mx_all && flags && w_rel1 && w_root1 &&
              w_rel2 && w_root2 && (cur_all || !history));
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
                                                  agg2_all, flags, T, B, history);
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
  const size_t lds = sizeof(float) * 2 * (size_t)N * F + sizeof(int) * (size_t)T;
  if (lds > 160 * 1024) return GCM_EUNSUPPORTED;   // T > ~24k steps: the caller's step-by-step schedule
  const int nper = (N * F + 255) / 256;
#define GCM_SCAN(P)                                                                              \
  {                                                                                              \
    gcm_allow_dynamic_lds((const void*)gcm_fused::k_gnodes_scan<P>, lds); \
    hipLaunchKernelGGL(gcm_fused::k_gnodes_scan<P>, dim3(B), dim3(256), lds, s, Q_all, pobs_all, \
                       g_nodes_T, cur_all, count_all, g_obs_all, g_nodes_0, T, B, N, F);         \
    return gcm_launch_status();                                                                  \
  }
  if (nper <= 4) GCM_SCAN(4)
  if (nper <= 8) GCM_SCAN(8)
  if (nper <= 16) GCM_SCAN(16)
  GCM_SCAN(32)
#undef GCM_SCAN
}
