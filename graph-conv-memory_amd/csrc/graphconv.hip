// DenseGraphConv forward / backward on the fp32 matrix cores of gfx950.
//
//   out = act( (adj @ x) @ w_rel^T + b_rel + x @ w_root^T )          (PyG DenseGraphConv)
//
// All products run on v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, so results
// stay inside the 1e-5 rtol contract).  One workgroup = WAVES waves = 32*WAVES
// rows of ONE graph; every wave owns a 32-row strip.  Operands are staged through
// LDS with row strides that are odd in dwords so that the per-lane fragment reads
// (lane = matrix row or column) are bank-conflict free with ds_read_b32.
//
// Shapes are arbitrary (tests use N=7, F=3 ...): tiles are zero padded in LDS and
// stores are masked.  Fi, Fo <= 128.
#include "fused_common.h"

// single-workgroup-per-graph kernels (fused_layer.hip), used whenever the graph fits one
int gcm_layer_fits(int N, int Fi, int Fo, int which);
int gcm_layer_fwd(const float* x, const float* adj, const float* w_rel, const float* b_rel,
                  const float* w_root, float* out, float* agg, int B, int N, int Fi, int Fo,
                  int act, hipStream_t s);
int gcm_layer_bwd(const float* g_out, const float* out, const float* x, const float* adj,
                  const float* agg, const float* w_rel, const float* w_root, float* g_x,
                  float* g_adj, float* slabs, int want_w, int B, int N, int Fi, int Fo, int act,
                  hipStream_t s);

namespace {

constexpr int KT = 32;  // K tile of the N-long contractions

__host__ __device__ constexpr int round32(int v) { return (v + 31) & ~31; }

// acc(32x32) += A(32 x K) * B(K x 32), operands in LDS.
//   A(i,k) = a[i*ais + k*aks]      B(k,j) = b[k*bks + j*bjs]
// Fragment map of v_mfma_f32_32x32x2_f32: lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31].
__device__ __forceinline__ void mma32(f32x16& acc, const float* a, int ais, int aks,
                                      const float* b, int bks, int bjs, int K, int li, int lh) {
  const float* ap = a + li * ais + lh * aks;
  const float* bp = b + lh * bks + li * bjs;
#pragma unroll 4
  for (int k = 0; k < K; k += 2) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k * aks], bp[k * bks], acc, 0, 0, 0);
  }
}

// accumulator element r of lane (li, lh) is C[row = (r&3) + 8*(r>>2) + 4*lh][col = li]
__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// cooperative zero-padded 2-D copy global -> LDS: dst[r*dstride + c] = src[(r0+r)*sstride + c0+c]
__device__ __forceinline__ void stage(float* dst, int dstride, const float* src, int sstride,
                                      int r0, int c0, int rows, int cols, int rmax, int cmax) {
  const int total = rows * cols;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int r = e / cols, c = e - r * cols;
    const int gr = r0 + r, gc = c0 + c;
    dst[r * dstride + c] = (gr < rmax && gc < cmax) ? src[(size_t)gr * sstride + gc] : 0.f;
  }
}
// transposed variant: dst[c*dstride + r] = src[(r0+r)*sstride + c0+c]
__device__ __forceinline__ void stage_t(float* dst, int dstride, const float* src, int sstride,
                                        int r0, int c0, int rows, int cols, int rmax, int cmax) {
  const int total = rows * cols;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int r = e / cols, c = e - r * cols;
    const int gr = r0 + r, gc = c0 + c;
    dst[c * dstride + r] = (gr < rmax && gc < cmax) ? src[(size_t)gr * sstride + gc] : 0.f;
  }
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <int WAVES, int NCT>
__global__ __launch_bounds__(64 * WAVES) void k_graphconv_fwd(
    const float* __restrict__ x, const float* __restrict__ adj, const float* __restrict__ w_rel,
    const float* __restrict__ b_rel, const float* __restrict__ w_root, float* __restrict__ out,
    float* __restrict__ agg_out, int N, int Fi, int Fo, int act) {
  constexpr int RB = 32 * WAVES;
  constexpr int FiP = 32 * NCT;
  const int b = blockIdx.y, r0 = blockIdx.x * RB;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const float* xg = x + (size_t)b * N * Fi;
  const float* ag = adj + (size_t)b * N * N;

  extern __shared__ float smem[];
  float* sAdj = smem;                       // [RB][KT+1]
  float* sX = sAdj + RB * (KT + 1);         // [KT][FiP]
  float* sAgg = sX + KT * FiP;              // [RB][FiP+1]
  float* sXr = sAgg + RB * (FiP + 1);       // [RB][FiP+1]
  float* sW = sXr + RB * (FiP + 1);         // [FiP][33]   one w^T tile ([k][n]) at a time: rel, then root

  // ---- phase A: agg = adj[rows, :] @ x -------------------------------------
  f32x16 acc[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  for (int k0 = 0; k0 < N; k0 += KT) {
    stage(sAdj, KT + 1, ag, N, r0, k0, RB, KT, N, N);
    stage(sX, FiP, xg, Fi, k0, 0, KT, FiP, N, Fi);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCT; ++c)
      mma32(acc[c], sAdj + wave * 32 * (KT + 1), KT + 1, 1, sX + c * 32, FiP, 1, KT, li, lh);
    __syncthreads();
  }

  // ---- hand agg over to LDS (it becomes an A operand), stage x rows --------
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wave * 32 + acc_row(r, lh), col = c * 32 + li;
      sAgg[row * (FiP + 1) + col] = acc[c][r];
      if (agg_out && r0 + row < N && col < Fi)
        agg_out[((size_t)b * N + r0 + row) * Fi + col] = acc[c][r];
    }
  stage(sXr, FiP + 1, xg, Fi, r0, 0, RB, FiP, N, Fi);

  // ---- phase B: out = agg @ w_rel^T + x @ w_root^T + b ---------------------
  for (int o0 = 0; o0 < Fo; o0 += 32) {
    __syncthreads();
    stage_t(sW, 33, w_rel, Fi, o0, 0, 32, FiP, Fo, Fi);
    __syncthreads();
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    mma32(o, sAgg + wave * 32 * (FiP + 1), FiP + 1, 1, sW, 33, 1, FiP, li, lh);
    __syncthreads();
    stage_t(sW, 33, w_root, Fi, o0, 0, 32, FiP, Fo, Fi);
    __syncthreads();
    mma32(o, sXr + wave * 32 * (FiP + 1), FiP + 1, 1, sW, 33, 1, FiP, li, lh);
    const int col = o0 + li;
    const float bias = (b_rel && col < Fo) ? b_rel[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = r0 + wave * 32 + acc_row(r, lh);
      if (row < N && col < Fo) out[((size_t)b * N + row) * Fo + col] = gcm_act(o[r] + bias, act);
    }
  }
}

size_t fwd_lds_bytes(int waves, int FiP) {
  const int RB = 32 * waves;
  return sizeof(float) * ((size_t)RB * (KT + 1) + KT * FiP + 2 * RB * (FiP + 1) + FiP * 33);
}

template <int WAVES>
int launch_fwd(int nct, dim3 grid, size_t lds, hipStream_t s, const float* x, const float* adj,
               const float* w_rel, const float* b_rel, const float* w_root, float* out, float* agg,
               int N, int Fi, int Fo, int act) {
#define GCM_FWD_CASE(NCT)                                                                      \
  case NCT: {                                                                                  \
    auto kern = k_graphconv_fwd<WAVES, NCT>;                                                   \
    gcm_allow_dynamic_lds((const void*)kern, lds); \
    hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES), lds, s, x, adj, w_rel, b_rel, w_root, out, \
                       agg, N, Fi, Fo, act);                                                   \
    break;                                                                                     \
  }
  switch (nct) {
    GCM_FWD_CASE(1)
    GCM_FWD_CASE(2)
    GCM_FWD_CASE(3)
    GCM_FWD_CASE(4)
    default: return GCM_EUNSUPPORTED;
  }
#undef GCM_FWD_CASE
  return gcm_launch_status();
}

int pick_waves(int N, int FiP, size_t (*lds_fn)(int, int)) {
  const int want = N > 64 ? 4 : (N > 32 ? 2 : 1);
  for (int w = want; w >= 1; w >>= 1)
    if (lds_fn(w, FiP) <= 160 * 1024) return w;
  return 0;
}

// ---------------------------------------------------------------------------
// backward, kernel 1 (row-local):  G = g_out * act'(out)
//   dAgg = G @ w_rel        -> ws_dagg [B,N,Fi]
//   g_x  = G @ w_root       (kernel 2 adds adj^T @ dAgg)
//   slab[b, blk] = { G^T @ agg, G^T @ x, colsum(G) }      (reduced by kernel 3)
//   g_adj[rows, :] = dAgg @ x^T                              (optional)
// ---------------------------------------------------------------------------
template <int WAVES, int NCT>
__global__ __launch_bounds__(64 * WAVES) void k_graphconv_bwd_rows(
    const float* __restrict__ g_out, const float* __restrict__ out, const float* __restrict__ x,
    const float* __restrict__ agg, const float* __restrict__ w_rel,
    const float* __restrict__ w_root, float* __restrict__ g_x, float* __restrict__ g_adj,
    float* __restrict__ ws_dagg, float* __restrict__ slabs, int N, int Fi, int Fo, int act,
    int want_w) {
  constexpr int RB = 32 * WAVES;
  constexpr int FiP = 32 * NCT;
  const int FoP = round32(Fo);
  const int b = blockIdx.y, r0 = blockIdx.x * RB;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const float* xg = x + (size_t)b * N * Fi;

  extern __shared__ float smem[];
  float* sG = smem;                          // [RB][FoP+1]
  float* sW = sG + RB * (FoP + 1);           // [FoP][FiP+1]  w (natural [fo][fi]) - reused rel/root
  float* sAgg = sW + FoP * (FiP + 1);        // [RB][FiP+1]   agg rows, later dAgg rows
  float* sXr = sAgg + RB * (FiP + 1);        // [RB][FiP+1]   x rows
  float* sXk = sXr + RB * (FiP + 1);         // [32][FiP+1]   x tile for g_adj

  // G = g_out * act'(out), zero padded
  {
    const int total = RB * FoP;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int r = e / FoP, c = e - r * FoP;
      float v = 0.f;
      if (r0 + r < N && c < Fo) {
        const size_t g = ((size_t)b * N + r0 + r) * Fo + c;
        v = g_out[g] * gcm_act_grad(out[g], act);
      }
      sG[r * (FoP + 1) + c] = v;
    }
  }
  stage(sAgg, FiP + 1, agg + (size_t)b * N * Fi, Fi, r0, 0, RB, FiP, N, Fi);
  stage(sXr, FiP + 1, xg, Fi, r0, 0, RB, FiP, N, Fi);
  stage(sW, FiP + 1, w_rel, Fi, 0, 0, FoP, FiP, Fo, Fi);
  __syncthreads();

  // ---- parameter-gradient slabs: [Fo x Fi] = G^T(Fo x RB) @ {agg, x}(RB x Fi), K = RB rows
  if (want_w) {
    float* slab = slabs + ((size_t)b * gridDim.x + blockIdx.x) * (2 * (size_t)Fo * Fi + Fo);
    const int n_ot = FoP / 32;
    const int tiles = n_ot * NCT * 2;  // (fo tile, fi tile, which operand)
    for (int t = wave; t < tiles; t += WAVES) {
      const int which = t & 1, ct = (t >> 1) % NCT, ot = (t >> 1) / NCT;
      f32x16 a;
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = 0.f;
      // A(i=fo,k=row) = sG[row][ot*32+fo];  B(k=row,j=fi) = src[row][ct*32+fi]
      mma32(a, sG + ot * 32, 1, FoP + 1, (which ? sXr : sAgg) + ct * 32, FiP + 1, 1, RB, li, lh);
      float* dst = slab + (which ? (size_t)Fo * Fi : 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int fo = ot * 32 + acc_row(r, lh), fi = ct * 32 + li;
        if (fo < Fo && fi < Fi) dst[(size_t)fo * Fi + fi] = a[r];
      }
    }
    for (int c = threadIdx.x; c < Fo; c += blockDim.x) {
      float s = 0.f;
      for (int r = 0; r < RB; ++r) s += sG[r * (FoP + 1) + c];
      slab[2 * (size_t)Fo * Fi + c] = s;
    }
  }
  __syncthreads();  // everyone is done reading sAgg as `agg`

  // ---- dAgg = G @ w_rel  (K = Fo) -> LDS + workspace ------------------------
  f32x16 d[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) d[c][r] = 0.f;
    mma32(d[c], sG + wave * 32 * (FoP + 1), FoP + 1, 1, sW + c * 32, FiP + 1, 1, FoP, li, lh);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wave * 32 + acc_row(r, lh), col = c * 32 + li;
      sAgg[row * (FiP + 1) + col] = d[c][r];
      if (r0 + row < N && col < Fi) ws_dagg[((size_t)b * N + r0 + row) * Fi + col] = d[c][r];
    }
  }
  __syncthreads();
  // ---- g_x (root part) = G @ w_root -----------------------------------------
  stage(sW, FiP + 1, w_root, Fi, 0, 0, FoP, FiP, Fo, Fi);
  __syncthreads();
  if (g_x) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
#pragma unroll
      for (int r = 0; r < 16; ++r) d[c][r] = 0.f;
      mma32(d[c], sG + wave * 32 * (FoP + 1), FoP + 1, 1, sW + c * 32, FiP + 1, 1, FoP, li, lh);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r0 + wave * 32 + acc_row(r, lh), col = c * 32 + li;
        if (row < N && col < Fi) g_x[((size_t)b * N + row) * Fi + col] = d[c][r];
      }
    }
  }
  // ---- g_adj[rows, j] = dAgg[rows, :] . x[j, :]  (K = Fi) --------------------
  if (g_adj) {
    for (int j0 = 0; j0 < N; j0 += 32) {
      __syncthreads();
      stage(sXk, FiP + 1, xg, Fi, j0, 0, 32, FiP, N, Fi);
      __syncthreads();
      f32x16 a;
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = 0.f;
      // A(i=row,k=fi) = sAgg[row][fi];  B(k=fi,j) = sXk[j][fi]
      mma32(a, sAgg + wave * 32 * (FiP + 1), FiP + 1, 1, sXk, 1, FiP + 1, FiP, li, lh);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r0 + wave * 32 + acc_row(r, lh), col = j0 + li;
        if (row < N && col < N) g_adj[((size_t)b * N + row) * N + col] = a[r];
      }
    }
  }
}

size_t bwd_rows_lds_bytes(int waves, int FiP, int FoP) {
  const int RB = 32 * waves;
  return sizeof(float) *
         ((size_t)RB * (FoP + 1) + FoP * (FiP + 1) + 2 * RB * (FiP + 1) + 32 * (FiP + 1));
}

// ---------------------------------------------------------------------------
// backward, kernel 2:  g_x[cols blk] += adj[:, blk]^T @ dAgg      (K = N)
// ---------------------------------------------------------------------------
template <int WAVES, int NCT>
__global__ __launch_bounds__(64 * WAVES) void k_graphconv_bwd_adjT(
    const float* __restrict__ adj, const float* __restrict__ ws_dagg, float* __restrict__ g_x,
    int N, int Fi) {
  constexpr int RB = 32 * WAVES;
  constexpr int FiP = 32 * NCT;
  const int b = blockIdx.y, i0 = blockIdx.x * RB;  // i0: block of adj COLUMNS = rows of g_x
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const float* ag = adj + (size_t)b * N * N;
  const float* dg = ws_dagg + (size_t)b * N * Fi;

  extern __shared__ float smem[];
  float* sA = smem;               // [KT][RB]     adj[k0+k][i0+i]  (A(i,k) read down a column)
  float* sD = sA + KT * RB;       // [KT][FiP]    dAgg[k0+k][:]

  f32x16 acc[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  for (int k0 = 0; k0 < N; k0 += KT) {
    stage(sA, RB, ag, N, k0, i0, KT, RB, N, N);
    stage(sD, FiP, dg, Fi, k0, 0, KT, FiP, N, Fi);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCT; ++c)
      mma32(acc[c], sA + wave * 32, 1, RB, sD + c * 32, FiP, 1, KT, li, lh);
    __syncthreads();
  }
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = i0 + wave * 32 + acc_row(r, lh), col = c * 32 + li;
      if (row < N && col < Fi) g_x[((size_t)b * N + row) * Fi + col] += acc[c][r];
    }
}

size_t bwd_adjT_lds_bytes(int waves, int FiP) {
  return sizeof(float) * ((size_t)KT * 32 * waves + KT * FiP);
}

// ---------------------------------------------------------------------------
// backward, kernel 3: sum the per-(graph, row block) slabs -> g_w_rel, g_w_root, g_b_rel
// deterministic (fixed order), no atomics
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce_slabs(const float* __restrict__ slabs, int n_slabs,
                                                      int slab_len, float* __restrict__ g_w_rel,
                                                      float* __restrict__ g_w_root,
                                                      float* __restrict__ g_b_rel, int FoFi, int Fo) {
  // block = 16 elements x 16 slab groups; every thread sums its slabs with 8 loads in flight,
  // the 16 partials of an element meet in LDS (fixed order => deterministic)
  __shared__ float part[256];
  const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  const int ec = e < slab_len ? e : slab_len - 1;
  float s = 0.f;
  for (int i0 = grp; i0 < n_slabs; i0 += 16 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 16 * u;
      const float t = slabs[(size_t)(i < n_slabs ? i : n_slabs - 1) * slab_len + ec];
      v[u] = i < n_slabs ? t : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  if (grp == 0 && e < slab_len) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += part[g * 16 + el];
    if (e < FoFi) { if (g_w_rel) g_w_rel[e] = v; }
    else if (e < 2 * FoFi) { if (g_w_root) g_w_root[e - FoFi] = v; }
    else if (g_b_rel) g_b_rel[e - 2 * FoFi] = v;
  }
}

// ===========================================================================
// GraphConv over CSR (sparse path).  The neighbour reduction is a gather of whole
// feature rows (coalesced 4*Fi-byte reads, L2-served for temporal graphs) into an LDS
// tile; the two linears then run on the matrix cores exactly like the dense layer.
// ===========================================================================
template <int NCT>
__global__ __launch_bounds__(256) void k_csr_graphconv_fwd(
    const float* __restrict__ x, const int64_t* __restrict__ row_ptr,
    const int64_t* __restrict__ col, const float* __restrict__ w,
    const uint8_t* __restrict__ mask, const float* __restrict__ w_rel,
    const float* __restrict__ b_rel, const float* __restrict__ w_root, float* __restrict__ out,
    float* __restrict__ agg_out, int64_t M, int Fi, int Fo, int act) {
  constexpr int RB = 128;
  constexpr int FiP = 32 * NCT;
  const int64_t r0 = (int64_t)blockIdx.x * RB;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;

  extern __shared__ float smem[];
  float* sAgg = smem;                     // [RB][FiP+1]
  float* sXr = sAgg + RB * (FiP + 1);     // [RB][FiP+1]
  float* sW = sXr + RB * (FiP + 1);       // [FiP][33]  one weight tile at a time (rel, then root)

  for (int idx = threadIdx.x; idx < RB * FiP; idx += 256) {
    const int r = idx / FiP, f = idx - r * FiP;
    const int64_t row = r0 + r;
    float a = 0.f, xv = 0.f;
    if (row < M && f < Fi && (!mask || mask[row])) {
      xv = x[(size_t)row * Fi + f];
      const int64_t e1 = row_ptr[row + 1];
      for (int64_t e = row_ptr[row]; e < e1; ++e) {
        const int64_t c = col[e];
        if (mask && !mask[c]) continue;
        const float m = x[(size_t)c * Fi + f];
        // mul then add, separately rounded: same arithmetic as msg = x_j * w; index_add
        a = __fadd_rn(a, w ? __fmul_rn(w[e], m) : m);
      }
      if (agg_out) agg_out[(size_t)row * Fi + f] = a;
    } else if (row < M && f < Fi && agg_out) {
      agg_out[(size_t)row * Fi + f] = 0.f;
    }
    sAgg[r * (FiP + 1) + f] = a;
    sXr[r * (FiP + 1) + f] = xv;
  }

  for (int o0 = 0; o0 < Fo; o0 += 32) {
    __syncthreads();
    stage_t(sW, 33, w_rel, Fi, o0, 0, 32, FiP, Fo, Fi);
    __syncthreads();
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    mma32(o, sAgg + wave * 32 * (FiP + 1), FiP + 1, 1, sW, 33, 1, FiP, li, lh);
    __syncthreads();
    stage_t(sW, 33, w_root, Fi, o0, 0, 32, FiP, Fo, Fi);
    __syncthreads();
    mma32(o, sXr + wave * 32 * (FiP + 1), FiP + 1, 1, sW, 33, 1, FiP, li, lh);
    const int c = o0 + li;
    const float bias = (b_rel && c < Fo) ? b_rel[c] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = r0 + wave * 32 + acc_row(r, lh);
      if (row < M && c < Fo) {
        const bool live = !mask || mask[row];
        out[(size_t)row * Fo + c] = live ? gcm_act(o[r] + bias, act) : 0.f;
      }
    }
  }
}

size_t csr_fwd_lds_bytes(int FiP) {
  return sizeof(float) * ((size_t)2 * 128 * (FiP + 1) + FiP * 33);
}

// transpose aggregation through the CSC view:  g_x[j] += sum_{k in col j} w * dAgg[rows[k]]
__global__ void k_csr_scatter_T(const float* __restrict__ dagg, const int64_t* __restrict__ col_ptr,
                                const int64_t* __restrict__ rows, const int64_t* __restrict__ perm,
                                const float* __restrict__ w, const uint8_t* __restrict__ mask,
                                float* __restrict__ g_x, int64_t M, int Fi) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * Fi) return;
  const int64_t j = i / Fi;
  const int f = i - j * Fi;
  if (mask && !mask[j]) return;
  float a = g_x[i];
  const int64_t k1 = col_ptr[j + 1];
  for (int64_t k = col_ptr[j]; k < k1; ++k) {
    const int64_t d = rows[k];
    if (mask && !mask[d]) continue;
    const float wv = w ? w[perm[k]] : 1.f;
    a = fmaf(wv, dagg[(size_t)d * Fi + f], a);
  }
  g_x[i] = a;
}

// edge-weight gradient: g_w[e] = < x[src(e)], dAgg[dst(e)] >
__global__ void k_csr_edge_grad(const float* __restrict__ x, const float* __restrict__ dagg,
                                const int64_t* __restrict__ col_ptr,
                                const int64_t* __restrict__ rows, const int64_t* __restrict__ perm,
                                const uint8_t* __restrict__ mask, float* __restrict__ g_w,
                                int64_t M, int Fi) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  for (int64_t k = col_ptr[j]; k < col_ptr[j + 1]; ++k) {
    const int64_t d = rows[k];
    float s = 0.f;
    if (!mask || (mask[j] && mask[d]))
      for (int f = 0; f < Fi; ++f) s = fmaf(x[(size_t)j * Fi + f], dagg[(size_t)d * Fi + f], s);
    g_w[perm[k]] = s;
  }
}


// ===========================================================================
// Flat-row kernels of the CSR path, second generation (Fi, Fo <= 64): every global load of a
// phase is issued before its first use (register staging with fixed trip counts, clamped
// unconditional addresses) - the first generation spent most of its time in per-element
// load -> store chains.
// ===========================================================================

// CSR GraphConv forward.  One workgroup = 128 destination rows.  The block's row_ptr slice and
// its (contiguous) col / w slices go to LDS first, so the neighbour gather has ONE level of
// dependent global loads (x rows) and the 16 row elements a thread owns are gathered together.
template <int NCT, int NHT>
__global__ __launch_bounds__(256) void k_csr_fwd2(
    const float* __restrict__ x, const int64_t* __restrict__ row_ptr,
    const int64_t* __restrict__ col, const float* __restrict__ w,
    const uint8_t* __restrict__ mask, const float* __restrict__ w_rel,
    const float* __restrict__ b_rel, const float* __restrict__ w_root, float* __restrict__ out,
    float* __restrict__ agg_out, int64_t M, int Fi, int Fo, int act) {
  using namespace gcm_fused;
  constexpr int RB = 128, FP = 32 * NCT, HP = 32 * NHT, FS = FP + 1, HS = HP + 1;
  constexpr int ECAP = 1024;             // edges of the block kept in LDS
  constexpr int PER = RB * FP / 256;     // (row, feature) elements per thread
  const int64_t r0 = (int64_t)blockIdx.x * RB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int rows = (int)min((int64_t)RB, M - r0);

  extern __shared__ float smem[];
  float* sAgg = smem;                     // [RB][FS]   agg, then (second phase) the x rows
  float* sW = sAgg + RB * FS;             // w_rel^T [f][HS] | w_root^T [f][HS]
  int64_t* sE0 = reinterpret_cast<int64_t*>(sW + 2 * FP * HS + ((2 * FP * HS) & 1));   // 8-B aligned
  int* sPtr = reinterpret_cast<int*>(sE0 + 1);            // [RB+1] edge offsets relative to e0
  int* sCol = sPtr + RB + 2;              // [ECAP] source rows (32-bit: M < 2^31)
  float* sWe = reinterpret_cast<float*>(sCol + ECAP);     // [ECAP] edge weights

  // ---- weights, own rows, row_ptr slice --------------------------------------------------
  Stage<HP, FP, true, false> st_wr, st_wo;
  st_wr.load(w_rel, Fo, Fi, Fi, tid);
  st_wo.load(w_root, Fo, Fi, Fi, tid);
  int64_t my_ptr = 0;
  if (tid <= RB) my_ptr = row_ptr[r0 + (tid < rows ? tid : rows)];
  float xv[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = tid + 256 * i, r = e / FP, f = e % FP;
    const float t = x[(size_t)(r0 + (r < rows ? r : rows - 1)) * Fi + (f < Fi ? f : Fi - 1)];
    const bool live = r < rows && f < Fi && (!mask || mask[r0 + (r < rows ? r : rows - 1)]);
    xv[i] = live ? t : 0.f;
  }
  st_wr.store(sW, HS, tid);
  st_wo.store(sW + FP * HS, HS, tid);
  if (tid == 0) *sE0 = my_ptr;
  __syncthreads();
  const int64_t e0 = *sE0;
  if (tid <= RB) sPtr[tid] = (int)(my_ptr - e0);
  __syncthreads();
  const int n_edges = sPtr[rows];
  const int n_lds = min(n_edges, ECAP);
  for (int k = tid; k < n_lds; k += 256) {
    sCol[k] = (int)col[e0 + k];
    sWe[k] = w ? w[e0 + k] : 1.f;
  }
  __syncthreads();
  // ---- gather: agg[r][f] = sum_e w_e * x[col_e][f], my PER elements advance together ---------
  float ag[PER];
  int beg[PER], deg[PER], dmax = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = tid + 256 * i, r = e / FP;
    ag[i] = 0.f;
    const bool row_live = r < rows && (!mask || mask[r0 + (r < rows ? r : 0)]);
    beg[i] = sPtr[r < rows ? r : 0];
    deg[i] = row_live ? sPtr[(r < rows ? r : 0) + 1] - beg[i] : 0;
    dmax = max(dmax, deg[i]);
  }
  // wave-wide maximum degree so that every lane runs the same trip count
  for (int o = 32; o > 0; o >>= 1) dmax = max(dmax, __shfl_xor(dmax, o));
  const int f_mine = (tid % FP) < Fi ? (tid % FP) : Fi - 1;   // 256 % FP == 0: fixed column
  // The hot loop holds ONLY unconditional loads (LDS index -> one global x read per element):
  // every data-dependent choice (mask present, weights present, edge list fits LDS) is made by
  // block-uniform branches outside it.
  const bool fits = n_edges <= ECAP;
  if (fits && !mask) {
    for (int d = 0; d < dmax; ++d) {
      float m[PER], we[PER];
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const bool on = d < deg[i];
        const int k = on ? beg[i] + d : 0;
        const int c = n_edges > 0 ? sCol[k] : 0;
        we[i] = on ? sWe[k] : 0.f;
        const float t = x[(size_t)c * Fi + f_mine];
        m[i] = on ? t : 0.f;
      }
#pragma unroll
      for (int i = 0; i < PER; ++i)   // mul then add, separately rounded (msg = x_j * w; index_add)
        ag[i] = __fadd_rn(ag[i], w ? __fmul_rn(we[i], m[i]) : m[i]);
    }
  } else {
    // general path: masked subgraph and / or more edges than the LDS slice holds
    for (int d = 0; d < dmax; ++d) {
#pragma unroll 1
      for (int i = 0; i < PER; ++i) {
        if (d >= deg[i]) continue;
        const int k = beg[i] + d;
        const int64_t c = col[e0 + k];
        if (mask && !mask[c]) continue;
        const float t = x[(size_t)c * Fi + f_mine];
        ag[i] = __fadd_rn(ag[i], w ? __fmul_rn(w[e0 + k], t) : t);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = tid + 256 * i, r = e / FP, f = e % FP;
    const float a = (f < Fi) ? ag[i] : 0.f;
    sAgg[r * FS + f] = a;
    if (agg_out && r < rows && f < Fi) agg_out[(size_t)(r0 + r) * Fi + f] = a;
  }
  __syncthreads();
  // ---- the two linears on the matrix cores: agg @ W_rel^T first, then the image is reused
  //      for the x rows and x @ W_root^T is added ------------------------------------------------
  f32x16 oo[NHT];
#pragma unroll
  for (int t = 0; t < NHT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) oo[t][r] = 0.f;
    mma32(oo[t], sAgg + wave * 32 * FS, FS, 1, sW + t * 32, HS, 1, FP, li, lh);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = tid + 256 * i, r = e / FP, f = e % FP;
    sAgg[r * FS + f] = xv[i];
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NHT; ++t) {
    f32x16 o = oo[t];
    mma32(o, sAgg + wave * 32 * FS, FS, 1, sW + FP * HS + t * 32, HS, 1, FP, li, lh);
    const int c = t * 32 + li;
    const float bias = (b_rel && c < Fo) ? b_rel[c] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wave * 32 + acc_row(r, lh);
      if (row < rows && c < Fo) {
        const bool live = !mask || mask[r0 + row];
        out[(size_t)(r0 + row) * Fo + c] = live ? gcm_act(o[r] + bias, act) : 0.f;
      }
    }
  }
}

// CSR GraphConv forward, third generation, for Fi, Fo in {32, 64} exactly: the kernel is bound by
// HBM (x, agg and out once each; the neighbour rows mostly hit in cache), so everything is shaped
// for bytes in flight - ONE WAVE owns 32 destination rows end to end (no workgroup barrier
// anywhere), every global access of the node rows is 16 bytes per lane, the weights live in
// registers for the whole life of the (persistent) wave, and the two linears run on the 32x32x2 fp32
// MFMA with a K order in which every lane reads CONTIGUOUS k (lane half kk owns k in
// [kk FI/2, (kk+1) FI/2): 16-byte LDS reads of the A operand, 16-byte global loads of the weights).
// HAS_W / HAS_MASK: edge weights / k-hop row mask present (compile time: a data-dependent choice
// inside the gather loop would serialise its loads).
template <int FI, int FO, bool HAS_W, bool HAS_MASK, bool CHK = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_csr_fwd3(
    const float* __restrict__ x, const int64_t* __restrict__ row_ptr,
    const int64_t* __restrict__ col, const float* __restrict__ w,
    const uint8_t* __restrict__ mask, const float* __restrict__ w_rel,
    const float* __restrict__ b_rel, const float* __restrict__ w_root, float* __restrict__ out,
    float* __restrict__ agg_out, int64_t M, uint32_t* __restrict__ flags, int act, int n_tiles) {
  // CHK: GCM_FLAG_NONFINITE is raised in *flags when an output value is not finite - the check SparseGCM makes
  // on the rows it returns (sparse_gcm.py:201-203), for callers that hand `out` back as it is (compile time: the
  // sixteen class tests per lane and tile cost the HBM-bound kernel 4 % when they sat behind a run-time pointer)
  constexpr int CPR = FI / 4;     // 16-byte chunks per row
  constexpr int RPP = 64 / CPR;   // rows per pass of the wave
  constexpr int NP = 32 / RPP;    // passes per 32-row tile
  constexpr int KH = FI / 2;      // k values per lane half
  constexpr int NT = FO / 32;     // 32-column output tiles
  constexpr int AS = FI + 4;      // LDS row stride (16-byte aligned rows)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int act_v = gcm_vgpr(act);
  extern __shared__ float smem[];
  float* sA = smem + (size_t)wave * 2 * 32 * AS;   // this wave's agg tile, then its x tile
  float* sX = sA + 32 * AS;

  // weights -> LDS once per (persistent) workgroup, in operand order: sW[(m, nt, kk, n)][KH (+4)] with
  // m = 0 W_rel, 1 W_root - lane (n = li, kk = lh) reads its KH contiguous k, 16 bytes at a time
  constexpr int WS = KH + 4;
  float* sW = smem + (size_t)4 * 2 * 32 * AS;
  for (int e = tid; e < 2 * FO * CPR; e += 256) {
    const int m = e / (FO * CPR), rem = e - m * FO * CPR, n = rem / CPR, q = rem % CPR;
    const float4 v = *reinterpret_cast<const float4*>((m ? w_root : w_rel) + (size_t)n * FI + 4 * q);
    const int kk = (4 * q) / KH, k0 = 4 * q - kk * KH;
    *reinterpret_cast<float4*>(sW + (size_t)(((m * NT + n / 32) * 2 + kk) * 32 + (n & 31)) * WS + k0) = v;
  }
  float bias[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bias[nt] = b_rel ? b_rel[nt * 32 + li] : 0.f;
  __syncthreads();
  const int lr = lane / CPR, c4 = (lane % CPR) * 4;
  const int n_waves = gridDim.x * 4;
  // A tile's own rows and row_ptr entries are fetched one trip ahead (while the previous tile is in
  // its MFMA / store phase): a trip then consists of two dependent round trips (column indices,
  // neighbour rows) instead of three, and the wave has loads in flight in every phase.
  float4 xv[NP];
  int64_t q0[NP], q1[NP];
  auto fetch = [&](int64_t r0) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int64_t r = r0 + i * RPP + lr;
      const int64_t rc = r < M ? r : M - 1;   // (past the end: loaded, never used)
      xv[i] = *reinterpret_cast<const float4*>(x + (size_t)rc * FI + c4);
      q0[i] = row_ptr[rc];
      q1[i] = row_ptr[rc + 1];
    }
  };
  fetch((int64_t)(blockIdx.x * 4 + wave) * 32);
#pragma unroll 1
  for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += n_waves) {
    const int64_t r0 = (int64_t)tile * 32;
    float4 ag[NP];
    int64_t p0[NP];
    int deg[NP], dmax = 0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int rl = i * RPP + lr;
      const int64_t r = r0 + rl;
      bool live = r < M;
      if (HAS_MASK) live = live && mask[r < M ? r : M - 1] != 0;
      p0[i] = q0[i];
      deg[i] = live ? (int)(q1[i] - q0[i]) : 0;
      const float4 xz = live ? xv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(sX + rl * AS + c4) = xz;   // the x tile goes to LDS right away
      ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      dmax = max(dmax, deg[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dmax = max(dmax, __shfl_xor(dmax, o));
    // neighbour gather: the wave's rows advance together, one edge each per trip
#pragma unroll 1
    for (int d = 0; d < dmax; ++d) {
      float4 m[NP];
      float we[NP];
      bool on[NP];
      int64_t cc[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) {   // every load of a round trip is issued before the first use
        on[i] = d < deg[i];
        const int64_t e = on[i] ? p0[i] + d : 0;   // (dmax > 0 implies E > 0)
        cc[i] = col[e];
        we[i] = HAS_W ? w[e] : 1.f;
      }
      asm volatile("" ::: "memory");
      uint8_t mk[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        m[i] = *reinterpret_cast<const float4*>(x + (size_t)cc[i] * FI + c4);
        mk[i] = HAS_MASK ? mask[cc[i]] : (uint8_t)1;
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < NP; ++i) {   // mul then add, separately rounded (msg = x_j * w; index_add)
        float4 t = m[i];
        if (HAS_W) t = make_float4(__fmul_rn(we[i], t.x), __fmul_rn(we[i], t.y), __fmul_rn(we[i], t.z),
                                   __fmul_rn(we[i], t.w));
        const bool take = on[i] && mk[i] != 0;
        ag[i] = make_float4(take ? __fadd_rn(ag[i].x, t.x) : ag[i].x, take ? __fadd_rn(ag[i].y, t.y) : ag[i].y,
                            take ? __fadd_rn(ag[i].z, t.z) : ag[i].z, take ? __fadd_rn(ag[i].w, t.w) : ag[i].w);
      }
    }
    fetch(r0 + (int64_t)n_waves * 32);   // the next tile of this wave
    const bool full = r0 + 32 <= M;   // (uniform) every row of the tile exists: unpredicated stores
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int rl = i * RPP + lr;
      *reinterpret_cast<float4*>(sA + rl * AS + c4) = ag[i];
    }
    if (agg_out) {
      float* ab = agg_out + (size_t)(r0 + lr) * FI + c4;
      if (full) {
#pragma unroll
        for (int i = 0; i < NP; ++i) *reinterpret_cast<float4*>(ab + (size_t)i * RPP * FI) = ag[i];
      } else {
#pragma unroll
        for (int i = 0; i < NP; ++i)
          if (r0 + i * RPP + lr < M) *reinterpret_cast<float4*>(ab + (size_t)i * RPP * FI) = ag[i];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // out = act(agg @ W_rel^T + x @ W_root^T + b)
    float4 a4[KH / 4], x4[KH / 4];
#pragma unroll
    for (int q = 0; q < KH / 4; ++q) {
      a4[q] = *reinterpret_cast<const float4*>(sA + li * AS + lh * KH + 4 * q);
      x4[q] = *reinterpret_cast<const float4*>(sX + li * AS + lh * KH + 4 * q);
    }
    uint8_t mrow[16];
    if (HAS_MASK) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = r0 + acc_row(r, lh);
        mrow[r] = mask[row < M ? row : M - 1];
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* pr = sW + (size_t)(((0 * NT + nt) * 2 + lh) * 32 + li) * WS;
      const float* po = sW + (size_t)(((1 * NT + nt) * 2 + lh) * 32 + li) * WS;
#pragma unroll
      for (int q = 0; q < KH / 4; ++q) {
        const float4 wv = *reinterpret_cast<const float4*>(pr + 4 * q);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].x, wv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].y, wv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].z, wv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q].w, wv.w, acc, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < KH / 4; ++q) {
        const float4 wv = *reinterpret_cast<const float4*>(po + 4 * q);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x4[q].x, wv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x4[q].y, wv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x4[q].z, wv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x4[q].w, wv.w, acc, 0, 0, 0);
      }
      // acc[r] is row acc_row(r, lh) = (r & 3) + 8 (r >> 2) + 4 lh: one base pointer, constant offsets
      float* ob = out + (size_t)(r0 + 4 * lh) * FO + nt * 32 + li;
      float ov[16];
      bool bad = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float t = gcm_act_sel(acc[r] + bias[nt], act_v);
        ov[r] = (!HAS_MASK || mrow[r]) ? t : 0.f;
        if (CHK) bad = bad || !isfinite(ov[r]);
      }
      // (rows past M in a partial tile are computed from clamped loads of real rows: finite iff those are)
      if (CHK && __any(bad) && lane == 0) atomicOr(flags, GCM_FLAG_NONFINITE);
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[((r & 3) + 8 * (r >> 2)) * FO] = ov[r];
      } else {
        const int left = (int)(M - r0) - 4 * lh;   // rows of this lane half inside M
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if ((r & 3) + 8 * (r >> 2) < left) ob[((r & 3) + 8 * (r >> 2)) * FO] = ov[r];
      }
    }
    __builtin_amdgcn_wave_barrier();   // the tiles are rewritten by the next trip
  }
}

// Row-local part of the backward for flat rows:  G = g_out * act'(out);
//   ws_dagg = G @ W_rel ; g_x = G @ W_root (the transpose aggregation is added later) ;
//   slab[block] = { G^T agg , G^T x , colsum G }.
template <int NCT, int NHT>
__global__ __launch_bounds__(256) void k_rows_bwd2(
    const float* __restrict__ g_out, const float* __restrict__ out, const float* __restrict__ x,
    const float* __restrict__ agg, const float* __restrict__ w_rel,
    const float* __restrict__ w_root, float* __restrict__ g_x, float* __restrict__ ws_dagg,
    float* __restrict__ slabs, int64_t M, int Fi, int Fo, int act, int want_w) {
  using namespace gcm_fused;
  constexpr int RB = 128, FP = 32 * NCT, HP = 32 * NHT, FS = FP + 1, HS = HP + 1;
  const int64_t r0 = (int64_t)blockIdx.x * RB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int rows = (int)min((int64_t)RB, M - r0);
  const float* gog = g_out + (size_t)r0 * Fo;
  const float* og = out + (size_t)r0 * Fo;
  const float* xg = x + (size_t)r0 * Fi;
  const float* a1g = agg + (size_t)r0 * Fi;

  extern __shared__ float smem[];
  float* sG = smem;                  // [RB][HS]
  float* sW = sG + RB * HS;          // w_rel [h][FS] | w_root [h][FS]
  float* sR = sW + 2 * HP * FS;      // [4][1024]
  float* sV = sR + 4 * 1024;         // [256]

  Stage<RB, HP, false, false> st_go, st_o;
  Stage<HP, FP, false, false> st_wr, st_wo;
  st_go.load(gog, rows, Fo, Fo, tid);
  st_o.load(og, rows, Fo, Fo, tid);
  st_wr.load(w_rel, Fo, Fi, Fi, tid);
  st_wo.load(w_root, Fo, Fi, Fi, tid);
  float part = 0.f;
#pragma unroll
  for (int i = 0; i < st_go.PER; ++i) {
    const float v = st_go.v[i] * gcm_act_grad(st_o.v[i], act);
    st_go.v[i] = v;
    part += v;
  }
  st_go.store(sG, HS, tid);
  sV[tid] = part;
  st_wr.store(sW, FS, tid);
  st_wo.store(sW + HP * FS, FS, tid);
  __syncthreads();
  const int r_base = wave * 32;
  if (want_w) {
    float* slab = slabs + (size_t)blockIdx.x * (2 * (size_t)Fo * Fi + Fo);
    if (tid < Fo) {
      constexpr int G = 256 / HP;
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < G; ++q) t += sV[q * HP + tid];
      slab[2 * (size_t)Fo * Fi + tid] = t;
    }
#pragma unroll 1
    for (int job = 0; job < 2 * NHT * NCT; ++job) {
      const int which = job & 1, ct = (job >> 1) % NCT, ht = (job >> 1) / NCT;
      const float* src = which ? xg : a1g;
      f32x16 a;
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = 0.f;
      float bq[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int row = r_base + 2 * s + lh, f = ct * 32 + li;
        const float t = src[(size_t)(row < rows ? row : rows - 1) * Fi + (f < Fi ? f : Fi - 1)];
        bq[s] = (row < rows && f < Fi) ? t : 0.f;
      }
      const float* ap = sG + (r_base + lh) * HS + ht * 32 + li;
#pragma unroll
      for (int s = 0; s < 16; ++s)
        a = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s * HS], bq[s], a, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) sR[wave * 1024 + acc_row(r, lh) * 32 + li] = a[r];
      __syncthreads();
      float* dst = slab + (which ? (size_t)Fo * Fi : 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, hh = ht * 32 + (e >> 5), ff = ct * 32 + (e & 31);
        if (hh < Fo && ff < Fi)
          dst[(size_t)hh * Fi + ff] = (sR[e] + sR[1024 + e]) + (sR[2048 + e] + sR[3072 + e]);
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int c = 0; c < NCT; ++c) {
    f32x16 d, g;
#pragma unroll
    for (int r = 0; r < 16; ++r) { d[r] = 0.f; g[r] = 0.f; }
    mma32(d, sG + r_base * HS, HS, 1, sW + c * 32, FS, 1, HP, li, lh);
    mma32(g, sG + r_base * HS, HS, 1, sW + HP * FS + c * 32, FS, 1, HP, li, lh);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = r_base + acc_row(r, lh), col = c * 32 + li;
      if (row < rows && col < Fi) {
        ws_dagg[(size_t)(r0 + row) * Fi + col] = d[r];
        if (g_x) g_x[(size_t)(r0 + row) * Fi + col] = g[r];
      }
    }
  }
}

// CSR GraphConv backward, third generation (Fi = Fo = 32, no k-hop mask, no edge-weight gradient):
// the row-local adjoint AND the transpose aggregation in one pass, shaped like k_csr_fwd3 (one wave
// per 32 rows, persistent, 16-byte accesses, no workgroup barrier in the loop).
//   G   = g_out * act'(out)
//   g_x[j] = G[j] W_root + (sum_{k in CSC column j} w_k G[sink_k]) W_rel        (NEED_X)
//   dW_rel += G^T agg,  dW_root += G^T x,  db += colsum G                       (accumulated in the
//   wave's registers over all its tiles; one slab per workgroup at the end, fixed order)
// The second generation wrote G W_rel for every row (33.5 MB at cfg4), read it back through the CSC
// gather and read-modify-wrote g_x: 325 MB per layer in two kernels; this one moves the four inputs and
// g_x once (168 MB) - the gathered rows are the neighbours' g_out / out rows, which mostly hit in cache.
template <bool HAS_W, bool NEED_X>
__global__ __launch_bounds__(256) void k_csr_bwd3(
    const float* __restrict__ g_out, const float* __restrict__ out, const float* __restrict__ x,
    const float* __restrict__ agg, const int64_t* __restrict__ col_ptr,
    const int64_t* __restrict__ rows_csc, const int64_t* __restrict__ perm,
    const float* __restrict__ w, const float* __restrict__ w_rel, const float* __restrict__ w_root,
    float* __restrict__ g_x, float* __restrict__ slabs, int64_t M, int act, int n_tiles) {
  constexpr int FI = 32, FO = 32, AS = 36, WS = 20;
  constexpr int NP = 4, RPP = 8;      // lane (lr = lane / 8, c4 = 4 (lane % 8)) owns rows lr + 8 i
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int act_v = gcm_vgpr(act);
  extern __shared__ float smem[];
  float* sG = smem + (size_t)wave * 3 * 32 * AS;   // G tile | agg tile (then the gathered G) | x tile
  float* sA = sG + 32 * AS;
  float* sX = sA + 32 * AS;
  // weights for g_x = [GT | G] @ [W_rel | W_root] (K = o): sW[(m, kk, f)][16 o (+4)] = W_m[kk 16 + s][f]
  float* sW = smem + (size_t)4 * 3 * 32 * AS;
  for (int e = tid; e < 2 * FO * FI; e += 256) {
    const int m = e / (FO * FI), rem = e - m * FO * FI, o = rem / FI, f = rem % FI;
    sW[(size_t)((m * 2 + o / 16) * 32 + f) * WS + (o & 15)] = (m ? w_root : w_rel)[(size_t)o * FI + f];
  }
  __syncthreads();
  const int lr = lane >> 3, c4 = (lane & 7) * 4;
  const int n_waves = gridDim.x * 4;
  f32x16 dwr, dwo;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dwr[r] = 0.f; dwo[r] = 0.f; }
  float4 db4 = make_float4(0.f, 0.f, 0.f, 0.f);   // column sums of G over this lane's rows
  auto act_grad4 = [&](const float4& g, const float4& y) {
    return make_float4(g.x * gcm_act_grad_sel(y.x, act_v), g.y * gcm_act_grad_sel(y.y, act_v),
                       g.z * gcm_act_grad_sel(y.z, act_v), g.w * gcm_act_grad_sel(y.w, act_v));
  };
#pragma unroll 1
  for (int tile = blockIdx.x * 4 + wave; tile < n_tiles; tile += n_waves) {
    const int64_t r0 = (int64_t)tile * 32;
    float4 gv[NP], yv[NP], xv[NP], av[NP];
    int64_t p0[NP];
    int deg[NP], dmax = 0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int64_t r = r0 + i * RPP + lr;
      const int64_t rc = r < M ? r : M - 1;
      gv[i] = *reinterpret_cast<const float4*>(g_out + (size_t)rc * FO + c4);
      yv[i] = *reinterpret_cast<const float4*>(out + (size_t)rc * FO + c4);
      xv[i] = *reinterpret_cast<const float4*>(x + (size_t)rc * FI + c4);
      av[i] = *reinterpret_cast<const float4*>(agg + (size_t)rc * FI + c4);
      if (NEED_X) {
        p0[i] = 0;
        deg[i] = 0;
        if (col_ptr) {   // (uniform; null: no edges at all)
          p0[i] = col_ptr[rc];
          deg[i] = r < M ? (int)(col_ptr[rc + 1] - p0[i]) : 0;
        }
        dmax = max(dmax, deg[i]);
      }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int rl = i * RPP + lr;
      const bool live = r0 + rl < M;
      float4 G = act_grad4(gv[i], yv[i]);
      if (!live) G = make_float4(0.f, 0.f, 0.f, 0.f);
      db4 = make_float4(db4.x + G.x, db4.y + G.y, db4.z + G.z, db4.w + G.w);
      *reinterpret_cast<float4*>(sG + rl * AS + c4) = G;
      *reinterpret_cast<float4*>(sA + rl * AS + c4) = live ? av[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(sX + rl * AS + c4) = live ? xv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // transpose aggregation of G over the CSC column of each row (the sinks this node feeds)
    float4 gt[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) gt[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (NEED_X) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) dmax = max(dmax, __shfl_xor(dmax, o));
#pragma unroll 1
      for (int d = 0; d < dmax; ++d) {
        int64_t sk[NP];
        float we[NP];
        bool on[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          on[i] = d < deg[i];
          const int64_t k = on[i] ? p0[i] + d : 0;   // (dmax > 0 implies E > 0)
          sk[i] = rows_csc[k];
          we[i] = HAS_W ? w[perm[k]] : 1.f;
        }
        asm volatile("" ::: "memory");
        float4 ng[NP], ny[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          ng[i] = *reinterpret_cast<const float4*>(g_out + (size_t)sk[i] * FO + c4);
          ny[i] = *reinterpret_cast<const float4*>(out + (size_t)sk[i] * FO + c4);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          float4 t = act_grad4(ng[i], ny[i]);
          const float ww = on[i] ? we[i] : 0.f;
          gt[i] = make_float4(fmaf(ww, t.x, gt[i].x), fmaf(ww, t.y, gt[i].y), fmaf(ww, t.z, gt[i].z),
                              fmaf(ww, t.w, gt[i].w));
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // dW_rel += G^T agg, dW_root += G^T x: M = o, N = f, K = the 32 rows (lane half kk owns rows kk 16 + s)
    {
      float ga[16], ba[16], bx[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = lh * 16 + q;
        ga[q] = sG[row * AS + li];
        ba[q] = sA[row * AS + li];
        bx[q] = sX[row * AS + li];
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        dwr = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[q], ba[q], dwr, 0, 0, 0);
        dwo = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[q], bx[q], dwo, 0, 0, 0);
      }
    }
    if (NEED_X) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < NP; ++i)   // the agg tile has been read: it now holds the gathered G
        *reinterpret_cast<float4*>(sA + (i * RPP + lr) * AS + c4) = gt[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // g_x = GT @ W_rel + G @ W_root: lane (row = li, kk) reads its 16 contiguous o
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* pt = sA + li * AS + lh * 16;
      const float* pg = sG + li * AS + lh * 16;
      const float* wr_ = sW + (size_t)((0 * 2 + lh) * 32 + li) * WS;
      const float* wo_ = sW + (size_t)((1 * 2 + lh) * 32 + li) * WS;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(pt + 4 * q);
        const float4 wv = *reinterpret_cast<const float4*>(wr_ + 4 * q);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wv.w, acc, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 a = *reinterpret_cast<const float4*>(pg + 4 * q);
        const float4 wv = *reinterpret_cast<const float4*>(wo_ + 4 * q);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wv.w, acc, 0, 0, 0);
      }
      float* ob = g_x + (size_t)(r0 + 4 * lh) * FI + li;
      if (r0 + 32 <= M) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[((r & 3) + 8 * (r >> 2)) * FI] = acc[r];
      } else {
        const int left = (int)(M - r0) - 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if ((r & 3) + 8 * (r >> 2) < left) ob[((r & 3) + 8 * (r >> 2)) * FI] = acc[r];
      }
    }
    __builtin_amdgcn_wave_barrier();   // the tiles are rewritten by the next trip
  }
  // ---- one slab per workgroup: { G^T agg [FO][FI] | G^T x [FO][FI] | colsum G [FO] }, the four waves
  // summed in fixed order.  acc[r] of dwr / dwo is entry (o = acc_row(r, lh), f = li).
  db4.x += __shfl_xor(db4.x, 8); db4.y += __shfl_xor(db4.y, 8); db4.z += __shfl_xor(db4.z, 8); db4.w += __shfl_xor(db4.w, 8);
  db4.x += __shfl_xor(db4.x, 16); db4.y += __shfl_xor(db4.y, 16); db4.z += __shfl_xor(db4.z, 16); db4.w += __shfl_xor(db4.w, 16);
  db4.x += __shfl_xor(db4.x, 32); db4.y += __shfl_xor(db4.y, 32); db4.z += __shfl_xor(db4.z, 32); db4.w += __shfl_xor(db4.w, 32);
  __syncthreads();                       // every wave is done with its tiles: LDS becomes the reduction buffer
  constexpr int SL = 2 * FO * FI + FO;
  float* red = smem + (size_t)wave * (SL + 32);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = acc_row(r, lh);
    red[o * FI + li] = dwr[r];
    red[FO * FI + o * FI + li] = dwo[r];
  }
  if (lane < 8) *reinterpret_cast<float4*>(red + 2 * FO * FI + 4 * lane) = db4;
  __syncthreads();
  float* slab = slabs + (size_t)blockIdx.x * SL;
  for (int e = tid; e < SL; e += 256)
    slab[e] = (smem[e] + smem[(SL + 32) + e]) + (smem[2 * (SL + 32) + e] + smem[3 * (SL + 32) + e]);
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" int gcm_dense_graphconv_fwd(const float* x, const float* adj, const float* w_rel,
                                       const float* b_rel, const float* w_root, float* out,
                                       float* agg, int B, int N, int Fi, int Fo, int act,
                                       gcm_stream_t stream) {
  GCM_REQUIRE(x && adj && w_rel && w_root && out);
  GCM_REQUIRE(B > 0 && N > 0 && Fi > 0 && Fo > 0);
  if (Fi > 128 || Fo > 128 || B > 65535) return GCM_EUNSUPPORTED;
  if (gcm_layer_fits(N, Fi, Fo, 0))
    return gcm_layer_fwd(x, adj, w_rel, b_rel, w_root, out, agg, B, N, Fi, Fo, act,
                         (hipStream_t)stream);
  const int FiP = round32(Fi);
  const int waves = pick_waves(N, FiP, fwd_lds_bytes);
  if (!waves) return GCM_EUNSUPPORTED;
  const int RB = 32 * waves;
  dim3 grid((N + RB - 1) / RB, B);
  const size_t lds = fwd_lds_bytes(waves, FiP);
  hipStream_t s = (hipStream_t)stream;
  switch (waves) {
    case 4: return launch_fwd<4>(FiP / 32, grid, lds, s, x, adj, w_rel, b_rel, w_root, out, agg, N, Fi, Fo, act);
    case 2: return launch_fwd<2>(FiP / 32, grid, lds, s, x, adj, w_rel, b_rel, w_root, out, agg, N, Fi, Fo, act);
    default: return launch_fwd<1>(FiP / 32, grid, lds, s, x, adj, w_rel, b_rel, w_root, out, agg, N, Fi, Fo, act);
  }
}

namespace {
struct BwdPlan {
  int waves, RB, nblk;
  size_t dagg_bytes, slab_len, slabs_bytes;
};
BwdPlan bwd_plan(int B, int N, int Fi, int Fo) {
  BwdPlan p;
  const int FiP = round32(Fi), FoP = round32(Fo);
  const int want = N > 64 ? 4 : (N > 32 ? 2 : 1);
  p.waves = 0;
  for (int w = want; w >= 1; w >>= 1)
    if (bwd_rows_lds_bytes(w, FiP, FoP) <= 160 * 1024) { p.waves = w; break; }
  p.RB = 32 * (p.waves ? p.waves : 1);
  p.nblk = (N + p.RB - 1) / p.RB;
  p.dagg_bytes = (((size_t)B * N * Fi * sizeof(float)) + 255) & ~(size_t)255;
  p.slab_len = 2 * (size_t)Fo * Fi + Fo;
  p.slabs_bytes = (size_t)B * p.nblk * p.slab_len * sizeof(float);
  return p;
}
}  // namespace

extern "C" size_t gcm_dense_graphconv_bwd_workspace_bytes(int B, int N, int Fi, int Fo) {
  if (B <= 0 || N <= 0 || Fi <= 0 || Fo <= 0) return 0;
  BwdPlan p = bwd_plan(B, N, Fi, Fo);
  return p.dagg_bytes + p.slabs_bytes;
}

extern "C" int gcm_dense_graphconv_bwd(const float* g_out, const float* out, const float* x,
                                       const float* adj, const float* agg, const float* w_rel,
                                       const float* w_root, float* g_x, float* g_adj,
                                       float* g_w_rel, float* g_b_rel, float* g_w_root,
                                       void* workspace, size_t workspace_bytes, int B, int N,
                                       int Fi, int Fo, int act, gcm_stream_t stream) {
  GCM_REQUIRE(g_out && out && x && adj && agg && w_rel && w_root && workspace);
  GCM_REQUIRE(B > 0 && N > 0 && Fi > 0 && Fo > 0);
  if (Fi > 128 || Fo > 128 || B > 65535) return GCM_EUNSUPPORTED;
  BwdPlan p = bwd_plan(B, N, Fi, Fo);
  if (!p.waves) return GCM_EUNSUPPORTED;
  if (workspace_bytes < p.dagg_bytes + p.slabs_bytes) return GCM_EWORKSPACE;
  float* ws_dagg = (float*)workspace;
  float* slabs = (float*)((char*)workspace + p.dagg_bytes);
  const int FiP = round32(Fi), FoP = round32(Fo), nct = FiP / 32;
  const int want_w = (g_w_rel || g_w_root || g_b_rel) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  if (gcm_layer_fits(N, Fi, Fo, g_adj ? 2 : 1)) {   // one kernel per layer + the slab reduction
    int rc = gcm_layer_bwd(g_out, out, x, adj, agg, w_rel, w_root, g_x, g_adj, slabs, want_w, B, N,
                           Fi, Fo, act, s);
    if (rc != GCM_OK || !want_w) return rc;
    hipLaunchKernelGGL(k_reduce_slabs, dim3(((int)p.slab_len + 15) / 16), dim3(256), 0, s, slabs, B,
                       (int)p.slab_len, g_w_rel, g_w_root, g_b_rel, Fo * Fi, Fo);
    return gcm_launch_status();
  }
  dim3 grid(p.nblk, B);
  const size_t lds1 = bwd_rows_lds_bytes(p.waves, FiP, FoP);
  const size_t lds2 = bwd_adjT_lds_bytes(p.waves, FiP);

#define GCM_BWD_LAUNCH(W, C)                                                                      \
  {                                                                                               \
    auto k1 = k_graphconv_bwd_rows<W, C>;                                                         \
    gcm_allow_dynamic_lds((const void*)k1, lds1); \
    hipLaunchKernelGGL(k1, grid, dim3(64 * W), lds1, s, g_out, out, x, agg, w_rel, w_root, g_x,   \
                       g_adj, ws_dagg, slabs, N, Fi, Fo, act, want_w);                            \
    if (g_x) {                                                                                    \
      auto k2 = k_graphconv_bwd_adjT<W, C>;                                                       \
      hipLaunchKernelGGL(k2, grid, dim3(64 * W), lds2, s, adj, ws_dagg, g_x, N, Fi);              \
    }                                                                                             \
  }
#define GCM_BWD_W(W)                                  \
  switch (nct) {                                      \
    case 1: GCM_BWD_LAUNCH(W, 1) break;               \
    case 2: GCM_BWD_LAUNCH(W, 2) break;               \
    case 3: GCM_BWD_LAUNCH(W, 3) break;               \
    default: GCM_BWD_LAUNCH(W, 4) break;              \
  }
  switch (p.waves) {
    case 4: GCM_BWD_W(4) break;
    case 2: GCM_BWD_W(2) break;
    default: GCM_BWD_W(1) break;
  }
#undef GCM_BWD_W
#undef GCM_BWD_LAUNCH
  int rc = gcm_launch_status();
  if (rc != GCM_OK) return rc;
  if (want_w) {
    const int slab_len = (int)p.slab_len;
    hipLaunchKernelGGL(k_reduce_slabs, dim3((slab_len + 15) / 16), dim3(256), 0, s, slabs,
                       B * p.nblk, slab_len, g_w_rel, g_w_root, g_b_rel, Fo * Fi, Fo);
    rc = gcm_launch_status();
  }
  return rc;
}

// ---------------------------------------------------------------------------
// C ABI: GraphConv over CSR
// ---------------------------------------------------------------------------
static int csr_fwd_impl(const float* x, const int64_t* row_ptr, const int64_t* col, const float* w,
                        const uint8_t* mask, const float* w_rel, const float* b_rel, const float* w_root,
                        float* out, float* agg, int64_t M, int Fi, int Fo, int act, uint32_t* flags,
                        gcm_stream_t stream);

extern "C" int gcm_csr_graphconv_fwd(const float* x, const int64_t* row_ptr, const int64_t* col,
                                     const float* w, const uint8_t* mask, const float* w_rel,
                                     const float* b_rel, const float* w_root, float* out,
                                     float* agg, int64_t M, int Fi, int Fo, int act,
                                     gcm_stream_t stream) {
  return csr_fwd_impl(x, row_ptr, col, w, mask, w_rel, b_rel, w_root, out, agg, M, Fi, Fo, act, nullptr, stream);
}

extern "C" int gcm_csr_graphconv_fwd_checked_supported(int64_t M, int Fi, int Fo) {
  return (Fi == 32 && (Fo == 32 || Fo == 64) && M >= 32 && M <= (int64_t)2147483647 - 256) ? 1 : 0;
}

extern "C" int gcm_csr_graphconv_fwd_checked(const float* x, const int64_t* row_ptr, const int64_t* col,
                                             const float* w, const uint8_t* mask, const float* w_rel,
                                             const float* b_rel, const float* w_root, float* out, float* agg,
                                             int64_t M, int Fi, int Fo, int act, uint32_t* flags,
                                             gcm_stream_t stream) {
  GCM_REQUIRE(flags);
  if (!gcm_csr_graphconv_fwd_checked_supported(M, Fi, Fo)) return GCM_EUNSUPPORTED;
  return csr_fwd_impl(x, row_ptr, col, w, mask, w_rel, b_rel, w_root, out, agg, M, Fi, Fo, act, flags, stream);
}

static int csr_fwd_impl(const float* x, const int64_t* row_ptr, const int64_t* col, const float* w,
                        const uint8_t* mask, const float* w_rel, const float* b_rel, const float* w_root,
                        float* out, float* agg, int64_t M, int Fi, int Fo, int act, uint32_t* flags,
                        gcm_stream_t stream) {
  GCM_REQUIRE(row_ptr && w_rel && w_root && M >= 0 && Fi > 0 && Fo > 0);
  if (M == 0) return GCM_OK;
  GCM_REQUIRE(x && out);
  if (Fi > 128 || Fo > 128 || M > (int64_t)2147483647 - 256) return GCM_EUNSUPPORTED;
  const int FiP = round32(Fi);
  dim3 grid((unsigned)((M + 127) / 128));
  hipStream_t s = (hipStream_t)stream;
  if (Fi == 32 && (Fo == 32 || Fo == 64) && M >= 32) {   // third generation: exact tiles
    const int n_tiles = (int)((M + 31) / 32);
    const size_t lds3 = sizeof(float) * (4 * 2 * 32 * ((size_t)Fi + 4) + 2 * (size_t)Fo * 2 * (Fi / 2 + 4));
#define GCM_CSR3(a, b_, hw, hm)                                                                 \
  if (Fi == a && Fo == b_ && (w != nullptr) == hw && (mask != nullptr) == hm) {                 \
    auto kern = k_csr_fwd3<a, b_, hw, hm>;                                                            \
    gcm_allow_dynamic_lds((const void*)kern, lds3);                                             \
    /* resident workgroups per CU (persistent waves): 3 waves per SIMD by registers, 160 KB of LDS */ \
    const int by_lds = (int)((160 * 1024) / lds3);                                              \
    const int per_cu = by_lds < 1 ? 1 : (by_lds > 3 ? 3 : by_lds);                              \
    const int cap = per_cu * gcm_cu_count();                                                    \
    const int blocks = (n_tiles + 3) / 4 < cap ? (n_tiles + 3) / 4 : cap;                       \
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds3, s, x, row_ptr, col, w, mask, w_rel, \
                       b_rel, w_root, out, agg, M, flags, act, n_tiles);                        \
    return gcm_launch_status();                                                                 \
  }
    if (flags && !w && !mask) {   // the checked form (gcm_csr_graphconv_fwd_checked)
#define GCM_CSR3C(a, b_)                                                                        \
  if (Fi == a && Fo == b_) {                                                                    \
    auto kern = k_csr_fwd3<a, b_, false, false, true>;                                          \
    gcm_allow_dynamic_lds((const void*)kern, lds3);                                             \
    const int by_lds = (int)((160 * 1024) / lds3);                                              \
    const int per_cu = by_lds < 1 ? 1 : (by_lds > 3 ? 3 : by_lds);                              \
    const int cap = per_cu * gcm_cu_count();                                                    \
    const int blocks = (n_tiles + 3) / 4 < cap ? (n_tiles + 3) / 4 : cap;                       \
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds3, s, x, row_ptr, col, w, mask, w_rel, \
                       b_rel, w_root, out, agg, M, flags, act, n_tiles);                        \
    return gcm_launch_status();                                                                 \
  }
      GCM_CSR3C(32, 32) GCM_CSR3C(32, 64)
#undef GCM_CSR3C
    }
    if (flags) return GCM_EUNSUPPORTED;   // (edge weights / k-hop mask: no checked form)
    GCM_CSR3(32, 32, false, false) GCM_CSR3(32, 32, true, false) GCM_CSR3(32, 32, false, true)
    GCM_CSR3(32, 32, true, true) GCM_CSR3(32, 64, false, false) GCM_CSR3(32, 64, true, false)
    GCM_CSR3(32, 64, false, true) GCM_CSR3(32, 64, true, true)
#undef GCM_CSR3
  }
  if (Fi <= 64 && Fo <= 64) {   // second-generation kernel
    const int NCT = FiP / 32, NHT = round32(Fo) / 32;
    const size_t lds2 = sizeof(float) * ((size_t)128 * (FiP + 1) + 2 * FiP * (32 * NHT + 1) + 1) + 8 +
                        sizeof(int) * (128 + 2 + 1024) + sizeof(float) * 1024;
#define GCM_CSR2(a, b_)                                                                         \
  if (NCT == a && NHT == b_) {                                                                  \
    auto kern = k_csr_fwd2<a, b_>;                                                              \
    gcm_allow_dynamic_lds((const void*)kern, lds2);                                                                                           \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds2, s, x, row_ptr, col, w, mask, w_rel, b_rel,  \
                       w_root, out, agg, M, Fi, Fo, act);                                       \
    return gcm_launch_status();                                                                 \
  }
    GCM_CSR2(1, 1) GCM_CSR2(1, 2) GCM_CSR2(2, 1) GCM_CSR2(2, 2)
#undef GCM_CSR2
  }
  const size_t lds = csr_fwd_lds_bytes(FiP);
#define GCM_CSR_FWD(NCT)                                                                        \
  {                                                                                             \
    auto kern = k_csr_graphconv_fwd<NCT>;                                                       \
    gcm_allow_dynamic_lds((const void*)kern, lds); \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, x, row_ptr, col, w, mask, w_rel, b_rel,   \
                       w_root, out, agg, M, Fi, Fo, act);                                       \
  }
  switch (FiP / 32) {
    case 1: GCM_CSR_FWD(1) break;
    case 2: GCM_CSR_FWD(2) break;
    case 3: GCM_CSR_FWD(3) break;
    default: GCM_CSR_FWD(4) break;
  }
#undef GCM_CSR_FWD
  return gcm_launch_status();
}

extern "C" size_t gcm_csr_graphconv_bwd_workspace_bytes(int64_t M, int Fi, int Fo) {
  if (M <= 0 || Fi <= 0 || Fo <= 0 || M > (int64_t)2147483647 - 256) return 0;
  BwdPlan p = bwd_plan(1, (int)M, Fi, Fo);
  return p.dagg_bytes + p.slabs_bytes;
}

extern "C" int gcm_csr_graphconv_bwd(const float* g_out, const float* out, const float* x,
                                     const float* agg, const int64_t* row_ptr, const int64_t* col,
                                     const int64_t* col_ptr, const int64_t* rows,
                                     const int64_t* perm, const float* w, const uint8_t* mask,
                                     const float* w_rel, const float* w_root, float* g_x,
                                     float* g_w, float* g_w_rel, float* g_b_rel, float* g_w_root,
                                     void* workspace, size_t workspace_bytes, int64_t M, int64_t E,
                                     int Fi, int Fo, int act, gcm_stream_t stream) {
  (void)row_ptr;
  (void)col;
  GCM_REQUIRE(g_out && out && x && agg && w_rel && w_root && workspace);
  GCM_REQUIRE(M > 0 && E >= 0 && Fi > 0 && Fo > 0);
  GCM_REQUIRE((col_ptr && rows) || E == 0 || (!g_x && !g_w));
  GCM_REQUIRE(!(w && (g_x || g_w) && E > 0) || perm);   // perm maps CSC entries to the CSR weights
  if (Fi > 128 || Fo > 128 || M > (int64_t)2147483647 - 256) return GCM_EUNSUPPORTED;
  BwdPlan p = bwd_plan(1, (int)M, Fi, Fo);
  if (!p.waves) return GCM_EUNSUPPORTED;
  if (workspace_bytes < p.dagg_bytes + p.slabs_bytes) return GCM_EWORKSPACE;
  float* ws_dagg = (float*)workspace;
  float* slabs = (float*)((char*)workspace + p.dagg_bytes);
  const int FiP = round32(Fi), FoP = round32(Fo), nct = FiP / 32;
  const int want_w = (g_w_rel || g_w_root || g_b_rel) ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(p.nblk, 1);
  const size_t lds1 = bwd_rows_lds_bytes(p.waves, FiP, FoP);
  const int N = (int)M;
  float* no_adj = nullptr;
  bool rows_done = false;
  if (Fi == 32 && Fo == 32 && !mask && !g_w && M >= 32 && (!g_x || E == 0 || (col_ptr && rows && perm))) {
    // third generation: row-local adjoint and transpose aggregation in one pass
    const int n_tiles = (int)((M + 31) / 32);
    const size_t lds3 = sizeof(float) * (4 * 3 * 32 * 36 + 2 * 2 * 32 * 20);
    const int cap = 2 * gcm_cu_count();   // two resident workgroups per CU (LDS)
    const int blocks = (n_tiles + 3) / 4 < cap ? (n_tiles + 3) / 4 : cap;
    const bool need_x = g_x != nullptr, gather = need_x && E > 0;
    // (no edges: the CSC view may be absent - the gather loop never runs, col_ptr is not read)
#define GCM_BWD3(hw, nx)                                                                        \
  if ((w != nullptr) == hw && need_x == nx) {                                                   \
    auto kern = k_csr_bwd3<hw, nx>;                                                             \
    gcm_allow_dynamic_lds((const void*)kern, lds3);                                             \
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds3, s, g_out, out, x, agg,              \
                       gather ? col_ptr : nullptr, rows, perm, w, w_rel, w_root, g_x, slabs, M, act, \
                       n_tiles);                                                                \
  }
    GCM_BWD3(false, false) GCM_BWD3(true, false) GCM_BWD3(false, true) GCM_BWD3(true, true)
#undef GCM_BWD3
    int rc3 = gcm_launch_status();
    if (rc3 != GCM_OK) return rc3;
    if (want_w) {
      const int slab_len = (int)p.slab_len;
      hipLaunchKernelGGL(k_reduce_slabs, dim3((slab_len + 15) / 16), dim3(256), 0, s, slabs, blocks,
                         slab_len, g_w_rel, g_w_root, g_b_rel, Fo * Fi, Fo);
    }
    return gcm_launch_status();
  }
  if (Fi <= 64 && Fo <= 64 && p.waves == 4) {   // second-generation row-local kernel
    const int NCT = FiP / 32, NHT = FoP / 32;
    const size_t lds2 = sizeof(float) * ((size_t)128 * (FoP + 1) + 2 * FoP * (FiP + 1) + 4 * 1024 + 256);
#define GCM_ROWS2(a, b_)                                                                        \
  if (NCT == a && NHT == b_) {                                                                  \
    auto kern = k_rows_bwd2<a, b_>;                                                             \
    gcm_allow_dynamic_lds((const void*)kern, lds2);                                                                                           \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds2, s, g_out, out, x, agg, w_rel, w_root, g_x,  \
                       ws_dagg, slabs, M, Fi, Fo, act, want_w);                                 \
    rows_done = true;                                                                           \
  }
    GCM_ROWS2(1, 1) GCM_ROWS2(1, 2) GCM_ROWS2(2, 1) GCM_ROWS2(2, 2)
#undef GCM_ROWS2
  }
  if (!rows_done) {
#define GCM_CSR_BWD(W, C)                                                                       \
  {                                                                                             \
    auto k1 = k_graphconv_bwd_rows<W, C>;                                                       \
    gcm_allow_dynamic_lds((const void*)k1, lds1); \
    hipLaunchKernelGGL(k1, grid, dim3(64 * W), lds1, s, g_out, out, x, agg, w_rel, w_root, g_x, \
                       no_adj, ws_dagg, slabs, N, Fi, Fo, act, want_w);                         \
  }
#define GCM_CSR_BWD_W(W)                           \
  switch (nct) {                                   \
    case 1: GCM_CSR_BWD(W, 1) break;               \
    case 2: GCM_CSR_BWD(W, 2) break;               \
    case 3: GCM_CSR_BWD(W, 3) break;               \
    default: GCM_CSR_BWD(W, 4) break;              \
  }
  switch (p.waves) {
    case 4: GCM_CSR_BWD_W(4) break;
    case 2: GCM_CSR_BWD_W(2) break;
    default: GCM_CSR_BWD_W(1) break;
  }
#undef GCM_CSR_BWD_W
#undef GCM_CSR_BWD
  }
  int rc = gcm_launch_status();
  if (rc != GCM_OK) return rc;
  if (g_x && E > 0) {
    const int64_t total = M * Fi;
    hipLaunchKernelGGL(k_csr_scatter_T, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       ws_dagg, col_ptr, rows, perm, w, mask, g_x, M, Fi);
  }
  if (g_w && E > 0) {
    hipLaunchKernelGGL(k_csr_edge_grad, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, x,
                       ws_dagg, col_ptr, rows, perm, mask, g_w, M, Fi);
  }
  if (want_w) {
    const int slab_len = (int)p.slab_len;
    hipLaunchKernelGGL(k_reduce_slabs, dim3((slab_len + 15) / 16), dim3(256), 0, s, slabs, p.nblk,
                       slab_len, g_w_rel, g_w_root, g_b_rel, Fo * Fi, Fo);
  }
  return gcm_launch_status();
}
