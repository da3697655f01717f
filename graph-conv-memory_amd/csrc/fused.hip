// Fused two-layer DenseGraphConv step for the canonical GCM GNN (README.md:52-62):
//
//     h1 = act1( (adj @ x) W_rel1^T + b1 + x W_root1^T )                      [N, H1]
//     mx = act2( (adj[cur,:] @ h1) W_rel2^T + b2 + h1[cur] W_root2^T )        [H2]   (gcm.py:314)
//
// One workgroup (4 waves) owns one graph: the whole adjacency (<= 128x128 fp32 = 64 KB), x and
// h1 live in LDS, adj is read from HBM exactly once for both layers, and the second layer is
// evaluated only on the row DenseGCM keeps (node_feats[b, num_nodes[b]]).  Each wave owns a
// 32-row strip: its adj rows are wave-private in LDS, so the aggregation needs no workgroup
// barrier; loads are all issued up front and consumed K-tile by K-tile, so the first MFMAs
// run while later tiles are still in flight.
//
// LDS images use row strides that are odd in dwords => the per-lane fragment reads
// (ds_read_b32, lane = matrix row or column) are bank-conflict free.
#include "gcm_common.h"

namespace {

__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// acc(32x32) += A(32xK) * B(Kx32), operands in LDS: A(i,k)=a[i*ais+k*aks], B(k,j)=b[k*bks+j*bjs]
__device__ __forceinline__ void mma32(f32x16& acc, const float* a, int ais, int aks,
                                      const float* b, int bks, int bjs, int K, int li, int lh) {
  const float* ap = a + li * ais + lh * aks;
  const float* bp = b + lh * bks + li * bjs;
#pragma unroll 8
  for (int k = 0; k < K; k += 2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k * aks], bp[k * bks], acc, 0, 0, 0);
}

struct Gnn2 {
  const float *w_rel1, *b_rel1, *w_root1;  // [H1,F], [H1], [H1,F]
  const float *w_rel2, *b_rel2, *w_root2;  // [H2,H1], [H2], [H2,H1]
  int act1, act2;
};

// LDS carve-up shared by forward and backward (floats)
template <int NT, int NCT, int NHT>
struct Lds {
  static constexpr int NP = 32 * NT, FS = 32 * NCT + 1, HS = 32 * NHT + 1;
  static constexpr int AS = FS > HS ? FS : HS;   // common stride of the agg / h1 image
  static constexpr int ADJ = NT * NP * 33;  // [col tile][row][33]
  static constexpr int X = NP * FS;         // [row][FS]
  static constexpr int AH = NP * AS;
  static constexpr int SV = 320;            // layer-2 vectors: 256 partials + 64
};

// adj[r][c] of the LDS image
template <int NP>
__device__ __forceinline__ int adj_at(int r, int c) {
  return ((c >> 5) * NP + r) * 33 + (c & 31);
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
template <int NT, int NCT, int NHT>
__global__ __launch_bounds__(256) void k_gnn2_row_fwd(
    const float* __restrict__ x, const float* __restrict__ adj, const int64_t* __restrict__ cur_idx,
    Gnn2 P, float* __restrict__ mx_out, float* __restrict__ h1_out, float* __restrict__ agg1_out,
    float* __restrict__ agg2_out, uint32_t* __restrict__ flags, int N, int F, int H1, int H2) {
  using L = Lds<NT, NCT, NHT>;
  constexpr int NP = L::NP, FS = L::FS, HS = L::HS, AS = L::AS, HP = 32 * NHT;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const float* xg = x + (size_t)b * N * F;
  const float* ag = adj + (size_t)b * N * N;
  int64_t cur64 = cur_idx[b];
  const int cur = cur64 < 0 ? 0 : (cur64 > N - 1 ? N - 1 : (int)cur64);

  extern __shared__ float smem[];
  float* sAdj = smem;
  float* sX = sAdj + L::ADJ;
  float* sAH = sX + L::X;              // agg, later h1 (both with stride AS)
  float* sW = sAH + L::AH;             // w_rel1^T [F][HS] then w_root1^T [F][HS]   (B(k=f, j=h))
  float* sV = sW + 2 * (32 * NCT) * HS;  // small vectors for layer 2: agg2[H1], pre2 scratch

  // ---- stage x and the layer-1 weights (shared by all waves) -----------------
  for (int e = tid; e < NP * 32 * NCT; e += 256) {
    const int r = e / (32 * NCT), c = e - r * (32 * NCT);
    sX[r * FS + c] = (r < N && c < F) ? xg[(size_t)r * F + c] : 0.f;
  }
  for (int e = tid; e < 32 * NHT * 32 * NCT; e += 256) {
    const int h = e / (32 * NCT), f = e - h * (32 * NCT);
    const bool ok = h < H1 && f < F;
    sW[f * HS + h] = ok ? P.w_rel1[(size_t)h * F + f] : 0.f;
    sW[(32 * NCT + f) * HS + h] = ok ? P.w_root1[(size_t)h * F + f] : 0.f;
  }
  // ---- this wave's 32 adjacency rows: issue every load first (wave-private region) ----
  const int r_base = wave * 32;
  const bool wave_live = wave < NT;
  float4 buf[NT * 4];
  const bool vec = (N & 3) == 0;
  if (wave_live) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = r_base + (lane >> 3) + 8 * q, c = t * 32 + (lane & 7) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < N) {
          const float* p = ag + (size_t)r * N + c;
          if (vec) {
            if (c < N) v = *reinterpret_cast<const float4*>(p);
          } else {
            if (c < N) v.x = p[0];
            if (c + 1 < N) v.y = p[1];
            if (c + 2 < N) v.z = p[2];
            if (c + 3 < N) v.w = p[3];
          }
        }
        buf[t * 4 + q] = v;
      }
  }
  __syncthreads();  // x and weights are in LDS (the adjacency loads are still in flight)

  // ---- layer 1, aggregation: agg = adj[rows,:] @ x, K tile by K tile ----------------
  f32x16 acc[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  if (wave_live) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = r_base + (lane >> 3) + 8 * q;
        float* d = sAdj + (t * NP + r) * 33 + (lane & 7) * 4;
        const float4 v = buf[t * 4 + q];
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < NCT; ++c)
        mma32(acc[c], sAdj + (t * NP + r_base) * 33, 33, 1, sX + (t * 32) * FS + c * 32, FS, 1, 32,
              li, lh);
    }
    // agg -> LDS (A operand of the linears) and -> HBM (saved for backward)
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r_base + acc_row(r, lh), col = c * 32 + li;
        sAH[row * AS + col] = acc[c][r];
        if (agg1_out && row < N && col < F) agg1_out[((size_t)b * N + row) * F + col] = acc[c][r];
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- layer 1, linears: h1 = act1(agg W_rel1^T + x W_root1^T + b1) ------------------
    f32x16 o[NHT];
#pragma unroll
    for (int t = 0; t < NHT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
      mma32(o[t], sAH + r_base * AS, AS, 1, sW + t * 32, HS, 1, 32 * NCT, li, lh);
      mma32(o[t], sX + r_base * FS, FS, 1, sW + (32 * NCT) * HS + t * 32, HS, 1, 32 * NCT, li, lh);
    }
    __builtin_amdgcn_wave_barrier();  // every lane is done reading this wave's agg rows
#pragma unroll
    for (int t = 0; t < NHT; ++t) {
      const int col = t * 32 + li;
      const float bias = (P.b_rel1 && col < H1) ? P.b_rel1[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r_base + acc_row(r, lh);
        const float v = (row < N && col < H1) ? gcm_act(o[t][r] + bias, P.act1) : 0.f;
        sAH[row * AS + col] = v;
        if (h1_out && row < N && col < H1) h1_out[((size_t)b * N + row) * H1 + col] = v;
      }
    }
  }
  __syncthreads();  // all of h1 and all of adj are in LDS

  // ---- layer 2 on row `cur` only -----------------------------------------------------
  // agg2[h] = sum_j adj[cur][j] * h1[j][h]: 256/HP partial sums per h, combined through LDS
  {
    constexpr int G = 256 / HP;
    const int g = tid / HP, h = tid - g * HP;
    float s = 0.f;
    for (int j = g; j < N; j += G) s = fmaf(sAdj[adj_at<NP>(cur, j)], sAH[j * AS + h], s);
    sV[64 + tid] = s;
    __syncthreads();
    if (tid < HP) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < G; ++q) t += sV[64 + q * HP + tid];
      sV[tid] = t;
      if (agg2_out && tid < H1) agg2_out[(size_t)b * H1 + tid] = t;
    }
  }
  __syncthreads();
  bool nonfinite = false;
  for (int o2 = tid; o2 < H2; o2 += 256) {
    float s = P.b_rel2 ? P.b_rel2[o2] : 0.f;
    const float* wr = P.w_rel2 + (size_t)o2 * H1;
    const float* wo = P.w_root2 + (size_t)o2 * H1;
    for (int k = 0; k < H1; ++k) s = fmaf(wr[k], sV[k], s);
    for (int k = 0; k < H1; ++k) s = fmaf(wo[k], sAH[cur * AS + k], s);
    const float v = gcm_act(s, P.act2);
    mx_out[(size_t)b * H2 + o2] = v;
    nonfinite |= !isfinite(v);
  }
  if (flags && __any(nonfinite) && lane == 0) atomicOr(flags, GCM_FLAG_NONFINITE);
}

template <int NT, int NCT, int NHT>
size_t fwd_lds_bytes() {
  using L = Lds<NT, NCT, NHT>;
  return sizeof(float) * ((size_t)L::ADJ + L::X + L::AH + 2 * (32 * NCT) * L::HS + L::SV);
}

template <int NT, int NCT, int NHT>
int launch_fwd(hipStream_t s, const float* x, const float* adj, const int64_t* cur, Gnn2 P,
               float* mx, float* h1, float* agg1, float* agg2, uint32_t* flags, int B, int N,
               int F, int H1, int H2) {
  const size_t lds = fwd_lds_bytes<NT, NCT, NHT>();
  if (lds > 160 * 1024) return GCM_EUNSUPPORTED;
  auto kern = k_gnn2_row_fwd<NT, NCT, NHT>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, s, x, adj, cur, P, mx, h1, agg1, agg2, flags,
                     N, F, H1, H2);
  return gcm_launch_status();
}


// ---------------------------------------------------------------------------
// backward of the fused step, state-advance adjoint included:
//   in : g_mx [B,H2], g_nodes_out [B,N,F] (gradient arriving from later steps, may be NULL)
//   out: g_nodes_in [B,N,F], g_obs [B,F], parameter-gradient slab of this graph
// slab layout (floats): dW_rel1 [H1*F] | dW_root1 [H1*F] | db1 [H1] | dW_rel2 [H2*H1] |
//                       dW_root2 [H2*H1] | db2 [H2]
// ---------------------------------------------------------------------------
template <int NT, int NCT, int NHT>
__global__ __launch_bounds__(256) void k_gnn2_row_bwd(
    const float* __restrict__ g_mx, const float* __restrict__ g_nodes_out,
    const float* __restrict__ x, const float* __restrict__ adj,
    const int64_t* __restrict__ cur_idx, const int64_t* __restrict__ num_nodes_in, Gnn2 P,
    const float* __restrict__ mx, const float* __restrict__ h1, const float* __restrict__ agg1,
    const float* __restrict__ agg2, float* __restrict__ g_nodes_in, float* __restrict__ g_obs,
    float* __restrict__ slabs, int accumulate, int N, int F, int H1, int H2) {
  using L = Lds<NT, NCT, NHT>;
  constexpr int NP = L::NP, FS = L::FS, HS = L::HS, FP = 32 * NCT, HP = 32 * NHT;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const float* xg = x + (size_t)b * N * F;
  const float* ag = adj + (size_t)b * N * N;
  const float* h1g = h1 + (size_t)b * N * H1;
  const float* a1g = agg1 + (size_t)b * N * F;
  int64_t cur64 = cur_idx[b];
  const int cur = cur64 < 0 ? 0 : (cur64 > N - 1 ? N - 1 : (int)cur64);
  const bool wrap = num_nodes_in[b] + 1 > N;
  float* slab = slabs + (size_t)b * (2 * (size_t)H1 * F + H1 + 2 * (size_t)H2 * H1 + H2);
  float* sl_rel1 = slab;
  float* sl_root1 = sl_rel1 + (size_t)H1 * F;
  float* sl_b1 = sl_root1 + (size_t)H1 * F;
  float* sl_rel2 = sl_b1 + H1;
  float* sl_root2 = sl_rel2 + (size_t)H2 * H1;
  float* sl_b2 = sl_root2 + (size_t)H2 * H1;

  extern __shared__ float smem[];
  float* sAdj = smem;                   // [col tile][row][33]
  float* sG = sAdj + L::ADJ;            // [NP][HS]   G1 = dh1 * act1'(h1)
  float* sD = sG + NP * HS;             // [NP][FS]   dAgg1
  float* sW = sD + NP * FS;             // w_rel1 [HP][FS] then w_root1 [HP][FS]   (B(k=h, j=f))
  float* sR = sW + 2 * HP * FS;         // [4][1024]  cross-wave reduction of the dW tiles
  float* sV = sR + 4 * 1024;            // d2 [<=256 -> uses sR instead], dagg2 [64], dh1cur [64] ...

  // ---- adjacency loads first (consumed after the barrier) ---------------------------
  const int r_base = wave * 32;
  const bool wave_live = wave < NT;
  float4 buf[NT * 4];
  const bool vec = (N & 3) == 0;
  if (wave_live) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = r_base + (lane >> 3) + 8 * q, c = t * 32 + (lane & 7) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < N) {
          const float* p = ag + (size_t)r * N + c;
          if (vec) {
            if (c < N) v = *reinterpret_cast<const float4*>(p);
          } else {
            if (c < N) v.x = p[0];
            if (c + 1 < N) v.y = p[1];
            if (c + 2 < N) v.z = p[2];
            if (c + 3 < N) v.w = p[3];
          }
        }
        buf[t * 4 + q] = v;
      }
  }
  for (int e = tid; e < HP * FP; e += 256) {
    const int h = e / FP, f = e - h * FP;
    const bool ok = h < H1 && f < F;
    sW[h * FS + f] = ok ? P.w_rel1[(size_t)h * F + f] : 0.f;
    sW[(HP + h) * FS + f] = ok ? P.w_root1[(size_t)h * F + f] : 0.f;
  }
  // ---- layer 2 backward (vector sized work) ------------------------------------------
  // d2[o] = g_mx[o] * act2'(mx[o])  -> sR[o] (H2 <= 256)
  for (int o = tid; o < H2; o += 256)
    sR[o] = g_mx[(size_t)b * H2 + o] * gcm_act_grad(mx[(size_t)b * H2 + o], P.act2);
  __syncthreads();
  for (int k = tid; k < H1; k += 256) {  // dagg2 = W_rel2^T d2 ; dh1cur = W_root2^T d2
    float s = 0.f, t = 0.f;
    for (int o = 0; o < H2; ++o) {
      s = fmaf(P.w_rel2[(size_t)o * H1 + k], sR[o], s);
      t = fmaf(P.w_root2[(size_t)o * H1 + k], sR[o], t);
    }
    sV[k] = s;
    sV[64 + k] = t;
  }
  for (int e = tid; e < H2 * H1; e += 256) {  // layer-2 parameter gradients
    const int o = e / H1, k = e - o * H1;
    const float d = sR[o];
    const float vr = d * agg2[(size_t)b * H1 + k], vo = d * h1g[(size_t)cur * H1 + k];
    sl_rel2[e] = accumulate ? sl_rel2[e] + vr : vr;
    sl_root2[e] = accumulate ? sl_root2[e] + vo : vo;
  }
  for (int o = tid; o < H2; o += 256) sl_b2[o] = accumulate ? sl_b2[o] + sR[o] : sR[o];
  __syncthreads();
  // ---- G1[j][h] = (adj[cur][j] * dagg2[h] + [j==cur] dh1cur[h]) * act1'(h1[j][h]) -------
  for (int e = tid; e < NP * HP; e += 256) {
    const int j = e / HP, h = e - j * HP;
    float v = 0.f;
    if (j < N && h < H1) {
      const float a = ag[(size_t)cur * N + j];
      const float d = a * sV[h] + (j == cur ? sV[64 + h] : 0.f);
      v = d * gcm_act_grad(h1g[(size_t)j * H1 + h], P.act1);
    }
    sG[j * HS + h] = v;
  }
  if (wave_live) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = r_base + (lane >> 3) + 8 * q;
        float* d = sAdj + (t * NP + r) * 33 + (lane & 7) * 4;
        const float4 v = buf[t * 4 + q];
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
  }
  __syncthreads();  // sG, sW, sAdj complete

  // ---- layer-1 parameter gradients: [H1 x F] = G1^T (H1 x N) @ {agg1, x} (N x F) ---------
  // every wave contracts over its own 32 rows; the four partial tiles meet in LDS.
  for (int job = 0; job < 2 * NHT * NCT; ++job) {
    const int which = job & 1, ct = (job >> 1) % NCT, ht = (job >> 1) / NCT;
    const float* src = which ? xg : a1g;
    f32x16 a;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
    if (wave_live) {
      // B(k=row, j=f) straight from HBM/L2: every element is used exactly once
      float bq[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int row = r_base + 2 * s + lh, f = ct * 32 + li;
        bq[s] = (row < N && f < F) ? src[(size_t)row * F + f] : 0.f;
      }
      const float* ap = sG + (r_base + lh) * HS + ht * 32 + li;   // A(i=h, k=row)
#pragma unroll
      for (int s = 0; s < 16; ++s)
        a = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s * HS], bq[s], a, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sR[wave * 1024 + acc_row(r, lh) * 32 + li] = a[r];
    __syncthreads();
    for (int e = tid; e < 1024; e += 256) {
      const int hh = ht * 32 + (e >> 5), ff = ct * 32 + (e & 31);
      if (hh < H1 && ff < F) {
        const float v = (sR[e] + sR[1024 + e]) + (sR[2048 + e] + sR[3072 + e]);
        float* dst = (which ? sl_root1 : sl_rel1) + (size_t)hh * F + ff;
        *dst = accumulate ? *dst + v : v;
      }
    }
    __syncthreads();
  }
  for (int h = tid; h < H1; h += 256) {  // db1 = column sums of G1
    float s = 0.f;
    for (int j = 0; j < N; ++j) s += sG[j * HS + h];
    sl_b1[h] = accumulate ? sl_b1[h] + s : s;
  }

  // ---- dAgg1 = G1 @ W_rel1 -> LDS ;  acc = G1 @ W_root1 (root part of dX) -----------------
  f32x16 acc[NCT];
  if (wave_live) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      f32x16 d;
#pragma unroll
      for (int r = 0; r < 16; ++r) { d[r] = 0.f; acc[c][r] = 0.f; }
      mma32(d, sG + r_base * HS, HS, 1, sW + c * 32, FS, 1, HP, li, lh);
      mma32(acc[c], sG + r_base * HS, HS, 1, sW + HP * FS + c * 32, FS, 1, HP, li, lh);
#pragma unroll
      for (int r = 0; r < 16; ++r) sD[(r_base + acc_row(r, lh)) * FS + c * 32 + li] = d[r];
    }
  }
  __syncthreads();
  // ---- dX[i] += sum_k adj[k][i] * dAgg1[k]   (A read down the columns of the adj image) ----
  if (wave_live) {
#pragma unroll
    for (int c = 0; c < NCT; ++c)
      mma32(acc[c], sAdj + (wave * NP) * 33, 1, 33, sD + c * 32, FS, 1, NP, li, lh);
    // ---- epilogue: add the gradient from later steps, undo insert + roll (gcm.py:262-278) ----
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r_base + acc_row(r, lh), col = c * 32 + li;
        if (row < N && col < F) {
          float v = acc[c][r];
          if (g_nodes_out) v += g_nodes_out[((size_t)b * N + row) * F + col];
          if (row == cur) {
            g_obs[(size_t)b * F + col] = v;     // the inserted row belongs to the observation
            if (!wrap) g_nodes_in[((size_t)b * N + row) * F + col] = 0.f;
          } else if (!wrap) {
            g_nodes_in[((size_t)b * N + row) * F + col] = v;
          } else {
            g_nodes_in[((size_t)b * N + row + 1) * F + col] = v;   // out[r] = in[r+1]
          }
        }
      }
  }
  if (wrap)  // in[0] was cleared before the roll: no gradient
    for (int c = tid; c < F; c += 256) g_nodes_in[(size_t)b * N * F + c] = 0.f;
}

template <int NT, int NCT, int NHT>
size_t bwd_lds_bytes() {
  using L = Lds<NT, NCT, NHT>;
  return sizeof(float) * ((size_t)L::ADJ + L::NP * L::HS + L::NP * L::FS + 2 * (32 * NHT) * L::FS +
                          4 * 1024 + L::SV);
}

template <int NT, int NCT, int NHT>
int launch_bwd(hipStream_t s, const float* g_mx, const float* g_nodes_out, const float* x,
               const float* adj, const int64_t* cur, const int64_t* nn_in, Gnn2 P, const float* mx,
               const float* h1, const float* agg1, const float* agg2, float* g_nodes_in,
               float* g_obs, float* slabs, int accumulate, int B, int N, int F, int H1, int H2) {
  const size_t lds = bwd_lds_bytes<NT, NCT, NHT>();
  if (lds > 160 * 1024) return GCM_EUNSUPPORTED;
  auto kern = k_gnn2_row_bwd<NT, NCT, NHT>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, s, g_mx, g_nodes_out, x, adj, cur, nn_in, P,
                     mx, h1, agg1, agg2, g_nodes_in, g_obs, slabs, accumulate, N, F, H1, H2);
  return gcm_launch_status();
}

// sum the per-graph slabs: out[e] = sum_b slabs[b][e]   (fixed order => deterministic)
__global__ void k_sum_slabs(const float* __restrict__ slabs, int n_slabs, int len,
                            float* __restrict__ out) {
  __shared__ float part[256];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63);
  const int q = threadIdx.x >> 6;
  float s = 0.f;
  if (e < len)
    for (int i = q; i < n_slabs; i += 4) s += slabs[(size_t)i * len + e];
  part[threadIdx.x] = s;
  __syncthreads();
  if (q == 0 && e < len)
    out[e] = (part[threadIdx.x] + part[threadIdx.x + 64]) +
             (part[threadIdx.x + 128] + part[threadIdx.x + 192]);
}

}  // namespace

extern "C" int gcm_dense_gnn2_row_supported(int N, int F, int H1, int H2) {
  if (N <= 0 || F <= 0 || H1 <= 0 || H2 <= 0) return 0;
  if (N > 128 || F > 64 || H1 > 64 || H2 > 256) return 0;
  const int NT = (N + 31) / 32, NCT = (F + 31) / 32, NHT = (H1 + 31) / 32;
  const int NP = 32 * NT, FS = 32 * NCT + 1, HS = 32 * NHT + 1;
  const size_t fwd = (size_t)NT * NP * 33 + (size_t)NP * FS + (size_t)NP * (FS > HS ? FS : HS) +
                     2 * (32 * NCT) * HS + 320;
  const size_t bwd = (size_t)NT * NP * 33 + (size_t)NP * HS + (size_t)NP * FS +
                     2 * (32 * NHT) * FS + 4 * 1024 + 320;
  return (fwd > bwd ? fwd : bwd) * sizeof(float) <= 160 * 1024 ? 1 : 0;
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
  Gnn2 P{w_rel1, b_rel1, w_root1, w_rel2, b_rel2, w_root2, act1, act2};
  hipStream_t s = (hipStream_t)stream;
  const int NT = (N + 31) / 32, NCT = (F + 31) / 32, NHT = (H1 + 31) / 32;
#define GCM_F(a, b_, c)                                                                          \
  if (NT == a && NCT == b_ && NHT == c)                                                          \
    return launch_fwd<a, b_, c>(s, x, adj, cur_idx, P, mx, h1, agg1, agg2, flags, B, N, F, H1, H2);
  GCM_F(1, 1, 1) GCM_F(1, 1, 2) GCM_F(1, 2, 1) GCM_F(1, 2, 2)
  GCM_F(2, 1, 1) GCM_F(2, 1, 2) GCM_F(2, 2, 1) GCM_F(2, 2, 2)
  GCM_F(3, 1, 1) GCM_F(3, 1, 2) GCM_F(3, 2, 1) GCM_F(3, 2, 2)
  GCM_F(4, 1, 1) GCM_F(4, 1, 2) GCM_F(4, 2, 1) GCM_F(4, 2, 2)
#undef GCM_F
  return GCM_EUNSUPPORTED;
}

extern "C" size_t gcm_dense_gnn2_param_count(int F, int H1, int H2) {
  return 2 * (size_t)H1 * F + H1 + 2 * (size_t)H2 * H1 + H2;
}

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
  Gnn2 P{w_rel1, b_rel1, w_root1, w_rel2, b_rel2, w_root2, act1, act2};
  hipStream_t s = (hipStream_t)stream;
  const int NT = (N + 31) / 32, NCT = (F + 31) / 32, NHT = (H1 + 31) / 32;
#define GCM_B(a, b_, c)                                                                         \
  if (NT == a && NCT == b_ && NHT == c)                                                         \
    return launch_bwd<a, b_, c>(s, g_mx, g_nodes_out, x, adj, cur_idx, num_nodes_in, P, mx, h1, \
                                agg1, agg2, g_nodes_in, g_obs, slabs, accumulate, B, N, F, H1,  \
                                H2);
  GCM_B(1, 1, 1) GCM_B(1, 1, 2) GCM_B(1, 2, 1) GCM_B(1, 2, 2)
  GCM_B(2, 1, 1) GCM_B(2, 1, 2) GCM_B(2, 2, 1) GCM_B(2, 2, 2)
  GCM_B(3, 1, 1) GCM_B(3, 1, 2) GCM_B(3, 2, 1) GCM_B(3, 2, 2)
  GCM_B(4, 1, 1) GCM_B(4, 1, 2) GCM_B(4, 2, 1) GCM_B(4, 2, 2)
#undef GCM_B
  return GCM_EUNSUPPORTED;
}

extern "C" int gcm_sum_slabs(const float* slabs, int n_slabs, int len, float* out,
                             gcm_stream_t stream) {
  GCM_REQUIRE(slabs && out && n_slabs > 0 && len > 0);
  hipLaunchKernelGGL(k_sum_slabs, dim3((len + 63) / 64), dim3(256), 0, (hipStream_t)stream, slabs,
                     n_slabs, len, out);
  return gcm_launch_status();
}
