// One DenseGCM step on the LIVE ROWS of the graph (gcm.py:262-321 with the canonical 2-layer
// DenseGraphConv GNN of README.md:52-62 and the index-writing selectors folded in).
//
// DenseGCM keeps one row of the last layer (mx = node_feats[b, cur], gcm.py:314), so
//     mx = act2( W_rel2 (sum_j adj[cur,j] h1[j]) + W_root2 h1[cur] + b2 )
// needs h1 only on the rows j with adj[cur, j] != 0 and on row cur: the "live rows" (4 of 128 for
// TemporalBackedge([1,2,4]), all rows <= cur for DenseEdge).  Layer 1 is evaluated on exactly those
// rows, h1[j] = act1(W_rel1 (adj[j,:] @ x) + W_root1 x[j] + b1), 16 at a time on the 16x16x4 fp32
// MFMA.  Exact: everything else is multiplied by zero in the reference.
//
// What that buys on the memory side: the kernel reads the node matrix, row cur of the adjacency and
// the live rows - not the [N,N] adjacency - and, when the caller donates the state
// (nodes_out == nodes_in, adj_out == adj_in), writes back only the inserted node, the selector's
// row / column entries and the count: the functional copy of gcm.py:262,278,286 (2 x 21 MB per
// step at B=256, N=128, F=32) disappears.  Without donation the same kernel streams the copy
// HBM -> registers -> HBM beside the row pipeline.  The overflow roll (gcm.py:323-355) is done
// by the graph's workgroup, in place when donated (every load lands before the first store).
//
// Saved for BPTT: per graph the live-row list, their coefficients adj[cur, j] and the rows
// h1[j] | agg1[j] | x[j], plus agg2 | h1[cur] and mx - a few KB per graph-step, which is ALL the
// time-parallel backward (rows_bptt.hip) reads.  The state itself is never needed again, which is
// what makes donation compatible with BPTT.
//
// One workgroup (4 waves) per graph.  N <= 128, N % 4 == 0, F % 4 == 0, F, H1, H2 <= 64.
#include "fused_common.h"
#include "rows_common.h"

namespace gcm_rows {

using gcm_fused::Edits;
using gcm_fused::Gnn2;

template <int FP, int HP, int H2P>
struct Lds {
  static constexpr int XS = FP + 16;       // x image [128][XS]: B operand, stride = 16 mod 32
  static constexpr int RS = 130;           // live adjacency rows [16][RS]: A operand, 2 mod 32
  static constexpr int AS = 2 * FP + 2;    // [agg1 | x[j]] rows [16][AS]: A operand, 2 mod 32
  static constexpr int W1S = 2 * FP + 2;   // W1 [h][rel f | root f]: B operand read as B[k][n=h]
  static constexpr int HS = HP + 1;        // h1 rows [16][HS]
  static constexpr int W2S = 2 * HP + 1;   // W2 [o][rel k | root k]
  static constexpr int X = 128 * XS, ROWS = 16 * RS, AGG = 16 * AS, W1 = HP * W1S, H1R = 16 * HS;
  static constexpr int W2 = H2P * W2S;
  // rowcur[128] | coef[128] | live j [128] | v [2*HP] | partials [256] | ints [16]
  static constexpr int MISC = 128 + 128 + 128 + 2 * HP + 256 + 16;
  static constexpr int TOTAL = X + ROWS + AGG + W1 + H1R + W2 + MISC;
};

template <int ADJ_PER, int NODE_PER>
__device__ __forceinline__ void store_copy(const float4 (&ca)[ADJ_PER], const float4 (&cn)[NODE_PER],
                                           float* ag, float* ng, int tid, int N, int N4, int F4) {
#pragma unroll
  for (int i = 0; i < ADJ_PER; ++i) {
    const int e4 = tid + 256 * i;
    if (e4 < N * N4) *reinterpret_cast<float4*>(ag + e4 * 4) = ca[i];
  }
#pragma unroll
  for (int i = 0; i < NODE_PER; ++i) {
    const int e4 = tid + 256 * i;
    if (e4 < N * F4) *reinterpret_cast<float4*>(ng + e4 * 4) = cn[i];
  }
}

template <int FP, int HP, int H2P>
__global__ __launch_bounds__(256) void k_step_rows(
    const float* __restrict__ obs, const float* nodes_in, const float* adj_in,
    const int64_t* count_in, float* nodes_out, float* adj_out, int64_t* count_out,
    int64_t* __restrict__ cur_out, Edits E, Gnn2 P, float* __restrict__ mx_out,
    float* __restrict__ saved, SavedLayout lay, uint32_t* __restrict__ flags, int N, int F, int H1,
    int H2) {
  using L = Lds<FP, HP, H2P>;
  constexpr int XS = L::XS, RS = L::RS, AS = L::AS, W1S = L::W1S, HS = L::HS, W2S = L::W2S;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m16 = lane & 15, kq = lane >> 4;
  const bool inplace = adj_out == adj_in;
  const int N4 = N >> 2, F4 = F >> 2;

  const int64_t n_in = count_in[b];
  const bool wrap = n_in + 1 > N;
  const int64_t c64 = wrap ? n_in - 1 : n_in;
  const int cur = c64 < 0 ? 0 : (c64 > N - 1 ? N - 1 : (int)c64);

  extern __shared__ float smem[];
  float* sX = smem;
  float* sRows = sX + L::X;
  float* sAgg = sRows + L::ROWS;
  float* sW1 = sAgg + L::AGG;
  float* sH1 = sW1 + L::W1;
  float* sW2 = sH1 + L::H1R;
  float* sRowCur = sW2 + L::W2;
  float* sCoef = sRowCur + 128;
  int* sLive = reinterpret_cast<int*>(sCoef + 128);
  float* sV = reinterpret_cast<float*>(sLive + 128);
  float* sPart = sV + 2 * HP;
  int* sInt = reinterpret_cast<int*>(sPart + 256);   // [0..1] live count of wave 0 / 1, [2] l_cur,
                                                     // [3] K-chunk mask of the current row group

  const float* ng_in = nodes_in + (size_t)b * N * F;
  const float* ag_in = adj_in + (size_t)b * N * N;
  float* ng = nodes_out + (size_t)b * N * F;
  float* ag = adj_out + (size_t)b * N * N;

  int lane_hop = -1, lane_dir = 0;   // lane i < n_hops: the i-th folded temporal edit
  if (lane < E.n_hops) {
    lane_hop = E.hops[lane & 15];
    lane_dir = E.dir[lane & 15];
  }
  const int n_hops = E.n_hops;
  const bool dense = E.dense != 0;

  // ---- the state copy / overflow roll ---------------------------------------------------------
  // out[r][c] = wrap ? in[r+1][c+1] (last row / column zero) : in[r][c]; nodes alike by rows.
  // Needed when the state is not donated (functional copy) or the graph overflows.  The loads are
  // issued here, the stores after the pipeline's own first loads (below) - except for a donated
  // overflow, where source and destination alias and everything must land first.
  constexpr int ADJ_PER = 16, NODE_PER = (128 * FP / 4 + 255) / 256;
  float4 ca[ADJ_PER], cn[NODE_PER];
  const bool need_copy = !inplace || wrap;   // uniform per workgroup
  if (need_copy) {
    const int sh = wrap ? 1 : 0;
#pragma unroll
    for (int i = 0; i < ADJ_PER; ++i) {
      const int e4 = tid + 256 * i;
      const int r = e4 / N4, c = (e4 - r * N4) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e4 < N * N4) {   // uniform per i for N = 128
        const int rs = r + sh < N ? r + sh : N - 1;
        const bool tail = wrap && c + 4 >= N;   // in[.][N] does not exist: shift in registers
        const float* p = ag_in + rs * N + c + (tail ? 0 : sh);
        __builtin_memcpy(&v, p, sizeof(float4));   // dword-aligned 16-byte load
        if (tail) v = make_float4(v.y, v.z, v.w, 0.f);
        if (r + sh >= N) v = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      ca[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NODE_PER; ++i) {
      const int e4 = tid + 256 * i;
      const int r = e4 / F4, c = (e4 - r * F4) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e4 < N * F4) {
        const int rs = r + sh < N ? r + sh : N - 1;
        v = *reinterpret_cast<const float4*>(ng_in + rs * F + c);
        if (r + sh >= N) v = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      cn[i] = v;
    }
  }
  // where the pipeline reads the (advanced, pre-selector) state from
  const float* ag_rd = ag_in;
  const float* ng_rd = ng_in;
  if (wrap) {
    if (inplace) {   // every load of the roll lands before the first store
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    store_copy<ADJ_PER, NODE_PER>(ca, cn, ag, ng, tid, N, N4, F4);
    __syncthreads();   // the rolled state is visible to the whole workgroup
    ag_rd = ag;
    ng_rd = ng;
  }

  // ---- phase A: every load that depends on cur only ----------------------------------------------
  // x image = node matrix with the observation in row cur (gcm.py:274); rows >= N and columns >= F
  // are zero (K / N padding of the MFMAs)
  {
    constexpr int PER = 128 * FP / 4 / 256;   // float4 per thread of the padded image
    float4 xv[PER];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e4 = tid + 256 * i, r = e4 / (FP / 4), c = (e4 % (FP / 4)) * 4;
      const bool ok = r < N && c < F;
      const int rc = r < N ? r : N - 1, cc = c < F ? c : F - 4;
      const float* src = r == cur ? obs + (size_t)b * F + cc : ng_rd + rc * F + cc;
      const float4 t = *reinterpret_cast<const float4*>(src);
      xv[i] = ok ? t : zero4;
    }
    // layer-1 weights as [h][rel f | root f] (padding zero), layer-2 as [o][rel k | root k]
    constexpr int PW1 = HP * 2 * FP / 256;
    float w1v[PW1];
#pragma unroll
    for (int i = 0; i < PW1; ++i) {
      const int e = tid + 256 * i, h = e / (2 * FP), k = e % (2 * FP);
      const int f = k < FP ? k : k - FP;
      const float* src = k < FP ? P.w_rel1 : P.w_root1;
      const float t = src[(h < H1 ? h : H1 - 1) * F + (f < F ? f : F - 1)];
      w1v[i] = (h < H1 && f < F) ? t : 0.f;
    }
    constexpr int PW2 = H2P * 2 * HP / 256;
    float w2v[PW2];
#pragma unroll
    for (int i = 0; i < PW2; ++i) {
      const int e = tid + 256 * i, o = e / (2 * HP), k = e % (2 * HP);
      const int kk = k < HP ? k : k - HP;
      const float* src = k < HP ? P.w_rel2 : P.w_root2;
      const float t = src[(o < H2 ? o : H2 - 1) * H1 + (kk < H1 ? kk : H1 - 1)];
      w2v[i] = (o < H2 && kk < H1) ? t : 0.f;
    }
    // row cur of the advanced adjacency, before the selectors (all zero for a state this code
    // produced; a caller's own state may hold anything)
    float rc_val = 0.f;
    if (tid < 128) {
      const float t = ag_rd[cur * N + (tid < N ? tid : N - 1)];
      rc_val = tid < N ? t : 0.f;
    }
    if (need_copy && !wrap) store_copy<ADJ_PER, NODE_PER>(ca, cn, ag, ng, tid, N, N4, F4);   // functional copy: stores behind the pipeline's loads
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e4 = tid + 256 * i, r = e4 / (FP / 4), c = (e4 % (FP / 4)) * 4;
      float* d = sX + r * XS + c;
      d[0] = xv[i].x; d[1] = xv[i].y; d[2] = xv[i].z; d[3] = xv[i].w;
    }
#pragma unroll
    for (int i = 0; i < PW1; ++i) {
      const int e = tid + 256 * i, h = e / (2 * FP), k = e % (2 * FP);
      sW1[h * W1S + k] = w1v[i];
    }
#pragma unroll
    for (int i = 0; i < PW2; ++i) {
      const int e = tid + 256 * i, o = e / (2 * HP), k = e % (2 * HP);
      sW2[o * W2S + k] = w2v[i];
    }
    // ---- phase B: selector writes on row cur (temporal.py:72-88, dense.py:16-21), live list ------
    float r_new = rc_val;
    bool pred = false;
    unsigned long long bal = 0;
    if (tid < 128) {
      const int j = tid;
      for (int i = 0; i < n_hops; ++i) {
        const int h = __builtin_amdgcn_readlane(lane_hop, i), d = __builtin_amdgcn_readlane(lane_dir, i);
        if (((d & GCM_DIR_FORWARD) || h == 0) && h >= 0 && cur >= h && j == cur - h) r_new = 1.f;
      }
      if (dense && j <= cur) r_new = 1.f;
      sRowCur[j] = r_new;
      pred = j < N && (r_new != 0.f || j == cur);
      bal = __ballot(pred);
      if (lane == 0) sInt[wave] = __popcll(bal);
    }
    if (tid == 0) sInt[3] = 0;
    __syncthreads();   // x image, weights, counts; the copy's stores are ordered before what follows
    if (tid < 128) {
      const int j = tid;
      const int pos = (wave ? sInt[0] : 0) + __popcll(bal & ((1ull << lane) - 1ull));
      if (pred) {
        sLive[pos] = j;
        sCoef[pos] = r_new;
        if (j == cur) sInt[2] = pos;
      }
      // the selector's entries go back to HBM behind the copy's stores
      if (j < N && r_new != rc_val) ag[cur * N + j] = r_new;
    } else {
      const int t2 = tid - 128;   // column cur: backward hops (temporal) / rows < cur (dense)
      if (wave == 2 && (lane_dir & GCM_DIR_BACKWARD) && lane_hop >= 0 && cur >= lane_hop)
        ag[(cur - lane_hop) * N + cur] = 1.f;   // lane i < n_hops holds hop i (lane_hop = -1 beyond)
      if (dense) {
        for (int r = t2; r < cur; r += 128) ag[r * N + cur] = 1.f;
      }
      if (t2 < F4) {   // the inserted node (gcm.py:274)
        *reinterpret_cast<float4*>(ng + cur * F + t2 * 4) =
            *reinterpret_cast<const float4*>(obs + (size_t)b * F + t2 * 4);
      }
    }
  }
  __syncthreads();   // live list
  const int Ltot = sInt[0] + sInt[1];
  const int l_cur = sInt[2];
  float* sv_rows = nullptr;
  if (saved) {
    sv_rows = saved + lay.o_rows + (size_t)b * N * lay.rw;
    if (tid == 0) {
      int* hdr = reinterpret_cast<int*>(saved + lay.o_hdr) + 4 * b;
      hdr[0] = Ltot; hdr[1] = l_cur; hdr[2] = cur; hdr[3] = wrap ? 1 : 0;
    }
    float* cf = saved + lay.o_coef + (size_t)b * N;
    for (int l = tid; l < Ltot; l += 256) cf[l] = sCoef[l];
  }

  // ---- the live rows, 16 at a time ---------------------------------------------------------------
  float a2 = 0.f, h1c = 0.f;   // tid < HP: agg2[tid], h1[cur][tid]
  const float bias1 = (P.b_rel1 && wave < HP / 16 && 16 * wave + m16 < H1) ? P.b_rel1[16 * wave + m16] : 0.f;
  const int act1_v = gcm_vgpr(P.act1), act2_v = gcm_vgpr(P.act2);
  const int n_groups = (Ltot + 15) >> 4;
#pragma unroll 1
  for (int g = 0; g < n_groups; ++g) {
    // -- C: adjacency rows of this group -> LDS (with the selector's column-cur entries applied in
    //       registers: the stores above may or may not have landed, both give the same row)
    {
      const int l = tid >> 4, c4 = tid & 15;
      const int lg = 16 * g + l;
      const bool valid = lg < Ltot;
      const int j = valid ? sLive[lg] : 0;
      bool colcur = dense && j < cur;   // does (j, cur) get an entry from the selectors?
      for (int i = 0; i < n_hops; ++i) {
        const int h = __builtin_amdgcn_readlane(lane_hop, i), d = __builtin_amdgcn_readlane(lane_dir, i);
        colcur |= (d & GCM_DIR_BACKWARD) && h >= 0 && cur >= h && j == cur - h;
      }
      unsigned nzbits = 0;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int c = (c4 + 16 * q) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < N) {
          v = *reinterpret_cast<const float4*>(ag_rd + j * N + c);
          if (j == cur) v = make_float4(sRowCur[c], sRowCur[c + 1], sRowCur[c + 2], sRowCur[c + 3]);
          else if (colcur) {
            const int k = cur - c;
            v.x = k == 0 ? 1.f : v.x;
            v.y = k == 1 ? 1.f : v.y;
            v.z = k == 2 ? 1.f : v.z;
            v.w = k == 3 ? 1.f : v.w;
          }
        }
        if (!valid) v = make_float4(0.f, 0.f, 0.f, 0.f);
        float* d = sRows + l * RS + c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        const bool nz = (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
        const unsigned long long bal = __ballot(nz);   // lane = 16*l' + c4: fold the 4 rows
        const unsigned m = (unsigned)((bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xffffull);
        nzbits |= m << (16 * q);
      }
      // x[j] beside agg1 in the A image of the linears
#pragma unroll
      for (int i = 0; i < FP / 16; ++i) {
        const int f = c4 + 16 * i;
        sAgg[l * AS + FP + f] = valid ? sX[j * XS + f] : 0.f;
      }
      if (lane == 0 && nzbits) atomicOr(reinterpret_cast<unsigned*>(&sInt[3]), nzbits);
    }
    __syncthreads();
    // -- D: agg1 = rows @ x over the non-zero 4-column chunks (exact: zero chunks add nothing)
    if (wave < FP / 16) {
      unsigned km = __builtin_amdgcn_readfirstlane((unsigned)sInt[3]);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* ap = sRows + m16 * RS + kq;
      const float* bp = sX + kq * XS + 16 * wave + m16;
      while (km) {
        const int c = __builtin_ctz(km);
        km &= km - 1;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * c], bp[4 * c * XS], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int l = 4 * kq + r, f = 16 * wave + m16;
        sAgg[l * AS + f] = acc[r];
      }
    }
    __syncthreads();
    // -- E: h1 = act1([agg1 | x[j]] @ [W_rel1 | W_root1]^T + b1)
    if (tid == 255) sInt[3] = 0;   // D has read the chunk mask (barrier above); C of the next group ORs after two more
    if (wave < HP / 16) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* ap = sAgg + m16 * AS + kq;
      const float* bp = sW1 + (16 * wave + m16) * W1S + kq;
      float av[2 * FP / 4], bv[2 * FP / 4];
#pragma unroll
      for (int s = 0; s < 2 * FP / 4; ++s) {
        av[s] = ap[4 * s];
        bv[s] = bp[4 * s];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 2 * FP / 4; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], bv[s], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int l = 4 * kq + r, h = 16 * wave + m16;
        sH1[l * HS + h] = gcm_act_sel(acc[r] + bias1, act1_v);
      }
    }
    __syncthreads();
    // -- F: layer-2 aggregation over this group; rows saved for BPTT
    if (tid < HP) {
      const int lmax = Ltot - 16 * g < 16 ? Ltot - 16 * g : 16;
      for (int l = 0; l < lmax; ++l) a2 = fmaf(sCoef[16 * g + l], sH1[l * HS + tid], a2);
      if (l_cur >= 16 * g && l_cur < 16 * g + 16) h1c = sH1[(l_cur - 16 * g) * HS + tid];
    }
    if (sv_rows) {
      const int rw = lay.rw;   // h1 [H1] | agg1 [F] | x[j] [F]
      for (int e = tid; e < 16 * rw; e += 256) {
        const int l = e / rw, k = e - l * rw;
        if (16 * g + l < Ltot) {
          const float v = k < H1 ? sH1[l * HS + k]
                                 : (k < H1 + F ? sAgg[l * AS + (k - H1)] : sAgg[l * AS + FP + (k - H1 - F)]);
          sv_rows[(size_t)(16 * g + l) * rw + k] = v;
        }
      }
    }
    __syncthreads();   // sRows / sAgg / sH1 are rewritten by the next group
  }

  // ---- layer 2 on row cur: mx = act2(W2c v + b2), v = agg2 | h1[cur] -------------------------------
  if (tid < HP) {
    sV[tid] = a2;
    sV[HP + tid] = h1c;
    if (saved && tid < H1) {
      float* v = saved + lay.o_v + (size_t)b * 2 * H1;
      v[tid] = a2;
      v[H1 + tid] = h1c;
    }
  }
  __syncthreads();
  {
    constexpr int G2 = 256 / H2P, KC = (2 * HP) / G2;
    const int gq = tid / H2P, o = tid - gq * H2P;
    float s = 0.f;
    const float* wrow = sW2 + o * W2S + gq * KC;
    const float* vv = sV + gq * KC;
#pragma unroll
    for (int k = 0; k < KC; ++k) s = fmaf(wrow[k], vv[k], s);
    sPart[tid] = s;
    __syncthreads();
    bool nonfinite = false;
    if (tid < H2) {
      float t = P.b_rel2 ? P.b_rel2[tid] : 0.f;
#pragma unroll
      for (int q = 0; q < G2; ++q) t += sPart[q * H2P + tid];
      const float v = gcm_act_sel(t, act2_v);
      mx_out[(size_t)b * H2 + tid] = v;
      nonfinite = !isfinite(v);
    }
    if (wave == 0) {   // H2 <= 64: every output lives in wave 0
      const bool any_bad = __any(nonfinite);
      if (any_bad && lane == 0) atomicOr(flags, GCM_FLAG_NONFINITE);
    }
  }
  if (tid == 0) {   // last: count_out may alias count_in, which every wave has read by now
    count_out[b] = cur + 1;
    if (cur_out) cur_out[b] = cur;
    const uint32_t f = (wrap ? GCM_FLAG_WRAPPED : 0u) | ((n_in < 0 || n_in > N) ? GCM_FLAG_BAD_COUNT : 0u);
    if (f) atomicOr(flags, f);
  }
}

template <int FP, int HP, int H2P>
int launch(hipStream_t s, const float* obs, const float* nodes_in, const float* adj_in,
           const int64_t* count_in, float* nodes_out, float* adj_out, int64_t* count_out,
           int64_t* cur_out, const Edits& E, const Gnn2& P, float* mx, float* saved,
           const SavedLayout& lay, uint32_t* flags, int B, int N, int F, int H1, int H2) {
  constexpr size_t lds = sizeof(float) * (size_t)Lds<FP, HP, H2P>::TOTAL;
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = k_step_rows<FP, HP, H2P>;
  if (lds > 64 * 1024) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    static bool attr_set[64] = {};   // per device (index clamped): the attribute is per context
    bool& done = attr_set[dev & 63];
    if (!done) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      done = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, s, obs, nodes_in, adj_in, count_in, nodes_out,
                     adj_out, count_out, cur_out, E, P, mx, saved, lay, flags, N, F, H1, H2);
  return gcm_launch_status();
}

}  // namespace gcm_rows

extern "C" int gcm_dense_rows_supported(int N, int F, int H1, int H2) {
  if (N <= 0 || F <= 0 || H1 <= 0 || H2 <= 0) return 0;
  if (N > 128 || F > 64 || H1 > 64 || H2 > 64) return 0;
  return ((N & 3) == 0 && (F & 3) == 0) ? 1 : 0;
}

extern "C" int gcm_dense_rows_layout(int B, int N, int F, int H1, int H2, size_t* out6) {
  GCM_REQUIRE(out6 && B > 0 && N > 0 && F > 0 && H1 > 0 && H2 > 0);
  const gcm_rows::SavedLayout lay = gcm_rows::make_layout(B, N, F, H1, H2);
  out6[0] = lay.total; out6[1] = lay.o_v; out6[2] = lay.o_hdr; out6[3] = lay.o_coef;
  out6[4] = lay.o_rows; out6[5] = (size_t)lay.rw;
  return GCM_OK;
}

extern "C" int gcm_dense_rows_step_fwd(const float* obs, const float* nodes_in, const float* adj_in,
                                       const int64_t* count_in, float* nodes_out, float* adj_out,
                                       int64_t* count_out, int64_t* cur_out,
                                       const gcm_selector_desc* selectors, int n_selectors,
                                       const float* params, int has_bias, int act1, int act2,
                                       float* mx, float* saved, uint32_t* flags, int B, int N,
                                       int F, int H1, int H2, gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes_in && adj_in && count_in && nodes_out && adj_out && count_out && params &&
              mx && flags);
  GCM_REQUIRE(B > 0 && (selectors || n_selectors == 0));
  GCM_REQUIRE((nodes_out == nodes_in) == (adj_out == adj_in));   // donate both or neither
  if (!gcm_dense_rows_supported(N, F, H1, H2)) return GCM_EUNSUPPORTED;
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
  const float* w_rel1 = params;
  const float* w_root1 = w_rel1 + (size_t)H1 * F;
  const float* b1 = w_root1 + (size_t)H1 * F;
  const float* w_rel2 = b1 + H1;
  const float* w_root2 = w_rel2 + (size_t)H2 * H1;
  const float* b2 = w_root2 + (size_t)H2 * H1;
  gcm_fused::Gnn2 P{w_rel1, (has_bias & 1) ? b1 : nullptr, w_root1, w_rel2,
                    (has_bias & 2) ? b2 : nullptr, w_root2, act1, act2};
  const gcm_rows::SavedLayout lay = gcm_rows::make_layout(B, N, F, H1, H2);
  hipStream_t s = (hipStream_t)stream;
  const int fp = F <= 32 ? 32 : 64, hp = H1 <= 32 ? 32 : 64, h2p = H2 <= 32 ? 32 : 64;
#define GCM_R(a, b_, c)                                                                          \
  if (fp == a && hp == b_ && h2p == c)                                                           \
    return gcm_rows::launch<a, b_, c>(s, obs, nodes_in, adj_in, count_in, nodes_out, adj_out,    \
                                      count_out, cur_out, E, P, mx, saved, lay, flags, B, N, F,  \
                                      H1, H2);
  GCM_R(32, 32, 32) GCM_R(32, 32, 64) GCM_R(32, 64, 32) GCM_R(32, 64, 64)
  GCM_R(64, 32, 32) GCM_R(64, 32, 64) GCM_R(64, 64, 32) GCM_R(64, 64, 64)
#undef GCM_R
  return GCM_EUNSUPPORTED;
}
