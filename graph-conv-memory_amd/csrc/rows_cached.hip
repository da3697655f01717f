// Cached live-row step (round 3): the DenseGCM step (gcm.py:262-321) for a chain of hidden states that started
// from EMPTY graphs, has not overflowed (fewer than N steps) and whose selectors only ever write row cur of the
// adjacency (TemporalBackedge with direction "forward", temporal.py:72-88; DenseEdge, dense.py:16-21, also writes
// COLUMN cur, "backward" / "both" hops write into older rows: their layer-1 rows change at every step).
//
// In such a chain the rows of layer 1 are FINAL once written: row j's adjacency entries are written at step j
// and never again, and they point at older nodes, whose features do not change.  k_step_rows re-evaluates the
// live rows (cur - hop) at every step, which costs it two dependent round trips (row cur of the adjacency, then
// the live rows' adjacency rows) before their inputs can even be fetched.  Here the chain keeps
//     cH [B,N,H1] = h1 of every node,  cA [B,N,F] = agg1,  cX [B,N,F] = the node itself
// and a step computes row cur alone: its selected set S follows from `cur` and the selectors' parameters, so
// every input (the |S| node rows, the |S| cached h1 rows, the observation) is fetched in ONE round trip behind
// the count; then agg1[cur], h1[cur], agg2 = sum_S h1[j], the belief - four small matrix-vector products.
// One wave per graph, four graphs per workgroup (the weights staged once per workgroup in LDS, rows at an
// odd stride: lane o reads row o conflict-free).  The state is advanced in place (donated): the observation
// into row cur of the node matrix, the selected entries into row cur of the adjacency, count + 1.
//
// The record of such a step (gcm_dense_rows_cached_layout) holds what the time-parallel backward cannot find in
// the caches: mx | v = agg2, h1[cur] | hdr (L, l_cur, cur, 0) | coef [B,N] | live [B,N] (the rows of S and row cur,
// ascending; coef = adj[cur, j], 0 for row cur without a self loop).  k_bptt_rows<.., 3> gathers the rows'
// (h1 | agg1 | x) from the caches.
#include <hip/hip_ext.h>

#include <algorithm>

#include "fused_common.h"
#include "gcm_common.h"
#include "rows_common.h"

#ifdef GCM_STAMPS
__device__ unsigned long long g_stamps[32];
extern "C" int gcm_debug_read_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
#endif

namespace gcm_rows {

// FP = F, HP = H1 (32 or 64: compile-time, so that the piece arithmetic is shifts and the matrix-vector loops are
// branch-free - a bounds check per element kept each LDS read behind its own branch, one round trip per element:
// 5 us of the first version).  H2 <= 64.
template <int FP, int HP>
__global__ __launch_bounds__(256) void k_step_rows_cached(
    const float* __restrict__ obs, float* __restrict__ nodes, float* __restrict__ adj, int64_t* __restrict__ count,
    gcm_fused::Edits E, const float* __restrict__ params, int act1, int act2, float* __restrict__ cH,
    float* __restrict__ cA, float* __restrict__ cX, float* __restrict__ saved, CachedLayout lay,
    uint32_t* __restrict__ flags, int B, int N, int H2, int cur_host) {
  constexpr int F = FP, H1 = HP, FS = FP + 1, HS = HP + 1;
  extern __shared__ float sW[];
  float* sR1 = sW;                 // W_rel1  [H1][FS]   (row o at an odd stride: lane o reads row o conflict-free)
  float* sT1 = sR1 + H1 * FS;      // W_root1 [H1][FS]
  float* sR2 = sT1 + H1 * FS;      // W_rel2  [64][HS]   (rows >= H2 never written: lanes >= H2 are masked)
  float* sT2 = sR2 + 64 * HS;      // W_root2 [64][HS]
  float* sVec = sT2 + 64 * HS;     // per wave: agg1 | x [2 FP], then agg2 | h1cur [2 HP]  (128-float slots)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.x * 4 + wave;
  const float* w_rel1 = params;
  const float* w_root1 = w_rel1 + H1 * F;
  const float* b1 = w_root1 + H1 * F;
  const float* w_rel2 = b1 + H1;
  const float* w_root2 = w_rel2 + (size_t)H2 * H1;
  const float* b2 = w_root2 + (size_t)H2 * H1;
  STAMP(0);
  // cur_host >= 0: the host knows the row (a chain from empty graphs: every graph holds as many nodes as the chain
  // has made steps) - no load of the count in front of everything else
  const bool on = b < B;
  const unsigned gb = (unsigned)(on ? b : 0);   // (32-bit offsets throughout: B N max(F, N) < 2^31 is checked by the host)
  const int64_t n64 = cur_host >= 0 ? (int64_t)cur_host : count[gb];
  // the weights in 16-byte pieces, every load in flight before the first LDS store: piece p = tid + 256 q of a
  // [R x C] matrix is (row p / (C / 4), columns 4 (p % (C / 4)) ..)
  constexpr int Q1 = H1 * F / 4 / 256, Q2 = 64 * H1 / 4 / 256;
  float4 w1r[Q1], w1t[Q1], w2r[Q2], w2t[Q2];
#pragma unroll
  for (int q = 0; q < Q1; ++q) {
    const int pce = tid + 256 * q;
    w1r[q] = *reinterpret_cast<const float4*>(w_rel1 + 4 * pce);
    w1t[q] = *reinterpret_cast<const float4*>(w_root1 + 4 * pce);
  }
#pragma unroll
  for (int q = 0; q < Q2; ++q) {
    const int pce = tid + 256 * q, o = pce / (H1 / 4);
    const int pc = o < H2 ? pce : 0;
    w2r[q] = *reinterpret_cast<const float4*>(w_rel2 + 4 * pc);
    w2t[q] = *reinterpret_cast<const float4*>(w_root2 + 4 * pc);
  }
  STAMP(8);
  const int fl = lane < F ? lane : F - 1, hl = lane < H1 ? lane : H1 - 1, ol = lane < H2 ? lane : H2 - 1;
  const float bias1 = b1[hl], bias2 = b2[ol];
  const float xc = obs[gb * F + fl];
  const int act1_v = gcm_vgpr(act1), act2_v = gcm_vgpr(act2);
  asm volatile("" ::: "memory");
  // ---- row cur and its selected set S (wave-uniform) -----------------------------------------------------
  const bool bad = n64 < 0 || n64 >= N;   // (>= N: the graph would roll - the host never sends such a chain here)
  const int cur = __builtin_amdgcn_readfirstlane(bad ? 0 : (int)n64);
  unsigned long long m0 = 0, m1 = 0;      // S without row cur
  bool self = false;
#pragma unroll
  for (int i = 0; i < 16; ++i) {           // (constant indices: the hop table arrives in one scalar load)
    const int h = E.hops[i], j = cur - h;
    const bool use = i < E.n_hops && h >= 0 && j >= 0;   // temporal.py:74: graphs with num_nodes >= hop
    self = self || (use && h == 0);
    const bool edge = use && h > 0;
    m0 |= (edge && j < 64) ? 1ull << (j & 63) : 0ull;
    m1 |= (edge && j >= 64) ? 1ull << ((j - 64) & 63) : 0ull;
  }
  STAMP(1);
  // ---- one round trip: the rows of S from the node matrix and from the h1 cache (the first four issued
  // together, more in further batches) ------------------------------------------------------------------------
  float xa[4], ha[4];
  unsigned long long a0 = m0, a1 = m1;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool any = (a0 | a1) != 0;
    const int j = !any ? 0 : (a0 ? __builtin_ctzll(a0) : 64 + __builtin_ctzll(a1));
    const bool low = a0 != 0;
    a0 &= low ? a0 - 1 : a0;
    a1 &= (low || !any) ? a1 : a1 - 1;
    const unsigned rj = gb * (unsigned)N + (unsigned)j;
    const float tx = nodes[rj * F + fl], th = cH[rj * H1 + hl];
    xa[q] = any ? tx : 0.f;
    ha[q] = any ? th : 0.f;
  }
  asm volatile("" ::: "memory");
  STAMP(2);
  // the weights into LDS (their loads are older than the input loads: no wait for those)
#pragma unroll
  for (int q = 0; q < Q1; ++q) {
    const int pce = tid + 256 * q, o = pce / (F / 4), c = 4 * (pce % (F / 4));
    float* d0 = sR1 + o * FS + c;
    float* d1 = sT1 + o * FS + c;
    d0[0] = w1r[q].x; d0[1] = w1r[q].y; d0[2] = w1r[q].z; d0[3] = w1r[q].w;
    d1[0] = w1t[q].x; d1[1] = w1t[q].y; d1[2] = w1t[q].z; d1[3] = w1t[q].w;
  }
#pragma unroll
  for (int q = 0; q < Q2; ++q) {
    const int pce = tid + 256 * q, o = pce / (H1 / 4), c = 4 * (pce % (H1 / 4));
    float* d0 = sR2 + o * HS + c;
    float* d1 = sT2 + o * HS + c;
    d0[0] = w2r[q].x; d0[1] = w2r[q].y; d0[2] = w2r[q].z; d0[3] = w2r[q].w;
    d1[0] = w2t[q].x; d1[1] = w2t[q].y; d1[2] = w2t[q].z; d1[3] = w2t[q].w;
  }
  float agg1 = (xa[0] + xa[1]) + (xa[2] + xa[3]), agg2 = (ha[0] + ha[1]) + (ha[2] + ha[3]);
  while (a0 | a1) {   // more than four selected rows
    const int j = a0 ? __builtin_ctzll(a0) : 64 + __builtin_ctzll(a1);
    if (a0) a0 &= a0 - 1; else a1 &= a1 - 1;
    const unsigned rj = gb * (unsigned)N + (unsigned)j;
    agg1 += nodes[rj * F + fl];
    agg2 += cH[rj * H1 + hl];
  }
  agg1 = lane < F ? agg1 + (self ? xc : 0.f) : 0.f;
  float* sv = sVec + wave * 128;
  if (lane < F) { sv[lane] = agg1; sv[F + lane] = xc; }
  asm volatile("" ::: "memory");   // (lanes exchange through LDS: the loads below are not this thread's stores - keep them apart)
  __syncthreads();   // the weights (and this wave's vector) are in LDS
  STAMP(3);
  if (!on) return;
  // ---- h1[cur][h] = act1(b1[h] + W_rel1[h,:] agg1 + W_root1[h,:] x), lane h: the row from LDS (one read per
  // element, all in flight), the vector as 16-byte broadcast reads - no cross-lane traffic --------------------------
  float rw2[H1], tw2[H1];   // layer 2's rows too: in flight under layer 1's arithmetic
  {
    const float* wr = sR2 + ol * HS;
    const float* wt = sT2 + ol * HS;
#pragma unroll
    for (int h = 0; h < H1; ++h) { rw2[h] = wr[h]; tw2[h] = wt[h]; }
  }
  float p1 = bias1;
  {
    const float* wr = sR1 + hl * FS;
    const float* wt = sT1 + hl * FS;
    float rw[F], tw[F];
#pragma unroll
    for (int f = 0; f < F; ++f) { rw[f] = wr[f]; tw[f] = wt[f]; }
    float pa = 0.f, pb = 0.f;
#pragma unroll
    for (int f4 = 0; f4 < F / 4; ++f4) {
      const float4 a = *reinterpret_cast<const float4*>(sv + 4 * f4);
      const float4 x = *reinterpret_cast<const float4*>(sv + F + 4 * f4);
      pa = fmaf(rw[4 * f4], a.x, pa); pb = fmaf(tw[4 * f4], x.x, pb);
      pa = fmaf(rw[4 * f4 + 1], a.y, pa); pb = fmaf(tw[4 * f4 + 1], x.y, pb);
      pa = fmaf(rw[4 * f4 + 2], a.z, pa); pb = fmaf(tw[4 * f4 + 2], x.z, pb);
      pa = fmaf(rw[4 * f4 + 3], a.w, pa); pb = fmaf(tw[4 * f4 + 3], x.w, pb);
    }
    p1 += pa + pb;
  }
  STAMP(4);
  const float h1c = lane < H1 ? gcm_act_sel(p1, act1_v) : 0.f;
  agg2 = lane < H1 ? agg2 + (self ? h1c : 0.f) : 0.f;
  if (lane < H1) { sv[lane] = agg2; sv[H1 + lane] = h1c; }   // (same wave wrote and read the slot: program order)
  asm volatile("" ::: "memory");
  // ---- mx[o] = act2(b2[o] + W_rel2[o,:] agg2 + W_root2[o,:] h1[cur]), lane o ---------------------------------------
  float p2 = bias2;
  {
    const float (&rw)[H1] = rw2;
    const float (&tw)[H1] = tw2;
    float pa = 0.f, pb = 0.f;
#pragma unroll
    for (int h4 = 0; h4 < H1 / 4; ++h4) {
      const float4 a = *reinterpret_cast<const float4*>(sv + 4 * h4);
      const float4 x = *reinterpret_cast<const float4*>(sv + H1 + 4 * h4);
      pa = fmaf(rw[4 * h4], a.x, pa); pb = fmaf(tw[4 * h4], x.x, pb);
      pa = fmaf(rw[4 * h4 + 1], a.y, pa); pb = fmaf(tw[4 * h4 + 1], x.y, pb);
      pa = fmaf(rw[4 * h4 + 2], a.z, pa); pb = fmaf(tw[4 * h4 + 2], x.z, pb);
      pa = fmaf(rw[4 * h4 + 3], a.w, pa); pb = fmaf(tw[4 * h4 + 3], x.w, pb);
    }
    p2 += pa + pb;
  }
  const float v = gcm_act_sel(p2, act2_v);
  STAMP(5);
  // ---- the state, the caches, the record ---------------------------------------------------------------------------
  const unsigned rc = gb * (unsigned)N + (unsigned)cur;
  if (!bad) {
    if (lane < F) {
      nodes[rc * F + lane] = xc;
      cX[rc * F + lane] = xc;
      cA[rc * F + lane] = agg1;
    }
    if (lane < H1) cH[rc * H1 + lane] = h1c;
    // row cur of the adjacency: the entries of S (and the self loop) become 1, what else is there stays
    float* arow = adj + rc * N;
    const unsigned long long s0 = m0 | ((self && cur < 64) ? 1ull << cur : 0ull);
    const unsigned long long s1 = m1 | ((self && cur >= 64) ? 1ull << (cur - 64) : 0ull);
    if (lane < N && ((s0 >> lane) & 1ull)) arow[lane] = 1.f;
    if (lane + 64 < N && ((s1 >> lane) & 1ull)) arow[lane + 64] = 1.f;
    if (lane == 0) count[b] = cur + 1;
  }
  STAMP(9);
  if (lane < H2) saved[gb * H2 + lane] = v;                         // mx: the head of the record
  if (lay.total) {
    if (lane < H1) {
      saved[lay.o_v + gb * 2 * H1 + lane] = agg2;
      saved[lay.o_v + gb * 2 * H1 + H1 + lane] = h1c;
    }
    STAMP(10);
    // live list: the rows of S and row cur, ascending; coef = adj[cur, j] (0 for row cur without a self loop)
    const unsigned long long l0 = m0 | (cur < 64 ? 1ull << cur : 0ull), l1 = m1 | (cur >= 64 ? 1ull << (cur - 64) : 0ull);
    int* live = reinterpret_cast<int*>(saved + lay.o_live) + gb * N;
    float* coef = saved + lay.o_coef + gb * N;
    const int j0 = lane, j1 = lane + 64;
    const bool in0 = (l0 >> lane) & 1ull, in1 = (l1 >> lane) & 1ull;
    const int pos0 = __popcll(l0 & ((1ull << lane) - 1ull));
    const int pos1 = __popcll(l0) + __popcll(l1 & ((1ull << lane) - 1ull));
    if (in0) { live[pos0] = j0; coef[pos0] = (j0 == cur && !self) ? 0.f : 1.f; }
    if (in1) { live[pos1] = j1; coef[pos1] = (j1 == cur && !self) ? 0.f : 1.f; }
    if (lane == 0) {
      int* hdr = reinterpret_cast<int*>(saved + lay.o_hdr) + 4 * gb;
      const int L = __popcll(l0) + __popcll(l1);
      const int l_cur = cur < 64 ? __popcll(l0 & ((1ull << cur) - 1ull)) : __popcll(l0) + __popcll(l1 & ((1ull << (cur - 64)) - 1ull));
      hdr[0] = L; hdr[1] = l_cur; hdr[2] = cur; hdr[3] = 0;
    }
  }
  STAMP(6);
  const bool nonfinite = __any(lane < H2 && !isfinite(v));
  if ((nonfinite || bad) && lane == 0)
    atomicOr(flags, (nonfinite ? GCM_FLAG_NONFINITE : 0u) | (bad ? GCM_FLAG_BAD_COUNT : 0u));
}

// ---------------------------------------------------------------------------------------------------------
// The same step with the weights read from a LANE-MAJOR image (k_cached_weight_image, made once per chain:
// image[m][k][lane] = W_m[lane][k], m = W_rel1, W_root1 (k < F), W_rel2, W_root2 (k < H1)): lane h's row of
// every matrix arrives as coalesced loads issued at kernel start together with the inputs - ONE memory round
// trip for everything, no LDS staging and no barrier.  One wave = one workgroup = one graph (256 workgroups
// spread the 64 KB of weight loads over all CUs' texture paths).
// ---------------------------------------------------------------------------------------------------------
__global__ void k_cached_weight_image(const float* __restrict__ params, float* __restrict__ image, int F, int H1,
                                      int H2) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;   // over [4][64][64]
  if (e >= 4 * 64 * 64) return;
  const int m = e >> 12, k = (e >> 6) & 63, lane = e & 63;
  const float* w_rel1 = params;
  const float* w_root1 = w_rel1 + (size_t)H1 * F;
  const float* w_rel2 = w_root1 + (size_t)H1 * F + H1;
  const float* w_root2 = w_rel2 + (size_t)H2 * H1;
  float v = 0.f;
  if (m < 2) { if (lane < H1 && k < F) v = (m == 0 ? w_rel1 : w_root1)[(size_t)lane * F + k]; }
  else { if (lane < H2 && k < H1) v = (m == 2 ? w_rel2 : w_root2)[(size_t)lane * H1 + k]; }
  image[e] = v;
  // ... and once more behind it with the two matrices of a layer interleaved, image2[layer][k][lane][rel | root]: a lane's
  // (W_rel[h][k], W_root[h][k]) as ONE 8-byte load into an adjacent register pair - what v_pk_fma_f32 wants
  // (k_step_rows_cached_img: 64 load instructions per step instead of 128, and none of the 128 register moves that
  // paired the operands of the packed products)
  image[4 * 64 * 64 + ((((m >> 1) * 64 + k) * 64 + lane) << 1) + (m & 1)] = v;
  // ... and, for widths <= 32 (where lanes 32 - 63 of the one-wave step hold no output), a third one that splits k over
  // the half-waves: image3[layer][k % 16][(k / 16) * 32 + h][rel | root] - lane h + 32 kh multiplies k = 16 kh .. 16 kh + 15
  // of output h: half the weight loads and half the dependent products per lane, one cross-half add per layer
  if (lane < 32 && k < 32)
    image[2 * 4 * 64 * 64 + ((((m >> 1) * 16 + (k & 15)) * 64 + (k >> 4) * 32 + lane) << 1) + (m & 1)] = v;
}

// SEL: the decisions of a distance selector arrive as a row (sel_row) and the selected rows beyond the first four
// are gathered eight per round trip; without it the kernel is the temporal-hops form exactly (cfg2's timed kernel).
// EX: F == FP and H1 == HP (compile-time widths: cfg2's / cfg3's timed kernels); otherwise the widths are the runtime
// Fr <= FP, H1r <= HP - the weight image is zero beyond them (k_cached_weight_image) and the operand vectors are
// written zero-padded, so the products run over FP / HP all the same.
// The layer-interleaved weight image (k_cached_weight_image: image2[layer][k][lane][rel | root], behind the lane-major one):
// lane h's (W_rel[h][k], W_root[h][k]) as one 8-byte load into an adjacent register pair, the operand vectors interleaved
// in LDS the same way ((agg, x)[k] pairs), the two products of a layer as ONE chain of packed fmas - the arithmetic of the
// two scalar chains pa += W_rel[h][k] agg[k], pb += W_root[h][k] x[k] (k ascending) and pa + pb, value for value.
template <int K>
__device__ __forceinline__ void load_pair_weights(f32x2 (&w)[K], const float* __restrict__ image, int layer, int lane) {
  const f32x2* i2 = reinterpret_cast<const f32x2*>(image + 4 * 64 * 64) + (size_t)layer * 64 * 64;
#pragma unroll
  for (int k = 0; k < K; ++k) w[k] = i2[k * 64 + lane];
}
template <int K>
__device__ __forceinline__ float pair_matvec(const f32x2 (&w)[K], const float* sv) {   // sv: (agg, x)[k], 16-byte aligned
  f32x2 acc = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < K / 2; ++j) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(sv + 4 * j);
    acc = w[2 * j] * f32x2{u[0], u[1]} + acc;
    acc = w[2 * j + 1] * f32x2{u[2], u[3]} + acc;
  }
  return acc[0] + acc[1];
}

// ... and with k split over the half-waves (image3; widths <= 32: lanes 32 - 63 hold no output): lane h + 32 kh takes
// k = 16 kh .. 16 kh + 15 of output h, the halves meet by one cross-half add
__device__ __forceinline__ void load_pair_weights_half(f32x2 (&w)[16], const float* __restrict__ image, int layer, int lane) {
  const f32x2* i3 = reinterpret_cast<const f32x2*>(image + 2 * 4 * 64 * 64) + layer * 16 * 64;
#pragma unroll
  for (int k = 0; k < 16; ++k) w[k] = i3[k * 64 + lane];
}
__device__ __forceinline__ float pair_matvec_half(const f32x2 (&w)[16], const float* sv, int kh) {
  f32x2 acc = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(sv + 32 * kh + 4 * j);
    acc = w[2 * j] * f32x2{u[0], u[1]} + acc;
    acc = w[2 * j + 1] * f32x2{u[2], u[3]} + acc;
  }
  const float ph = acc[0] + acc[1];
  return gcm_xor32_add(ph);
}

// The forward temporal hops of the step as ONE 128-bit mask, built on the host: bit (128 - h) of (rev_hi : rev_lo) for
// every hop 0 < h < 128, so that the source rows of row cur - bit j for j = cur - h >= 0 - are that mask shifted right
// by 128 - cur: half a dozen scalar instructions in the kernel.  (The first form walked the sixteen hop slots of the
// Edits argument with compares and 64-bit selects: 864 of the kernel's 1481 instructions were that loop's scalar code,
// in a kernel that is one wave issuing an instruction every four cycles - and 136 bytes of kernel arguments.)
struct HopMask {
  unsigned long long rev_lo, rev_hi;
  int hop4[4];   // n4 >= 0: the distinct hops 0 < h < 128, DESCENDING (sources ascending), 0 in the unused slots
  int n4;        // their number when there are at most four, else -1 (the kernel then walks the mask)
  int self;      // a hop of 0 (the row aggregates itself)
};
inline HopMask make_hop_mask(const gcm_fused::Edits& E) {
  HopMask m{0ull, 0ull, {0, 0, 0, 0}, 0, 0};
  int distinct[128], nd = 0;
  for (int i = 0; i < E.n_hops; ++i) {
    const int h = E.hops[i];
    if (h == 0) m.self = 1;
    if (h <= 0 || h >= 128) continue;
    const int bit = 128 - h;
    const bool seen = bit >= 64 ? ((m.rev_hi >> (bit - 64)) & 1ull) != 0 : ((m.rev_lo >> bit) & 1ull) != 0;
    if (bit >= 64) m.rev_hi |= 1ull << (bit - 64);
    else m.rev_lo |= 1ull << bit;
    if (!seen) distinct[nd++] = h;
  }
  if (nd <= 4) {
    for (int a = 0; a < nd; ++a)               // descending
      for (int c = a + 1; c < nd; ++c)
        if (distinct[c] > distinct[a]) { const int t = distinct[a]; distinct[a] = distinct[c]; distinct[c] = t; }
    for (int a = 0; a < nd; ++a) m.hop4[a] = distinct[a];
    m.n4 = nd;
  } else {
    m.n4 = -1;
  }
  return m;
}

// BK2 (round 6): a SECOND wave of the graph's workgroup does everything that does not depend on the arithmetic - the
// observation into row cur of the node matrix and of the node cache, the selector's entries in row cur of the adjacency,
// the count, the record's live list / coefficients / header - while wave 0 computes.  They never meet (no barrier, no
// LDS between them): all of it follows from `cur` (the host's: cur_host >= 0 is required) and the hop mask.  ~120 of the
// one-wave kernel's ~600 instructions leave its instruction stream - a stream that runs at about eight cycles an
// instruction when it is alone on its SIMD.
template <int FP, int HP, bool SEL, bool EX = true, bool V4 = false, bool HS = false, bool BK2 = false>   // HS: H2 <= 32 too (host)
__device__ __forceinline__ void step_rows_cached_img_body(
    const float* __restrict__ obs, float* __restrict__ nodes, float* __restrict__ adj, int64_t* __restrict__ count,
    const HopMask& hm, const float* __restrict__ params, const float* __restrict__ image, int act1, int act2,
    float* __restrict__ cH, float* __restrict__ cA, float* __restrict__ cX, float* __restrict__ saved,
    const CachedLayout& lay, uint32_t* __restrict__ flags, int B, int N, int H2, int cur_host,
    const float* __restrict__ sel_row, int Fr = FP, int H1r = HP) {
  const int F = EX ? FP : Fr, H1 = EX ? HP : H1r;
  __shared__ __attribute__((aligned(16))) float sv[128];
  const int lane = BK2 ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
  const unsigned gb = blockIdx.x;
  // the sources of row cur from the hop mask (see HopMask)
  auto hop_sources = [&](const int cur, unsigned long long& m0, unsigned long long& m1) {
    const int sft = 128 - cur;   // (cur in 0 .. 127: 1 .. 128)
    if (sft >= 64) {
      m1 = 0ull;
      m0 = sft >= 128 ? 0ull : hm.rev_hi >> (sft - 64);
    } else {
      m0 = (hm.rev_lo >> sft) | (hm.rev_hi << (64 - sft));
      m1 = hm.rev_hi >> sft;
    }
  };
  // what follows from cur and the masks alone: the donated state's entries and the record's live list
  auto bookkeeping = [&](const int cur, const bool bad, const unsigned long long m0, const unsigned long long m1,
                         const float xc) {
    const bool self = hm.self != 0;
    const unsigned rc = gb * (unsigned)N + (unsigned)cur;
    if (!bad) {
      if (lane < F) {
        nodes[rc * F + lane] = xc;
        cX[rc * F + lane] = xc;
      }
      float* arow = adj + (size_t)rc * N;
      const unsigned long long s0 = m0 | ((self && cur < 64) ? 1ull << cur : 0ull);
      const unsigned long long s1 = m1 | ((self && cur >= 64) ? 1ull << (cur - 64) : 0ull);
      if (lane < N && ((s0 >> lane) & 1ull)) arow[lane] = 1.f;
      if (lane + 64 < N && ((s1 >> lane) & 1ull)) arow[lane + 64] = 1.f;
      if (lane == 0) count[gb] = cur + 1;
    }
    if (lay.total) {
      const unsigned long long l0 = m0 | (cur < 64 ? 1ull << cur : 0ull), l1 = m1 | (cur >= 64 ? 1ull << (cur - 64) : 0ull);
      int* live = reinterpret_cast<int*>(saved + lay.o_live) + gb * N;
      float* coef = saved + lay.o_coef + gb * N;
      const int j0 = lane, j1 = lane + 64;
      const bool in0 = (l0 >> lane) & 1ull, in1 = (l1 >> lane) & 1ull;
      const int pos0 = __popcll(l0 & ((1ull << lane) - 1ull));
      const int pos1 = __popcll(l0) + __popcll(l1 & ((1ull << lane) - 1ull));
      if (in0) { live[pos0] = j0; coef[pos0] = (j0 == cur && !self) ? 0.f : 1.f; }
      if (in1) { live[pos1] = j1; coef[pos1] = (j1 == cur && !self) ? 0.f : 1.f; }
      if (lane == 0) {
        int* hdr = reinterpret_cast<int*>(saved + lay.o_hdr) + 4 * gb;
        const int L = __popcll(l0) + __popcll(l1);
        const int l_cur = cur < 64 ? __popcll(l0 & ((1ull << cur) - 1ull)) : __popcll(l0) + __popcll(l1 & ((1ull << (cur - 64)) - 1ull));
        hdr[0] = L; hdr[1] = l_cur; hdr[2] = cur; hdr[3] = 0;
      }
    }
  };
  if (BK2 && threadIdx.x >= 64) {   // the bookkeeping wave (cur_host >= 0: the launcher's condition; no SEL form)
    const float xo_ = obs[gb * F + (lane < F ? lane : F - 1)];
    const bool bad_ = cur_host < 0 || cur_host >= N;
    const int cur_ = bad_ ? 0 : cur_host;
    unsigned long long b0, b1_;
    hop_sources(cur_, b0, b1_);
    bookkeeping(cur_, bad_, b0, b1_, xo_);
    if (bad_ && lane == 0) atomicOr(flags, GCM_FLAG_BAD_COUNT);
    return;
  }
  const float* b1 = params + 2 * H1 * F;
  const float* b2 = b1 + H1 + 2 * (size_t)H2 * H1;
  const int64_t n64 = cur_host >= 0 ? (int64_t)cur_host : count[gb];
  // decisions of a distance selector that ran ahead of this kernel (gcm_edge_distance_pre: 1 / 0 per row j < cur,
  // entries beyond unspecified; distance.py:31-37 writes row cur alone - the cached argument holds)
  float sel0 = 0.f, sel1 = 0.f;
  if (SEL) {
    sel0 = sel_row[gb * (unsigned)N + (unsigned)(lane < N ? lane : N - 1)];
    sel1 = sel_row[gb * (unsigned)N + (unsigned)(lane + 64 < N ? lane + 64 : N - 1)];
  }
  // every weight load in flight at once (coalesced: lane h reads element h of row k of the image)
  float r1[FP], t1[FP], r2[HP], t2[HP];
  f32x2 w1[V4 ? FP : 1], w2[V4 ? HP : 1];   // V4 (the interleaved image): (rel, root) pairs of layer 1 / 2
  constexpr bool HALF = V4 && EX && FP == 32 && HP == 32 && HS;   // k split over the half-waves (image3): 16 pairs a lane and layer
  const int kh = lane >> 5;
  if (HALF) {
    const f32x2* i3 = reinterpret_cast<const f32x2*>(image + 2 * 4 * 64 * 64);
#pragma unroll
    for (int k = 0; k < 16; ++k) { w1[k] = i3[k * 64 + lane]; w2[k] = i3[(16 + k) * 64 + lane]; }
  } else if (V4) {
    const f32x2* i2 = reinterpret_cast<const f32x2*>(image + 4 * 64 * 64);
#pragma unroll
    for (int k = 0; k < FP; ++k) w1[k] = i2[k * 64 + lane];
#pragma unroll
    for (int k = 0; k < HP; ++k) w2[k] = i2[(64 + k) * 64 + lane];
  } else {
#pragma unroll
    for (int k = 0; k < FP; ++k) { r1[k] = image[k * 64 + lane]; t1[k] = image[4096 + k * 64 + lane]; }
#pragma unroll
    for (int k = 0; k < HP; ++k) { r2[k] = image[2 * 4096 + k * 64 + lane]; t2[k] = image[3 * 4096 + k * 64 + lane]; }
  }
  const int fl = lane < F ? lane : F - 1, hl = lane < H1 ? lane : H1 - 1, ol = lane < H2 ? lane : H2 - 1;
  const float bias1 = b1[hl], bias2 = b2[ol];
  const float xc = obs[gb * F + fl];
  const int act1_v = gcm_vgpr(act1), act2_v = gcm_vgpr(act2);
  asm volatile("" ::: "memory");
  const bool bad = n64 < 0 || n64 >= N;
  const int cur = __builtin_amdgcn_readfirstlane(bad ? 0 : (int)n64);
  unsigned long long m0, m1;
  const bool self = hm.self != 0;
  hop_sources(cur, m0, m1);
  if (SEL) {
    m0 |= __ballot(lane < cur && lane < N && sel0 != 0.f);
    m1 |= __ballot(lane + 64 < cur && lane + 64 < N && sel1 != 0.f);
  }
  float xa[4], ha[4];
  unsigned long long a0 = m0, a1 = m1;
  float agg1, agg2;
  if (V4 && !SEL && hm.n4 >= 0) {
    // at most four hops (the temporal form's usual case): the source rows are cur - hop, straight from the hop slots -
    // no bit scans.  Hops descending = sources ascending; the slots WITH a source are a contiguous run (a larger hop
    // runs out of rows first, unused slots are at the end), so the sum below is the mask walk's (v0 + v1) + (v2 + v3)
    // over the compacted list in every case but one: the first slot without a source and the other three with one.
    bool lead = false;   // slot 0 is a hop without a source (cur < hop)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int h = hm.hop4[q], j = cur - h;
      const bool any = h > 0 && j >= 0;
      if (q == 0) lead = h > 0 && j < 0;
      const unsigned rj = gb * (unsigned)N + (unsigned)(any ? j : 0);
      const float tx = nodes[rj * F + fl], th = cH[rj * H1 + hl];
      xa[q] = any ? tx : 0.f;
      ha[q] = any ? th : 0.f;
    }
    const float n1 = (xa[0] + xa[1]) + (xa[2] + xa[3]), n2 = (ha[0] + ha[1]) + (ha[2] + ha[3]);
    const float c1 = (xa[1] + xa[2]) + xa[3], c2 = (ha[1] + ha[2]) + ha[3];
    agg1 = lead ? c1 : n1;
    agg2 = lead ? c2 : n2;
    a0 = 0ull;
    a1 = 0ull;
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool any = (a0 | a1) != 0;
      const int j = !any ? 0 : (a0 ? __builtin_ctzll(a0) : 64 + __builtin_ctzll(a1));
      const bool low = a0 != 0;
      a0 &= low ? a0 - 1 : a0;
      a1 &= (low || !any) ? a1 : a1 - 1;
      const unsigned rj = gb * (unsigned)N + (unsigned)j;
      const float tx = nodes[rj * F + fl], th = cH[rj * H1 + hl];
      xa[q] = any ? tx : 0.f;
      ha[q] = any ? th : 0.f;
    }
    agg1 = (xa[0] + xa[1]) + (xa[2] + xa[3]);
    agg2 = (ha[0] + ha[1]) + (ha[2] + ha[3]);
  }
  // further rows (a distance selector's clusters; more than four hops): eight per round trip, added in ascending
  // order - the sums a row-at-a-time loop makes
  while (!SEL && (a0 | a1)) {
    const int j = a0 ? __builtin_ctzll(a0) : 64 + __builtin_ctzll(a1);
    if (a0) a0 &= a0 - 1; else a1 &= a1 - 1;
    const unsigned rj = gb * (unsigned)N + (unsigned)j;
    agg1 += nodes[rj * F + fl];
    agg2 += cH[rj * H1 + hl];
  }
  while (SEL && (a0 | a1)) {
    float bx[8], bh[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bool any = (a0 | a1) != 0;
      const int j = !any ? 0 : (a0 ? __builtin_ctzll(a0) : 64 + __builtin_ctzll(a1));
      const bool low = a0 != 0;
      a0 &= low ? a0 - 1 : a0;
      a1 &= (low || !any) ? a1 : a1 - 1;
      const unsigned rj = gb * (unsigned)N + (unsigned)j;
      const float tx = nodes[rj * F + fl], th = cH[rj * H1 + hl];
      bx[q] = any ? tx : 0.f;
      bh[q] = any ? th : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) { agg1 += bx[q]; agg2 += bh[q]; }
  }
  agg1 = lane < F ? agg1 + (self ? xc : 0.f) : 0.f;
  if (V4) {
    if (lane < FP) *reinterpret_cast<f32x2*>(sv + 2 * lane) = f32x2{agg1, (EX || lane < F) ? xc : 0.f};
    // (the stores are 2-vectors, the loads below 4-vectors: to the compiler's type-based alias analysis they do not
    //  alias, and the half-wave form's loads of layer 2 came out of layer 1's registers - a compiler barrier per store)
    asm volatile("" ::: "memory");
  } else if (lane < FP) { sv[lane] = agg1; sv[FP + lane] = (EX || lane < F) ? xc : 0.f; }
  asm volatile("" ::: "memory");   // (lanes exchange through LDS: a compiler barrier per exchange)
  // (one wave: its LDS operations execute in order - the broadcast reads below see these writes)
  float p1 = bias1;
  if (HALF) {   // this half-wave's sixteen k of both chains, then the other half's
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(sv + 32 * kh + 4 * j);
      acc = w1[2 * j] * f32x2{u[0], u[1]} + acc;
      acc = w1[2 * j + 1] * f32x2{u[2], u[3]} + acc;
    }
    const float ph = acc[0] + acc[1];
    p1 += gcm_xor32_add(ph);
  } else if (V4) {   // (pa, pb) as one packed accumulator: the same two chains, value for value
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < FP / 2; ++j) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(sv + 4 * j);   // (agg1, x)[2 j], (agg1, x)[2 j + 1]
      acc = w1[2 * j] * f32x2{u[0], u[1]} + acc;
      acc = w1[2 * j + 1] * f32x2{u[2], u[3]} + acc;
    }
    p1 += acc[0] + acc[1];
  } else {
    float pa = 0.f, pb = 0.f;
#pragma unroll
    for (int f4 = 0; f4 < FP / 4; ++f4) {
      const float4 a = *reinterpret_cast<const float4*>(sv + 4 * f4);
      const float4 x = *reinterpret_cast<const float4*>(sv + FP + 4 * f4);
      pa = fmaf(r1[4 * f4], a.x, pa); pb = fmaf(t1[4 * f4], x.x, pb);
      pa = fmaf(r1[4 * f4 + 1], a.y, pa); pb = fmaf(t1[4 * f4 + 1], x.y, pb);
      pa = fmaf(r1[4 * f4 + 2], a.z, pa); pb = fmaf(t1[4 * f4 + 2], x.z, pb);
      pa = fmaf(r1[4 * f4 + 3], a.w, pa); pb = fmaf(t1[4 * f4 + 3], x.w, pb);
    }
    p1 += pa + pb;
  }
  const float h1c = lane < H1 ? gcm_act_sel(p1, act1_v) : 0.f;
  agg2 = lane < H1 ? agg2 + (self ? h1c : 0.f) : 0.f;
  if (V4) {
    if (lane < HP) *reinterpret_cast<f32x2*>(sv + 2 * lane) = f32x2{agg2, h1c};
    asm volatile("" ::: "memory");
  } else if (lane < HP) { sv[lane] = agg2; sv[HP + lane] = h1c; }
  asm volatile("" ::: "memory");
  float p2 = bias2;
  if (HALF) {
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(sv + 32 * kh + 4 * j);
      acc = w2[2 * j] * f32x2{u[0], u[1]} + acc;
      acc = w2[2 * j + 1] * f32x2{u[2], u[3]} + acc;
    }
    const float ph = acc[0] + acc[1];
    p2 += gcm_xor32_add(ph);
  } else if (V4) {
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < HP / 2; ++j) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(sv + 4 * j);
      acc = w2[2 * j] * f32x2{u[0], u[1]} + acc;
      acc = w2[2 * j + 1] * f32x2{u[2], u[3]} + acc;
    }
    p2 += acc[0] + acc[1];
  } else {
    float pa = 0.f, pb = 0.f;
#pragma unroll
    for (int h4 = 0; h4 < HP / 4; ++h4) {
      const float4 a = *reinterpret_cast<const float4*>(sv + 4 * h4);
      const float4 x = *reinterpret_cast<const float4*>(sv + HP + 4 * h4);
      pa = fmaf(r2[4 * h4], a.x, pa); pb = fmaf(t2[4 * h4], x.x, pb);
      pa = fmaf(r2[4 * h4 + 1], a.y, pa); pb = fmaf(t2[4 * h4 + 1], x.y, pb);
      pa = fmaf(r2[4 * h4 + 2], a.z, pa); pb = fmaf(t2[4 * h4 + 2], x.z, pb);
      pa = fmaf(r2[4 * h4 + 3], a.w, pa); pb = fmaf(t2[4 * h4 + 3], x.w, pb);
    }
    p2 += pa + pb;
  }
  const float v = gcm_act_sel(p2, act2_v);
  const unsigned rc = gb * (unsigned)N + (unsigned)cur;
  if (!bad) {
    if (lane < F) cA[rc * F + lane] = agg1;
    if (lane < H1) cH[rc * H1 + lane] = h1c;
  }
  if (lane < H2) saved[gb * H2 + lane] = v;
  if (lay.total && lane < H1) {
    saved[lay.o_v + gb * 2 * H1 + lane] = agg2;
    saved[lay.o_v + gb * 2 * H1 + H1 + lane] = h1c;
  }
  if (!BK2) bookkeeping(cur, bad, m0, m1, xc);   // (BK2: the second wave's)
  const bool nonfinite = __any(lane < H2 && !isfinite(v));
  if ((nonfinite || (bad && !BK2)) && lane == 0)
    atomicOr(flags, (nonfinite ? GCM_FLAG_NONFINITE : 0u) | ((bad && !BK2) ? GCM_FLAG_BAD_COUNT : 0u));
}

// The two kernels over that body.  (Separate signatures on purpose: with the decision row's pointer as a ninth
// pointer argument of the temporal form too, the same instructions ran 170 ns per launch slower - the kernel-argument
// segment is fetched differently - which was 3 % of cfg2's timed region.)
template <int FP, int HP>
__global__ __launch_bounds__(64) void k_step_rows_cached_img(
    const float* __restrict__ obs, float* __restrict__ nodes, float* __restrict__ adj, int64_t* __restrict__ count,
    HopMask E, const float* __restrict__ params, const float* __restrict__ image, int act1, int act2,
    float* __restrict__ cH, float* __restrict__ cA, float* __restrict__ cX, float* __restrict__ saved,
    CachedLayout lay, uint32_t* __restrict__ flags, int B, int N, int H2, int cur_host) {
  step_rows_cached_img_body<FP, HP, false>(obs, nodes, adj, count, E, params, image, act1, act2, cH, cA, cX, saved, lay,
                                           flags, B, N, H2, cur_host, nullptr);
}
template <int FP, int HP, bool HS = false>   // ... the weights from the layer-interleaved image (image2 / image3), packed products
__global__ __launch_bounds__(64) void k_step_rows_cached_img4(
    const float* __restrict__ obs, float* __restrict__ nodes, float* __restrict__ adj, int64_t* __restrict__ count,
    HopMask E, const float* __restrict__ params, const float* __restrict__ image, int act1, int act2,
    float* __restrict__ cH, float* __restrict__ cA, float* __restrict__ cX, float* __restrict__ saved,
    CachedLayout lay, uint32_t* __restrict__ flags, int B, int N, int H2, int cur_host) {
  step_rows_cached_img_body<FP, HP, false, true, true, HS>(obs, nodes, adj, count, E, params, image, act1, act2, cH, cA,
                                                           cX, saved, lay, flags, B, N, H2, cur_host, nullptr);
}
// ... with the bookkeeping on a second wave (BK2 above): F = H1 = 32, H2 <= 32, cur_host >= 0
__global__ __launch_bounds__(128) void k_step_rows_cached_img4b(
    const float* __restrict__ obs, float* __restrict__ nodes, float* __restrict__ adj, int64_t* __restrict__ count,
    HopMask E, const float* __restrict__ params, const float* __restrict__ image, int act1, int act2,
    float* __restrict__ cH, float* __restrict__ cA, float* __restrict__ cX, float* __restrict__ saved,
    CachedLayout lay, uint32_t* __restrict__ flags, int B, int N, int H2, int cur_host) {
  step_rows_cached_img_body<32, 32, false, true, true, true, true>(obs, nodes, adj, count, E, params, image, act1, act2,
                                                                   cH, cA, cX, saved, lay, flags, B, N, H2, cur_host,
                                                                   nullptr);
}
template <int FP, int HP>
__global__ __launch_bounds__(64) void k_step_rows_cached_sel(
    const float* __restrict__ obs, float* __restrict__ nodes, float* __restrict__ adj, int64_t* __restrict__ count,
    HopMask E, const float* __restrict__ params, const float* __restrict__ image, int act1, int act2,
    float* __restrict__ cH, float* __restrict__ cA, float* __restrict__ cX, float* __restrict__ saved,
    CachedLayout lay, uint32_t* __restrict__ flags, int B, int N, int H2, int cur_host,
    const float* __restrict__ sel_row) {
  step_rows_cached_img_body<FP, HP, true>(obs, nodes, adj, count, E, params, image, act1, act2, cH, cA, cX, saved, lay,
                                          flags, B, N, H2, cur_host, sel_row);
}
// any widths F <= FP, H1 <= HP (round 4: the observation / hidden sizes that are not 32 or 64 - BASELINE cfg1's
// F = 8 among them - took the general live-row kernel before)
template <int FP, int HP, bool SEL>
__global__ __launch_bounds__(64) void k_step_rows_cached_gen(
    const float* __restrict__ obs, float* __restrict__ nodes, float* __restrict__ adj, int64_t* __restrict__ count,
    HopMask E, const float* __restrict__ params, const float* __restrict__ image, int act1, int act2,
    float* __restrict__ cH, float* __restrict__ cA, float* __restrict__ cX, float* __restrict__ saved,
    CachedLayout lay, uint32_t* __restrict__ flags, int B, int N, int H2, int cur_host,
    const float* __restrict__ sel_row, int F, int H1) {
  step_rows_cached_img_body<FP, HP, SEL, false>(obs, nodes, adj, count, E, params, image, act1, act2, cH, cA, cX,
                                                saved, lay, flags, B, N, H2, cur_host, sel_row, F, H1);
}

// ---------------------------------------------------------------------------------------------------------
// The cached step in the STEADY STATE of such a chain (round 4): t_abs >= N steps made, every graph is full and
// every step drops the oldest node (gcm.py:263-271, 323-355: zero row / column 0, roll nodes by -1 and the
// adjacency by (-1, -1), insert at row N - 1).  For a chain whose selectors are forward temporal hops only
// (temporal.py:72-88) and that started from empty graphs:
//   * the adjacency is the band adj[i, i - h] = 1 (i >= h) on every full graph, and the roll maps that band onto
//     itself (new[i, i - h] = old[i + 1, i + 1 - h]; the entries of column 0 fall off, row N - 1 is zeroed and
//     gets the same hops back from the selector): the step leaves the adjacency and the count UNTOUCHED - the
//     reference rewrites all N^2 entries per graph and step to arrive at the same values;
//   * the caches become rings: node t lives in slot t mod N, the rows the new node aggregates from are the nodes
//     t_abs - h, whose layer-1 rows are still final when N > 2 max(hop): a node's own sources are at most
//     max(hop) older, and only nodes older than t_abs - N have been dropped;
//   * the node matrix does roll (nodes[i] <- nodes[i + 1], the observation into row N - 1): a second wave of the
//     graph's workgroup moves it in place - every load of the graph before its first store, one wave, in order.
// Wave 0 is the step proper (the body of k_step_rows_cached_img in ring coordinates, its node rows from the cX
// ring instead of the matrix the other wave is moving).  A ring slot is overwritten N steps later, so the
// record of such a step is the GENERAL live-row record (rows_common.h: SavedLayout - the rows h1 | agg1 | x of the
// live rows travel in the record, ascending, row cur last), which gcm_dense_rows_bptt reads like any other.
// hops: sorted descending, distinct, 0 < h < N (host: gcm_dense_rows_step_cached_roll); self: a hop of 0.
// ---------------------------------------------------------------------------------------------------------
template <int FP, int HP, bool HS = false>   // HS: F = H1 = 32 and H2 <= 32 (host): the half-wave form of the products
__global__ __launch_bounds__(192) void k_step_rows_cached_roll(
    const float* __restrict__ obs, float* __restrict__ nodes, gcm_fused::Edits E, int self_i,
    const float* __restrict__ params, const float* __restrict__ image, int act1, int act2, float* __restrict__ cH,
    float* __restrict__ cA, float* __restrict__ cX, float* __restrict__ saved, SavedLayout lay, int record,
    uint32_t* __restrict__ flags, int B, int N, int H2, int slot_new) {
  // slot_new = t_abs mod N (host): the ring slot of the new node; node t_abs - h sits h slots behind it
  constexpr int F = FP, H1 = HP;
  __shared__ __attribute__((aligned(16))) float sv[128];
  const int lane = threadIdx.x & 63;
  const unsigned gb = blockIdx.x;
  if (threadIdx.x >= 64 && threadIdx.x < 128) {
    // ---- wave 1: the node matrix, in place: row r <- row r + 1, row N - 1 <- the observation -------------------
    constexpr int PER = 128 * FP / 4 / 64;                 // float4 per lane at N = 128
    const int F4 = F / 4, total = N * F4, moved = total - F4;
    // (an array of HIP's float4 STRUCT of this length is not promoted to registers by hipcc 7.2 - it went through
    //  scratch memory; the compiler's own vector type is)
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f* g4 = reinterpret_cast<v4f*>(nodes + (size_t)gb * N * F);
    const v4f* o4 = reinterpret_cast<const v4f*>(obs + (size_t)gb * F);
    v4f v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = lane + 64 * i;
      const int src = idx < moved ? idx + F4 : total - 1;   // (clamped: an unconditional load)
      v[i] = g4[src];
    }
    const v4f ov = o4[lane < F4 ? lane : F4 - 1];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // source and destination alias: every load has landed
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = lane + 64 * i;
      if (idx < moved) g4[idx] = v[i];
    }
    if (lane < F4) g4[moved + lane] = ov;
    if (gb == 0 && lane == 0) atomicOr(flags, GCM_FLAG_WRAPPED);   // gcm.py:264-266: the caller's one-time notice
    return;
  }
  if (threadIdx.x >= 128) {
    // ---- wave 2 (round 6): what does not depend on the step's arithmetic - the selected rows' h1 | agg1 | x from the
    // rings into the record (they travel in it: a ring slot is overwritten N steps later), its coefficients and header,
    // the observation into the node ring.  ~60 instructions and a third of the loads off wave 0's stream.
    const int F_ = FP, H1_ = HP;
    const int fl2 = lane < F_ ? lane : F_ - 1, hl2 = lane < H1_ ? lane : H1_ - 1;
    const float xo = obs[gb * F_ + fl2];
    const unsigned rc2 = gb * (unsigned)N + (unsigned)slot_new;
    if (lane < F_) cX[rc2 * F_ + lane] = xo;
    if (record) {
      const int nh = E.n_hops;
      float* rows2 = saved + lay.o_rows + (size_t)gb * N * lay.rw;
      for (int q = 0; q < nh; ++q) {
        int hq = 0;   // (a run-time index into the kernel-argument array would move it to scratch)
#pragma unroll
        for (int i = 0; i < 16; ++i) hq = q == i ? E.hops[i] : hq;
        const int sl = slot_new - hq + (slot_new < hq ? N : 0);
        const unsigned rj = gb * (unsigned)N + (unsigned)sl;
        const float tx = cX[rj * F_ + fl2], th = cH[rj * H1_ + hl2], ta = cA[rj * F_ + fl2];
        float* row = rows2 + (size_t)q * lay.rw;
        if (lane < H1_) row[lane] = th;
        if (lane < F_) { row[H1_ + lane] = ta; row[H1_ + F_ + lane] = tx; }
      }
      float* coef = saved + lay.o_coef + (size_t)gb * N;
      if (lane <= nh) coef[lane] = (lane < nh || self_i != 0) ? 1.f : 0.f;
      if (lane == 0) {
        int* hdr = reinterpret_cast<int*>(saved + lay.o_hdr) + 4 * gb;
        hdr[0] = nh + 1; hdr[1] = nh; hdr[2] = N - 1; hdr[3] = 1;
      }
    }
    return;
  }
  // ---- wave 0: the step on row cur = N - 1, ring coordinates -----------------------------------------------------
  const float* b1 = params + 2 * H1 * F;
  const float* b2 = b1 + H1 + 2 * (size_t)H2 * H1;
  constexpr bool HALF = HS && FP == 32 && HP == 32;
  const int kh = lane >> 5;
  f32x2 w1[HALF ? 16 : F], w2[HALF ? 16 : H1];   // (rel, root) pairs of layer 1 / 2: the interleaved image
  if constexpr (HALF) {
    load_pair_weights_half(w1, image, 0, lane);
    load_pair_weights_half(w2, image, 1, lane);
  } else {
    load_pair_weights(w1, image, 0, lane);
    load_pair_weights(w2, image, 1, lane);
  }
  const int fl = lane < F ? lane : F - 1, hl = lane < H1 ? lane : H1 - 1, ol = lane < H2 ? lane : H2 - 1;
  const float bias1 = b1[hl], bias2 = b2[ol];
  const float xc = obs[gb * F + fl];
  const int act1_v = gcm_vgpr(act1), act2_v = gcm_vgpr(act2);
  const int n_h = E.n_hops;
  const bool self = self_i != 0;
  // the selected rows, ascending in node age (hops descending): the first four in one round trip with everything
  // above, further ones four at a time
  float xa[4], ha[4];
  int slot[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool on = q < n_h;
    const int h = on ? E.hops[q] : 0;
    slot[q] = slot_new - h + (slot_new < h ? N : 0);        // (0 < h < N)
    const unsigned rj = gb * (unsigned)N + (unsigned)slot[q];
    const float tx = cX[rj * F + fl], th = cH[rj * H1 + hl];
    xa[q] = on ? tx : 0.f;
    ha[q] = on ? th : 0.f;
  }
  asm volatile("" ::: "memory");
  float* rows = saved + lay.o_rows + (size_t)gb * N * lay.rw;
  const bool rec = record != 0;
  float agg1 = (xa[0] + xa[1]) + (xa[2] + xa[3]), agg2 = (ha[0] + ha[1]) + (ha[2] + ha[3]);
  for (int q = 4; q < n_h; ++q) {   // (their record rows: wave 2)
    int hq = 0;   // (a run-time index into the kernel-argument array would move it to scratch)
#pragma unroll
    for (int i = 4; i < 16; ++i) hq = q == i ? E.hops[i] : hq;
    const int sl = slot_new - hq + (slot_new < hq ? N : 0);
    const unsigned rj = gb * (unsigned)N + (unsigned)sl;
    agg1 += cX[rj * F + fl];
    agg2 += cH[rj * H1 + hl];
  }
  agg1 = lane < F ? agg1 + (self ? xc : 0.f) : 0.f;
  if (lane < F) *reinterpret_cast<f32x2*>(sv + 2 * lane) = f32x2{agg1, xc};
  asm volatile("" ::: "memory");   // (2-vector stores, 4-vector loads: no alias to the compiler's type-based analysis)
  float p1 = bias1;
  if constexpr (HALF) p1 += pair_matvec_half(w1, sv, kh);
  else p1 += pair_matvec(w1, sv);
  const float h1c = lane < H1 ? gcm_act_sel(p1, act1_v) : 0.f;
  agg2 = lane < H1 ? agg2 + (self ? h1c : 0.f) : 0.f;
  if (lane < H1) *reinterpret_cast<f32x2*>(sv + 2 * lane) = f32x2{agg2, h1c};
  asm volatile("" ::: "memory");
  float p2 = bias2;
  if constexpr (HALF) p2 += pair_matvec_half(w2, sv, kh);
  else p2 += pair_matvec(w2, sv);
  const float v = gcm_act_sel(p2, act2_v);
  const unsigned rc = gb * (unsigned)N + (unsigned)slot_new;      // the new node's ring slot (node t_abs - N leaves)
  if (lane < F) cA[rc * F + lane] = agg1;                           // (cX[rc]: wave 2)
  if (lane < H1) cH[rc * H1 + lane] = h1c;
  if (lane < H2) saved[gb * H2 + lane] = v;                         // mx: the head of the record
  if (rec) {
    if (lane < H1) {
      saved[lay.o_v + gb * 2 * H1 + lane] = agg2;
      saved[lay.o_v + gb * 2 * H1 + H1 + lane] = h1c;
    }
    float* row = rows + (size_t)n_h * lay.rw;                       // row cur: last in the list (the others: wave 2)
    if (lane < H1) row[lane] = h1c;
    if (lane < F) { row[H1 + lane] = agg1; row[H1 + F + lane] = xc; }
  }
  const bool nonfinite = __any(lane < H2 && !isfinite(v));
  if (nonfinite && lane == 0) atomicOr(flags, GCM_FLAG_NONFINITE);
}

// ---------------------------------------------------------------------------------------------------------
// SparseGCM, stepwise use (sparse_gcm.py:72-212 called with one new node per graph: x [B, 1, F], taus in {0, 1}) with a
// TemporalEdge selector (sparse_edge_selectors/temporal.py:18-63), in a chain that started from empty graphs: the
// same argument - edges only ever point from a new node to older ones, so the layer-1 row of a node is final once
// written - and the same step: the new node's rows of both layers from the chain's caches, instead of flattening
// the whole batch, building CSR / CSC views of it and running both GraphConv layers over every stored node.
// The node's row index is T[b] (read: the graphs of a SparseGCM batch need not move in lockstep); taus[b] = 0:
// no node this step (zero output row, empty record).
// ---------------------------------------------------------------------------------------------------------
template <int FP, int HP>
__global__ __launch_bounds__(64) void k_sparse_step_cached(
    const float* __restrict__ x, const int64_t* __restrict__ T, const int64_t* __restrict__ taus, gcm_fused::Edits E,
    const float* __restrict__ params, const float* __restrict__ image, int act1, int act2, float* __restrict__ cH,
    float* __restrict__ cA, float* __restrict__ cX, float* __restrict__ mx, float* __restrict__ saved,
    CachedLayout lay, uint32_t* __restrict__ flags, int B, int N, int H2) {
  constexpr int F = FP, H1 = HP;
  __shared__ __attribute__((aligned(16))) float sv[128];
  const int lane = threadIdx.x;
  const unsigned gb = blockIdx.x;
  const float* b1 = params + 2 * H1 * F;
  const float* b2 = b1 + H1 + 2 * (size_t)H2 * H1;
  const int64_t n64 = T[gb], tau = taus[gb];
  f32x2 w1[F], w2[H1];   // (rel, root) pairs of layer 1 / 2: the interleaved image
  load_pair_weights(w1, image, 0, lane);
  load_pair_weights(w2, image, 1, lane);
  const int fl = lane < F ? lane : F - 1, hl = lane < H1 ? lane : H1 - 1, ol = lane < H2 ? lane : H2 - 1;
  const float bias1 = b1[hl], bias2 = b2[ol];
  const float xc = x[gb * F + fl];
  const int act1_v = gcm_vgpr(act1), act2_v = gcm_vgpr(act2);
  asm volatile("" ::: "memory");
  const bool on = tau > 0;
  const bool bad = on && (n64 < 0 || n64 >= N);
  const bool rec = on && !bad;
  const int cur = __builtin_amdgcn_readfirstlane(rec ? (int)n64 : 0);
  // the selected rows: cur - hop for every (distinct, host-sorted descending) hop with a source (temporal.py:
  // t - h >= 0) - straight from the hop table (any graph size; no row masks), every load issued together
  float xa[16], ha[16];
  unsigned valid = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int h = E.hops[i], j = cur - h;
    const bool ok = rec && i < E.n_hops && h > 0 && j >= 0;
    valid |= ok ? 1u << i : 0u;
    const unsigned rj = gb * (unsigned)N + (unsigned)(ok ? j : 0);
    const float tx = cX[rj * F + fl], th = cH[rj * H1 + hl];
    xa[i] = ok ? tx : 0.f;
    ha[i] = ok ? th : 0.f;
  }
  float agg1 = 0.f, agg2 = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { agg1 += xa[i]; agg2 += ha[i]; }   // (hops descending: sources ascending)
  agg1 = lane < F ? agg1 : 0.f;
  if (lane < F) *reinterpret_cast<f32x2*>(sv + 2 * lane) = f32x2{agg1, xc};
  asm volatile("" ::: "memory");   // (2-vector stores, 4-vector loads: no alias to the compiler's type-based analysis)
  const float p1 = bias1 + pair_matvec(w1, sv);
  const float h1c = lane < H1 ? gcm_act_sel(p1, act1_v) : 0.f;
  agg2 = lane < H1 ? agg2 : 0.f;
  if (lane < H1) *reinterpret_cast<f32x2*>(sv + 2 * lane) = f32x2{agg2, h1c};
  asm volatile("" ::: "memory");
  const float p2 = bias2 + pair_matvec(w2, sv);
  const float v = rec ? gcm_act_sel(p2, act2_v) : 0.f;   // (no node: the padded output row is zero)
  const unsigned rc = gb * (unsigned)N + (unsigned)cur;
  if (rec) {
    if (lane < F) { cX[rc * F + lane] = xc; cA[rc * F + lane] = agg1; }
    if (lane < H1) cH[rc * H1 + lane] = h1c;
  }
  if (lane < H2) {
    mx[gb * H2 + lane] = v;
    saved[gb * H2 + lane] = v;
  }
  if (lay.total) {
    if (lane < H1) {
      saved[lay.o_v + gb * 2 * H1 + lane] = rec ? agg2 : 0.f;
      saved[lay.o_v + gb * 2 * H1 + H1 + lane] = rec ? h1c : 0.f;
    }
    // live list: the selected rows (ascending) and row cur behind them; coef = 1 for an edge, 0 for row cur
    int* live = reinterpret_cast<int*>(saved + lay.o_live) + (size_t)gb * N;
    float* coef = saved + lay.o_coef + (size_t)gb * N;
    const int Ls = __popc(valid);
    int myh = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) myh = lane == i ? E.hops[i] : myh;
    if (lane < 16 && ((valid >> lane) & 1u)) {
      const int pos = __popc(valid & ((1u << lane) - 1u));
      live[pos] = cur - myh;
      coef[pos] = 1.f;
    }
    if (lane == 16 && rec) { live[Ls] = cur; coef[Ls] = 0.f; }
    if (lane == 0) {
      int* hdr = reinterpret_cast<int*>(saved + lay.o_hdr) + 4 * gb;
      hdr[0] = rec ? Ls + 1 : 0; hdr[1] = Ls; hdr[2] = cur; hdr[3] = 0;
    }
  }
  const bool nonfinite = __any(lane < H2 && !isfinite(v));
  if ((nonfinite || bad) && lane == 0)
    atomicOr(flags, (nonfinite ? GCM_FLAG_NONFINITE : 0u) | (bad ? GCM_FLAG_SPARSE_OVERFLOW : 0u));
}

}  // namespace gcm_rows

static int cached_supported(const gcm_selector_desc* selectors, int n_selectors, int has_bias, int N, int F, int H1,
                            int H2, bool allow_distance) {
  if (!gcm_dense_rows_supported(N, F, H1, H2) || F > 64 || H1 > 64 || H2 > 64 || N > 128) return 0;
  if (has_bias & (GCM_GNN_HAS_DEG_TERM | GCM_GNN_HAS_PE_TABLE | GCM_GNN_RECORD_DX)) return 0;
  int hops = 0, dist = 0;
  for (int i = 0; i < n_selectors; ++i) {
    const gcm_selector_desc& d = selectors[i];
    // only row cur may be written: DenseEdge (dense.py:16-21) also writes column cur, "backward" / "both" hops
    // write into older rows - their layer-1 rows change at every step.  A distance selector (distance.py:31-37)
    // that is not bidirectional writes row cur alone: its decisions arrive as a row (gcm_edge_distance_pre).
    if (d.kind == GCM_SEL_DISTANCE && allow_distance && !d.bidirectional && dist == 0) { ++dist; continue; }
    if (d.kind != GCM_SEL_TEMPORAL || d.direction != GCM_DIR_FORWARD) return 0;
    hops += d.n_hops;
  }
  return hops <= 16;
}

extern "C" int gcm_dense_rows_cached_supported(const gcm_selector_desc* selectors, int n_selectors, int has_bias,
                                               int N, int F, int H1, int H2) {
  return cached_supported(selectors, n_selectors, has_bias, N, F, H1, H2, false);
}

extern "C" int gcm_dense_rows_cached_supported_ws(const gcm_selector_desc* selectors, int n_selectors, int has_bias,
                                                  int N, int F, int H1, int H2) {
  return cached_supported(selectors, n_selectors, has_bias, N, F, H1, H2, true);
}

extern "C" int gcm_dense_rows_cached_layout(int B, int N, int F, int H1, int H2, size_t* out5) {
  GCM_REQUIRE(out5 && B > 0 && N > 0);
  const gcm_rows::CachedLayout l = gcm_rows::make_cached_layout(B, N, H1, H2);
  out5[0] = l.total; out5[1] = l.o_v; out5[2] = l.o_hdr; out5[3] = l.o_coef; out5[4] = l.o_live;
  (void)F;
  return GCM_OK;
}

/* floats a weight image takes (gcm_dense_rows_cached_weight_image writes all of it): the lane-major layout, the
 * layer-interleaved one and the half-wave split of the widths <= 32 */
extern "C" size_t gcm_dense_rows_cached_weight_image_floats(void) { return 2 * 4 * 64 * 64 + 2 * 16 * 64 * 2; }

extern "C" int gcm_dense_rows_cached_weight_image(const float* params, float* image, int F, int H1, int H2,
                                                  gcm_stream_t stream) {
  GCM_REQUIRE(params && image && F > 0 && F <= 64 && H1 > 0 && H1 <= 64 && H2 > 0 && H2 <= 64);
  hipLaunchKernelGGL(gcm_rows::k_cached_weight_image, dim3(4 * 64 * 64 / 256), dim3(256), 0, (hipStream_t)stream, params,
                     image, F, H1, H2);
  return gcm_launch_status();
}

// EuclideanEdge alone: the distance kernel and the cached step are ONE launch when the shapes allow
// (gcm_edge_distance_step_cached), unless the caller asks for the two-launch form (GCM_STEP_TWO_LAUNCH in has_bias:
// the A/B of tests and tools - a per-call argument, the library keeps no switch)
static bool one_launch_distance(const gcm_selector_desc* selectors, int n_selectors, int has_bias, int B, int N, int F,
                                int H1, int H2) {
  if (n_selectors != 1 || selectors[0].kind != GCM_SEL_DISTANCE || selectors[0].mode != GCM_DIST_EUCLID_CROSSBATCH ||
      (has_bias & GCM_STEP_TWO_LAUNCH))
    return false;
  return gcm_edge_distance_step_cached_supported(selectors[0].cur_rows ? selectors[0].n_cur_rows : B, B, N, F, H1, H2) != 0;
}

extern "C" int gcm_dense_rows_cached_launches(const gcm_selector_desc* selectors, int n_selectors, int has_bias, int B,
                                              int N, int F, int H1, int H2) {
  if (B <= 0 || !gcm_dense_rows_cached_supported_ws(selectors, n_selectors, has_bias, N, F, H1, H2)) return 0;
  if (one_launch_distance(selectors, n_selectors, has_bias, B, N, F, H1, H2)) return 1;
  int n = 1;
  for (int i = 0; i < n_selectors; ++i)
    if (selectors[i].kind == GCM_SEL_DISTANCE) ++n;
  return n;
}

extern "C" int gcm_dense_rows_step_cached(const float* obs, float* nodes, float* adj, int64_t* count,
                                          const gcm_selector_desc* selectors, int n_selectors, const float* params,
                                          const float* weight_image,
                                          int has_bias, int act1, int act2, float* cache_h1, float* cache_agg1,
                                          float* cache_nodes, float* saved, int record, int cur_host, uint32_t* flags,
                                          int B, int N, int F, int H1, int H2, gcm_stream_t stream) {
  if (!gcm_dense_rows_cached_supported(selectors, n_selectors, has_bias, N, F, H1, H2)) return GCM_EUNSUPPORTED;
  return gcm_dense_rows_step_cached_ws(obs, nodes, adj, count, selectors, n_selectors, params, weight_image, has_bias,
                                       act1, act2, cache_h1, cache_agg1, cache_nodes, saved, record, cur_host, flags,
                                       nullptr, 0, B, N, F, H1, H2, stream);
}

extern "C" int gcm_dense_rows_step_cached_ws(const float* obs, float* nodes, float* adj, int64_t* count,
                                             const gcm_selector_desc* selectors, int n_selectors,
                                             const float* params, const float* weight_image, int has_bias, int act1,
                                             int act2, float* cache_h1, float* cache_agg1, float* cache_nodes,
                                             float* saved, int record, int cur_host, uint32_t* flags, void* workspace,
                                             size_t workspace_bytes, int B, int N, int F, int H1, int H2,
                                             gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes && adj && count && params && cache_h1 && cache_agg1 && cache_nodes && saved && flags);
  GCM_REQUIRE(B > 0 && (selectors || n_selectors == 0));
  if (!gcm_dense_rows_cached_supported_ws(selectors, n_selectors, has_bias, N, F, H1, H2)) return GCM_EUNSUPPORTED;
  if ((size_t)B * N * (size_t)(F > N ? F : N) >= ((size_t)1 << 31)) return GCM_EUNSUPPORTED;   // 32-bit offsets inside
  gcm_fused::Edits E{};
  const float* sel_row = nullptr;
  // EuclideanEdge alone: the distance kernel and the step as ONE launch when the shapes allow
  if (weight_image && one_launch_distance(selectors, n_selectors, has_bias, B, N, F, H1, H2)) {
    const gcm_selector_desc& d = selectors[0];
    const gcm_rows::CachedLayout l = gcm_rows::make_cached_layout(B, N, H1, H2);
    const size_t lay5[5] = {l.total, l.o_v, l.o_hdr, l.o_coef, l.o_live};
    const int rc = gcm_edge_distance_step_cached(obs, nodes, adj, count, d.max_distance, d.dist_param, d.cur_rows,
                                                 d.n_cur_rows, params, weight_image, act1, act2, cache_h1, cache_agg1,
                                                 cache_nodes, saved, lay5, record, cur_host, nullptr, flags, B, N, F, H1,
                                                 H2, stream);
    if (rc != GCM_EUNSUPPORTED) return rc;
  }
  for (int i = 0; i < n_selectors; ++i) {
    const gcm_selector_desc& d = selectors[i];
    if (d.kind == GCM_SEL_DISTANCE) {
      // the distance selector runs first, on the state as it comes in, and hands its row over (as for
      // gcm_dense_rows_step_fwd_ws); the decision row is the head of the workspace
      GCM_REQUIRE(workspace && weight_image);
      if (workspace_bytes < gcm_dense_rows_step_workspace_bytes(selectors, n_selectors, B, N, F)) return GCM_EWORKSPACE;
      float* row = (float*)workspace;
      const int rc = gcm_edge_distance_pre_ex(nodes, count, obs, row, d.mode, d.max_distance, d.dist_param, d.a0, d.a1,
                                              d.b0, d.b1, d.cur_rows, d.n_cur_rows, row + (size_t)B * N,
                                              workspace_bytes - sizeof(float) * (size_t)B * N, B, N, F, stream);
      if (rc) return rc;
      sel_row = row;
      continue;
    }
    for (int k = 0; k < d.n_hops; ++k) {
      E.hops[E.n_hops] = d.hops[k];
      E.dir[E.n_hops++] = d.direction;
    }
  }
  gcm_rows::CachedLayout lay = gcm_rows::make_cached_layout(B, N, H1, H2);
  if (!record) lay.total = 0;
  const gcm_rows::HopMask HM = gcm_rows::make_hop_mask(E);
  if (weight_image) {
#define GCM_RI(a, b_)                                                                                            \
  if (F == a && H1 == b_) {                                                                                      \
    if (sel_row)                                                                                                 \
      hipLaunchKernelGGL((gcm_rows::k_step_rows_cached_sel<a, b_>), dim3(B), dim3(64), 0, (hipStream_t)stream,       \
                         obs, nodes, adj, count, HM, params, weight_image, act1, act2, cache_h1, cache_agg1,          \
                         cache_nodes, saved, lay, flags, B, N, H2, cur_host, sel_row);                               \
    else if ((has_bias & GCM_STEP_IMG_V4) && a == 32 && b_ == 32 && H2 <= 32 && cur_host >= 0 &&                  \
             !(has_bias & GCM_STEP_ONE_WAVE))                                                                     \
      hipLaunchKernelGGL(gcm_rows::k_step_rows_cached_img4b, dim3(B), dim3(128), 0, (hipStream_t)stream, obs, nodes,  \
                         adj, count, HM, params, weight_image, act1, act2, cache_h1, cache_agg1, cache_nodes, saved,  \
                         lay, flags, B, N, H2, cur_host);                                                             \
    else if ((has_bias & GCM_STEP_IMG_V4) && a == 32 && b_ == 32 && H2 <= 32)                                     \
      hipLaunchKernelGGL((gcm_rows::k_step_rows_cached_img4<a, b_, true>), dim3(B), dim3(64), 0,                     \
                         (hipStream_t)stream, obs, nodes, adj, count, HM, params, weight_image, act1, act2,           \
                         cache_h1, cache_agg1, cache_nodes, saved, lay, flags, B, N, H2, cur_host);                   \
    else if (has_bias & GCM_STEP_IMG_V4)                                                                          \
      hipLaunchKernelGGL((gcm_rows::k_step_rows_cached_img4<a, b_>), dim3(B), dim3(64), 0, (hipStream_t)stream,      \
                         obs, nodes, adj, count, HM, params, weight_image, act1, act2, cache_h1, cache_agg1,          \
                         cache_nodes, saved, lay, flags, B, N, H2, cur_host);                                        \
    else                                                                                                         \
      hipLaunchKernelGGL((gcm_rows::k_step_rows_cached_img<a, b_>), dim3(B), dim3(64), 0, (hipStream_t)stream,       \
                         obs, nodes, adj, count, HM, params, weight_image, act1, act2, cache_h1, cache_agg1,          \
                         cache_nodes, saved, lay, flags, B, N, H2, cur_host);                                        \
    return gcm_launch_status();                                                                                  \
  }
    GCM_RI(32, 32) GCM_RI(64, 32) GCM_RI(32, 64) GCM_RI(64, 64)
#undef GCM_RI
    // other widths: the padded form
    const int fp = F <= 32 ? 32 : 64, hp = H1 <= 32 ? 32 : 64;
#define GCM_RG(a, b_)                                                                                            \
  if (fp == a && hp == b_) {                                                                                     \
    if (sel_row)                                                                                                 \
      hipLaunchKernelGGL((gcm_rows::k_step_rows_cached_gen<a, b_, true>), dim3(B), dim3(64), 0, (hipStream_t)stream, \
                         obs, nodes, adj, count, HM, params, weight_image, act1, act2, cache_h1, cache_agg1,          \
                         cache_nodes, saved, lay, flags, B, N, H2, cur_host, sel_row, F, H1);                        \
    else                                                                                                         \
      hipLaunchKernelGGL((gcm_rows::k_step_rows_cached_gen<a, b_, false>), dim3(B), dim3(64), 0,                    \
                         (hipStream_t)stream, obs, nodes, adj, count, HM, params, weight_image, act1, act2, cache_h1, \
                         cache_agg1, cache_nodes, saved, lay, flags, B, N, H2, cur_host, (const float*)nullptr, F,   \
                         H1);                                                                                        \
    return gcm_launch_status();                                                                                  \
  }
    GCM_RG(32, 32) GCM_RG(64, 32) GCM_RG(32, 64) GCM_RG(64, 64)
#undef GCM_RG
    return GCM_EUNSUPPORTED;
  }
  const size_t lds = sizeof(float) * (2 * (size_t)H1 * (F + 1) + 2 * (size_t)64 * (H1 + 1) + 4 * 128);
#define GCM_RC(a, b_)                                                                                           \
  if (F == a && H1 == b_) {                                                                                     \
    auto kern = gcm_rows::k_step_rows_cached<a, b_>;                                                            \
    gcm_allow_dynamic_lds((const void*)kern, lds);                                                              \
    hipLaunchKernelGGL(kern, dim3((B + 3) / 4), dim3(256), lds, (hipStream_t)stream, obs, nodes, adj, count, E, \
                       params, act1, act2, cache_h1, cache_agg1, cache_nodes, saved, lay, flags, B, N, H2,      \
                       cur_host);                                                                              \
    return gcm_launch_status();                                                                                 \
  }
  GCM_RC(32, 32) GCM_RC(64, 32) GCM_RC(32, 64) GCM_RC(64, 64)
#undef GCM_RC
  return GCM_EUNSUPPORTED;
}

// hops of forward temporal selectors as the steady-state step wants them: distinct, descending, 0 < h < N; -> count,
// -1 when the configuration has no steady-state cached form
static int roll_hops(const gcm_selector_desc* selectors, int n_selectors, int has_bias, int N, int F, int H1, int H2,
                     int* hops16, int* self) {
  if (!cached_supported(selectors, n_selectors, has_bias, N, F, H1, H2, false)) return -1;
  if ((F != 32 && F != 64) || (H1 != 32 && H1 != 64)) return -1;   // (k_step_rows_cached_roll is specialised on them)
  int n = 0, mx = 0;
  *self = 0;
  for (int i = 0; i < n_selectors; ++i)
    for (int k = 0; k < selectors[i].n_hops; ++k) {
      const int h = selectors[i].hops[k];
      if (h < 0 || h > N - 1) continue;   // (temporal.py:74: no source for this hop in a graph of N nodes)
      if (h == 0) { *self = 1; continue; }
      bool seen = false;
      for (int q = 0; q < n; ++q) seen = seen || hops16[q] == h;
      if (!seen) hops16[n++] = h;
      mx = h > mx ? h : mx;
    }
  if (N <= 2 * mx) return -1;             // the rows the new node aggregates from must not have lost a source
  std::sort(hops16, hops16 + n, [](int a, int b) { return a > b; });
  return n;
}

extern "C" int gcm_dense_rows_cached_roll_supported(const gcm_selector_desc* selectors, int n_selectors, int has_bias,
                                                    int N, int F, int H1, int H2) {
  int hops[16], self;
  return roll_hops(selectors, n_selectors, has_bias, N, F, H1, H2, hops, &self) >= 0 ? 1 : 0;
}

extern "C" int gcm_dense_rows_step_cached_roll(const float* obs, float* nodes, const gcm_selector_desc* selectors,
                                               int n_selectors, const float* params, const float* weight_image,
                                               int has_bias, int act1, int act2, float* cache_h1, float* cache_agg1,
                                               float* cache_nodes, float* saved, int record, int t_abs, uint32_t* flags,
                                               int B, int N, int F, int H1, int H2, gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes && params && weight_image && cache_h1 && cache_agg1 && cache_nodes && saved && flags);
  GCM_REQUIRE(B > 0 && (selectors || n_selectors == 0) && t_abs >= N);
  if ((size_t)B * N * (size_t)(F > N ? F : N) >= ((size_t)1 << 31)) return GCM_EUNSUPPORTED;   // 32-bit offsets inside
  gcm_fused::Edits E{};
  int hops[16], self = 0;
  const int n = roll_hops(selectors, n_selectors, has_bias, N, F, H1, H2, hops, &self);
  if (n < 0) return GCM_EUNSUPPORTED;
  for (int i = 0; i < n; ++i) {
    E.hops[i] = hops[i];
    E.dir[i] = GCM_DIR_FORWARD;
  }
  E.n_hops = n;
  const gcm_rows::SavedLayout lay = gcm_rows::make_layout(B, N, F, H1, H2);
#define GCM_RR(a, b_)                                                                                             \
  if (F == a && H1 == b_) {                                                                                       \
    if (a == 32 && b_ == 32 && H2 <= 32)                                                                          \
      hipLaunchKernelGGL((gcm_rows::k_step_rows_cached_roll<a, b_, true>), dim3(B), dim3(192), 0,                     \
                         (hipStream_t)stream, obs, nodes, E, self, params, weight_image, act1, act2, cache_h1,        \
                         cache_agg1, cache_nodes, saved, lay, record, flags, B, N, H2, t_abs % N);                    \
    else                                                                                                          \
      hipLaunchKernelGGL((gcm_rows::k_step_rows_cached_roll<a, b_>), dim3(B), dim3(192), 0, (hipStream_t)stream,      \
                         obs, nodes, E, self, params, weight_image, act1, act2, cache_h1, cache_agg1, cache_nodes,    \
                         saved, lay, record, flags, B, N, H2, t_abs % N);                                             \
    return gcm_launch_status();                                                                                   \
  }
  GCM_RR(32, 32) GCM_RR(64, 32) GCM_RR(32, 64) GCM_RR(64, 64)
#undef GCM_RR
  return GCM_EUNSUPPORTED;
}

#ifdef GCM_DEBUG_ABI   // libgcm_hip_debug.so only (include/gcm_hip_debug.h)
/* Measurement aid (bench.py), the cached twin of gcm_debug_time_rows_rollout: T cached steps (T <= N) of a rollout
 * from empty graphs enqueued back to back from C, each launch bracketed by the caller's HIP events recorded by the
 * dispatch itself (hipExtLaunchKernelGGL start / stop events = the kernel begin / end timestamps rocprofv3
 * --kernel-trace reports).  Temporal selectors with forward hops only; F, H1 in {32, 64}; weight_image as above. */
extern "C" int gcm_debug_time_cached_rollout(const float* obs_all, float* nodes, float* adj, int64_t* count,
                                             const gcm_selector_desc* selectors, int n_selectors,
                                             const float* params, const float* weight_image, int has_bias, int act1,
                                             int act2, float* cache_h1, float* cache_agg1, float* cache_nodes,
                                             float* const* saved_per_step, uint32_t* flags, void* const* start_events,
                                             void* const* stop_events, int T, int B, int N, int F, int H1, int H2,
                                             gcm_stream_t stream) {
  GCM_REQUIRE(obs_all && nodes && adj && count && params && weight_image && cache_h1 && cache_agg1 && cache_nodes &&
              saved_per_step && start_events && stop_events && flags && T > 0 && T <= N);
  if (!gcm_dense_rows_cached_supported(selectors, n_selectors, has_bias, N, F, H1, H2)) return GCM_EUNSUPPORTED;
  gcm_fused::Edits E{};
  for (int i = 0; i < n_selectors; ++i)
    for (int k = 0; k < selectors[i].n_hops; ++k) {
      E.hops[E.n_hops] = selectors[i].hops[k];
      E.dir[E.n_hops++] = selectors[i].direction;
    }
  const gcm_rows::CachedLayout lay = gcm_rows::make_cached_layout(B, N, H1, H2);
  const gcm_rows::HopMask HM = gcm_rows::make_hop_mask(E);
  for (int t = 0; t < T; ++t) {
#define GCM_RT(a, b_)                                                                                           \
  if (F == a && H1 == b_)                                                                                       \
    hipExtLaunchKernelGGL((gcm_rows::k_step_rows_cached_img<a, b_>), dim3(B), dim3(64), 0, (hipStream_t)stream,  \
                          (hipEvent_t)start_events[t], (hipEvent_t)stop_events[t], 0,                            \
                          obs_all + (size_t)t * B * F, nodes, adj, count, HM, params, weight_image, act1, act2,  \
                          cache_h1, cache_agg1, cache_nodes, saved_per_step[t], lay, flags, B, N, H2, t);
    GCM_RT(32, 32) GCM_RT(64, 32) GCM_RT(32, 64) GCM_RT(64, 64)
#undef GCM_RT
    const int rc = gcm_launch_status();
    if (rc) return rc;
  }
  return GCM_OK;
}
#endif   // GCM_DEBUG_ABI

/* SparseGCM stepwise (x [B, 1, F], taus in {0, 1}) in a chain from empty graphs, TemporalEdge selector with hops >= 1
 * (HOST array): the new node's belief from the chain's caches (see k_sparse_step_cached).  T: node counts BEFORE the
 * step.  mx [B, H2]; saved: the record (gcm_dense_rows_cached_layout; written in full with record != 0, else mx only).
 * The state itself (node matrix, COO adjacency, T) is advanced by the caller with the usual entry points. */
extern "C" int gcm_sparse_step_cached(const float* x, const int64_t* T, const int64_t* taus, const int32_t* hops_host,
                                      int n_hops, const float* params, const float* weight_image, int act1, int act2,
                                      float* cache_h1, float* cache_agg1, float* cache_nodes, float* mx, float* saved,
                                      int record, uint32_t* flags, int B, int N, int F, int H1, int H2,
                                      gcm_stream_t stream) {
  GCM_REQUIRE(x && T && taus && hops_host && params && weight_image && cache_h1 && cache_agg1 && cache_nodes && mx &&
              saved && flags && B > 0 && N > 0 && n_hops >= 0);
  if (n_hops > 16 || H2 > 64 || H2 <= 0 || (F != 32 && F != 64) || (H1 != 32 && H1 != 64))
    return GCM_EUNSUPPORTED;
  if ((size_t)B * N * 64 >= ((size_t)1 << 31)) return GCM_EUNSUPPORTED;
  gcm_fused::Edits E{};
  for (int i = 0; i < n_hops; ++i) {
    if (hops_host[i] < 1) return GCM_EUNSUPPORTED;   // (a hop of 0 would be a self loop: "Causality violated")
    E.hops[E.n_hops] = hops_host[i];
    E.dir[E.n_hops++] = GCM_DIR_FORWARD;
  }
  gcm_rows::CachedLayout lay = gcm_rows::make_cached_layout(B, N, H1, H2);
  if (!record) lay.total = 0;
#define GCM_SC(a, b_)                                                                                         \
  if (F == a && H1 == b_) {                                                                                   \
    hipLaunchKernelGGL((gcm_rows::k_sparse_step_cached<a, b_>), dim3(B), dim3(64), 0, (hipStream_t)stream, x, T, \
                       taus, E, params, weight_image, act1, act2, cache_h1, cache_agg1, cache_nodes, mx, saved,   \
                       lay, flags, B, N, H2);                                                                     \
    return gcm_launch_status();                                                                               \
  }
  GCM_SC(32, 32) GCM_SC(64, 32) GCM_SC(32, 64) GCM_SC(64, 64)
#undef GCM_SC
  return GCM_EUNSUPPORTED;
}
