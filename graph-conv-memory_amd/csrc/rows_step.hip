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
#include <hip/hip_ext.h>

#include <algorithm>

#include "fused_common.h"
#include "rows_common.h"
#include "state_copy.h"
#include "rows_state_waves.h"

#ifdef GCM_STAMPS   // diagnostic build only (make stamps5, tools/kstamp_rows.py)
__device__ unsigned long long g_stamps[32];
extern "C" int gcm_debug_read_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
#endif

namespace gcm_rows {

using gcm_fused::Edits;
using gcm_fused::Gnn2;
using gcm_state::load_copy;
using gcm_state::store_copy;

template <int FP, int HP, int H2P>
struct Lds {
  static constexpr int XS = FP + 16;   // x image [128][XS]: B operand of the aggregation (stride = 16 mod 32)
  static constexpr int RS = 130;       // live adjacency rows [16][RS]: A operand (stride = 2 mod 32)
  static constexpr int AS = FP + 4;    // agg1 rows [16][AS], read 16 bytes at a time
  static constexpr int HS = HP + 1;    // h1 rows [16][HS]
  static constexpr int X = 128 * XS, ROWS = 16 * RS, AGG = 16 * AS, H1R = 16 * HS;
  // rowcur [128] | coef [128] | live j [128] | v [2 HP] | candidates [4][32] |
  // agg2 partials (double) [4 waves][4][16] | ints [16] | row sums of the group's live rows [16]
  static constexpr int MISC = 3 * 128 + 2 * HP + 128 + 512 + 16 + 16;
  static constexpr int TOTAL = X + ROWS + AGG + H1R + MISC;
};

__device__ __forceinline__ float f4_at(const float4& v, int i) {   // i: compile time after unrolling
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

// The live-row list.  Slot 0 is row cur; then the rows the folded temporal selectors connect it to
// (cur - hop for forward / both hops, in hop order, duplicates dropped): these CANDIDATES depend on
// cur alone, so their adjacency rows are fetched before row cur of the adjacency has even arrived.
// Rows that are live for any other reason (DenseEdge; entries a caller's own state holds in row
// cur) are appended in ascending order once the row is known, and fetched then.  The order of the
// list is the summation order of layer 2 and of the backward.
//
// The kernel runs at one wave per SIMD, so it is bound by instruction issue and by the chain of
// dependent latencies, not by bytes:
//   * everything that does not depend on the count (x rows, the weights - straight into the
//     registers that feed the MFMAs, no LDS staging -, biases, the functional copy's loads) is in
//     flight before the count has arrived; then row cur and the candidate rows;
//   * the linears use a K order in which every lane group owns CONTIGUOUS k (16-byte LDS reads of
//     the A operand, 16-byte global loads of the B operand);
//   * four workgroup barriers on the common path; the selector's HBM entries, the inserted node
//     and (functional state) the copy's stores go last, off the critical path.
// NX: graph size as a compile-time constant (0: run time); EXACT: F, H1, H2 are the padded sizes.
// The specialisations fold the clamps, masks and most of the address arithmetic away - a third of
// the instructions of a kernel that is bound by instruction issue.
template <int FP, int HP, int H2P, bool FUNC, int NX, bool EXACT>
__global__ __launch_bounds__(256) void k_step_rows(
    const float* __restrict__ obs, const float* nodes_in, const float* adj_in,
    const int64_t* count_in, float* nodes_out, float* adj_out, int64_t* count_out,
    int64_t* __restrict__ cur_out, Edits E, Gnn2 P, float* __restrict__ mx_out,
    float* __restrict__ saved, SavedLayout lay, uint32_t* __restrict__ flags, int N_, int F_,
    int H1_, int H2_, const float* __restrict__ c1, const float* __restrict__ pe,
    const float* __restrict__ sel_row, int Bn) {
  using L = Lds<FP, HP, H2P>;
  const int N = NX ? NX : N_, F = EXACT ? FP : F_, H1 = EXACT ? HP : H1_, H2 = EXACT ? H2P : H2_;
  constexpr int XS = L::XS, RS = L::RS, AS = L::AS, HS = L::HS;
  constexpr int HB = HP / 16, FB = FP / 16;
  constexpr int KL = FP / 2;                          // k values per lane group in the linears
  constexpr int KG = 256 / H2P, KC = 2 * HP / KG;     // layer 2: KG adjacent lanes share an output
  const int b = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m16 = lane & 15, kq = lane >> 4;
  const int N4 = N >> 2, F4 = F >> 2;
  const int o2 = tid / KG, kg = tid % KG;

  extern __shared__ float smem[];
  float* sX = smem;
  float* sRows = sX + L::X;
  float* sAgg = sRows + L::ROWS;
  float* sH1 = sAgg + L::AGG;
  float* sRowCur = sH1 + L::H1R;
  float* sCoef = sRowCur + 128;
  int* sLive = reinterpret_cast<int*>(sCoef + 128);
  float* sV = reinterpret_cast<float*>(sLive + 128);
  int* sCand = reinterpret_cast<int*>(sV + 2 * HP);   // [4 waves][32]: each wave's own copy
  double* sA2 = reinterpret_cast<double*>(sCand + 128);   // [4 waves][4 kq][16]
  int* sInt = reinterpret_cast<int*>(sA2 + 256);   // [0..1] extra live rows of wave 0 / 1, [3] K-chunk mask
  float* sDeg = reinterpret_cast<float*>(sInt + 16);
  // a folded preprocessor / positional encoding (generic shapes only): see gcm_dense_rows_step_fwd
  const bool fold_deg = !EXACT && c1 != nullptr, fold_pe = !EXACT && pe != nullptr;
  // decisions of a distance selector that ran ahead of this kernel (gcm_edge_distance_pre): 1 / 0 per
  // image row j < cur (entries beyond are unspecified)
  const bool has_sel = sel_row != nullptr;

  const float* ng_in = nodes_in + (size_t)b * N * F;
  const float* ag_in = adj_in + (size_t)b * N * N;
  float* ng = nodes_out + (size_t)b * N * F;
  float* ag = adj_out + (size_t)b * N * N;

  if (FUNC && blockIdx.x >= Bn) {   // blocks B ..: the functional state advance (GCM_STATE_WGS workgroups per graph)
#if defined(GCM_EXP) && GCM_EXP == 1
    return;
#endif
    const int q = blockIdx.x - Bn, bs = q % Bn, ks = q / Bn;
    advance_state_waves<FP>(obs, nodes_in + (size_t)bs * N * F, adj_in + (size_t)bs * N * N, count_in,
                            nodes_out + (size_t)bs * N * F, adj_out + (size_t)bs * N * N, count_out, cur_out, E,
                            flags, sel_row, bs, ks, tid, N, F);
    return;
  }
#if defined(GCM_EXP) && GCM_EXP == 2
  if (FUNC) return;
#endif

  STAMP(0);
  // ---- loads that do not depend on the count ----------------------------------------------------
  // first in the queue (loads return in order): the folded hops of this lane, then the count
  const int n_hops = E.n_hops;
  int lane_h = -1, cdir = 0;
  if (lane >= 1 && lane <= n_hops) {
    lane_h = E.hops[(lane - 1) & 15];
    cdir = E.dir[(lane - 1) & 15];
  }
  const int64_t n_in = count_in[b];
  // x rows (unshifted; the overflow shift and the observation are applied when the image is written)
  constexpr int XROWS = 256 / (FP / 4), PER = 128 / XROWS;   // rows per pass, float4 per thread
  const int xc = (tid % (FP / 4)) * 4, xr0 = tid / (FP / 4);
  const int xcc = xc < F ? xc : F - 4;
  float4 xv[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int r = xr0 + i * XROWS;
    xv[i] = *reinterpret_cast<const float4*>(ng_in + (r < N ? r : N - 1) * F + xcc);
  }
  const float4 obv = *reinterpret_cast<const float4*>(obs + (size_t)b * F + xcc);
  // layer-1 weights, B operand of the linears: lane (h = 16 (wave % HB) + m16, kq) owns k in
  // [kq KL, (kq + 1) KL) of [W_rel1 | W_root1][h] - contiguous in memory
  const int hcol = 16 * (wave % HB) + m16;
  float4 w1r[KL / 4];
  {
    const float* src = (kq < 2 ? P.w_rel1 : P.w_root1) + (hcol < H1 ? hcol : H1 - 1) * F;
#pragma unroll
    for (int q = 0; q < KL / 4; ++q) {
      const int c = (kq & 1) * KL + 4 * q;
      w1r[q] = *reinterpret_cast<const float4*>(src + (c < F ? c : F - 4));
    }
  }
  // layer-2 weights: thread (o2, kg) owns k in [kg KC, (kg + 1) KC) of [W_rel2 | W_root2][o2]
  float w2r[KC];
  {
    const bool root = kg * KC >= HP;
    const int kk0 = root ? kg * KC - HP : kg * KC;
    const float* src = (root ? P.w_root2 : P.w_rel2) + (o2 < H2 ? o2 : H2 - 1) * H1;
#pragma unroll
    for (int k = 0; k < KC; ++k) w2r[k] = src[kk0 + k < H1 ? kk0 + k : H1 - 1];
  }
  // biases (the packed vector always holds the slots; zeros when a layer has no bias)
  const float bias1 = P.b_rel1[hcol < H1 ? hcol : H1 - 1];
  const float bias2 = P.b_rel2[o2 < H2 ? o2 : H2 - 1];
  const float c1v = fold_deg ? c1[hcol < H1 ? hcol : H1 - 1] : 0.f;
  const float srv = has_sel ? sel_row[(size_t)b * N + min(tid & 127, N - 1)] : 0.f;
  constexpr int ADJ_PER = 16, NODE_PER = (128 * FP / 4 + 255) / 256;   // (the in-place roll below)
  // kernel arguments used late: in registers now (a scalar load at its point of use is a round trip)
  const int act1_v = gcm_vgpr(P.act1), act2_v = gcm_vgpr(P.act2);
  const size_t lay_v = lay.o_v, lay_hdr = lay.o_hdr, lay_coef = lay.o_coef, lay_rows = lay.o_rows;
  const size_t lay_ar = saved ? lay.o_arows : 0, lay_lv = lay.o_live;
  const int rw = lay.rw;
  const int dense_i = E.dense;
  asm volatile("" ::"s"(lay_v), "s"(lay_rows), "s"(saved), "s"(mx_out), "s"(dense_i), "s"(N), "s"(count_out),
               "s"(flags));
  asm volatile("" ::: "memory");   // compiler barrier: the loads above stay above, unpredicated

  const bool wrap = n_in + 1 > N;
  const int64_t c64 = wrap ? n_in - 1 : n_in;
  const int cur = c64 < 0 ? 0 : (c64 > N - 1 ? N - 1 : (int)c64);
  const int sh = wrap ? 1 : 0;
  STAMP(1);
  // ---- overflow (gcm.py:323-355): roll the state, in place when it is donated (every load lands
  // before the first store).  Rare; its own loads, stores and barriers. ----------------------------
  if (wrap && !FUNC) {
    float4 ra[ADJ_PER], rn[NODE_PER];
    load_copy<ADJ_PER, NODE_PER, true>(ra, rn, ag_in, ng_in, tid, N, N4, F, F4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // source and destination alias
    __syncthreads();
    store_copy<ADJ_PER, NODE_PER>(ra, rn, ag, ng, tid, N, N4, F4, true);
    __syncthreads();   // the rolled state is visible to the whole workgroup
  }
  // Functional state and the graph rolls: the rolled adjacency is never materialised for this
  // workgroup - rolled[j][c] = in[j + 1][c + 1], last row and column zero - it reads the old state
  // through shifted (dword-aligned 16-byte) addresses; same instructions either way, the shift is data.
  const bool shifted = FUNC && wrap;
  const int rsh = shifted ? 1 : 0;
  const float* ag_rd = (wrap && !FUNC) ? ag : ag_in;   // the advanced adjacency, before the selectors
  auto adj_row4 = [&](int j, int c) -> float4 {   // columns c .. c+3 of row j (c clamped by the caller)
    const bool tail = shifted && c + 4 >= N;      // in[.][N] does not exist: shifted in registers
    const int jr = j + rsh < N ? j + rsh : N - 1;
    float4 v;
    __builtin_memcpy(&v, ag_rd + jr * N + c + ((shifted && !tail) ? 1 : 0), sizeof(float4));
    if (tail) v = make_float4(v.y, v.z, v.w, 0.f);
    if (shifted && j + 1 >= N) v = make_float4(0.f, 0.f, 0.f, 0.f);
    return v;
  };

  // ---- row cur of the advanced adjacency, before the selectors (all zero for a state this code
  // produced; a caller's own state may hold anything): heads the longest dependent chain ------------
  float rc_val = ag_rd[cur * N + min(tid & 127, N - 1)];
  if (shifted) rc_val = 0.f;   // (row N - 1 of a rolled adjacency)
  // ---- candidates (every wave computes them for itself: no barrier) ------------------------------
  // lane 0: row cur; lane i in [1, n_hops]: cur - hop[i-1] when that hop writes into row cur
  const bool dense = dense_i != 0;
  int cj = cur;
  bool cvalid = lane == 0;
  if (lane >= 1 && lane <= n_hops) {
    cj = cur - lane_h;
    cvalid = ((cdir & GCM_DIR_FORWARD) || lane_h == 0) && lane_h >= 0 && lane_h <= cur;
  }
  // row that gets a column-cur entry from a backward / both hop, per hop lane
  const int lane_colrow = (lane >= 1 && lane <= n_hops && (cdir & GCM_DIR_BACKWARD) && cj >= 0 && cj <= cur)
                              ? cj : -1;
  // duplicates dropped (a later lane that names the row of an earlier valid one), and for row
  // j = tid & 127 of the phase below: is it a candidate?  One pass over the hop lanes, vector ops only.
  const int jrow = tid & 127;
  bool is_cand = jrow == cur, hop0 = false, dup = false;
  for (int k = 1; k <= n_hops; ++k) {
    const int jk = __builtin_amdgcn_readlane(cj, k);
    const bool vk = __builtin_amdgcn_readlane((int)cvalid, k) != 0;   // (before any removal: a
    hop0 |= __builtin_amdgcn_readlane(lane_h, k) == 0;                //  duplicate of a duplicate
    is_cand |= vk && jk == jrow;                                      //  is a duplicate as well)
    dup |= vk && jk == cj && k < lane;
  }
  dup |= lane >= 1 && cj == cur;   // hop 0 names row cur itself (lane 0)
  cvalid = cvalid && !dup;
  const unsigned long long cbal = __ballot(cvalid);
  const int C = __popcll(cbal);                       // 1 .. 17
  const int cpos = __popcll(cbal & ((1ull << lane) - 1ull));
  if (cvalid) sCand[wave * 32 + cpos] = cj;
  const int Cs = C < 16 ? C : 16;                     // rows fetched ahead (group 0)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // candidate rows: thread (l = tid / 16, c4 = tid % 16) holds columns 4 c4 .. and 4 (c4 + 16) ..
  const int sl = tid >> 4, c4 = tid & 15;
  const int sj = sl < Cs ? sCand[wave * 32 + sl] : cur;
  float4 sp[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c = (c4 + 16 * q) * 4;
    sp[q] = adj_row4(sj, c < N ? c : N - 4);
  }
  asm volatile("" ::: "memory");
  STAMP(2);
  // ---- x image: node matrix after the roll, the observation in row cur (gcm.py:274), zero padding
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int r = xr0 + i * XROWS;
    // image row of loaded row r: r - sh; the row dropped by the roll (r = 0) writes row cur instead
    const bool is_obs = wrap ? r == 0 : r == cur;
    const int ir = is_obs ? cur : r - sh;
    const bool ok = r < N && xc < F;
    const float4 t = xv[i];
    float4 v = make_float4(is_obs ? obv.x : t.x, is_obs ? obv.y : t.y, is_obs ? obv.z : t.z,
                           is_obs ? obv.w : t.w);
    v = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    if (fold_pe && ok && ir <= cur) {   // gcm.py:120-131, mode "add": rows up to and including cur
      const float4 e = *reinterpret_cast<const float4*>(pe + ir * F + xc);
      v = make_float4(v.x + e.x, v.y + e.y, v.z + e.z, v.w + e.w);
    }
    *reinterpret_cast<float4*>(sX + (r < N ? ir : r) * XS + xc) = v;
  }
  // ---- row cur after the selectors (temporal.py:72-88, dense.py:16-21), extra live rows ----------
  rc_val = (tid < 128 && tid < N) ? rc_val : 0.f;
  float r_new = rc_val;
  bool xpred = false;
  unsigned long long xbal = 0;
  if (tid < 128) {
    const int j = tid;
    if (is_cand && j != cur) r_new = 1.f;            // a forward / both hop writes (cur, j)
    if (dense && j <= cur) r_new = 1.f;
    if (j == cur && hop0) r_new = 1.f;
    if (has_sel && j < cur && srv != 0.f) r_new = 1.f;   // distance.py:31-37
    sRowCur[j] = r_new;
    sCoef[j] = 0.f;                                  // entries beyond the live list stay zero
    xpred = j < N && r_new != 0.f && !is_cand;       // live, and not fetched ahead
    xbal = __ballot(xpred);
    if (lane == 0) sInt[wave] = __popcll(xbal);
  }
  if (tid == 128) sInt[3] = 0;
  STAMP(3);
  __syncthreads();   // #1: x image, row cur, counts
  STAMP(4);
  const int n_extra = sInt[0] + sInt[1];
  const int Ltot = C + n_extra;
  const bool slow = Ltot > Cs;   // rows that were not fetched ahead (uniform)
  if (tid < 128) {
    if (wave == 0 && cvalid) {             // the candidates, in hop order
      sLive[cpos] = cj;
      sCoef[cpos] = sRowCur[cj];
    }
    if (xpred) {                           // the others, ascending
      const int pos = C + (wave ? sInt[0] : 0) + __popcll(xbal & ((1ull << lane) - 1ull));
      sLive[pos] = tid;
      sCoef[pos] = r_new;
    }
  }
  if (slow) __syncthreads();   // #2 (rare): the appended rows are fetched through the list

  // ---- the live rows, 16 at a time ---------------------------------------------------------------
  double a2p = 0.0;   // waves < HB: this lane's part of agg2[hcol] (fp64: up to N terms)
  float h1c = 0.f;    // waves < HB, kq == 0: h1[cur][hcol]
  const int n_groups = (Ltot + 15) >> 4;
  float* sv_rows = saved ? saved + lay_rows + (size_t)b * N * rw : nullptr;
#pragma unroll 1
  for (int g = 0; g < n_groups; ++g) {
    STAMP(5);
    // -- C: adjacency rows of this group -> LDS, the selector's column-cur entries applied in
    //       registers (the selector's stores go to HBM at the end: the rows read here are older)
    {
      const int l = sl, lg = 16 * g + l;
      const bool valid = lg < Ltot;
      const bool ahead = g == 0 && l < Cs;
      const int j = ahead ? sj : (valid ? sLive[lg] : 0);
      bool colcur = dense && j < cur;   // does (j, cur) get an entry from the selectors?
      for (int k = 1; k <= n_hops; ++k) colcur |= __builtin_amdgcn_readlane(lane_colrow, k) == j;
      unsigned nzbits = 0;
      float dsum = 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int c = (c4 + 16 * q) * 4;
        float4 v = sp[q];
        if (!ahead && valid) v = adj_row4(j, c < N ? c : N - 4);
        if (j == cur) v = make_float4(sRowCur[c], sRowCur[c + 1], sRowCur[c + 2], sRowCur[c + 3]);
        else if (colcur) {
          const int k = cur - c;
          v.x = k == 0 ? 1.f : v.x;
          v.y = k == 1 ? 1.f : v.y;
          v.z = k == 2 ? 1.f : v.z;
          v.w = k == 3 ? 1.f : v.w;
        }
        if (!valid || c >= N) v = make_float4(0.f, 0.f, 0.f, 0.f);
        float2* d = reinterpret_cast<float2*>(sRows + l * RS + c);
        d[0] = make_float2(v.x, v.y);
        d[1] = make_float2(v.z, v.w);
        if (lay_ar && valid && c < N)   // GCM_GNN_RECORD_DX: the row as layer 1 aggregates it
          *reinterpret_cast<float4*>(saved + lay_ar + ((size_t)b * N + lg) * N + c) = v;
        dsum += (v.x + v.y) + (v.z + v.w);
        const bool nz = (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
        const unsigned long long bal = __ballot(nz);   // lane = 16 l' + c4: fold the wave's 4 rows
        const unsigned m = (unsigned)((bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xffffull);
        nzbits |= m << (16 * q);
      }
      if (lane == 0 && nzbits) atomicOr(reinterpret_cast<unsigned*>(&sInt[3]), nzbits);
      if (lay_ar && valid && c4 == 0) reinterpret_cast<int*>(saved + lay_lv)[(size_t)b * N + lg] = j;
      if (fold_deg) {   // row sum of live row l: the 16 lanes that hold it
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) dsum += __shfl_xor(dsum, d);
        if (c4 == 0) sDeg[l] = dsum;
      }
    }
    STAMP(6);
    __syncthreads();   // #3
    STAMP(7);
    // -- D: agg1 = rows @ x over the non-zero 4-column chunks (exact: zero chunks add nothing)
    if (wave < FB) {
      unsigned km = __builtin_amdgcn_readfirstlane((unsigned)sInt[3]);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* ap = sRows + m16 * RS + kq;
      const float* bp = sX + kq * XS + 16 * wave + m16;
      while (km) {   // four chunks per trip: the LDS reads of a trip are in flight together
        float av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool has = km != 0;
          const int c = has ? __builtin_ctz(km) : 0;
          km &= km - 1;
          const float a = ap[4 * c], bb = bp[4 * c * XS];
          av[q] = has ? a : 0.f;
          bv[q] = has ? bb : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) sAgg[(4 * kq + r) * AS + 16 * wave + m16] = acc[r];
    }
    STAMP(8);
    __syncthreads();   // #4
    STAMP(9);
    // -- E: h1 = act1([agg1 | x[j]] @ [W_rel1 | W_root1]^T + b1); this lane's part of agg2
    if (tid == 255) sInt[3] = 0;   // D has read the chunk mask; the next group's C ORs after #5
    if (wave < HB) {
      // A operand: lane (l = m16, kq) reads its KL contiguous k - agg1[l] (kq < 2) or x[j_l] (kq >= 2)
      const int lg = 16 * g + m16;
      const int jl = lg < Ltot ? sLive[lg] : cur;
      const float* abase = (kq < 2 ? sAgg + m16 * AS : sX + jl * XS) + (kq & 1) * KL;
      float4 a4[KL / 4];
#pragma unroll
      for (int q = 0; q < KL / 4; ++q) a4[q] = *reinterpret_cast<const float4*>(abase + 4 * q);
      float cf[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) cf[r] = sCoef[16 * g + 4 * kq + r];
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < KL / 4; ++q) {   // two chains: the 16x16x4 MFMA has 40 cycles of latency
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q].x, w1r[q].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q].y, w1r[q].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q].z, w1r[q].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[q].w, w1r[q].w, acc1, 0, 0, 0);
      }
      float hv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pre = acc0[r] + acc1[r] + bias1;
        if (fold_deg) pre = fmaf(sDeg[4 * kq + r], c1v, pre);
        const float t = gcm_act_sel(pre, act1_v);
        hv[r] = hcol < H1 ? t : 0.f;                 // padding columns stay out of layer 2
        sH1[(4 * kq + r) * HS + hcol] = hv[r];
        a2p = fma((double)cf[r], (double)hv[r], a2p);   // coef is zero beyond the live list
      }
      if (g == 0 && kq == 0) h1c = hv[0];            // row cur is slot 0 of the list
      if (g == n_groups - 1) {                       // the four lane groups of a column meet in LDS
        sA2[(wave * 4 + kq) * 16 + m16] = a2p;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (kq == 0) {
          const double t = (sA2[(wave * 4 + 0) * 16 + m16] + sA2[(wave * 4 + 1) * 16 + m16]) +
                           (sA2[(wave * 4 + 2) * 16 + m16] + sA2[(wave * 4 + 3) * 16 + m16]);
          sV[hcol] = (float)t;
          sV[HP + hcol] = h1c;
        }
      }
    }
    STAMP(10);
    __syncthreads();   // #5
    STAMP(11);
    if (sv_rows) {   // thread -> (row l, columns k = c4 + 16 i): h1 [H1] | agg1 [F] | x[j] [F]
      const int l = sl;
      if (16 * g + l < Ltot) {
        const int j = sLive[16 * g + l];
        float* dst = sv_rows + (size_t)(16 * g + l) * rw;
#pragma unroll
        for (int i = 0; i < HB; ++i) {
          const int k = c4 + 16 * i;
          if (k < H1) dst[k] = sH1[l * HS + k];
        }
#pragma unroll
        for (int i = 0; i < FB; ++i) {
          const int k = c4 + 16 * i;
          if (k < F) {
            dst[H1 + k] = sAgg[l * AS + k];
            dst[H1 + F + k] = sX[j * XS + k];
          }
        }
        if (fold_deg && c4 == 0) saved[lay.o_deg + (size_t)b * N + 16 * g + l] = sDeg[l];
      }
    }
    if (g + 1 < n_groups) __syncthreads();   // the images are rewritten by the next group
  }

  // ---- layer 2 on row cur: mx = act2(W2c v + b2), v = agg2 | h1[cur]; KG adjacent lanes per output --
  {
    const float* vv = sV + kg * KC;
    float xq[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) xq[k] = vv[k];
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < KC; ++k) t = fmaf(w2r[k], xq[k], t);
    // sum over the KG (4 or 8) adjacent lanes of an output on the DPP path (ds_bpermute butterflies: three
    // dependent LDS round trips on the step's critical path)
    static_assert(KG == 4 || KG == 8, "layer-2 lane groups");
#define GCM_DPP_ADD(v, ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, false))
    GCM_DPP_ADD(t, 0xB1);                 // quad_perm [1,0,3,2]
    GCM_DPP_ADD(t, 0x4E);                 // quad_perm [2,3,0,1]
    if (KG == 8) GCM_DPP_ADD(t, 0x141);   // row_half_mirror: the other quad of the 8
#undef GCM_DPP_ADD
    const float v = gcm_act_sel(t + bias2, act2_v);
    const bool mine = kg == 0 && o2 < H2;
    if (mine) mx_out[(size_t)b * H2 + o2] = v;
    const bool any_bad = __any(mine && !isfinite(v));
    if (any_bad && lane == 0) atomicOr(flags, GCM_FLAG_NONFINITE);
  }
  if (saved) {
    if (tid == 0) {
      int* hdr = reinterpret_cast<int*>(saved + lay_hdr) + 4 * b;
      hdr[0] = Ltot; hdr[1] = 0; hdr[2] = cur; hdr[3] = wrap ? 1 : 0;
    }
    float* cf = saved + lay_coef + (size_t)b * N;
    for (int l = tid; l < Ltot; l += 256) cf[l] = sCoef[l];
    if (tid < 2 * H1) {
      const int k = tid < H1 ? tid : HP + (tid - H1);
      saved[lay_v + (size_t)b * 2 * H1 + tid] = sV[k];
    }
  }
  STAMP(12);
  // ---- the donated state, off the critical path: the selector's entries, the inserted node and the
  // count ------------------------------------------------------------------------------------------
  if (FUNC) return;   // (functional state: written by the state waves)
  if (tid < 128) {
    if (tid < N && r_new != rc_val) ag[cur * N + tid] = r_new;
  } else {
    const int t2 = tid - 128;   // column cur: backward hops (temporal) / rows < cur (dense)
    if (wave == 2 && lane_colrow >= 0) ag[lane_colrow * N + cur] = 1.f;
    if (dense) {
      for (int r = t2; r < cur; r += 128) ag[r * N + cur] = 1.f;
    }
    if (t2 < F4) {   // the inserted node (gcm.py:274)
      *reinterpret_cast<float4*>(ng + cur * F + t2 * 4) =
          *reinterpret_cast<const float4*>(obs + (size_t)b * F + t2 * 4);
    }
    if (t2 == 127) {   // count_out may alias count_in: every wave read it long ago
      count_out[b] = cur + 1;
      if (cur_out) cur_out[b] = cur;
      const uint32_t f = (wrap ? GCM_FLAG_WRAPPED : 0u) | ((n_in < 0 || n_in > N) ? GCM_FLAG_BAD_COUNT : 0u);
      if (f) atomicOr(flags, f);
    }
  }
}

#ifdef GCM_DEBUG_ABI   // libgcm_hip_debug.so only (include/gcm_hip_debug.h): the product library keeps no state
static thread_local hipEvent_t t_start = nullptr, t_stop = nullptr;   // gcm_debug_time_next_launch
#endif

template <int FP, int HP, int H2P, int NX, bool EXACT>
int launch(hipStream_t s, const float* obs, const float* nodes_in, const float* adj_in,
           const int64_t* count_in, float* nodes_out, float* adj_out, int64_t* count_out,
           int64_t* cur_out, const Edits& E, const Gnn2& P, float* mx, float* saved,
           const SavedLayout& lay, uint32_t* flags, int B, int N, int F, int H1, int H2,
           const float* c1 = nullptr, const float* pe = nullptr, const float* sel_row = nullptr) {
  constexpr size_t lds = sizeof(float) * (size_t)Lds<FP, HP, H2P>::TOTAL;
  static_assert(lds <= 160 * 1024, "LDS budget");
  const bool func = adj_out != adj_in;
  auto kern = func ? k_step_rows<FP, HP, H2P, true, NX, EXACT> : k_step_rows<FP, HP, H2P, false, NX, EXACT>;
  gcm_allow_dynamic_lds((const void*)kern, lds);
#ifdef GCM_DEBUG_ABI
  if (t_start && t_stop) {   // one-shot: events recorded by the dispatch itself
    hipExtLaunchKernelGGL(kern, dim3(func ? (1 + GCM_STATE_WGS) * B : B), dim3(256), lds, s, t_start, t_stop, 0, obs, nodes_in,
                          adj_in, count_in, nodes_out, adj_out, count_out, cur_out, E, P, mx, saved, lay, flags,
                          N, F, H1, H2, c1, pe, sel_row, B);
    t_start = t_stop = nullptr;
    return gcm_launch_status();
  }
#endif
  hipLaunchKernelGGL(kern, dim3(func ? (1 + GCM_STATE_WGS) * B : B), dim3(256), lds, s, obs, nodes_in, adj_in, count_in,
                     nodes_out, adj_out, count_out, cur_out, E, P, mx, saved, lay, flags, N, F, H1, H2, c1, pe,
                     sel_row, B);
  return gcm_launch_status();
}

}  // namespace gcm_rows

#ifdef GCM_DEBUG_ABI
extern "C" int gcm_debug_time_next_launch(void* start_event, void* stop_event) {
  gcm_rows::t_start = (hipEvent_t)start_event;
  gcm_rows::t_stop = (hipEvent_t)stop_event;
  return GCM_OK;
}

/* Measurement aid (bench.py): T steps of gcm_dense_rows_step_fwd enqueued back to back from C on the
 * evolving donated state - the launch cadence of a replayed HIP graph, which an interpreter loop does
 * not reach (the gaps it leaves let the clocks sag and the caches cool) - every launch bracketed by the
 * caller's HIP events, recorded by the dispatch itself (gcm_debug_time_next_launch). */
extern "C" int gcm_debug_time_rows_rollout(const float* obs_all, float* nodes, float* adj, int64_t* count,
                                           const gcm_selector_desc* selectors, int n_selectors,
                                           const float* params, int has_bias, int act1, int act2,
                                           float* const* saved_per_step, uint32_t* flags,
                                           void* const* start_events, void* const* stop_events, int T,
                                           int B, int N, int F, int H1, int H2, gcm_stream_t stream) {
  GCM_REQUIRE(obs_all && saved_per_step && start_events && stop_events && T > 0);
  for (int t = 0; t < T; ++t) {
    gcm_debug_time_next_launch(start_events[t], stop_events[t]);
    const int rc = gcm_dense_rows_step_fwd(obs_all + (size_t)t * B * F, nodes, adj, count, nodes, adj, count,
                                           nullptr, selectors, n_selectors, params, has_bias, act1, act2,
                                           saved_per_step[t], saved_per_step[t], flags, B, N, F, H1, H2, stream);
    if (rc) return rc;
  }
  return GCM_OK;
}
#endif   // GCM_DEBUG_ABI

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

extern "C" int gcm_dense_rows_layout_dx(int B, int N, int F, int H1, int H2, size_t* out8) {
  GCM_REQUIRE(out8 && B > 0 && N > 0 && F > 0 && H1 > 0 && H2 > 0);
  const gcm_rows::SavedLayout lay = gcm_rows::make_layout(B, N, F, H1, H2, true);
  out8[0] = lay.total; out8[1] = lay.o_v; out8[2] = lay.o_hdr; out8[3] = lay.o_coef;
  out8[4] = lay.o_rows; out8[5] = (size_t)lay.rw; out8[6] = lay.o_live; out8[7] = lay.o_arows;
  return GCM_OK;
}

extern "C" size_t gcm_dense_rows_step_workspace_bytes(const gcm_selector_desc* selectors,
                                                      int n_selectors, int B, int N, int F) {
  size_t need = 0;
  for (int i = 0; selectors && i < n_selectors; ++i)
    if (selectors[i].kind == GCM_SEL_DISTANCE)
      need = sizeof(float) * (size_t)B * N +
             gcm_edge_distance_workspace_bytes(selectors[i].mode, std::max(B, selectors[i].n_cur_rows), N, F);
  return need;
}

extern "C" int gcm_dense_rows_step_fwd(const float* obs, const float* nodes_in, const float* adj_in,
                                       const int64_t* count_in, float* nodes_out, float* adj_out,
                                       int64_t* count_out, int64_t* cur_out,
                                       const gcm_selector_desc* selectors, int n_selectors,
                                       const float* params, int has_bias, int act1, int act2,
                                       float* mx, float* saved, uint32_t* flags, int B, int N,
                                       int F, int H1, int H2, gcm_stream_t stream) {
  return gcm_dense_rows_step_fwd_ws(obs, nodes_in, adj_in, count_in, nodes_out, adj_out, count_out,
                                    cur_out, selectors, n_selectors, params, has_bias, act1, act2, mx,
                                    saved, flags, nullptr, 0, B, N, F, H1, H2, stream);
}

extern "C" int gcm_dense_rows_step_fwd_ws(const float* obs, const float* nodes_in, const float* adj_in,
                                          const int64_t* count_in, float* nodes_out, float* adj_out,
                                          int64_t* count_out, int64_t* cur_out,
                                          const gcm_selector_desc* selectors, int n_selectors,
                                          const float* params, int has_bias, int act1, int act2,
                                          float* mx, float* saved, uint32_t* flags, void* workspace,
                                          size_t workspace_bytes, int B, int N, int F, int H1, int H2,
                                          gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes_in && adj_in && count_in && nodes_out && adj_out && count_out && params &&
              mx && flags);
  GCM_REQUIRE(B > 0 && (selectors || n_selectors == 0));
  GCM_REQUIRE((nodes_out == nodes_in) == (adj_out == adj_in));   // donate both or neither
  if (!gcm_dense_rows_supported(N, F, H1, H2)) return GCM_EUNSUPPORTED;
  gcm_fused::Edits E{};
  const float* sel_row = nullptr;
  for (int i = 0; i < n_selectors; ++i) {
    const gcm_selector_desc& d = selectors[i];
    if (d.kind == GCM_SEL_DISTANCE) {
      // the distance selector runs first, on the state as it comes in, and hands its row over
      if (sel_row || d.bidirectional) return GCM_EUNSUPPORTED;
      GCM_REQUIRE(workspace);
      if (workspace_bytes < gcm_dense_rows_step_workspace_bytes(selectors, n_selectors, B, N, F))
        return GCM_EWORKSPACE;
      float* row = (float*)workspace;
      const int rc = gcm_edge_distance_pre_ex(nodes_in, count_in, obs, row, d.mode, d.max_distance,
                                              d.dist_param, d.a0, d.a1, d.b0, d.b1, d.cur_rows, d.n_cur_rows,
                                              row + (size_t)B * N,
                                              workspace_bytes - sizeof(float) * (size_t)B * N, B, N, F, stream);
      if (rc) return rc;
      sel_row = row;
      continue;
    }
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
  // folded preprocessor bias / positional encoding: extra sections behind the GNN parameters
  const float* c1 = (has_bias & GCM_GNN_HAS_DEG_TERM) ? b2 + H2 : nullptr;
  const float* pe = (has_bias & GCM_GNN_HAS_PE_TABLE) ? b2 + H2 + (c1 ? H1 : 0) : nullptr;
  const bool folded = c1 || pe || sel_row;
  // the bias slots are always there (zeros when a layer has none): read unconditionally
  gcm_fused::Gnn2 P{w_rel1, b1, w_root1, w_rel2, b2, w_root2, act1, act2};
  const gcm_rows::SavedLayout lay = gcm_rows::make_layout(B, N, F, H1, H2, (has_bias & GCM_GNN_RECORD_DX) != 0);
  hipStream_t s = (hipStream_t)stream;
  const int fp = F <= 32 ? 32 : 64, hp = H1 <= 32 ? 32 : 64, h2p = H2 <= 32 ? 32 : 64;
  // tile-exact specialisations of the common shapes
#define GCM_RX(a, n)                                                                               \
  if (!folded && F == a && H1 == a && H2 == a && N == n)                                                      \
    return gcm_rows::launch<a, a, a, n, true>(s, obs, nodes_in, adj_in, count_in, nodes_out,       \
                                              adj_out, count_out, cur_out, E, P, mx, saved, lay,   \
                                              flags, B, N, F, H1, H2);
  GCM_RX(32, 128) GCM_RX(32, 64) GCM_RX(32, 32) GCM_RX(64, 128)
#undef GCM_RX
  // cfg3's shape (observations of 64, hidden 32, 128 nodes), with or without a distance selector's row
  if (!c1 && !pe && F == 64 && H1 == 32 && H2 == 32 && N == 128)
    return gcm_rows::launch<64, 32, 32, 128, true>(s, obs, nodes_in, adj_in, count_in, nodes_out, adj_out,
                                                   count_out, cur_out, E, P, mx, saved, lay, flags, B, N, F,
                                                   H1, H2, nullptr, nullptr, sel_row);
#define GCM_R(a, b_, c)                                                                          \
  if (fp == a && hp == b_ && h2p == c)                                                           \
    return gcm_rows::launch<a, b_, c, 0, false>(s, obs, nodes_in, adj_in, count_in, nodes_out,   \
                                                adj_out, count_out, cur_out, E, P, mx, saved,    \
                                                lay, flags, B, N, F, H1, H2, c1, pe, sel_row);
  GCM_R(32, 32, 32) GCM_R(32, 32, 64) GCM_R(32, 64, 32) GCM_R(32, 64, 64)
  GCM_R(64, 32, 32) GCM_R(64, 32, 64) GCM_R(64, 64, 32) GCM_R(64, 64, 64)
#undef GCM_R
  return GCM_EUNSUPPORTED;
}
