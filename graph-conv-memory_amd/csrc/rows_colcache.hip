// The DenseGCM step (gcm.py:262-321) for chains whose selectors also write COLUMN cur of the adjacency:
// DenseEdge (dense.py:16-21: the new node <-> every earlier node, plus a self edge) and TemporalBackedge with
// direction "backward" / "both" (temporal.py:72-88: adj[cur - hop, cur] = 1).  There the layer-1 rows of OLDER
// nodes are no longer final - row j gains the source `cur` - which is what kept these selectors on the general
// live-row kernel (rows_step.hip: it re-aggregates every live row from the adjacency at every step: cur^2 F
// work per graph with DenseEdge, five barriers per 16 rows).
//
// In a chain that started from EMPTY graphs on a donated state and has made fewer than N steps (nothing has
// overflowed; every graph holds exactly `cur` nodes, and the host knows cur), a column write is a RANK-1
// correction of what the chain already knows:
//
//     agg1[j] = sum_k adj[j,k] x[k]      gains   + x[cur]    for the rows j the selectors give the entry (j, cur)
//     root[j] = W_root1 x[j] + b1        is final once node j is written
//     h1[j]   = act1(W_rel1 agg1[j] + root[j])
//
// so the chain keeps agg1 [B,N,F] and root [B,N,H1] per node and a step is: the new row's aggregate (a masked
// column sum of the node matrix), the rank-1 update of the touched rows, ONE [rows x F] . [F x H1] product on the
// fp32 matrix cores for the live rows (32 rows per wave: v_mfma_f32_32x32x2_f32 with both operands straight from
// global memory into the registers that feed it - lane (row, half) owns F/2 contiguous k of its row, the same k
// of W_rel1's row `col` as the B operand), the activation, agg2, and layer 2 on row cur.  Exact: the same sums as
// the reference's adj @ x (ones and zeros), accumulated in ascending source order.  cur^2 F per graph-step
// becomes cur F (update) + 2 cur F H1 (product).
//
// Which rows: the host folds the selector chain and cur into two 128-bit masks -
//   srow: the sources of row cur (bit cur = a self edge),  scol: the older rows that gain the source cur.
// The state (nodes, adj, count) is advanced in place; the record is the GENERAL live-row record of rows_common.h
// (slot 0 = row cur, then the other live rows ascending), read by k_bptt_rows<.., 0> unchanged.
//
// One workgroup (4 waves) per graph.  F, H1 in {32, 64}, H2 <= 64, N <= 128.
#include <type_traits>

#include "fused_common.h"
#include "rows_common.h"
#include "rows_state_waves.h"

#ifdef GCM_STAMPS   // diagnostic build only (make stamps11, tools/kstamp_colcache.py)
__device__ unsigned long long g_stamps[32];
extern "C" int gcm_debug_read_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
#endif

namespace gcm_rows {

using gcm_fused::acc_row;
using gcm_fused::Gnn2;

struct RowMask {
  unsigned long long lo, hi;
};
// bits [32 w, 32 w + 32) of the mask (w compile-time or wave-uniform: scalar code)
__device__ __forceinline__ unsigned mword(const RowMask& m, int w) {
  return (unsigned)((w & 2 ? m.hi : m.lo) >> ((w & 1) * 32));
}
__device__ __forceinline__ bool mbit(const RowMask& m, int j) {
  return ((j < 64 ? m.lo >> j : m.hi >> (j - 64)) & 1ull) != 0;
}
// set bits below position j
__device__ __forceinline__ int mrank(const RowMask& m, int j) {
  if (j < 64) return __popcll(m.lo & ((1ull << j) - 1ull));
  return __popcll(m.lo) + __popcll(m.hi & ((1ull << (j - 64)) - 1ull));
}
// LDS exchange between the waves of the workgroup: waits for this wave's LDS operations only - __syncthreads() also
// drains the wave's global loads and STORES (vmcnt(0)), which here would put every barrier behind the record's stores
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ssrc: the STORED rows the new row aggregates from;  self: the new row's own self edge;  scol: the stored rows that gain
// the new node as a source;  sdrop (steady state): the stored rows that lose the dropped node as a source - all in RING
// coordinates (slot of a node = its chain index mod N; below N steps slot = graph row).  cur: the new node's slot;
// rot: the slot of graph row 0 as the state comes in (0 below N steps);  n_slots: slots in use after the step.
// FUNC: functional state (the reference's default: gcm.py:262,278,286 clone it every step) - the state advance (copy,
// roll, the selectors' entries, the observation, the count) runs in GCM_STATE_WGS extra workgroups per graph of the same
// launch (rows_state_waves.h: blocks >= Bn; `E` = the selector chain as they want it), the compute workgroups read the
// incoming state and write the chain's caches and the record only.
template <int FK, int HK, int O2T, bool FUNC>
__global__ __launch_bounds__(256) void k_step_colcache(
    const float* __restrict__ obs, const float* nodes_in, const float* adj_in, const int64_t* count_in, float* nodes,
    float* adj, int64_t* count, const RowMask ssrc, const int self, const RowMask scol, const RowMask sdrop,
    const int cur, const int rot, const int steady, const Gnn2 P, float* __restrict__ cA, float* __restrict__ cR,
    float* __restrict__ saved, const SavedLayout lay, uint32_t* __restrict__ flags, const int N, const int H2,
    const Edits E, const int Bn) {
  if (FUNC && (int)blockIdx.x >= Bn) {
    const int q = blockIdx.x - Bn, bs = q % Bn, ks = q / Bn;
    advance_state_waves<FK>(obs, nodes_in + (size_t)bs * N * FK, adj_in + (size_t)bs * N * N, count_in,
                            nodes + (size_t)bs * N * FK, adj + (size_t)bs * N * N, count, nullptr, E, flags, nullptr, bs,
                            ks, threadIdx.x, N, FK);
    return;
  }
  constexpr int C4 = FK / 4;       // 16-byte pieces of a node row
  constexpr int RG = 256 / C4;     // node rows per pass of the workgroup (32 or 16)
  constexpr int XP = 128 / RG;     // node rows per thread
  constexpr int KH = FK / 2;       // k per half-wave
  constexpr int KQ = KH / 4;
  constexpr int CT = HK / 32;      // 32-column tiles of layer 1
  constexpr int PS = FK + 4;       // stride of the partial-sum image
  constexpr int SPLIT = 64 / FK;   // lanes per feature in the row-group sum (2 or 1)
  constexpr int GP = RG / SPLIT;   // row groups per lane there (16)
  __shared__ __attribute__((aligned(16))) float sPart[RG * PS];
  __shared__ __attribute__((aligned(16))) float sAggc[FK];
  __shared__ float sRcur[HK];
  __shared__ __attribute__((aligned(16))) float sV[2 * HK];
  __shared__ float sA2[4 * HK];

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t gb = (size_t)b;
  const float* ng_in = nodes_in + gb * N * FK;   // the state as it comes in (FUNC: never written)
  float* ng = nodes + gb * N * FK;               // donated: the same matrix, advanced in place
  float* ag = adj + gb * N * N;
  float* cAg = cA + gb * N * FK;
  float* cRg = cR + gb * N * HK;
  const bool rec = lay.total != 0;
  const int rw = lay.rw;
  float* sv_rows = saved + lay.o_rows + gb * N * rw;
  const int n_slots = steady ? N : cur + 1;       // slots in use after the step = graph rows read below
  // wave-uniform: this wave's 32 slots hold a row the step touches (a live row, a row whose aggregate changes, the new
  // row) - with a handful of temporal hops most tiles hold none and skip their loads, product and activation
  const bool tile_on = 32 * wave < n_slots && ((mword(ssrc, wave) | mword(scol, wave) | mword(sdrop, wave)) != 0u ||
                                               (cur >> 5) == wave);

  STAMP(0);
  // ---- every load of the step, in the order of use, before anything waits ------------------------------------------
  const int64_t n_in = count_in[b];
  // steady state: the dropped node - graph row 0 - as the A-operand lanes hold a row.  FIRST in the queue: loads return
  // in order, so a wave that has its node rows (barrier #1) has this one too - the roll overwrites row 0 behind that
  // barrier.
  f32x4 xo[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) xo[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (steady) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) xo[q] = *reinterpret_cast<const f32x4*>(ng_in + lh * KH + 4 * q);
  }
  asm volatile("" ::: "memory");   // (the compiler keeps them in front of the node rows' loads)
  // node rows (graph coordinates, the state as it comes in), one 16-byte piece per (row group, piece) thread: the new
  // row's aggregate, the record's x section and - in the steady state - the roll of the node matrix
  const int c4 = tid % C4, rg = tid / C4;
  f32x4 xr[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) xr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    // (uniform: passes beyond the stored rows are skipped.  The zero fill sits BEFORE the loads: as the other arm of
    //  the branch it made hipcc wait for every load in flight - vmcnt(0) - right behind the first pass)
    const int row = rg + RG * i;
    if (i == 0 || RG * i < n_slots)   // (pass 0 always holds a row: no branch - hipcc waited behind its conditional form)
      xr[i] = *reinterpret_cast<const f32x4*>(ng_in + (row < N ? row : N - 1) * FK + 4 * c4);
  }
  const f32x4 obq = *reinterpret_cast<const f32x4*>(obs + gb * FK + 4 * c4);
  // the observation as the A-operand lanes hold a row: k in [lh KH, (lh + 1) KH)
  f32x4 xa[KQ];
#pragma unroll
  for (int q = 0; q < KQ; ++q) xa[q] = *reinterpret_cast<const f32x4*>(obs + gb * FK + lh * KH + 4 * q);
  // wave 3: W_root1 (the new node's root row)
  f32x4 wr[CT][KQ];
  float b1v[CT];
  if (wave == 3) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
      for (int q = 0; q < KQ; ++q)
        wr[ct][q] = *reinterpret_cast<const f32x4*>(P.w_root1 + (32 * ct + li) * FK + lh * KH + 4 * q);
      b1v[ct] = P.b_rel1[32 * ct + li];
    }
  }
  // the A operand: slot r = 32 wave + li, k in [lh KH, (lh + 1) KH) - agg1 of the stored rows from the chain's cache;
  // the B operand: W_rel1[col][k], col = 32 ct + li, the same k; root[slot][col] of the rows the accumulators hold
  const int r = 32 * wave + li;
  const int rc = r < N ? r : N - 1;
  f32x4 ca[KQ], wb[CT][KQ], crq[CT][4];
  const int NQ = N >> 2;   // the root cache is kept in quads of rows: cR[b][slot / 4][col][slot % 4] (one 16-byte load
                           // per four accumulator rows of a lane)
  if (tile_on) {
#pragma unroll
    for (int q = 0; q < KQ; ++q) ca[q] = *reinterpret_cast<const f32x4*>(cAg + rc * FK + lh * KH + 4 * q);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int q = 0; q < KQ; ++q)
        wb[ct][q] = *reinterpret_cast<const f32x4*>(P.w_rel1 + (32 * ct + li) * FK + lh * KH + 4 * q);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {   // rows 8 q4 + 4 lh + (0 .. 3) of the tile: one quad of the root cache
        const int quad = 8 * wave + 2 * q4 + lh;
        crq[ct][q4] = *reinterpret_cast<const f32x4*>(cRg + ((quad < NQ ? quad : NQ - 1) * HK + 32 * ct + li) * 4);
      }
  }
  // wave 0: layer 2
  f32x4 w2[O2T][HK / 4];
  float b2v[O2T];
  if (wave == 0) {
#pragma unroll
    for (int ot = 0; ot < O2T; ++ot) {
      const int o = 32 * ot + li < H2 ? 32 * ot + li : H2 - 1;
      const float* src = (lh ? P.w_root2 : P.w_rel2) + o * HK;
#pragma unroll
      for (int q = 0; q < HK / 4; ++q) w2[ot][q] = *reinterpret_cast<const f32x4*>(src + 4 * q);
      b2v[ot] = P.b_rel2[o];
    }
  }
  const int act1 = P.act1, act2 = P.act2;
  asm volatile("" ::: "memory");
  STAMP(1);

  // A chain from empty graphs holds min(steps made, N) nodes in every graph; anything else (a caller edited the count)
  // leaves the graph untouched and raises the flag (uniform per workgroup: nothing has been stored yet)
  if (n_in != (int64_t)(steady ? N : cur)) {
    if (tid == 0) atomicOr(flags, GCM_FLAG_BAD_COUNT);
    return;
  }
  STAMP(2);
  // the stored sources as four 32-slot words (scalars) and the record position of each word's first slot: slot 0 of
  // the record is the new row, the stored live rows follow in ascending SLOT order
  const unsigned sw0 = mword(ssrc, 0), sw1 = mword(ssrc, 1), sw2 = mword(ssrc, 2), sw3 = mword(ssrc, 3);
  const int pc1 = 1 + __popc(sw0), pc2 = pc1 + __popc(sw1), pc3 = pc2 + __popc(sw2);
  const int L = pc3 + __popc(sw3);   // the new row + its stored sources
  auto sword = [&](int w) { return w == 0 ? sw0 : (w == 1 ? sw1 : (w == 2 ? sw2 : sw3)); };
  auto sbase = [&](int w) { return w == 0 ? 1 : (w == 1 ? pc1 : (w == 2 ? pc2 : pc3)); };

  // ---- the new row's aggregate: sum of the selected node rows, ascending inside a thread, then over the row groups.
  // Graph row i of the incoming state sits in slot i + rot (mod N); the new node takes the place of graph row `cur`
  // (below N steps) / of the dropped graph row 0 (steady state: that thread carries the observation instead).
  auto new_row_sum = [&](auto ringc) {
    constexpr bool RING = decltype(ringc)::value;   // rot != 0: slot = (row + rot) mod N; else slot = row and the
                                                    // mask word of a pass is a compile-time choice of a scalar
    f32x4 part = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      if (RG * i >= n_slots) continue;   // uniform
      const int row = rg + RG * i;
      const bool is_cur = steady ? row == 0 : row == cur;
      bool stored;     // a stored source of the new row
      unsigned pos;    // its record row
      if (RING) {
        int slot = row + rot;
        slot -= slot >= N ? N : 0;
        stored = row < n_slots && !is_cur && mbit(ssrc, slot);
        pos = 1u + (unsigned)mrank(ssrc, slot);
      } else {
        const int w = (RG * i) >> 5, bit = row & 31;     // w: compile time
        const unsigned word = sword(w);
        stored = !is_cur && ((word >> bit) & 1u) != 0;   // (bits beyond the slots in use are never set)
        pos = (unsigned)sbase(w) + (unsigned)__popc(word & ((1u << bit) - 1u));
      }
      const f32x4 v = is_cur ? obq : xr[i];
      if (stored || (is_cur && self != 0)) part += v;
      if (rec && (stored || is_cur))
        *reinterpret_cast<f32x4*>(sv_rows + (is_cur ? 0u : pos) * (unsigned)rw + HK + FK + 4 * c4) = v;
    }
    *reinterpret_cast<f32x4*>(sPart + rg * PS + 4 * c4) = part;
  };
  if (rot) new_row_sum(std::true_type{});
  else new_row_sum(std::false_type{});
  // the new node's root row (wave 3 holds no stored row until 96 slots are in use)
  if (wave == 3) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        t = fmaf(wr[ct][q].x, xa[q].x, t);
        t = fmaf(wr[ct][q].y, xa[q].y, t);
        t = fmaf(wr[ct][q].z, xa[q].z, t);
        t = fmaf(wr[ct][q].w, xa[q].w, t);
      }
      t = gcm_xor32_add(t);
      const float v = t + b1v[ct];
      if (lh == 0) {
        sRcur[32 * ct + li] = v;
        cRg[((cur >> 2) * HK + 32 * ct + li) * 4 + (cur & 3)] = v;
      }
    }
  }
  STAMP(3);
  lds_barrier();   // #1: every wave has its node rows in registers
  STAMP(4);
  // the node matrix of the donated state: the observation into row cur (gcm.py:274) - in the steady state behind the
  // overflow roll (gcm.py:323-355: row i <- row i + 1), in place: every wave's loads of it landed before barrier #1
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    if (FUNC || RG * i >= n_slots) continue;   // (functional state: the state workgroups write the new matrix)
    const int row = rg + RG * i;
    if (steady) {
      if (row < N) *reinterpret_cast<f32x4*>(ng + (row == 0 ? N - 1 : row - 1) * FK + 4 * c4) = row == 0 ? obq : xr[i];
    } else if (row == cur) {
      *reinterpret_cast<f32x4*>(ng + cur * FK + 4 * c4) = obq;
    }
  }
  if (wave == 2) {   // the row groups' partial sums -> agg1[cur]: lane (feature, part) takes GP groups
    const int f = lane % FK, part = lane / FK;
    float v[GP];
#pragma unroll
    for (int g = 0; g < GP; ++g) v[g] = sPart[(part * GP + g) * PS + f];
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < GP; ++g) s += v[g];
    if (SPLIT == 2) s = gcm_xor32_add(s);
    if (part == 0) {
      sAggc[f] = s;
      cAg[cur * FK + f] = s;
      if (rec) sv_rows[HK + f] = s;
    }
  }
  lds_barrier();   // #2
  STAMP(5);

  // ---- layer 1 of the live rows: [32 slots x F] . [F x H1] per wave ------------------------------------------------
  if (tile_on) {
    const unsigned sr_t = sword(wave), sc_t = mword(scol, wave), sd_t = mword(sdrop, wave);   // scalars
    const int rb = sbase(wave);
    const bool cur_in = (cur >> 5) == wave;          // the new row's slot lies in this tile
    const int cur_bit = cur_in ? cur & 31 : 99;
    {
      const bool upd = ((sc_t >> li) & 1u) != 0;     // the row gains the source cur
      const bool drp = ((sd_t >> li) & 1u) != 0;     // steady state: the row loses the dropped node
      const bool lrow = ((sr_t >> li) & 1u) != 0;    // a stored live row
      const unsigned pos = (unsigned)rb + (unsigned)__popc(sr_t & ((1u << li) - 1u));
      float* rdst = sv_rows + pos * (unsigned)rw + HK + lh * KH;
      float a[KH];
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        f32x4 v = ca[q];
        const f32x4 xq = xa[q], xd = xo[q];
        if (steady) v -= f32x4{drp ? xd.x : 0.f, drp ? xd.y : 0.f, drp ? xd.z : 0.f, drp ? xd.w : 0.f};   // (uniform)
        v += f32x4{upd ? xq.x : 0.f, upd ? xq.y : 0.f, upd ? xq.z : 0.f, upd ? xq.w : 0.f};
        if (upd || drp) *reinterpret_cast<f32x4*>(cAg + r * FK + lh * KH + 4 * q) = v;
        if (li == cur_bit) v = *reinterpret_cast<const f32x4*>(sAggc + lh * KH + 4 * q);
        if (rec && lrow) *reinterpret_cast<f32x4*>(rdst + 4 * q) = v;
        a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
      }
      STAMP(6);
      f32x16 acc[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
#pragma unroll
      for (int s = 0; s < KH; ++s)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const f32x4 w = wb[ct][s >> 2];
          const float wv = (s & 3) == 0 ? w.x : ((s & 3) == 1 ? w.y : ((s & 3) == 2 ? w.z : w.w));
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wv, acc[ct], 0, 0, 0);
        }
      STAMP(7);
      // the activation, this lane's part of agg2, h1[cur], the record's h1 rows.
      // (One wave = one instruction stream, and a dependent VALU chain runs at about half its issue rate: the four
      //  rows of a block are evaluated side by side and nothing in the loop branches - a row without a record row
      //  stores into the record's unused `deg` section instead of being masked off.)
      auto epilogue = [&](auto actf, auto recc) {
        constexpr bool REC = decltype(recc)::value;
        const unsigned live_t = sr_t | (cur_in ? 1u << cur_bit : 0u);   // (scalar) slots of the tile with a record row
        const unsigned srl = lh ? sr_t >> 4 : sr_t;   // bit c: the source bit of this lane's row c + 4 lh
        const bool cur_lane = cur_in && ((cur_bit >> 2) & 1) == lh;   // the new row sits in this lane's accumulators
        const unsigned dump = (unsigned)(lay.o_deg - lay.o_rows) + (unsigned)(b * N + (li < N ? li : N - 1));   // from sv_rows0
        float* sv_rows0 = saved + lay.o_rows;   // (uniform base, 32-bit offsets: the host checked the record's size)
        const unsigned gofs = (unsigned)b * (unsigned)N * (unsigned)rw;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const float rcur = sRcur[32 * ct + li];
          float a2 = 0.f, hc = 0.f;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            if (((live_t >> (8 * q4)) & 0xffu) == 0u) continue;   // uniform: no live row among slots 8 q4 .. 8 q4 + 7
            // record position (in floats) of this lane's first stored live row of the block; the next ones follow
            const unsigned below = lh ? (1u << (8 * q4 + 4)) - 1u : (1u << (8 * q4)) - 1u;
            unsigned off = gofs + ((unsigned)rb + (unsigned)__popc(sr_t & below)) * (unsigned)rw + 32u * ct + (unsigned)li;
            const f32x4 rq = crq[ct][q4];
            const bool cur_blk = (cur_bit >> 3) == q4;   // uniform: the block that holds the new row
            float h[4];
            bool st[4];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
              const bool is_cur = cur_blk && 8 * q4 + ii + 4 * lh == cur_bit;
              const float rr = ii == 0 ? rq.x : (ii == 1 ? rq.y : (ii == 2 ? rq.z : rq.w));
              h[ii] = acc[ct][4 * q4 + ii] + (is_cur ? rcur : rr);
            }
            actf(h);   // the four rows side by side
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
              const bool is_cur = cur_blk && 8 * q4 + ii + 4 * lh == cur_bit;
              st[ii] = ((srl >> (8 * q4 + ii)) & 1u) != 0;   // a stored source: in agg2, and it has a record row
              a2 += (st[ii] || (is_cur && self != 0)) ? h[ii] : 0.f;
              hc = is_cur ? h[ii] : hc;
            }
            if (REC) {
#pragma unroll
              for (int ii = 0; ii < 4; ++ii) {
                sv_rows0[st[ii] ? off : dump] = h[ii];
                off += st[ii] ? (unsigned)rw : 0u;
              }
            }
          }
          a2 = gcm_xor32_add(a2);
          if (lh == 0) sA2[wave * HK + 32 * ct + li] = a2;
          if (cur_lane) {
            sV[HK + 32 * ct + li] = hc;
            if (REC) sv_rows[32 * ct + li] = hc;   // record row 0: the new row
          }
        }
      };
      auto with_act = [&](auto recc) {
        // (a wave-uniform "no lane holds a small value: skip gcm_tanh's polynomial arm" was tried: with 256 values a
        //  wave some lane almost always does - 0.13 us slower)
        if (act1 == GCM_ACT_TANH) epilogue([](float (&v)[4]) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = gcm_tanh(v[i]);
        }, recc);
        else if (act1 == GCM_ACT_RELU) epilogue([](float (&v)[4]) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
        }, recc);
        else epilogue([](float (&)[4]) {}, recc);
      };
      if (rec) with_act(std::true_type{});
      else with_act(std::false_type{});
    }
  }
  else if (lh == 0) {   // (a tile without work: its part of agg2 is zero)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) sA2[wave * HK + 32 * ct + li] = 0.f;
  }
  STAMP(8);
  lds_barrier();   // #3
  STAMP(9);

  // ---- layer 2 on the new row (wave 0); the rest of the donated state (the other waves) ---------------------------
  if (wave == 0) {
    const int n_tiles = (n_slots + 31) >> 5;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      if (lh == 0) {
        double t = (double)sA2[32 * ct + li];
        for (int w = 1; w < n_tiles; ++w) t += (double)sA2[w * HK + 32 * ct + li];
        sV[32 * ct + li] = (float)t;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const float* vv = sV + lh * HK;   // lanes 0-31: W_rel2 . agg2, lanes 32-63: W_root2 . h1[cur]
    bool bad = false;
#pragma unroll
    for (int ot = 0; ot < O2T; ++ot) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < HK / 4; ++q) {
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(vv + 4 * q);
        t = fmaf(w2[ot][q].x, x4.x, t);
        t = fmaf(w2[ot][q].y, x4.y, t);
        t = fmaf(w2[ot][q].z, x4.z, t);
        t = fmaf(w2[ot][q].w, x4.w, t);
      }
      t = gcm_xor32_add(t);
      const float y = gcm_act(t + b2v[ot], act2);
      const int o = 32 * ot + li;
      const bool mine = lh == 0 && o < H2;
      if (mine) saved[gb * H2 + o] = y;   // (the record starts with the belief states: mx IS saved[0 .. B H2))
      bad |= mine && !isfinite(y);
    }
    if (__any(bad) && lane == 0) atomicOr(flags, GCM_FLAG_NONFINITE);
    STAMP(10);
    if (rec) {
      for (int k = lane; k < 2 * HK; k += 64) saved[lay.o_v + gb * 2 * HK + k] = sV[k];
      if (lane == 0) {
        int* hdr = reinterpret_cast<int*>(saved + lay.o_hdr) + 4 * b;
        hdr[0] = L; hdr[1] = 0; hdr[2] = steady ? N - 1 : cur; hdr[3] = steady;
      }
    }
    STAMP(11);
  } else {
    const int t2 = tid - 64;   // 0 .. 191
    if (rec) {   // coef: adj[cur, j_l] - the self edge for record row 0, one for every other live row
      float* cf = saved + lay.o_coef + gb * N;
      for (int l = t2; l < L; l += 192) cf[l] = l == 0 ? (self ? 1.f : 0.f) : 1.f;
    }
    if (FUNC) {
      // (the state workgroups write the whole new state, flags included)
    } else if (!steady) {
      // the selectors' entries (below N steps slot = graph row); in the steady state the adjacency is a fixed point
      // of roll + selectors - a Toeplitz pattern built by the same hops at every step - and stays as it is
      for (int j = t2; j <= cur; j += 192) {
        if (j == cur ? self != 0 : mbit(ssrc, j)) ag[cur * N + j] = 1.f;   // row cur (temporal.py:76-81, dense.py:18,20)
        if (j < cur && mbit(scol, j)) ag[j * N + cur] = 1.f;              // column cur (temporal.py:82-87, dense.py:19)
      }
      if (t2 == 191) count[b] = cur + 1;
    } else if (b == 0 && t2 == 191) {
      atomicOr(flags, GCM_FLAG_WRAPPED);   // every graph drops its oldest node (gcm.py:263-271); the count stays N
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same step with EIGHT waves per graph at F = H1 = 32 (the bench shape): 16-slot tiles on v_mfma_f32_16x16x4_f32.
// k_step_colcache's phases are one wave's dependent instruction stream each (in-kernel stamps: ~8 cycles an instruction
// at one wave per SIMD - nothing fills the bubbles of a dependent chain); with two waves per SIMD the other wave does,
// and every per-lane loop halves (8 activations a lane instead of 16, two node rows a thread instead of four).
// Lane (m = lane & 15, g = lane >> 4) of wave w: slot 16 w + m as the A operand's row, k in [8 g, 8 g + 8) of it and of
// W_rel1[col] (col = 16 ct + m) - the matrix instruction's k index is g, so instruction s pairs A[m][8 g + s] with
// B[8 g + s][col]; its accumulators hold slots 16 w + 4 g + (0 .. 3) at column col = one quad of the root cache.
// Same arguments, caches, record and results as k_step_colcache<32, 32, O2T, FUNC> (the sums of a row's 32 products are
// associated differently: eight per lane group instead of sixteen per half).
// ---------------------------------------------------------------------------------------------------------------------
template <int O2T, bool FUNC>
__global__ __launch_bounds__(512) void k_step_colcache8(
    const float* __restrict__ obs, const float* nodes_in, const float* adj_in, const int64_t* count_in, float* nodes,
    float* adj, int64_t* count, const RowMask ssrc, const int self, const RowMask scol, const RowMask sdrop,
    const int cur, const int rot, const int steady, const Gnn2 P, float* __restrict__ cA, float* __restrict__ cR,
    float* __restrict__ saved, const SavedLayout lay, uint32_t* __restrict__ flags, const int N, const int H2,
    const Edits E, const int Bn) {
  constexpr int FK = 32, HK = 32;
  if (FUNC && (int)blockIdx.x >= Bn) {
    // (the state advance is written for GCM_STATE_WGS workgroups of 256 threads per graph: here the two halves of ONE
    //  512-thread workgroup - it has no workgroup barrier and its cross-lane traffic stays inside a wave)
    static_assert(GCM_STATE_WGS == 2, "two 256-thread halves per state workgroup");
    const int bs = blockIdx.x - Bn, ks = threadIdx.x >> 8;
    advance_state_waves<FK>(obs, nodes_in + (size_t)bs * N * FK, adj_in + (size_t)bs * N * N, count_in,
                            nodes + (size_t)bs * N * FK, adj + (size_t)bs * N * N, count, nullptr, E, flags, nullptr,
                            bs, ks, threadIdx.x & 255, N, FK);
    return;
  }
  constexpr int C4 = 8, RG = 64, XP = 2, PS = FK + 4;
  __shared__ __attribute__((aligned(16))) float sPart[RG * PS];
  __shared__ __attribute__((aligned(16))) float sAggc[FK];
  __shared__ float sRcur[HK];
  __shared__ __attribute__((aligned(16))) float sV[2 * HK];
  __shared__ float sA2[8 * HK];

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t gb = (size_t)b;
  const float* ng_in = nodes_in + gb * N * FK;
  float* ng = nodes + gb * N * FK;
  float* ag = adj + gb * N * N;
  float* cAg = cA + gb * N * FK;
  float* cRg = cR + gb * N * HK;
  const bool rec = lay.total != 0;
  const int rw = lay.rw;
  float* sv_rows = saved + lay.o_rows + gb * N * rw;
  const int n_slots = steady ? N : cur + 1;
  // this wave's 16 slots as mask bits (scalars)
  auto half16 = [&](const RowMask& k) { return (mword(k, wave >> 1) >> (16 * (wave & 1))) & 0xffffu; };
  const unsigned sr_t = half16(ssrc), sc_t = half16(scol), sd_t = half16(sdrop);
  const bool cur_in = (cur >> 4) == wave;
  const bool tile_on = 16 * wave < n_slots && ((sr_t | sc_t | sd_t) != 0u || cur_in);

  // ---- every load of the step, in the order of use ------------------------------------------------------------------
  const int64_t n_in = count_in[b];
  f32x4 xo[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) xo[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (steady) {
#pragma unroll
    for (int q = 0; q < 2; ++q) xo[q] = *reinterpret_cast<const f32x4*>(ng_in + 8 * g + 4 * q);
  }
  asm volatile("" ::: "memory");
  const int c4 = tid % C4, rg = tid / C4;   // node rows rg and rg + 64, piece c4
  f32x4 xr[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) xr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int row = rg + RG * i;
    if (i == 0 || RG * i < n_slots)
      xr[i] = *reinterpret_cast<const f32x4*>(ng_in + (row < N ? row : N - 1) * FK + 4 * c4);
  }
  const f32x4 obq = *reinterpret_cast<const f32x4*>(obs + gb * FK + 4 * c4);
  f32x4 xa[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) xa[q] = *reinterpret_cast<const f32x4*>(obs + gb * FK + 8 * g + 4 * q);
  // wave 7: W_root1 (the new node's root row)
  f32x4 wr[2][2];
  float b1v[2];
  if (wave == 7) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
        wr[ct][q] = *reinterpret_cast<const f32x4*>(P.w_root1 + (16 * ct + m) * FK + 8 * g + 4 * q);
      b1v[ct] = P.b_rel1[16 * ct + m];
    }
  }
  const int r = 16 * wave + m;
  const int rc = r < N ? r : N - 1;
  const int NQ = N >> 2;
  f32x4 ca[2], wb[2][2], crq[2];
  if (tile_on) {
#pragma unroll
    for (int q = 0; q < 2; ++q) ca[q] = *reinterpret_cast<const f32x4*>(cAg + rc * FK + 8 * g + 4 * q);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        wb[ct][q] = *reinterpret_cast<const f32x4*>(P.w_rel1 + (16 * ct + m) * FK + 8 * g + 4 * q);
    const int quad = 4 * wave + g;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
      crq[ct] = *reinterpret_cast<const f32x4*>(cRg + ((quad < NQ ? quad : NQ - 1) * HK + 16 * ct + m) * 4);
  }
  // wave 0: layer 2 (lane & 31 = output, lane >> 5 = W_rel2 . agg2 / W_root2 . h1[cur])
  const int li = lane & 31, lh = lane >> 5;
  f32x4 w2[O2T][HK / 4];
  float b2v[O2T];
  if (wave == 0) {
#pragma unroll
    for (int ot = 0; ot < O2T; ++ot) {
      const int o = 32 * ot + li < H2 ? 32 * ot + li : H2 - 1;
      const float* src = (lh ? P.w_root2 : P.w_rel2) + o * HK;
#pragma unroll
      for (int q = 0; q < HK / 4; ++q) w2[ot][q] = *reinterpret_cast<const f32x4*>(src + 4 * q);
      b2v[ot] = P.b_rel2[o];
    }
  }
  const int act1 = P.act1, act2 = P.act2;
  asm volatile("" ::: "memory");

  if (n_in != (int64_t)(steady ? N : cur)) {
    if (tid == 0) atomicOr(flags, GCM_FLAG_BAD_COUNT);
    return;
  }
  const unsigned sw0 = mword(ssrc, 0), sw1 = mword(ssrc, 1), sw2 = mword(ssrc, 2), sw3 = mword(ssrc, 3);
  const int pc1 = 1 + __popc(sw0), pc2 = pc1 + __popc(sw1), pc3 = pc2 + __popc(sw2);
  const int L = pc3 + __popc(sw3);
  auto sword = [&](int w) { return w == 0 ? sw0 : (w == 1 ? sw1 : (w == 2 ? sw2 : sw3)); };
  auto sbase = [&](int w) { return w == 0 ? 1 : (w == 1 ? pc1 : (w == 2 ? pc2 : pc3)); };

  // ---- the new row's aggregate ----------------------------------------------------------------------------------------
  auto new_row_sum = [&](auto ringc) {
    constexpr bool RING = decltype(ringc)::value;
    f32x4 part = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      if (RG * i >= n_slots) continue;   // uniform
      const int row = rg + RG * i;
      const bool is_cur = steady ? row == 0 : row == cur;
      bool stored;
      unsigned pos;
      if (RING) {
        int slot = row + rot;
        slot -= slot >= N ? N : 0;
        stored = row < n_slots && !is_cur && mbit(ssrc, slot);
        pos = 1u + (unsigned)mrank(ssrc, slot);
      } else {
        const int w = 2 * i + (rg >> 5), bit = row & 31;   // (row = rg + 64 i: word 2 i or 2 i + 1)
        const unsigned word = sword(w);
        stored = !is_cur && ((word >> bit) & 1u) != 0;
        pos = (unsigned)sbase(w) + (unsigned)__popc(word & ((1u << bit) - 1u));
      }
      const f32x4 v = is_cur ? obq : xr[i];
      if (stored || (is_cur && self != 0)) part += v;
      if (rec && (stored || is_cur))
        *reinterpret_cast<f32x4*>(sv_rows + (is_cur ? 0u : pos) * (unsigned)rw + HK + FK + 4 * c4) = v;
    }
    *reinterpret_cast<f32x4*>(sPart + rg * PS + 4 * c4) = part;
  };
  if (rot) new_row_sum(std::true_type{});
  else new_row_sum(std::false_type{});
  if (wave == 7) {   // the new node's root row: lane (col m, group g) takes eight k
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        t = fmaf(wr[ct][q].x, xa[q].x, t);
        t = fmaf(wr[ct][q].y, xa[q].y, t);
        t = fmaf(wr[ct][q].z, xa[q].z, t);
        t = fmaf(wr[ct][q].w, xa[q].w, t);
      }
      t = gcm_xor16_add(t);
      t = gcm_xor32_add(t);
      const float v = t + b1v[ct];
      if (g == 0) {
        sRcur[16 * ct + m] = v;
        cRg[((cur >> 2) * HK + 16 * ct + m) * 4 + (cur & 3)] = v;
      }
    }
  }
  lds_barrier();   // #1
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    if (FUNC || RG * i >= n_slots) continue;
    const int row = rg + RG * i;
    if (steady) {
      if (row < N) *reinterpret_cast<f32x4*>(ng + (row == 0 ? N - 1 : row - 1) * FK + 4 * c4) = row == 0 ? obq : xr[i];
    } else if (row == cur) {
      *reinterpret_cast<f32x4*>(ng + cur * FK + 4 * c4) = obq;
    }
  }
  if (wave == 6) {   // 64 row groups -> agg1[cur]: lane (feature, half) takes 32 groups
    const int f = lane & 31, part = lane >> 5;
    float v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = sPart[(part * 32 + k) * PS + f];
    float s0 = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) s0 += v[k];
    s0 = gcm_xor32_add(s0);
    if (part == 0) {
      sAggc[f] = s0;
      cAg[cur * FK + f] = s0;
      if (rec) sv_rows[HK + f] = s0;
    }
  }
  lds_barrier();   // #2

  // ---- layer 1 of the live rows: [16 slots x 32] . [32 x 32] per wave ----------------------------------------------
  if (tile_on) {
    const int rb = sbase(wave >> 1) + ((wave & 1) ? __popc(sword(wave >> 1) & 0xffffu) : 0);
    const int cur_bit = cur_in ? cur & 15 : 99;
    const bool upd = ((sc_t >> m) & 1u) != 0, drp = ((sd_t >> m) & 1u) != 0, lrow = ((sr_t >> m) & 1u) != 0;
    const unsigned pos = (unsigned)rb + (unsigned)__popc(sr_t & ((1u << m) - 1u));
    float* rdst = sv_rows + pos * (unsigned)rw + HK + 8 * g;
    float a[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x4 v = ca[q];
      const f32x4 xq = xa[q], xd = xo[q];
      if (steady) v -= f32x4{drp ? xd.x : 0.f, drp ? xd.y : 0.f, drp ? xd.z : 0.f, drp ? xd.w : 0.f};
      v += f32x4{upd ? xq.x : 0.f, upd ? xq.y : 0.f, upd ? xq.z : 0.f, upd ? xq.w : 0.f};
      if (upd || drp) *reinterpret_cast<f32x4*>(cAg + r * FK + 8 * g + 4 * q) = v;
      if (m == cur_bit) v = *reinterpret_cast<const f32x4*>(sAggc + 8 * g + 4 * q);
      if (rec && lrow) *reinterpret_cast<f32x4*>(rdst + 4 * q) = v;
      a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
    f32x4 acc[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const f32x4 w = wb[ct][s >> 2];
        const float wv = (s & 3) == 0 ? w.x : ((s & 3) == 1 ? w.y : ((s & 3) == 2 ? w.z : w.w));
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], wv, acc[ct], 0, 0, 0);
      }
    // the activation of this lane's four slots (4 g + 0 .. 3) at columns m and 16 + m, side by side, no branch inside
    auto epilogue = [&](auto actf, auto recc) {
      constexpr bool REC = decltype(recc)::value;
      const unsigned dump = (unsigned)(lay.o_deg - lay.o_rows) + (unsigned)(b * N + (m < N ? m : N - 1));
      float* sv_rows0 = saved + lay.o_rows;
      const unsigned gofs = (unsigned)b * (unsigned)N * (unsigned)rw;
      const unsigned off0 = gofs + ((unsigned)rb + (unsigned)__popc(sr_t & ((1u << (4 * g)) - 1u))) * (unsigned)rw + (unsigned)m;
      const bool cur_grp = cur_in && (cur_bit >> 2) == g;   // the new row sits in this lane's accumulators
      float h[2][4];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const float rcur = sRcur[16 * ct + m];
        const f32x4 rq = crq[ct];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const bool is_cur = cur_grp && (cur_bit & 3) == ii;
          const float rr = ii == 0 ? rq.x : (ii == 1 ? rq.y : (ii == 2 ? rq.z : rq.w));
          h[ct][ii] = acc[ct][ii] + (is_cur ? rcur : rr);
        }
      }
      actf(h);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float a2 = 0.f, hc = 0.f;
        unsigned off = off0 + 16u * ct;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const bool is_cur = cur_grp && (cur_bit & 3) == ii;
          const bool st = ((sr_t >> (4 * g + ii)) & 1u) != 0;
          a2 += (st || (is_cur && self != 0)) ? h[ct][ii] : 0.f;
          hc = is_cur ? h[ct][ii] : hc;
          if (REC) {
            sv_rows0[st ? off : dump] = h[ct][ii];
            off += st ? (unsigned)rw : 0u;
          }
        }
        a2 = gcm_xor16_add(a2);
        a2 = gcm_xor32_add(a2);
        if (g == 0) sA2[wave * HK + 16 * ct + m] = a2;
        if (cur_grp) {
          sV[HK + 16 * ct + m] = hc;
          if (REC) sv_rows[16 * ct + m] = hc;   // record row 0: the new row
        }
      }
    };
    auto with_act = [&](auto recc) {
      if (act1 == GCM_ACT_TANH) epilogue([](float (&v)[2][4]) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i) v[c][i] = gcm_tanh(v[c][i]);
      }, recc);
      else if (act1 == GCM_ACT_RELU) epilogue([](float (&v)[2][4]) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i) v[c][i] = v[c][i] > 0.f ? v[c][i] : 0.f;
      }, recc);
      else epilogue([](float (&)[2][4]) {}, recc);
    };
    if (rec) with_act(std::true_type{});
    else with_act(std::false_type{});
  } else if (g == 0) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) sA2[wave * HK + 16 * ct + m] = 0.f;
  }
  lds_barrier();   // #3

  if (wave == 0) {
    const int n_tiles = (n_slots + 15) >> 4;
    if (lh == 0) {
      double t = (double)sA2[li];
      for (int w = 1; w < n_tiles; ++w) t += (double)sA2[w * HK + li];
      sV[li] = (float)t;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const float* vv = sV + lh * HK;
    bool bad = false;
#pragma unroll
    for (int ot = 0; ot < O2T; ++ot) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < HK / 4; ++q) {
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(vv + 4 * q);
        t = fmaf(w2[ot][q].x, x4.x, t);
        t = fmaf(w2[ot][q].y, x4.y, t);
        t = fmaf(w2[ot][q].z, x4.z, t);
        t = fmaf(w2[ot][q].w, x4.w, t);
      }
      t = gcm_xor32_add(t);
      const float y = gcm_act(t + b2v[ot], act2);
      const int o = 32 * ot + li;
      const bool mine = lh == 0 && o < H2;
      if (mine) saved[gb * H2 + o] = y;
      bad |= mine && !isfinite(y);
    }
    if (__any(bad) && lane == 0) atomicOr(flags, GCM_FLAG_NONFINITE);
    if (rec) {
      for (int k = lane; k < 2 * HK; k += 64) saved[lay.o_v + gb * 2 * HK + k] = sV[k];
      if (lane == 0) {
        int* hdr = reinterpret_cast<int*>(saved + lay.o_hdr) + 4 * b;
        hdr[0] = L; hdr[1] = 0; hdr[2] = steady ? N - 1 : cur; hdr[3] = steady;
      }
    }
  } else {
    const int t2 = tid - 64;   // 0 .. 447
    if (rec) {
      float* cf = saved + lay.o_coef + gb * N;
      for (int l = t2; l < L; l += 448) cf[l] = l == 0 ? (self ? 1.f : 0.f) : 1.f;
    }
    if (FUNC) {
    } else if (!steady) {
      for (int j = t2; j <= cur; j += 448) {
        if (j == cur ? self != 0 : mbit(ssrc, j)) ag[cur * N + j] = 1.f;
        if (j < cur && mbit(scol, j)) ag[j * N + cur] = 1.f;
      }
      if (t2 == 447) count[b] = cur + 1;
    } else if (b == 0 && t2 == 447) {
      atomicOr(flags, GCM_FLAG_WRAPPED);
    }
  }
}

}  // namespace gcm_rows

// The selector chain and the number of steps the chain has made -> the masks of k_step_colcache in ring coordinates.
// Graph coordinates first: c = the graph row the new node lands in (t below N steps, N - 1 after), sources / column
// rows among the rows < c, the rows that lose the dropped node among the incoming rows >= 1; then graph row i of the
// state AFTER the step sits in slot (rot + 1 + i) mod N in the steady state (rot = t mod N: the slot the dropped node
// vacates), in slot i before it.
struct ColMasks {
  gcm_rows::RowMask ssrc, scol, sdrop;
  int self, cur, rot, steady;
  bool writes_column;
};
static bool colcache_masks(const gcm_selector_desc* selectors, int n_selectors, int N, int t, ColMasks* out) {
  ColMasks m{};
  auto set = [](gcm_rows::RowMask& k, int j) {
    if (j < 64) k.lo |= 1ull << j;
    else k.hi |= 1ull << (j - 64);
  };
  m.steady = t >= N ? 1 : 0;
  m.rot = m.steady ? t % N : 0;
  const int c = m.steady ? N - 1 : t;
  m.cur = m.steady ? m.rot : t;
  auto post = [&](int i) { return m.steady ? (m.rot + 1 + i) % N : i; };   // slot of graph row i after the step
  auto pre = [&](int i) { return (m.rot + i) % N; };                       // ... of graph row i as the state comes in
  for (int i = 0; i < n_selectors; ++i) {
    const gcm_selector_desc& d = selectors[i];
    if (d.kind == GCM_SEL_TEMPORAL) {
      if (d.n_hops < 0 || d.n_hops > 16) return false;
      for (int k = 0; k < d.n_hops; ++k) {
        const int h = d.hops[k];
        if (h < 0) return false;
        if (d.direction & GCM_DIR_BACKWARD) m.writes_column = m.writes_column || h > 0;
        if (h > c) continue;                                       // temporal.py:74: num_nodes >= hop
        if (h == 0) { m.self = 1; continue; }
        if (d.direction & GCM_DIR_FORWARD) set(m.ssrc, post(c - h));
        if (d.direction & GCM_DIR_BACKWARD) set(m.scol, post(c - h));
        if (m.steady && (d.direction & GCM_DIR_FORWARD) && h <= N - 1) set(m.sdrop, pre(h));   // entry (h, 0)
      }
    } else if (d.kind == GCM_SEL_DENSE) {
      m.writes_column = true;
      m.self = 1;
      for (int j = 0; j < c; ++j) { set(m.ssrc, post(j)); set(m.scol, post(j)); }
      if (m.steady) for (int j = 1; j < N; ++j) set(m.sdrop, pre(j));
    } else {
      return false;
    }
  }
  *out = m;
  return true;
}

extern "C" int gcm_dense_rows_colcache_supported(const gcm_selector_desc* selectors, int n_selectors, int has_bias,
                                                 int N, int F, int H1, int H2) {
  if (N <= 0 || N > 128 || (N & 3) || !(F == 32 || F == 64) || !(H1 == 32 || H1 == 64) || H2 <= 0 || H2 > 64) return 0;
  if (has_bias & ~(3 | GCM_STEP_FOUR_WAVES)) return 0;   // (no folded preprocessor / positional encoding, no dx record)
  if (n_selectors <= 0 || !selectors) return 0;
  ColMasks m;
  if (!colcache_masks(selectors, n_selectors, N, 0, &m)) return 0;
  return m.writes_column ? 1 : 0;   // (chains that only ever write row cur have the one-wave cached step of rows_cached.hip)
}

static int colcache_launch(const float* obs, const float* nodes_in, const float* adj_in, const int64_t* count_in,
                           float* nodes_out, float* adj_out, int64_t* count_out, const gcm_selector_desc* selectors,
                           int n_selectors, const float* params, int has_bias, int act1, int act2, float* cache_agg1,
                           float* cache_root, float* saved, int record, int cur_host, uint32_t* flags, int B, int N,
                           int F, int H1, int H2, gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes_in && adj_in && count_in && nodes_out && adj_out && count_out && params && cache_agg1 &&
              cache_root && saved && flags);
  GCM_REQUIRE(B > 0 && cur_host >= 0);
  GCM_REQUIRE((nodes_out == nodes_in) == (adj_out == adj_in) && (nodes_out == nodes_in) == (count_out == count_in));
  if (!gcm_dense_rows_colcache_supported(selectors, n_selectors, has_bias, N, F, H1, H2)) return GCM_EUNSUPPORTED;
  if ((size_t)B * N * (size_t)(F > N ? F : N) >= ((size_t)1 << 31)) return GCM_EUNSUPPORTED;
  ColMasks m;
  if (!colcache_masks(selectors, n_selectors, N, cur_host, &m)) return GCM_EUNSUPPORTED;
  gcm_fused::Edits E{};
  for (int i = 0; i < n_selectors; ++i) {   // (the state workgroups of the functional form take the chain as hops)
    const gcm_selector_desc& d = selectors[i];
    if (d.kind == GCM_SEL_DENSE) E.dense = 1;
    for (int k = 0; d.kind == GCM_SEL_TEMPORAL && k < d.n_hops; ++k) {
      if (E.n_hops >= 16) return GCM_EUNSUPPORTED;
      E.hops[E.n_hops] = d.hops[k];
      E.dir[E.n_hops++] = d.direction;
    }
  }
  const float* w_rel1 = params;
  const float* w_root1 = w_rel1 + (size_t)H1 * F;
  const float* b1 = w_root1 + (size_t)H1 * F;
  const float* w_rel2 = b1 + H1;
  const float* w_root2 = w_rel2 + (size_t)H2 * H1;
  const float* b2 = w_root2 + (size_t)H2 * H1;
  const gcm_fused::Gnn2 P{w_rel1, b1, w_root1, w_rel2, b2, w_root2, act1, act2};
  gcm_rows::SavedLayout lay = gcm_rows::make_layout(B, N, F, H1, H2, false);
  if (lay.total >= ((size_t)1 << 32)) return GCM_EUNSUPPORTED;   // (32-bit float offsets into the record)
  if (!record) lay.total = 0;
  hipStream_t s = (hipStream_t)stream;
  const bool func = nodes_out != nodes_in;
  if (F == 32 && H1 == 32 && !(has_bias & GCM_STEP_FOUR_WAVES)) {   // eight waves per graph (16-slot tiles)
#define GCM_C8(c)                                                                                                   \
  if ((H2 <= 32 ? 1 : 2) == c) {                                                                                    \
    if (func)                                                                                                       \
      hipLaunchKernelGGL((gcm_rows::k_step_colcache8<c, true>), dim3(2 * B), dim3(512), 0, s, obs,                      \
                         nodes_in, adj_in, count_in, nodes_out, adj_out, count_out, m.ssrc, m.self, m.scol, m.sdrop,    \
                         m.cur, m.rot, m.steady, P, cache_agg1, cache_root, saved, lay, flags, N, H2, E, B);            \
    else                                                                                                            \
      hipLaunchKernelGGL((gcm_rows::k_step_colcache8<c, false>), dim3(B), dim3(512), 0, s, obs, nodes_in, adj_in,       \
                         count_in, nodes_out, adj_out, count_out, m.ssrc, m.self, m.scol, m.sdrop, m.cur, m.rot,        \
                         m.steady, P, cache_agg1, cache_root, saved, lay, flags, N, H2, E, B);                          \
    return gcm_launch_status();                                                                                     \
  }
    GCM_C8(1) GCM_C8(2)
#undef GCM_C8
  }
#define GCM_CC(a, b_, c)                                                                                          \
  if (F == a && H1 == b_ && (H2 <= 32 ? 1 : 2) == c) {                                                           \
    if (func)                                                                                                     \
      hipLaunchKernelGGL((gcm_rows::k_step_colcache<a, b_, c, true>), dim3((1 + GCM_STATE_WGS) * B), dim3(256), 0, s, \
                         obs, nodes_in, adj_in, count_in, nodes_out, adj_out, count_out, m.ssrc, m.self, m.scol,      \
                         m.sdrop, m.cur, m.rot, m.steady, P, cache_agg1, cache_root, saved, lay, flags, N, H2, E, B); \
    else                                                                                                          \
      hipLaunchKernelGGL((gcm_rows::k_step_colcache<a, b_, c, false>), dim3(B), dim3(256), 0, s, obs, nodes_in,       \
                         adj_in, count_in, nodes_out, adj_out, count_out, m.ssrc, m.self, m.scol, m.sdrop, m.cur,     \
                         m.rot, m.steady, P, cache_agg1, cache_root, saved, lay, flags, N, H2, E, B);                 \
    return gcm_launch_status();                                                                                   \
  }
  GCM_CC(32, 32, 1) GCM_CC(32, 32, 2) GCM_CC(64, 32, 1) GCM_CC(64, 32, 2)
  GCM_CC(32, 64, 1) GCM_CC(32, 64, 2) GCM_CC(64, 64, 1) GCM_CC(64, 64, 2)
#undef GCM_CC
  return GCM_EUNSUPPORTED;
}

extern "C" int gcm_dense_rows_step_colcache(const float* obs, float* nodes, float* adj, int64_t* count,
                                            const gcm_selector_desc* selectors, int n_selectors, const float* params,
                                            int has_bias, int act1, int act2, float* cache_agg1, float* cache_root,
                                            float* saved, int record, int cur_host, uint32_t* flags, int B, int N,
                                            int F, int H1, int H2, gcm_stream_t stream) {
  return colcache_launch(obs, nodes, adj, count, nodes, adj, count, selectors, n_selectors, params, has_bias, act1, act2,
                         cache_agg1, cache_root, saved, record, cur_host, flags, B, N, F, H1, H2, stream);
}

extern "C" int gcm_dense_rows_step_colcache_functional(const float* obs, const float* nodes_in, const float* adj_in,
                                                       const int64_t* count_in, float* nodes_out, float* adj_out,
                                                       int64_t* count_out, const gcm_selector_desc* selectors,
                                                       int n_selectors, const float* params, int has_bias, int act1,
                                                       int act2, float* cache_agg1, float* cache_root, float* saved,
                                                       int record, int cur_host, uint32_t* flags, int B, int N, int F,
                                                       int H1, int H2, gcm_stream_t stream) {
  GCM_REQUIRE(nodes_out != nodes_in && adj_out != adj_in && count_out != count_in);
  return colcache_launch(obs, nodes_in, adj_in, count_in, nodes_out, adj_out, count_out, selectors, n_selectors, params,
                         has_bias, act1, act2, cache_agg1, cache_root, saved, record, cur_host, flags, B, N, F, H1, H2,
                         stream);
}
