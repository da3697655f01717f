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

template <int FK, int HK, int O2T>
__global__ __launch_bounds__(256) void k_step_colcache(
    const float* __restrict__ obs, float* __restrict__ nodes, float* __restrict__ adj, int64_t* __restrict__ count,
    const RowMask srow, const RowMask scol, const int cur, const Gnn2 P, float* __restrict__ cA,
    float* __restrict__ cR, float* __restrict__ saved, const SavedLayout lay, uint32_t* __restrict__ flags,
    const int N, const int H2) {
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
  float* ng = nodes + gb * N * FK;
  float* ag = adj + gb * N * N;
  float* cAg = cA + gb * N * FK;
  float* cRg = cR + gb * N * HK;
  const bool rec = lay.total != 0;
  const int rw = lay.rw;
  float* sv_rows = saved + lay.o_rows + gb * N * rw;
  const bool tile_on = 32 * wave <= cur;          // wave-uniform: this wave's 32 rows hold a row <= cur

  STAMP(0);
  // ---- every load of the step, in the order of use, before anything waits ------------------------------------------
  const int64_t n_in = count[b];
  // node rows, one 16-byte piece per (row group, piece) thread: the new row's aggregate and the record's x section
  const int c4 = tid % C4, rg = tid / C4;
  f32x4 xr[XP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int row = rg + RG * i;
    if (RG * i <= cur)   // (uniform: passes beyond the stored rows are skipped)
      xr[i] = *reinterpret_cast<const f32x4*>(ng + (row < N ? row : N - 1) * FK + 4 * c4);
    else
      xr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
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
  // the A operand: row r = 32 wave + li, k in [lh KH, (lh + 1) KH) - agg1 of the stored rows from the chain's cache;
  // the B operand: W_rel1[col][k], col = 32 ct + li, the same k; root[j][col] of the rows the accumulators hold
  const int r = 32 * wave + li;
  const int rc = r < N ? r : N - 1;
  f32x4 ca[KQ], wb[CT][KQ], crq[CT][4];
  const int NQ = N >> 2;   // the root cache is kept in quads of rows: cR[b][row / 4][col][row % 4] (one 16-byte load
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

  // A chain from empty graphs holds `cur` nodes in every graph; anything else (a caller edited the count) leaves the
  // graph untouched and raises the flag (uniform per workgroup: nothing has been stored yet)
  if (n_in != (int64_t)cur) {
    if (tid == 0) atomicOr(flags, GCM_FLAG_BAD_COUNT);
    return;
  }
  STAMP(2);
  // the masks as four 32-row words (scalars) and the live-list position of each word's first row: slot 0 is row cur,
  // the stored live rows follow in ascending order
  const unsigned sw0 = mword(srow, 0), sw1 = mword(srow, 1), sw2 = mword(srow, 2), sw3 = mword(srow, 3);
  const unsigned selfbit = mbit(srow, cur) ? 1u : 0u;
  const int pc1 = 1 + __popc(sw0), pc2 = pc1 + __popc(sw1), pc3 = pc2 + __popc(sw2);
  const int L = pc3 + __popc(sw3) - (int)selfbit;   // row cur + the stored sources
  auto sword = [&](int w) { return w == 0 ? sw0 : (w == 1 ? sw1 : (w == 2 ? sw2 : sw3)); };
  auto sbase = [&](int w) { return w == 0 ? 1 : (w == 1 ? pc1 : (w == 2 ? pc2 : pc3)); };

  // ---- the new row's aggregate: sum of the selected node rows, ascending inside a thread, then over the row groups
  {
    f32x4 part = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      if (RG * i > cur) continue;   // uniform
      const int row = rg + RG * i;
      const int w = (RG * i) >> 5, bit = row & 31;     // w: compile time
      const unsigned word = sword(w);
      const bool is_cur = row == cur;
      const f32x4 v = is_cur ? obq : xr[i];
      const bool src = ((word >> bit) & 1u) != 0;      // (bits beyond cur are never set)
      if (src) part += v;
      if (rec && (is_cur || src)) {
        const unsigned slot = is_cur ? 0u : (unsigned)sbase(w) + (unsigned)__popc(word & ((1u << bit) - 1u));
        *reinterpret_cast<f32x4*>(sv_rows + slot * (unsigned)rw + HK + FK + 4 * c4) = v;
      }
    }
    *reinterpret_cast<f32x4*>(sPart + rg * PS + 4 * c4) = part;
  }
  // the new node's root row (wave 3 holds no stored row until cur >= 96)
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
      t += __shfl_xor(t, 32);
      const float v = t + b1v[ct];
      if (lh == 0) {
        sRcur[32 * ct + li] = v;
        cRg[((cur >> 2) * HK + 32 * ct + li) * 4 + (cur & 3)] = v;
      }
    }
  }
  STAMP(3);
  lds_barrier();   // #1
  STAMP(4);
  if (wave == 2) {   // the row groups' partial sums -> agg1[cur]: lane (feature, part) takes GP groups
    const int f = lane % FK, part = lane / FK;
    float v[GP];
#pragma unroll
    for (int g = 0; g < GP; ++g) v[g] = sPart[(part * GP + g) * PS + f];
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < GP; ++g) s += v[g];
    if (SPLIT == 2) s += __shfl_xor(s, 32);
    if (part == 0) {
      sAggc[f] = s;
      cAg[cur * FK + f] = s;
      if (rec) sv_rows[HK + f] = s;
    }
  }
  lds_barrier();   // #2
  STAMP(5);

  // ---- layer 1 of the live rows: [32 rows x F] . [F x H1] per wave ------------------------------------------------
  if (tile_on) {
    const unsigned sr_t = sword(wave), sc_t = mword(scol, wave);   // scalars
    const int rb = sbase(wave);
    const int cur_bit = cur - 32 * wave;                            // row cur inside this tile: 0 .. 31
    {
      const bool upd = ((sc_t >> li) & 1u) != 0;        // the row gains the source cur (stored rows only)
      const bool lrow = ((sr_t >> li) & 1u) != 0 && li != cur_bit;   // a stored live row
      const int slot = rb + __popc(sr_t & ((1u << li) - 1u));
      float* rdst = sv_rows + (unsigned)slot * (unsigned)rw + HK + lh * KH;
      float a[KH];
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        f32x4 v = ca[q];
        const f32x4 xq = xa[q];
        v += f32x4{upd ? xq.x : 0.f, upd ? xq.y : 0.f, upd ? xq.z : 0.f, upd ? xq.w : 0.f};
        if (upd) *reinterpret_cast<f32x4*>(cAg + r * FK + lh * KH + 4 * q) = v;
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
      // the activation, this lane's part of agg2, h1[cur], the record's h1 rows.  One wave = one instruction stream:
      // the activation is a uniform branch around the loop (not selects inside it), positions come from the tile's
      // mask word
      // (One wave = one instruction stream, and a dependent VALU chain runs at about half its issue rate: the four
      //  rows of a block are evaluated side by side and nothing in the loop branches - a row without a record row
      //  stores into the record's unused `deg` section instead of being masked off.)
      auto epilogue = [&](auto actf, auto recc) {
        constexpr bool REC = decltype(recc)::value;
        const unsigned live_t = sr_t | (cur_bit < 32 ? 1u << cur_bit : 0u);   // (scalar) rows of the tile with a record row
        const unsigned srl = lh ? sr_t >> 4 : sr_t;   // bit c: the source bit of this lane's row c + 4 lh
        const bool cur_lane = cur_bit < 32 && ((cur_bit >> 2) & 1) == lh;   // row cur sits in this lane's accumulators
        const unsigned dump = (unsigned)(lay.o_deg - lay.o_rows) + (unsigned)(b * N + (li < N ? li : N - 1));   // from sv_rows0
        float* sv_rows0 = saved + lay.o_rows;   // (uniform base, 32-bit offsets: B N rw < 2^31 is checked by the host)
        const unsigned gofs = (unsigned)b * (unsigned)N * (unsigned)rw;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const float rcur = sRcur[32 * ct + li];
          float a2 = 0.f, hc = 0.f;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            if (((live_t >> (8 * q4)) & 0xffu) == 0u) continue;   // uniform: no live row among rows 8 q4 .. 8 q4 + 7
            // record position (in floats) of this lane's first stored live row of the block; the next ones follow
            const unsigned below = lh ? (1u << (8 * q4 + 4)) - 1u : (1u << (8 * q4)) - 1u;
            unsigned off = gofs + ((unsigned)rb + (unsigned)__popc(sr_t & below)) * (unsigned)rw + 32u * ct + (unsigned)li;
            const f32x4 rq = crq[ct][q4];
            const bool cur_blk = (cur_bit >> 3) == q4;   // uniform: the block that holds row cur
            float h[4];
            bool st[4];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
              const bool is_cur = cur_blk && 8 * q4 + ii + 4 * lh == cur_bit;
              const float rr = ii == 0 ? rq.x : (ii == 1 ? rq.y : (ii == 2 ? rq.z : rq.w));
              h[ii] = actf(acc[ct][4 * q4 + ii] + (is_cur ? rcur : rr));
              const bool src = ((srl >> (8 * q4 + ii)) & 1u) != 0;
              a2 += src ? h[ii] : 0.f;
              hc = is_cur ? h[ii] : hc;
              st[ii] = src && !is_cur;   // a stored live row: it has a record row (row cur's is slot 0, below)
            }
            if (REC) {
#pragma unroll
              for (int ii = 0; ii < 4; ++ii) {
                sv_rows0[st[ii] ? off : dump] = h[ii];
                off += st[ii] ? (unsigned)rw : 0u;
              }
            }
          }
          a2 += __shfl_xor(a2, 32);
          if (lh == 0) sA2[wave * HK + 32 * ct + li] = a2;
          if (cur_lane) {
            sV[HK + 32 * ct + li] = hc;
            if (REC) sv_rows[32 * ct + li] = hc;   // slot 0: row cur
          }
        }
      };
      auto with_act = [&](auto recc) {
        if (act1 == GCM_ACT_TANH) epilogue([](float v) { return gcm_tanh(v); }, recc);
        else if (act1 == GCM_ACT_RELU) epilogue([](float v) { return v > 0.f ? v : 0.f; }, recc);
        else epilogue([](float v) { return v; }, recc);
      };
      if (rec) with_act(std::true_type{});
      else with_act(std::false_type{});
    }
  }
  STAMP(8);
  lds_barrier();   // #3
  STAMP(9);

  // ---- layer 2 on row cur (wave 0); the donated state (the other waves) ---------------------------------------------
  if (wave == 0) {
    const int wmax = cur >> 5;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      if (lh == 0) {
        double t = (double)sA2[32 * ct + li];
        for (int w = 1; w <= wmax; ++w) t += (double)sA2[w * HK + 32 * ct + li];
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
      t += __shfl_xor(t, 32);
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
        hdr[0] = L; hdr[1] = 0; hdr[2] = cur; hdr[3] = 0;
      }
    }
    STAMP(11);
  } else {
    const int t2 = tid - 64;   // 0 .. 191
    if (rec) {   // coef: adj[cur, j_l] - the self edge for slot 0, one for every other live row
      float* cf = saved + lay.o_coef + gb * N;
      for (int l = t2; l < L; l += 192) cf[l] = l == 0 ? (selfbit ? 1.f : 0.f) : 1.f;
    }
    for (int j = t2; j <= cur; j += 192) {
      if (mbit(srow, j)) ag[cur * N + j] = 1.f;              // row cur (temporal.py:76-81, dense.py:18,20)
      if (j < cur && mbit(scol, j)) ag[j * N + cur] = 1.f;   // column cur (temporal.py:82-87, dense.py:19)
    }
    if (t2 < C4) *reinterpret_cast<f32x4*>(ng + cur * FK + 4 * t2) =
        *reinterpret_cast<const f32x4*>(obs + gb * FK + 4 * t2);   // gcm.py:274
    if (t2 == 191) count[b] = cur + 1;
  }
}

}  // namespace gcm_rows

// the selector chain and the row the new node lands in -> the sources of row cur / the rows that gain the source cur
static bool colcache_masks(const gcm_selector_desc* selectors, int n_selectors, int cur, gcm_rows::RowMask* srow,
                           gcm_rows::RowMask* scol, bool* writes_column) {
  gcm_rows::RowMask r{0, 0}, c{0, 0};
  auto set = [](gcm_rows::RowMask& m, int j) {
    if (j < 64) m.lo |= 1ull << j;
    else m.hi |= 1ull << (j - 64);
  };
  bool col = false;
  for (int i = 0; i < n_selectors; ++i) {
    const gcm_selector_desc& d = selectors[i];
    if (d.kind == GCM_SEL_TEMPORAL) {
      if (d.n_hops < 0 || d.n_hops > 16) return false;
      for (int k = 0; k < d.n_hops; ++k) {
        const int h = d.hops[k];
        if (h < 0) return false;
        if (d.direction & GCM_DIR_BACKWARD) col = col || h > 0;
        if (h > cur) continue;                                     // temporal.py:74: num_nodes >= hop
        if ((d.direction & GCM_DIR_FORWARD) || h == 0) set(r, cur - h);
        if ((d.direction & GCM_DIR_BACKWARD) && h > 0) set(c, cur - h);
      }
    } else if (d.kind == GCM_SEL_DENSE) {
      col = true;
      for (int j = 0; j <= cur; ++j) set(r, j);
      for (int j = 0; j < cur; ++j) set(c, j);
    } else {
      return false;
    }
  }
  *srow = r;
  *scol = c;
  if (writes_column) *writes_column = col;
  return true;
}

extern "C" int gcm_dense_rows_colcache_supported(const gcm_selector_desc* selectors, int n_selectors, int has_bias,
                                                 int N, int F, int H1, int H2) {
  if (N <= 0 || N > 128 || (N & 3) || !(F == 32 || F == 64) || !(H1 == 32 || H1 == 64) || H2 <= 0 || H2 > 64) return 0;
  if (has_bias & ~3) return 0;   // (no folded preprocessor / positional encoding, no observation-gradient record)
  if (n_selectors <= 0 || !selectors) return 0;
  gcm_rows::RowMask a, c;
  bool col = false;
  if (!colcache_masks(selectors, n_selectors, 0, &a, &c, &col)) return 0;
  return col ? 1 : 0;   // (chains that only ever write row cur have the one-wave cached step of rows_cached.hip)
}

extern "C" int gcm_dense_rows_step_colcache(const float* obs, float* nodes, float* adj, int64_t* count,
                                            const gcm_selector_desc* selectors, int n_selectors, const float* params,
                                            int has_bias, int act1, int act2, float* cache_agg1, float* cache_root,
                                            float* saved, int record, int cur_host, uint32_t* flags, int B, int N,
                                            int F, int H1, int H2, gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes && adj && count && params && cache_agg1 && cache_root && saved && flags);
  GCM_REQUIRE(B > 0 && cur_host >= 0);
  if (!gcm_dense_rows_colcache_supported(selectors, n_selectors, has_bias, N, F, H1, H2)) return GCM_EUNSUPPORTED;
  if (cur_host >= N) return GCM_EUNSUPPORTED;   // (a full graph rolls: the caches stop being valid)
  if ((size_t)B * N * (size_t)(F > N ? F : N) >= ((size_t)1 << 31)) return GCM_EUNSUPPORTED;
  gcm_rows::RowMask srow, scol;
  if (!colcache_masks(selectors, n_selectors, cur_host, &srow, &scol, nullptr)) return GCM_EUNSUPPORTED;
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
#define GCM_CC(a, b_, c)                                                                                          \
  if (F == a && H1 == b_ && (H2 <= 32 ? 1 : 2) == c) {                                                           \
    hipLaunchKernelGGL((gcm_rows::k_step_colcache<a, b_, c>), dim3(B), dim3(256), 0, s, obs, nodes, adj, count,     \
                       srow, scol, cur_host, P, cache_agg1, cache_root, saved, lay, flags, N, H2);                  \
    return gcm_launch_status();                                                                                   \
  }
  GCM_CC(32, 32, 1) GCM_CC(32, 32, 2) GCM_CC(64, 32, 1) GCM_CC(64, 32, 2)
  GCM_CC(32, 64, 1) GCM_CC(32, 64, 2) GCM_CC(64, 64, 1) GCM_CC(64, 64, 2)
#undef GCM_CC
  return GCM_EUNSUPPORTED;
}
