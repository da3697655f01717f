// The functional state advance of the live-row step kernels as extra workgroups of the same launch (rows_step.hip:
// k_step_rows<.., FUNC = true>; rows_colcache.hip: k_step_colcache<.., FUNC = true>).
#pragma once
#include "fused_common.h"
#include "state_copy.h"

namespace gcm_rows {

using gcm_fused::Edits;
using gcm_state::load_copy;
using gcm_state::store_copy;

// Functional state (distinct output buffers): the state advance of gcm.py:262-287 - copy, overflow
// roll, the selectors' entries, the inserted node, the count - is a pure function of the incoming
// state, independent of the GNN.  It runs in EXTRA WORKGROUPS of the same launch (blocks >= B,
// GCM_STATE_WGS per graph), which stream the graph's 80 KB HBM -> registers -> HBM with the edits applied in registers;
// the graph's compute workgroup (block b) only reads the old state - through shifted addresses when
// the graph rolls - so the two never wait for each other.  (Round 2 moved the copy through the compute
// waves' registers: its stores had to wait for the end of the kernel - loads and stores share one
// in-order counter per wave - 11.8 us against 6.2 us donated.  Extra waves in the SAME workgroup do
// not work either: s_barrier counts every wave that has not terminated, so the compute waves' first
// barrier waited for the whole copy - measured: copy alone 8.4 us, compute alone 5.9 us, both 12.2 us.)
#ifndef GCM_STATE_CH
#define GCM_STATE_CH 2
#endif
#ifndef GCM_STATE_WGS
#define GCM_STATE_WGS 2   // state workgroups per graph (each moves every GCM_STATE_WGS-th 4 KB slice)
#endif
template <int FP>
__device__ __forceinline__ void advance_state_waves(
    const float* __restrict__ obs, const float* ng_in, const float* ag_in, const int64_t* count_in,
    float* ng, float* ag, int64_t* count_out, int64_t* cur_out, const Edits& E,
    uint32_t* __restrict__ flags, const float* __restrict__ sel_row, int b, int ks, int t, int N, int F) {
  // items: float4 number t + 256 q of the adjacency (q < 16) and of the node matrix (q < NODE_ALL),
  // workgroup ks of the graph takes q = ks, ks + KS, ...; moved in chunks of CH with two chunks of loads
  // in flight while a chunk is edited and stored - every workgroup of the launch starts at the same
  // time, so without this the whole chip reads, then the whole chip writes
  constexpr int KS = GCM_STATE_WGS;
  constexpr int ADJ_ALL = 16, NODE_ALL = (128 * FP / 4 + 255) / 256;
  constexpr int ADJ_PER = ADJ_ALL / KS, NODE_PER = (NODE_ALL + KS - 1) / KS;
  static_assert(ADJ_ALL % KS == 0, "state workgroups per graph");
  constexpr int NITEM = ADJ_PER + NODE_PER, CH = GCM_STATE_CH, NCH = (NITEM + CH - 1) / CH;
  const int lane = t & 63;
  const int N4 = N >> 2, F4 = F >> 2;
  const int lim_a = N * N4, lim_n = N * F4;
  const int n_hops = E.n_hops;
  int lane_h = -1, cdir = 0;
  if (lane >= 1 && lane <= n_hops) {
    lane_h = E.hops[(lane - 1) & 15];
    cdir = E.dir[(lane - 1) & 15];
  }
  const int64_t n_in = count_in[b];
  const bool dense = E.dense != 0;
  float4 buf[NITEM];
  auto e4_of = [&](int i) { return t + 256 * ((i < ADJ_PER ? i : i - ADJ_PER) * KS + ks); };
  // the plain copy's loads (16-byte aligned), from clamped addresses
  auto issue0 = [&](int i) {
    if (i < ADJ_PER) buf[i] = *reinterpret_cast<const float4*>(ag_in + 4 * min(e4_of(i), lim_a - 1));
    else buf[i] = *reinterpret_cast<const float4*>(ng_in + 4 * min(e4_of(i), lim_n - 1));
  };
  // the overflow roll's (gcm.py:323-355): out[r][c] = in[r + 1][c + 1], dword-aligned 16-byte loads,
  // the last column shifted in registers at store time
  auto issue1 = [&](int i) {
    if (i < ADJ_PER) {
      const int e4 = min(e4_of(i), lim_a - 1);
      const int r = e4 / N4, c = (e4 - r * N4) * 4;
      const bool tail = c + 4 >= N;
      __builtin_memcpy(&buf[i], ag_in + (r + 1 < N ? r + 1 : N - 1) * N + c + (tail ? 0 : 1), sizeof(float4));
    } else {
      const int e4 = min(e4_of(i), lim_n - 1);
      const int r = e4 / F4, c = (e4 - r * F4) * 4;
      buf[i] = *reinterpret_cast<const float4*>(ng_in + (r + 1 < N ? r + 1 : N - 1) * F + c);
    }
  };
#pragma unroll
  for (int i = 0; i < 2 * CH && i < NITEM; ++i) issue0(i);   // (no overflow: assumed)
  asm volatile("" ::: "memory");
  const bool wrap = n_in + 1 > N;
  const int64_t c64 = wrap ? n_in - 1 : n_in;
  const int cur = c64 < 0 ? 0 : (c64 > N - 1 ? N - 1 : (int)c64);
  const int sh = wrap ? 1 : 0;
  if (wrap) {
#pragma unroll
    for (int i = 0; i < 2 * CH && i < NITEM; ++i) issue1(i);
  }
  // the folded temporal hops as two bit sets over the node index (scalar code): entries (cur, j) of
  // forward / both hops, rows j that get a (j, cur) entry from backward / both hops
  unsigned long long cand0 = 0, cand1 = 0, col0 = 0, col1 = 0;
  bool hop0 = false;
  for (int k = 1; k <= n_hops; ++k) {
    const int h = __builtin_amdgcn_readlane(lane_h, k), d = __builtin_amdgcn_readlane(cdir, k);
    if (h < 0 || h > cur) continue;
    if (h == 0) {
      hop0 = true;
      continue;
    }
    const int j = cur - h;
    const unsigned long long bit = 1ull << (j & 63);
    if (d & GCM_DIR_FORWARD) (j < 64 ? cand0 : cand1) |= bit;
    if (d & GCM_DIR_BACKWARD) (j < 64 ? col0 : col1) |= bit;
  }
  auto finish = [&](int i) {   // edits in registers, then the store
    float4 v = buf[i];
    const int e4 = e4_of(i);
    if (i < ADJ_PER) {
      const int r = e4 / N4, c = (e4 - r * N4) * 4;
      if (wrap && c + 4 >= N) v = make_float4(v.y, v.z, v.w, 0.f);
      if (r + sh >= N) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r == cur) {   // temporal.py:72-88 (forward), dense.py:16-21, distance.py:31-37
        float4 sr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sel_row) sr = *reinterpret_cast<const float4*>(sel_row + (size_t)b * N + (c < N ? c : N - 4));
        float vv[4] = {v.x, v.y, v.z, v.w};
        const float ss[4] = {sr.x, sr.y, sr.z, sr.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int j = c + k;
          const bool in_cand = (((j < 64 ? cand0 : cand1) >> (j & 63)) & 1ull) != 0;
          // (bitwise on purpose: hipcc 7.2 lowered the short-circuit form of this condition to a branch
          //  tree that dropped its last term - tools/_dbg/dbg_sel.py, the g3 fixtures)
          const bool set = in_cand | (dense & (j <= cur)) | ((j == cur) & hop0) | ((j < cur) & (ss[k] != 0.f));
          vv[k] = set ? 1.f : vv[k];
        }
        v = make_float4(vv[0], vv[1], vv[2], vv[3]);
      } else {          // column cur: backward hops, DenseEdge's rows < cur
        const int k = cur - c;
        if (k >= 0 && k < 4) {
          const bool in_col = (((r < 64 ? col0 : col1) >> (r & 63)) & 1ull) != 0;
          const bool set = in_col | (dense & (r < cur));
          v.x = (set & (k == 0)) ? 1.f : v.x;
          v.y = (set & (k == 1)) ? 1.f : v.y;
          v.z = (set & (k == 2)) ? 1.f : v.z;
          v.w = (set & (k == 3)) ? 1.f : v.w;
        }
      }
      if (e4 < lim_a) *reinterpret_cast<float4*>(ag + e4 * 4) = v;
    } else {
      const int r = e4 / F4, c = (e4 - r * F4) * 4;
      if (r + sh >= N) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r == cur) v = *reinterpret_cast<const float4*>(obs + (size_t)b * F + c);   // gcm.py:274
      if (e4 < lim_n) *reinterpret_cast<float4*>(ng + e4 * 4) = v;
    }
  };
  if (!wrap) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
#pragma unroll
      for (int i = (k + 2) * CH; i < (k + 3) * CH && i < NITEM; ++i) issue0(i);
      asm volatile("" ::: "memory");   // the next chunk's loads are in the queue before this chunk's stores
#pragma unroll
      for (int i = k * CH; i < (k + 1) * CH && i < NITEM; ++i) finish(i);
      asm volatile("" ::: "memory");
    }
  } else {
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
#pragma unroll
      for (int i = (k + 2) * CH; i < (k + 3) * CH && i < NITEM; ++i) issue1(i);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = k * CH; i < (k + 1) * CH && i < NITEM; ++i) finish(i);
      asm volatile("" ::: "memory");
    }
  }
  if (t == 0 && ks == 0) {
    count_out[b] = cur + 1;
    if (cur_out) cur_out[b] = cur;
    const uint32_t f = (wrap ? GCM_FLAG_WRAPPED : 0u) | ((n_in < 0 || n_in > N) ? GCM_FLAG_BAD_COUNT : 0u);
    if (f) atomicOr(flags, f);
  }
}

}  // namespace gcm_rows
