// The tile arithmetic both EuclideanEdge kernels share (distance.hip: k_euclid_mfma2, one step per launch;
// euclid_tp.hip: k_euclid_tp, all T steps of a rollout) - one definition, so that both produce the same value for the
// same (stored node, step) pair and a time-parallel rollout decides exactly like T single steps.
//
//   acc(i = current row b', j = node) = C' N'   with C'(i, .) = [c_i | |c_i|^2, 1] streamed from the LDS image of the
//   current rows (MFMA A operand) and N'(., j) = [-2 n_j | 1, |n_j|^2] in registers (B operand): the accumulator IS the
//   squared distance |c|^2 + |n|^2 - 2 c.n.  In the 32x32 accumulator layout a lane holds ONE node (column j = lane & 31)
//   and sixteen current rows, so sqrt and the sum over b' stay inside the lane: no cross-lane reduction, one partial
//   sum register per wave instead of sixteen (round 4 had the operands the other way round: sixteen row sums per lane
//   and a DPP butterfly over the 32 columns per row block and step - 1.6 us of a 19 us kernel).
//
// The A operand passes through a window of W registers (one ds_read_b32 per MFMA, W ahead of its use) instead of all
// KQ + 1 at once: what keeps 16-wave workgroups (128 registers) free of spills at F = 64.
#pragma once
#include "gcm_common.h"

#ifndef GCM_CHAIN_W
#define GCM_CHAIN_W 8   // registers of the A-operand window (tools/ab_euclid.py varies it)
#endif

// one element of the epilogue: sqrt(max(d^2, 0)) as max(sqrt(d^2), 0) - v_sqrt_f32 (1 ulp; a NaN for the -1e-7 a
// self-distance rounds to) and ONE v_max_f32 (max of a NaN and 0 is 0; sqrt first saves the canonicalising max)
__device__ __forceinline__ float gcm_dist_elem(const float d2) { return fmaxf(__builtin_amdgcn_sqrtf(d2), 0.f); }

// PIPE: the epilogue of the wave's PREVIOUS tile (accp -> its sum over the sixteen current rows, returned in psum) is
// issued between the MFMAs of this tile's chain, in program order pinned by scheduling barriers: it runs under the
// matrix work instead of behind it (gcm_dist_tile_sum_full's arithmetic, value for value).
template <int KQ, int W, bool PIPE>
__device__ __forceinline__ void gcm_dist_chain(f32x16& acc, const float (&nv)[KQ + 1], const float* cp, const int cs,
                                               const float c_last, const f32x16& accp, float& psum) {
  static_assert(W <= KQ && (KQ == 16 || KQ == 32), "window");
  float cw[W];
#pragma unroll
  for (int q = 0; q < W; ++q) cw[q] = cp[2 * q * cs];
  float s = 0.f, e = 0.f;
#pragma unroll
  for (int q = 0; q < KQ; ++q) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cw[q % W], nv[q], acc, 0, 0, 0);
    if (q + W < KQ) cw[q % W] = cp[2 * (q + W) * cs];
    if (PIPE) {
      if (KQ == 16) {
        s += gcm_dist_elem(accp[q]);
      } else if ((q & 1) == 0) {
        e = __builtin_amdgcn_sqrtf(accp[q >> 1]);
      } else {
        s += fmaxf(e, 0.f);
      }
    }
#ifndef GCM_CHAIN_NOBAR   // (tools/ab_euclid.py: the A/B of this barrier)
    __builtin_amdgcn_sched_barrier(0);   // program order is the schedule: one ds_read (W ahead) and one epilogue piece per MFMA
#endif
  }
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c_last, nv[KQ], acc, 0, 0, 0);
  if (PIPE) psum = s;
}

// sum over the tile's sixteen current rows of this lane's node, rows in register order
__device__ __forceinline__ float gcm_dist_tile_sum_full(const f32x16& acc) {
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += gcm_dist_elem(acc[r]);
  return s;
}
// ... of the batch's last, partial tile: rows i >= n_rows hold no graph - masked (branch-free: one basic block)
__device__ __forceinline__ float gcm_dist_tile_sum_masked(const f32x16& acc, const int lh, const int n_rows) {
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float keep = (r & 3) + 8 * (r >> 2) + 4 * lh < n_rows ? 1.f : 0.f;
    s = fmaf(keep, gcm_dist_elem(acc[r]), s);
  }
  return s;
}
__device__ __forceinline__ float gcm_dist_tile_sum(const f32x16& acc, const int lh, const int n_rows) {
  return n_rows >= 32 ? gcm_dist_tile_sum_full(acc) : gcm_dist_tile_sum_masked(acc, lh, n_rows);
}

// the step's summed distance of one node from the per-(column tile, lane half) partial sums [4][2][rb_stride] in LDS,
// fixed order
__device__ __forceinline__ float gcm_dist_total(const float* part, const int rb_stride, const int slot, const int tiles) {
  float tot = part[slot] + part[rb_stride + slot];
  for (int q = 1; q < tiles; ++q) {
    tot += part[2 * q * rb_stride + slot];
    tot += part[(2 * q + 1) * rb_stride + slot];
  }
  return tot;
}
