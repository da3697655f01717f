// Shared helpers for the gfx950 kernels behind include/gcm_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gcm_hip.h"

#define GCM_REQUIRE(cond) \
  do {                    \
    if (!(cond)) return GCM_EINVAL; \
  } while (0)

static inline int gcm_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GCM_OK : (int)e;
}

// hipFuncAttributeMaxDynamicSharedMemorySize (needed above 64 KB of dynamic LDS) is a property of a
// kernel ON A DEVICE: the largest size granted so far is remembered per (kernel, device), behind a
// mutex - one process may drive several devices from several threads.  Defined in state.hip.
void gcm_allow_dynamic_lds(const void* kernel, size_t bytes);
// compute units of the current device (cached per device; persistent kernels size their grids by it)
int gcm_cu_count();

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// 16-byte WRITE-THROUGH store (sc1) of data this launch does not read again: the bytes leave for memory while the
// kernel computes instead of sitting dirty in the XCD's L2 until the write-back at the end of the launch, which nothing
// overlaps (16.8 MB a steady-state LearnedEdge step: 0.75 us of its 14).  rsrc: a buffer descriptor of the graph's
// matrix (wave-uniform), off: byte offset.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wt_store4(__amdgpu_buffer_rsrc_t rsrc, int off, float x, float y, float z, float w) {
  const u32x4 v = {__float_as_uint(x), __float_as_uint(y), __float_as_uint(z), __float_as_uint(w)};
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off, 0, 16);
}

// tanh with <= ~5e-7 relative error in a dozen instructions (the ocml tanhf costs several
// hundred cycles per call and dominated the fused epilogue): (1-e)/(1+e), e = exp(-2|x|), away
// from zero; the odd Taylor polynomial through x^9 below |x| = 0.25 where 1-e would cancel.
__device__ __forceinline__ float gcm_tanh(float x) {
  const float ax = fabsf(x);
  const float e = __expf(-2.f * ax);
  const float big = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e);
  const float x2 = x * x;
  const float small =
      ax * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 62.f / 2835.f, -17.f / 315.f), 2.f / 15.f),
                         -1.f / 3.f), 1.f);
  return copysignf(ax < 0.25f ? small : big, x);
}

// Cross-lane sums / maxima on the DPP path (full-rate VALU) instead of __shfl_xor (ds_bpermute: every step of
// a butterfly is an LDS round trip of ~100 cycles on the critical path of the single-wave kernels here).
#define GCM_DPP_F(v, ctrl, rmask, old) \
  __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), ctrl, rmask, 0xF, false))
// value of the lane's xor-1 neighbour (quad_perm [1,0,3,2])
__device__ __forceinline__ float gcm_lane_xor1(float v) { return GCM_DPP_F(v, 0xB1, 0xF, 0.f); }
// sum / max over the 64 lanes, the same value returned to every lane
__device__ __forceinline__ float gcm_wave_sum(float v) {
  v += GCM_DPP_F(v, 0xB1, 0xF, 0.f);    // quad_perm [1,0,3,2]
  v += GCM_DPP_F(v, 0x4E, 0xF, 0.f);    // quad_perm [2,3,0,1]
  v += GCM_DPP_F(v, 0x141, 0xF, 0.f);   // row_half_mirror
  v += GCM_DPP_F(v, 0x140, 0xF, 0.f);   // row_mirror: every lane holds its row's (16 lanes) sum
  v += GCM_DPP_F(v, 0x142, 0xA, 0.f);   // row_bcast15 -> rows 1, 3
  v += GCM_DPP_F(v, 0x143, 0xC, 0.f);   // row_bcast31 -> rows 2, 3: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// v + the value of lane ^ 16 / lane ^ 32, in every lane, on gfx950's row / half swaps (v_permlane16_swap exchanges the
// odd 16-lane rows of its first operand with the even rows of its second, v_permlane32_swap the upper half of the
// first with the lower half of the second: from two copies of v one register ends up with the even rows / lower half
// everywhere, the other with the odd rows / upper half).  VALU instructions; `gcm_xor16_add(v)` is a
// ds_bpermute, an LDS round trip each.  Bit-identical to it (one commutative add of the same two numbers).
typedef unsigned gcm_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float gcm_xor16_add(float v) {
  const unsigned u = __float_as_uint(v);
  const gcm_u32x2 r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float gcm_xor32_add(float v) {
  const unsigned u = __float_as_uint(v);
  const gcm_u32x2 r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float gcm_wave_max(float v) {
  v = fmaxf(v, GCM_DPP_F(v, 0xB1, 0xF, v));
  v = fmaxf(v, GCM_DPP_F(v, 0x4E, 0xF, v));
  v = fmaxf(v, GCM_DPP_F(v, 0x141, 0xF, v));
  v = fmaxf(v, GCM_DPP_F(v, 0x140, 0xF, v));
  v = fmaxf(v, GCM_DPP_F(v, 0x142, 0xA, v));
  v = fmaxf(v, GCM_DPP_F(v, 0x143, 0xC, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float gcm_act(float v, int act) {
  if (act == GCM_ACT_TANH) return gcm_tanh(v);
  if (act == GCM_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}
// Same function with the activation code in a VGPR (gcm_vgpr): compiles to selects instead of the
// scalar branch tree above - for latency-bound single-wave code, where taken branches cost more
// than the few extra VALU instructions.
__device__ __forceinline__ int gcm_vgpr(int x) {
  int y;
  asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "s"(x));
  return y;
}
__device__ __forceinline__ float gcm_act_sel(float v, int act_v) {
  const float t = gcm_tanh(v);
  const float r = v > 0.f ? v : 0.f;
  float out = act_v == GCM_ACT_TANH ? t : v;
  out = act_v == GCM_ACT_RELU ? r : out;
  return out;
}
// d act / d pre, expressed through the activation OUTPUT y (activation code in a VGPR: selects)
__device__ __forceinline__ float gcm_act_grad_sel(float y, int act_v) {
  float g = 1.f;
  g = act_v == GCM_ACT_TANH ? 1.f - y * y : g;
  g = act_v == GCM_ACT_RELU ? (y > 0.f ? 1.f : 0.f) : g;
  return g;
}
// d act / d pre, expressed through the activation OUTPUT y
__device__ __forceinline__ float gcm_act_grad(float y, int act) {
  if (act == GCM_ACT_TANH) return 1.f - y * y;
  if (act == GCM_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  return 1.f;
}
