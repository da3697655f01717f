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

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float gcm_act(float v, int act) {
  if (act == GCM_ACT_TANH) return tanhf(v);
  if (act == GCM_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}
// d act / d pre, expressed through the activation OUTPUT y
__device__ __forceinline__ float gcm_act_grad(float y, int act) {
  if (act == GCM_ACT_TANH) return 1.f - y * y;
  if (act == GCM_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  return 1.f;
}
