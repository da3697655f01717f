// Register-staged copy of one graph's state (adjacency [N,N], node matrix [N,F]) by a 256-thread
// workgroup, with the overflow roll (gcm.py:323-355) folded in: used by the live-row step
// (rows_step.hip, functional state) and by the LearnedEdge advance + select kernel (learned_step.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace gcm_state {

// the state copy's stores; the overflow fix-ups (last column shifted in registers, last row zero)
// are applied here so that nothing depends on the loaded values while the loads are being issued
template <int ADJ_PER, int NODE_PER>
__device__ __forceinline__ void store_copy(const float4 (&ca)[ADJ_PER], const float4 (&cn)[NODE_PER],
                                           float* ag, float* ng, int tid, int N, int N4, int F4,
                                           bool wrap) {
  const int lim_a = N * N4, lim_n = N * F4, sh = wrap ? 1 : 0;
#pragma unroll
  for (int i = 0; i < ADJ_PER; ++i) {
    const int e4 = tid + 256 * i;
    const int r = e4 / N4, c = (e4 - r * N4) * 4;
    float4 v = ca[i];
    if (wrap && c + 4 >= N) v = make_float4(v.y, v.z, v.w, 0.f);
    if (r + sh >= N) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e4 < lim_a) *reinterpret_cast<float4*>(ag + e4 * 4) = v;
  }
#pragma unroll
  for (int i = 0; i < NODE_PER; ++i) {
    const int e4 = tid + 256 * i;
    const int r = e4 / F4;
    float4 v = cn[i];
    if (r + sh >= N) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e4 < lim_n) *reinterpret_cast<float4*>(ng + e4 * 4) = v;
  }
}

// out[r][c] = in[r + sh][c + sh]: the loads of the roll (sh = 1, dword-aligned 16-byte loads) or of
// the plain copy (sh = 0), from clamped addresses
template <int ADJ_PER, int NODE_PER, bool WRAP>
__device__ __forceinline__ void load_copy(float4 (&ca)[ADJ_PER], float4 (&cn)[NODE_PER],
                                          const float* ag_in, const float* ng_in, int tid, int N,
                                          int N4, int F, int F4) {
  const int lim_a = N * N4, lim_n = N * F4;
#pragma unroll
  for (int i = 0; i < ADJ_PER; ++i) {
    const int e4 = min(tid + 256 * i, lim_a - 1);
    if (!WRAP) {
      ca[i] = *reinterpret_cast<const float4*>(ag_in + 4 * e4);
    } else {
      const int r = e4 / N4, c = (e4 - r * N4) * 4;
      const int rs = r + 1 < N ? r + 1 : N - 1;
      const bool tail = c + 4 >= N;   // in[.][N] does not exist: shifted at store time
      __builtin_memcpy(&ca[i], ag_in + rs * N + c + (tail ? 0 : 1), sizeof(float4));
    }
  }
#pragma unroll
  for (int i = 0; i < NODE_PER; ++i) {
    const int e4 = min(tid + 256 * i, lim_n - 1);
    if (!WRAP) {
      cn[i] = *reinterpret_cast<const float4*>(ng_in + 4 * e4);
    } else {
      const int r = e4 / F4, c = (e4 - r * F4) * 4;
      cn[i] = *reinterpret_cast<const float4*>(ng_in + (r + 1 < N ? r + 1 : N - 1) * F + c);
    }
  }
}

}  // namespace gcm_state
