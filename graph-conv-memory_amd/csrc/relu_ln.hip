// y = LayerNorm(relu(x)) over the last dimension F <= 64, many rows (the two hidden layers of the
// LearnedEdge edge network, learned.py:38-51: Linear - ReLU - LayerNorm on B*N candidate rows).
// torch runs this as a relu kernel + vectorized_layer_norm (35 us at [32768, 32]) forward and four
// kernels backward (relu_backward, cuComputeGradInput, cuComputePartGradGammaBeta + GradGammaBeta:
// 60 us); it is 8 MB of traffic, a few microseconds.  One thread group of F lanes owns a row, the
// row statistics are wavefront shuffles; gamma/beta gradients are per-workgroup slabs summed in a
// fixed order (deterministic).  eps and the biased variance follow torch.nn.LayerNorm.
#include "gcm_common.h"

namespace {

constexpr int RL_ROWS = 256;   // rows per workgroup (backward slabs: one per workgroup)

template <int FP>   // lanes per row: 32 or 64 (F <= FP)
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int m = FP / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

template <int FP>
__global__ __launch_bounds__(256) void k_relu_ln_fwd(const float* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     float* __restrict__ y, int64_t M, int F,
                                                     float eps) {
  constexpr int RPW = 256 / FP;           // rows handled per pass
  const int tid = threadIdx.x, f = tid % FP, sub = tid / FP;
  const bool on = f < F;
  const float g = on ? gamma[f] : 0.f, be = on ? beta[f] : 0.f;
  const float invF = 1.f / (float)F;
  for (int64_t r = (int64_t)blockIdx.x * RL_ROWS + sub; r < (int64_t)(blockIdx.x + 1) * RL_ROWS && r < M;
       r += RPW) {
    const float v = on ? x[r * F + f] : 0.f;
    const float a = v > 0.f ? v : 0.f;
    const float mean = row_sum<FP>(a) * invF;
    const float d = on ? a - mean : 0.f;
    const float var = row_sum<FP>(d * d) * invF;
    const float rstd = rsqrtf(var + eps);
    if (on) y[r * F + f] = d * rstd * g + be;
  }
}

// dx = relu'(x) * LN'(relu(x)) dy ;  slab of this workgroup: dgamma [F] | dbeta [F]
template <int FP>
__global__ __launch_bounds__(256) void k_relu_ln_bwd(const float* __restrict__ dy,
                                                     const float* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     float* __restrict__ dx, float* __restrict__ slabs,
                                                     int64_t M, int F, float eps) {
  constexpr int RPW = 256 / FP;
  __shared__ float sg[256], sb[256];
  const int tid = threadIdx.x, f = tid % FP, sub = tid / FP;
  const bool on = f < F;
  const float g = on ? gamma[f] : 0.f;
  const float invF = 1.f / (float)F;
  float dg = 0.f, db = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * RL_ROWS + sub; r < (int64_t)(blockIdx.x + 1) * RL_ROWS && r < M;
       r += RPW) {
    const float v = on ? x[r * F + f] : 0.f;
    const float go = on ? dy[r * F + f] : 0.f;
    const float a = v > 0.f ? v : 0.f;
    const float mean = row_sum<FP>(a) * invF;
    const float d = on ? a - mean : 0.f;
    const float var = row_sum<FP>(d * d) * invF;
    const float rstd = rsqrtf(var + eps);
    const float xh = d * rstd;
    const float gx = go * g;                                  // d/d xhat
    const float m1 = row_sum<FP>(gx) * invF;
    const float m2 = row_sum<FP>(gx * xh) * invF;
    const float da = rstd * (gx - m1 - xh * m2);
    if (on) dx[r * F + f] = v > 0.f ? da : 0.f;
    dg += go * xh;
    db += go;
  }
  sg[tid] = dg;
  sb[tid] = db;
  __syncthreads();
  if (tid < FP && tid < F) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      a += sg[q * FP + tid];
      b += sb[q * FP + tid];
    }
    slabs[(size_t)blockIdx.x * 2 * F + tid] = a;
    slabs[(size_t)blockIdx.x * 2 * F + F + tid] = b;
  }
}

}  // namespace

extern "C" int gcm_relu_layernorm_fwd(const float* x, const float* gamma, const float* beta,
                                      float* y, int64_t M, int F, float eps, gcm_stream_t stream) {
  GCM_REQUIRE(x && gamma && beta && y && M >= 0 && F > 0);
  if (F > 64) return GCM_EUNSUPPORTED;
  if (M == 0) return GCM_OK;
  const unsigned grid = (unsigned)((M + RL_ROWS - 1) / RL_ROWS);
  hipStream_t s = (hipStream_t)stream;
  if (F <= 32)
    hipLaunchKernelGGL(k_relu_ln_fwd<32>, dim3(grid), dim3(256), 0, s, x, gamma, beta, y, M, F, eps);
  else
    hipLaunchKernelGGL(k_relu_ln_fwd<64>, dim3(grid), dim3(256), 0, s, x, gamma, beta, y, M, F, eps);
  return gcm_launch_status();
}

extern "C" size_t gcm_relu_layernorm_bwd_workspace_bytes(int64_t M, int F) {
  if (M <= 0 || F <= 0) return 0;
  return sizeof(float) * (size_t)((M + RL_ROWS - 1) / RL_ROWS) * 2 * F;
}

extern "C" int gcm_relu_layernorm_bwd(const float* dy, const float* x, const float* gamma, float* dx,
                                      float* dgamma_dbeta, void* workspace, size_t workspace_bytes,
                                      int64_t M, int F, float eps, gcm_stream_t stream) {
  GCM_REQUIRE(dy && x && gamma && dx && dgamma_dbeta && workspace && M > 0 && F > 0);
  if (F > 64) return GCM_EUNSUPPORTED;
  if (workspace_bytes < gcm_relu_layernorm_bwd_workspace_bytes(M, F)) return GCM_EWORKSPACE;
  const unsigned grid = (unsigned)((M + RL_ROWS - 1) / RL_ROWS);
  hipStream_t s = (hipStream_t)stream;
  float* slabs = (float*)workspace;
  if (F <= 32)
    hipLaunchKernelGGL(k_relu_ln_bwd<32>, dim3(grid), dim3(256), 0, s, dy, x, gamma, dx, slabs, M, F, eps);
  else
    hipLaunchKernelGGL(k_relu_ln_bwd<64>, dim3(grid), dim3(256), 0, s, dy, x, gamma, dx, slabs, M, F, eps);
  int rc = gcm_launch_status();
  if (rc) return rc;
  return gcm_sum_slabs(slabs, (int)grid, 2 * F, dgamma_dbeta, stream);
}
