// Weight gradient of a skinny Linear: dW[o][i] = sum_m dY[m][o] * X[m][i], db[o] = sum_m dY[m][o]
// for M >> O, I (the LearnedEdge edge network scores B*N candidate rows with 64 -> 32 -> 32 -> 1
// linears, learned.py:38-51).  A library GEMM sees a [O x M] x [M x I] product with K = 32768 and a
// 32 x 64 output: one or two workgroups do all the work (204 us at cfg5).  Here the rows are split
// over the grid: every workgroup contracts 128 rows on the matrix cores into a partial [O x I]
// slab, gcm_sum_slabs adds the slabs in a fixed order (deterministic, no atomics).
#include "fused_common.h"

namespace {

constexpr int ROWS = 128;   // rows per workgroup: 4 waves x 32

// slab layout: dW [O*I] | db [O]
template <int NOT, int NIT>   // 32-wide tiles of O and I
__global__ __launch_bounds__(256) void k_skinny_wgrad(const float* __restrict__ dy,
                                                      const float* __restrict__ x,
                                                      float* __restrict__ slabs, int M, int O, int I) {
  __shared__ float sR[4][NOT * NIT * 1024 + NOT * 32];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int r0 = blockIdx.x * ROWS + wave * 32;
  // A(i=o, k=row) = dY[row][o]; B(k=row, j=i) = X[row][i]: both straight from HBM, each element once
  float a[NOT][16], bsum[NOT];
#pragma unroll
  for (int ot = 0; ot < NOT; ++ot) {
    const int o = ot * 32 + li;
    bsum[ot] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int row = r0 + 2 * s + lh;
      const float v = dy[(size_t)(row < M ? row : M - 1) * O + (o < O ? o : O - 1)];
      a[ot][s] = (row < M && o < O) ? v : 0.f;
      bsum[ot] += a[ot][s];
    }
    bsum[ot] += __shfl_xor(bsum[ot], 32);   // both row parities
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int i = it * 32 + li;
    float b[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int row = r0 + 2 * s + lh;
      const float v = x[(size_t)(row < M ? row : M - 1) * I + (i < I ? i : I - 1)];
      b[s] = (row < M && i < I) ? v : 0.f;
    }
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 16; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ot][s], b[s], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sR[wave][(ot * NIT + it) * 1024 + gcm_fused::acc_row(r, lh) * 32 + li] = acc[r];
    }
  }
  if (lh == 0) {
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) sR[wave][NOT * NIT * 1024 + ot * 32 + li] = bsum[ot];
  }
  __syncthreads();
  float* slab = slabs + (size_t)blockIdx.x * ((size_t)O * I + O);
  for (int e = tid; e < NOT * NIT * 1024; e += 256) {
    const int blk = e >> 10, ot = blk / NIT, it = blk % NIT, o = ot * 32 + ((e >> 5) & 31),
              i = it * 32 + (e & 31);
    if (o < O && i < I) slab[o * I + i] = (sR[0][e] + sR[1][e]) + (sR[2][e] + sR[3][e]);
  }
  for (int o = tid; o < O; o += 256) {
    const int e = NOT * NIT * 1024 + o;
    slab[O * I + o] = (sR[0][e] + sR[1][e]) + (sR[2][e] + sR[3][e]);
  }
}

}  // namespace

extern "C" size_t gcm_skinny_wgrad_workspace_bytes(int M, int O, int I) {
  if (M <= 0 || O <= 0 || I <= 0) return 0;
  return sizeof(float) * (size_t)((M + ROWS - 1) / ROWS) * ((size_t)O * I + O);
}

extern "C" int gcm_skinny_wgrad(const float* dy, const float* x, float* dw_db, void* workspace,
                                size_t workspace_bytes, int M, int O, int I, gcm_stream_t stream) {
  GCM_REQUIRE(dy && x && dw_db && workspace && M > 0 && O > 0 && I > 0);
  if (O > 64 || I > 64) return GCM_EUNSUPPORTED;
  if (workspace_bytes < gcm_skinny_wgrad_workspace_bytes(M, O, I)) return GCM_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int G = (M + ROWS - 1) / ROWS, NOT = (O + 31) / 32, NIT = (I + 31) / 32;
  float* slabs = (float*)workspace;
#define GCM_W(a, b)                                                                           \
  if (NOT == a && NIT == b)                                                                   \
    hipLaunchKernelGGL((k_skinny_wgrad<a, b>), dim3(G), dim3(256), 0, s, dy, x, slabs, M, O, I);
  GCM_W(1, 1) GCM_W(1, 2) GCM_W(2, 1) GCM_W(2, 2)
#undef GCM_W
  int rc = gcm_launch_status();
  if (rc) return rc;
  return gcm_sum_slabs(slabs, G, O * I + O, dw_db, stream);
}
