// y = x W^T + b (or x W for the input gradient) on MANY rows of NARROW layers (<= 64 features in and
// out): the Linear layers of the LearnedEdge edge network (reference: src/gcm/edge_selectors/learned.py:38-51,
// 240-262 - M = one row per candidate edge) and the re-projection of PositionalEncoding(mode="cat")
// (src/gcm/gcm.py:133-140).  A library GEMM sees a [M x 64] x [64 x 64] problem and runs it at a few
// percent of the machine; this is an HBM-bound row stream: read 4 I bytes, write 4 O bytes per row.
//
// One workgroup = 128 rows (4 waves x 32), the weight matrix staged once per workgroup in LDS, the product
// on v_mfma_f32_32x32x2f32.  Optional epilogue: ReLU + LayerNorm of the row (learned.py:41-43), written to a
// second output while the pre-activation goes to the first (the backward wants both).
#include "fused_common.h"
#include "gcm_common.h"

namespace {

using gcm_fused::acc_row;
using gcm_fused::mma32;

constexpr int TR = 128;   // rows per workgroup
constexpr int XS = 65;    // LDS row stride of the row tile and of the weight image

// TRANS = false: B(k = i, j = o) = W[o][i]  (y = x W^T);  true: B(k = o, j = i) = W[o][i]  (gx = g W)
template <bool TRANS>
__global__ __launch_bounds__(256) void k_rows_linear(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, int64_t M, int K, int Nout, int w_cols, int ldy,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float* __restrict__ h) {
  __shared__ float sX[TR * XS];
  __shared__ float sB[64 * XS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * TR;
  const int rows = (int)((M - row0) < TR ? (M - row0) : TR);
  const int Kp = (K + 1) & ~1;
  const int nt_n = (Nout + 31) >> 5;

  // weight image sB[k][j], zero padded to Kp x 32 nt_n
  {
    for (int e = tid; e < 64 * 64; e += 256) {
      const int k = e >> 6, j = e & 63;
      float v = 0.f;
      if (k < K && j < Nout) v = TRANS ? w[(size_t)k * w_cols + j] : w[(size_t)j * w_cols + k];
      if (k < Kp && j < 32 * nt_n) sB[k * XS + j] = v;
    }
  }
  // row tile (contiguous in memory: rows x K floats)
  {
    const float* src = x + row0 * K;
    const int total = rows * K;
    for (int e = tid; e < TR * K; e += 256) {
      const int r = e / K, c = e - r * K;
      sX[r * XS + c] = e < total ? src[e] : 0.f;
    }
    if (Kp != K)
      for (int r = tid; r < TR; r += 256) sX[r * XS + K] = 0.f;
  }
  __syncthreads();

  f32x16 acc[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    if (nt < nt_n) mma32(acc[nt], sX + 32 * wave * XS, XS, 1, sB + 32 * nt, XS, 1, Kp, li, lh);
  }
  if (gamma) __syncthreads();   // the row tile is about to be overwritten by the pre-activations
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    if (nt >= nt_n) break;
    const int col = 32 * nt + li;
    const float bv = (bias && col < Nout) ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = 32 * wave + acc_row(r, lh);
      const float v = acc[nt][r] + bv;
      if (rr < rows && col < Nout) y[(row0 + rr) * ldy + col] = v;
      if (gamma) sX[rr * XS + col] = v;
    }
  }
  if (!gamma) return;
  __syncthreads();
  // ReLU + LayerNorm over the Nout columns of each row: two adjacent threads per row
  {
    const int row = tid >> 1, half = tid & 1;
    const int c0 = half * 32, c1 = (c0 + 32 < Nout) ? c0 + 32 : Nout;
    float s = 0.f;
    for (int c = c0; c < c1; ++c) {
      const float v = sX[row * XS + c];
      s += v > 0.f ? v : 0.f;
    }
    s += __shfl_xor(s, 1);
    const float mean = s / (float)Nout;
    float q = 0.f;
    for (int c = c0; c < c1; ++c) {
      const float v = sX[row * XS + c];
      const float d = (v > 0.f ? v : 0.f) - mean;
      q = fmaf(d, d, q);
    }
    q += __shfl_xor(q, 1);
    const float rstd = rsqrtf(q / (float)Nout + eps);
    if (row < rows)
      for (int c = c0; c < c1; ++c) {
        const float v = sX[row * XS + c];
        sX[row * XS + c] = fmaf(((v > 0.f ? v : 0.f) - mean) * rstd, gamma[c], beta[c]);
      }
  }
  __syncthreads();
  {
    float* dst = h + row0 * Nout;
    const int total = rows * Nout;
    for (int e = tid; e < total; e += 256) {
      const int r = e / Nout, c = e - r * Nout;
      dst[e] = sX[r * XS + c];
    }
  }
}

// PositionalEncoding(mode="cat") (gcm.py:133-140): out[b, i] = i <= n_b ? [pe[i, :cat] | proj[b, i]] : x[b, i]
// (out already holds proj in its columns cat.. - k_rows_linear wrote it there with ldy = F).
__global__ void k_posenc_cat_finish(const float* __restrict__ x, const float* __restrict__ pe, int pe_ld,
                                    const int64_t* __restrict__ num_nodes, float* __restrict__ out, int B, int N,
                                    int F, int cat) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)B * N * F) return;
  const int f = (int)(e % F);
  const int64_t bi = e / F;
  const int i = (int)(bi % N), b = (int)(bi / N);
  const bool live = i <= num_nodes[b];
  if (!live) out[e] = x[e];
  else if (f < cat) out[e] = pe[(size_t)i * pe_ld + f];
}
// adjoint: g_x = dead rows of g_out; g_proj [B*N, F - cat] = live rows of g_out[:, cat:]
__global__ void k_posenc_cat_bwd(const float* __restrict__ g_out, const int64_t* __restrict__ num_nodes,
                                 float* __restrict__ g_x, float* __restrict__ g_proj, int B, int N, int F, int cat) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)B * N * F) return;
  const int f = (int)(e % F);
  const int64_t bi = e / F;
  const int i = (int)(bi % N), b = (int)(bi / N);
  const bool live = i <= num_nodes[b];
  const float g = g_out[e];
  g_x[e] = live ? 0.f : g;
  if (f >= cat) g_proj[bi * (F - cat) + (f - cat)] = live ? g : 0.f;
}

}  // namespace

extern "C" int gcm_rows_linear(const float* x, const float* w, const float* bias, float* y, int64_t M, int I,
                               int O, int transpose, int ldy, const float* gamma, const float* beta, float eps,
                               float* h, gcm_stream_t stream) {
  GCM_REQUIRE(x && w && y && M >= 0 && I > 0 && O > 0);
  GCM_REQUIRE((gamma != nullptr) == (h != nullptr) && (gamma != nullptr) == (beta != nullptr));
  if (I > 64 || O > 64) return GCM_EUNSUPPORTED;
  if (M == 0) return GCM_OK;
  const int K = transpose ? O : I, Nout = transpose ? I : O;   // W is [O x I] either way
  if (ldy <= 0) ldy = Nout;
  GCM_REQUIRE(ldy >= Nout);
  const dim3 grid((unsigned)((M + TR - 1) / TR));
  if (transpose)
    hipLaunchKernelGGL(k_rows_linear<true>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, M, K, Nout, I,
                       ldy, gamma, beta, eps, h);
  else
    hipLaunchKernelGGL(k_rows_linear<false>, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, M, K, Nout, I,
                       ldy, gamma, beta, eps, h);
  return gcm_launch_status();
}

extern "C" int gcm_posenc_cat_finish(const float* x, const float* pe, int pe_ld, const int64_t* num_nodes,
                                     float* out, int B, int N, int F, int cat_dim, gcm_stream_t stream) {
  GCM_REQUIRE(x && pe && num_nodes && out && B > 0 && N > 0 && F > 0 && cat_dim >= 0 && cat_dim < F);
  const int64_t total = (int64_t)B * N * F;
  hipLaunchKernelGGL(k_posenc_cat_finish, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     x, pe, pe_ld, num_nodes, out, B, N, F, cat_dim);
  return gcm_launch_status();
}

extern "C" int gcm_posenc_cat_bwd(const float* g_out, const int64_t* num_nodes, float* g_x, float* g_proj, int B,
                                  int N, int F, int cat_dim, gcm_stream_t stream) {
  GCM_REQUIRE(g_out && num_nodes && g_x && g_proj && B > 0 && N > 0 && F > 0 && cat_dim >= 0 && cat_dim < F);
  const int64_t total = (int64_t)B * N * F;
  hipLaunchKernelGGL(k_posenc_cat_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     g_out, num_nodes, g_x, g_proj, B, N, F, cat_dim);
  return gcm_launch_status();
}
