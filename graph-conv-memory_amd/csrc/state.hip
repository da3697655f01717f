// DenseGCM state kernels: insert + overflow wrap, belief-row gather, and the
// index-writing edge selectors.  Pure HBM-bound byte movement: one pass over
// nodes/adj with 16-byte accesses where alignment allows, no LDS needed.
#include "gcm_common.h"

// ---------------------------------------------------------------------------
// state advance (gcm.py:262-278, 323-355)
// ---------------------------------------------------------------------------
// grid.x = row chunks, grid.y = graph, grid.z = plane (0 nodes, 1 adj, 2 weights)
// A plane is [N rows][W cols]; square planes (adj, weights) shift rows AND cols
// on wrap, the node plane shifts rows only.
template <bool BWD>
__global__ __launch_bounds__(256) void k_state_advance(
    const float* __restrict__ nodes_in, const float* __restrict__ adj_in,
    const float* __restrict__ w_in, const int64_t* __restrict__ num_nodes_in,
    const float* __restrict__ x, float* __restrict__ nodes_out, float* __restrict__ adj_out,
    float* __restrict__ w_out, float* __restrict__ g_x, int64_t* __restrict__ cur_idx_out,
    int64_t* __restrict__ num_nodes_out, uint32_t* __restrict__ flags, int N, int F,
    int rows_per_block) {
  const int b = blockIdx.y;
  const int plane = blockIdx.z;
  const int64_t n_in = num_nodes_in[b];
  const bool bad = n_in < 0 || n_in > N;
  const bool wrap = n_in + 1 > N;
  int64_t cur = wrap ? n_in - 1 : n_in;
  if (cur < 0) cur = 0;
  if (cur > N - 1) cur = N - 1;  // keeps every access in bounds when `bad`

  const float* src;
  float* dst;
  int W;
  bool square;
  if (plane == 0) { src = nodes_in; dst = nodes_out; W = F; square = false; }
  else if (plane == 1) { src = adj_in; dst = adj_out; W = N; square = true; }
  else { src = w_in; dst = w_out; W = N; square = true; }
  if (src == nullptr || dst == nullptr) return;
  src += (size_t)b * N * W;
  dst += (size_t)b * N * W;

  if (!BWD && plane == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    cur_idx_out[b] = cur;
    num_nodes_out[b] = cur + 1;
    uint32_t f = (wrap ? GCM_FLAG_WRAPPED : 0u) | (bad ? GCM_FLAG_BAD_COUNT : 0u);
    if (f) atomicOr(flags, f);
  }

  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(N, r0 + rows_per_block);
  const int total = (r1 - r0) * W;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int r = r0 + e / W;
    const int c = e - (e / W) * W;
    float v;
    if (!BWD) {
      // forward: out[r][c] = wrapped ? in[r+1][c(+1)] (last row / col zero) : in[r][c]
      if (!wrap) {
        v = src[(size_t)r * W + c];
      } else {
        const int sr = r + 1, sc = square ? c + 1 : c;
        v = (sr < N && sc < W) ? src[(size_t)sr * W + sc] : 0.f;
      }
      if (plane == 0 && r == (int)cur) v = x[(size_t)b * F + c];
    } else {
      // backward: g_in[r][c] = wrapped ? g_out[r-1][c(-1)] (row/col 0 zero) : g_out[r][c],
      // except that the forward OVERWROTE out[cur] with x: that row feeds g_x, not g_in.
      int orow, ocol;
      bool live;
      if (!wrap) { orow = r; ocol = c; live = true; }
      else { orow = r - 1; ocol = square ? c - 1 : c; live = orow >= 0 && ocol >= 0; }
      v = live ? src[(size_t)orow * W + ocol] : 0.f;
      if (plane == 0 && live && orow == (int)cur) v = 0.f;
    }
    dst[(size_t)r * W + c] = v;
  }
  if (BWD && plane == 0 && g_x != nullptr && blockIdx.x == 0) {
    for (int c = threadIdx.x; c < F; c += blockDim.x)
      g_x[(size_t)b * F + c] = src[(size_t)cur * F + c];
  }
}


// 16-byte version of the forward copy (N % 4 == 0 and F % 4 == 0): one float4 per thread and
// step; the rolled source (one row down, one column right) is read with a dword-aligned
// 16-byte load, the last column group is shifted in registers.
__global__ __launch_bounds__(256) void k_state_advance_vec(
    const float* __restrict__ nodes_in, const float* __restrict__ adj_in,
    const float* __restrict__ w_in, const int64_t* __restrict__ num_nodes_in,
    const float* __restrict__ x, float* __restrict__ nodes_out, float* __restrict__ adj_out,
    float* __restrict__ w_out, int64_t* __restrict__ cur_idx_out,
    int64_t* __restrict__ num_nodes_out, uint32_t* __restrict__ flags, int N, int F,
    int rows_per_block) {
  const int b = blockIdx.y;
  const int plane = blockIdx.z;
  const int64_t n_in = num_nodes_in[b];
  const bool bad = n_in < 0 || n_in > N;
  const bool wrap = n_in + 1 > N;
  int64_t cur64 = wrap ? n_in - 1 : n_in;
  const int cur = cur64 < 0 ? 0 : (cur64 > N - 1 ? N - 1 : (int)cur64);
  const float* src;
  float* dst;
  int W;
  bool square;
  if (plane == 0) { src = nodes_in; dst = nodes_out; W = F; square = false; }
  else if (plane == 1) { src = adj_in; dst = adj_out; W = N; square = true; }
  else { src = w_in; dst = w_out; W = N; square = true; }
  if (src == nullptr || dst == nullptr) return;
  src += (size_t)b * N * W;
  dst += (size_t)b * N * W;
  if (plane == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    cur_idx_out[b] = cur;
    num_nodes_out[b] = cur + 1;
    const uint32_t f = (wrap ? GCM_FLAG_WRAPPED : 0u) | (bad ? GCM_FLAG_BAD_COUNT : 0u);
    if (f) atomicOr(flags, f);
  }
  const int W4 = W >> 2;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(N, r0 + rows_per_block);
  const int total = (r1 - r0) * W4;
  const int sh = wrap ? 1 : 0;
  for (int e = threadIdx.x; e < total; e += blockDim.x) {
    const int r = r0 + e / W4, c = (e % W4) * 4;
    const int rs = r + sh < N ? r + sh : N - 1;
    const int csh = square ? sh : 0;
    const bool tail = csh && c + 4 >= W;
    float4 v;
    __builtin_memcpy(&v, src + (size_t)rs * W + c + (tail ? 0 : csh), sizeof(float4));
    if (tail) v = make_float4(v.y, v.z, v.w, 0.f);
    if (r + sh >= N) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (plane == 0 && r == cur) v = *reinterpret_cast<const float4*>(x + (size_t)b * F + c);
    *reinterpret_cast<float4*>(dst + (size_t)r * W + c) = v;
  }
}

extern "C" int gcm_state_advance_fwd(const float* nodes_in, const float* adj_in,
                                     const float* weights_in, const int64_t* num_nodes_in,
                                     const float* x, float* nodes_out, float* adj_out,
                                     float* weights_out, int64_t* cur_idx_out,
                                     int64_t* num_nodes_out, uint32_t* flags, int B, int N, int F,
                                     gcm_stream_t stream) {
  GCM_REQUIRE(nodes_in && num_nodes_in && x && nodes_out && cur_idx_out && num_nodes_out && flags);
  GCM_REQUIRE(B > 0 && N > 0 && F > 0);
  GCM_REQUIRE((adj_in == nullptr) == (adj_out == nullptr));
  GCM_REQUIRE((weights_in == nullptr) == (weights_out == nullptr));
  const int rows_per_block = 16;
  const int planes = weights_in ? 3 : (adj_in ? 2 : 1);
  dim3 grid((N + rows_per_block - 1) / rows_per_block, B, planes);
  auto aligned16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  if ((N & 3) == 0 && (F & 3) == 0 && aligned16(nodes_in) && aligned16(nodes_out) && aligned16(x) &&
      aligned16(adj_in) && aligned16(adj_out) && aligned16(weights_in) && aligned16(weights_out)) {
    hipLaunchKernelGGL(k_state_advance_vec, grid, dim3(256), 0, (hipStream_t)stream, nodes_in,
                       adj_in, weights_in, num_nodes_in, x, nodes_out, adj_out, weights_out,
                       cur_idx_out, num_nodes_out, flags, N, F, rows_per_block);
    return gcm_launch_status();
  }
  hipLaunchKernelGGL(k_state_advance<false>, grid, dim3(256), 0, (hipStream_t)stream, nodes_in,
                     adj_in, weights_in, num_nodes_in, x, nodes_out, adj_out, weights_out,
                     (float*)nullptr, cur_idx_out, num_nodes_out, flags, N, F, rows_per_block);
  return gcm_launch_status();
}

extern "C" int gcm_state_advance_bwd(const float* g_nodes_out, const float* g_plane_out,
                                     const int64_t* num_nodes_in, float* g_nodes_in,
                                     float* g_plane_in, float* g_x, int B, int N, int F,
                                     gcm_stream_t stream) {
  GCM_REQUIRE(g_nodes_out && num_nodes_in && g_nodes_in);
  GCM_REQUIRE(B > 0 && N > 0 && F > 0);
  GCM_REQUIRE((g_plane_out == nullptr) == (g_plane_in == nullptr));
  const int rows_per_block = 16;
  dim3 grid((N + rows_per_block - 1) / rows_per_block, B, g_plane_out ? 2 : 1);
  hipLaunchKernelGGL(k_state_advance<true>, grid, dim3(256), 0, (hipStream_t)stream, g_nodes_out,
                     g_plane_out, (const float*)nullptr, num_nodes_in, (const float*)nullptr,
                     g_nodes_in, g_plane_in, (float*)nullptr, g_x, (int64_t*)nullptr,
                     (int64_t*)nullptr, (uint32_t*)nullptr, N, F, rows_per_block);
  return gcm_launch_status();
}

// ---------------------------------------------------------------------------
// belief row gather (gcm.py:309-318)
// ---------------------------------------------------------------------------
__global__ void k_gather_rows_fwd(const float* __restrict__ feats,
                                  const int64_t* __restrict__ cur_idx, float* __restrict__ out,
                                  uint32_t* __restrict__ flags, int B, int N, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool nonfinite = false;
  if (i < B * H) {
    const int b = i / H, c = i - b * H;
    int64_t r = cur_idx[b];
    r = r < 0 ? 0 : (r > N - 1 ? N - 1 : r);
    const float v = feats[((size_t)b * N + r) * H + c];
    out[i] = v;
    nonfinite = !isfinite(v);
  }
  if (__any(nonfinite) && (threadIdx.x & 63) == 0) atomicOr(flags, GCM_FLAG_NONFINITE);
}

__global__ void k_gather_rows_bwd(const float* __restrict__ g_out,
                                  const int64_t* __restrict__ cur_idx, float* __restrict__ g_feats,
                                  int B, int N, int H) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * N * H;
  if (i >= total) return;
  const int c = i % H;
  const int r = (i / H) % N;
  const int b = i / ((size_t)H * N);
  int64_t cur = cur_idx[b];
  cur = cur < 0 ? 0 : (cur > N - 1 ? N - 1 : cur);
  g_feats[i] = (r == (int)cur) ? g_out[(size_t)b * H + c] : 0.f;
}

extern "C" int gcm_gather_rows_fwd(const float* feats, const int64_t* cur_idx, float* out,
                                   uint32_t* flags, int B, int N, int H, gcm_stream_t stream) {
  GCM_REQUIRE(feats && cur_idx && out && flags && B > 0 && N > 0 && H > 0);
  const int total = B * H;
  hipLaunchKernelGGL(k_gather_rows_fwd, dim3((total + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, feats, cur_idx, out, flags, B, N, H);
  return gcm_launch_status();
}

extern "C" int gcm_gather_rows_bwd(const float* g_out, const int64_t* cur_idx, float* g_feats,
                                   int B, int N, int H, gcm_stream_t stream) {
  GCM_REQUIRE(g_out && cur_idx && g_feats && B > 0 && N > 0 && H > 0);
  const size_t total = (size_t)B * N * H;
  hipLaunchKernelGGL(k_gather_rows_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, g_out, cur_idx, g_feats, B, N, H);
  return gcm_launch_status();
}

// ---------------------------------------------------------------------------
// index-writing selectors
// ---------------------------------------------------------------------------
struct HopList {
  int32_t h[16];
};

__global__ void k_edge_temporal(float* __restrict__ adj, const int64_t* __restrict__ cur_idx,
                                HopList hops, int n_hops, int direction, int B, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n_hops) return;
  const int b = i / n_hops;
  const int hop = hops.h[i - b * n_hops];
  const int64_t cur = cur_idx[b];
  // temporal.py:74 - valid when num_nodes >= hop; python-style negative hops are not supported
  if (cur < 0 || cur >= N || cur < hop || hop < 0) return;
  const int64_t past = cur - hop;
  float* a = adj + (size_t)b * N * N;
  if (direction & GCM_DIR_FORWARD) a[cur * N + past] = 1.f;
  if (direction & GCM_DIR_BACKWARD) a[past * N + cur] = 1.f;
}

extern "C" int gcm_edge_temporal(float* adj, const int64_t* cur_idx, const int32_t* hops_host,
                                 int n_hops, int direction, int B, int N, gcm_stream_t stream) {
  GCM_REQUIRE(adj && cur_idx && hops_host && B > 0 && N > 0);
  GCM_REQUIRE(direction >= 1 && direction <= 3);
  if (n_hops <= 0) return GCM_OK;
  if (n_hops > 16) return GCM_EUNSUPPORTED;
  HopList hl;
  for (int i = 0; i < 16; ++i) hl.h[i] = i < n_hops ? hops_host[i] : 0;
  const int total = B * n_hops;
  hipLaunchKernelGGL(k_edge_temporal, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     adj, cur_idx, hl, n_hops, direction, B, N);
  return gcm_launch_status();
}

// dense.py:16-21 - row cur[0..cur] = 1 (includes the self edge), col cur[0..cur) = 1
__global__ void k_edge_dense(float* __restrict__ adj, const int64_t* __restrict__ cur_idx, int N) {
  const int b = blockIdx.x;
  const int64_t cur = cur_idx[b];
  if (cur < 0 || cur >= N) return;
  float* a = adj + (size_t)b * N * N;
  for (int j = threadIdx.x; j <= cur; j += blockDim.x) {
    a[cur * N + j] = 1.f;
    if (j < cur) a[(size_t)j * N + cur] = 1.f;
  }
}

extern "C" int gcm_edge_dense(float* adj, const int64_t* cur_idx, int B, int N,
                              gcm_stream_t stream) {
  GCM_REQUIRE(adj && cur_idx && B > 0 && N > 0);
  hipLaunchKernelGGL(k_edge_dense, dim3(B), dim3(128), 0, (hipStream_t)stream, adj, cur_idx, N);
  return gcm_launch_status();
}

// PositionalEncoding mode="add" (gcm.py:120-131)
__global__ void k_posenc_add(float* __restrict__ x, const float* __restrict__ pe,
                             const int64_t* __restrict__ num_nodes, int N, int F, int d_model,
                             int max_len) {
  const int b = blockIdx.y;
  const int64_t last = num_nodes[b];   // rows 0..last inclusive are live
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * F) return;
  const int n = i / F, f = i - n * F;
  if (n <= last && n < max_len) x[(size_t)b * N * F + i] += pe[(size_t)n * d_model + f];
}

extern "C" int gcm_posenc_add(float* x, const float* pe, const int64_t* num_nodes, int B, int N,
                              int F, int d_model, int max_len, gcm_stream_t stream) {
  GCM_REQUIRE(x && pe && num_nodes && B > 0 && N > 0 && F > 0 && d_model >= F && max_len > 0);
  if (B > 65535) return GCM_EUNSUPPORTED;
  hipLaunchKernelGGL(k_posenc_add, dim3((N * F + 255) / 256, B), dim3(256), 0, (hipStream_t)stream,
                     x, pe, num_nodes, N, F, d_model, max_len);
  return gcm_launch_status();
}

#include <map>
#include <mutex>
#include <utility>

void gcm_allow_dynamic_lds(const void* kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return;
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> granted;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  size_t& g = granted[{kernel, dev}];
  if (bytes > g) {
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    g = bytes;
  }
}

int gcm_cu_count() {
  static std::mutex mu;
  static std::map<int, int> cus;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  int& c = cus[dev];
  if (!c) {
    hipDeviceProp_t prop;
    c = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            ? prop.multiProcessorCount : 256;
  }
  return c;
}

extern "C" int gcm_version(void) { return 103; }
// bumped whenever an existing entry point's SIGNATURE or a caller-allocated buffer's size changes (new entry points
// alone do not bump it): a binding built against another value must not call into this library
extern "C" int gcm_abi_version(void) { return GCM_ABI_VERSION; }

extern "C" const char* gcm_status_string(int code) {
  switch (code) {
    case GCM_OK: return "ok";
    case GCM_EINVAL: return "invalid argument (null pointer or non-positive size)";
    case GCM_EUNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case GCM_EWORKSPACE: return "workspace too small";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}
