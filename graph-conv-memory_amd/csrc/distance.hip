// Fused pairwise-distance + threshold + adjacency-row write for the Distance
// edge selectors (edge_selectors/distance.py:18-81).
//
//   EUCLID_CROSSBATCH  d[b,j] = mean_{b'} || cur[b'] - nodes[b,j] ||        (EuclideanEdge: the
//                      reference's cdist([B,F],[B,N,F]).mean(1) averages over ALL graphs b')
//   L2_PERGRAPH        d[b,j] = || cur[b, a0:a1] - nodes[b, j, b0:b1] ||     (SpatialEdge)
//   COSINE_SIM         d[b,j] = <cur[b]/max(|cur[b]|,eps), nodes[b,j]/max(|nodes[b,j]|,eps)>
//
// then adj[b, cur_b, j] = 1 for every j < cur_b with d[b,j] < max_distance.
// The distance matrix never leaves the chip unless dist_out is given.
#include "fused_common.h"
#include "euclid_chain.h"

#ifdef GCM_STAMPS   // diagnostic build only (make stamps6, tools/kstamp_euclid.py)
__device__ unsigned long long g_stamps[32];
extern "C" int gcm_debug_read_stamps(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n);
}
#define DSTAMP(i) do { if (blockIdx.y == (gridDim.y >> 1)) STAMP(i); } while (0)
#else
#define DSTAMP(i)
#endif

namespace {

// Where a kernel finds the graph.  Either the ADVANCED state (count == nullptr): cur_idx[b] is the row
// that holds the current node.  Or the state BEFORE the step (count = num_nodes going in, obs = the
// new nodes): the current node is obs[b], it will land in row cur = min(count, N-1), and after the
// overflow roll (count + 1 > N) image row j is stored row j + 1.  Same arithmetic either way: the
// second form lets the live-row step (rows_step.hip) run the distance selectors ahead of its own
// state advance, with the decisions handed over as a row [B, N] instead of adjacency writes.
//
// cur_rows != nullptr: the "current nodes" the distances are taken against come from outside - n_cur rows
// [n_cur, F], the all-gathered current nodes of EVERY rank of a batch-sharded run (EuclideanEdge's
// mean runs over all graphs b' of the GLOBAL batch, distance.py:48-49) - instead of the B local ones.
struct View {
  const float* nodes;
  const int64_t* cur_idx;
  const int64_t* count;
  const float* obs;
  const float* cur_rows;
  int n_cur;
};
__device__ __forceinline__ int view_cur(const View& v, int b, int N, int& sh) {
  sh = 0;
  int64_t c;
  if (v.count) {
    const int64_t n = v.count[b];
    sh = n + 1 > N ? 1 : 0;
    c = sh ? n - 1 : n;
  } else {
    c = v.cur_idx[b];
  }
  return (int)(c < 0 ? 0 : (c > N - 1 ? N - 1 : c));
}
__device__ __forceinline__ const float* view_cur_row(const View& v, int b, int cur, int N, int F) {
  return v.count ? v.obs + (size_t)b * F : v.nodes + ((size_t)b * N + cur) * F;
}
// decision for image row j: adjacency entry (advanced state) or the row handed to the step kernel
__device__ __forceinline__ void view_emit(float* adj, float* sel_row, int b, int cur, int j, int N,
                                          bool hit, int bidirectional) {
  if (sel_row) {
    sel_row[(size_t)b * N + j] = hit ? 1.f : 0.f;
  } else if (hit) {
    adj[((size_t)b * N + cur) * N + j] = 1.f;
    if (bidirectional) adj[((size_t)b * N + j) * N + cur] = 1.f;
  }
}

// gather cur rows (scaled) into workspace: ws_cur [B, F]
__global__ void k_gather_cur(View vw, const float* __restrict__ dist_param,
                             float* __restrict__ ws_cur, int Bc, int N, int F) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Bc * F) return;
  const int b = i / F, f = i - b * F;
  float v;
  if (vw.cur_rows) {
    v = vw.cur_rows[i];
  } else {
    int sh;
    const int c = view_cur(vw, b, N, sh);
    v = view_cur_row(vw, b, c, N, F)[f];
  }
  ws_cur[i] = dist_param ? v / dist_param[0] : v;
}

// One workgroup = 128 threads = 128 node rows j of one graph b; each thread keeps its
// (scaled) node row in registers (FP floats), the cur rows stream through LDS in chunks
// and are read as broadcast 16-byte vectors.
template <int FP>
__global__ __launch_bounds__(128) void k_euclid_crossbatch(
    View vw, const float* __restrict__ ws_cur, const float* __restrict__ dist_param,
    float* __restrict__ adj, float* __restrict__ sel_row, float* __restrict__ dist_out,
    float max_distance, int bidirectional, int Bc, int N, int F) {
  const float* __restrict__ nodes = vw.nodes;
  constexpr int CH = 32;  // cur rows per LDS chunk
  __shared__ __attribute__((aligned(16))) float sC[CH * FP];
  const int b = blockIdx.y;
  const int j = blockIdx.x * 128 + threadIdx.x;
  int sh;
  const int cur = view_cur(vw, b, N, sh);
  // rows >= cur can never become edges (distance.py:31-33); skip whole blocks of them
  const bool block_live = (int64_t)blockIdx.x * 128 < cur || dist_out != nullptr;
  if (!block_live) return;
  const bool live = j < N && (j < cur || dist_out != nullptr);
  const int js = j + sh < N ? j + sh : N - 1;   // stored row of image row j

  float n[FP];
#pragma unroll
  for (int f = 0; f < FP; ++f) {
    float v = (live && f < F) ? nodes[((size_t)b * N + js) * F + f] : 0.f;
    n[f] = dist_param ? v / dist_param[0] : v;
  }
  float total = 0.f;
  for (int c0 = 0; c0 < Bc; c0 += CH) {   // Bc current rows: the local batch, or every rank's (sharded)
    __syncthreads();
    for (int e = threadIdx.x; e < CH * FP; e += 128) {
      const int r = e / FP, f = e - r * FP;
      sC[e] = (c0 + r < Bc && f < F) ? ws_cur[(size_t)(c0 + r) * F + f] : 0.f;
    }
    __syncthreads();
    const int rows = min(CH, Bc - c0);
    for (int r = 0; r < rows; ++r) {
      const f32x4* cp = reinterpret_cast<const f32x4*>(sC + r * FP);
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < FP / 4; ++q) {
        const f32x4 c = cp[q];
        const float d0 = c[0] - n[4 * q], d1 = c[1] - n[4 * q + 1];
        const float d2 = c[2] - n[4 * q + 2], d3 = c[3] - n[4 * q + 3];
        s = fmaf(d0, d0, s);
        s = fmaf(d1, d1, s);
        s = fmaf(d2, d2, s);
        s = fmaf(d3, d3, s);
      }
      total += sqrtf(s);
    }
  }
  if (!live) return;
  const float d = total / (float)Bc;
  if (dist_out) dist_out[(size_t)b * N + j] = d;
  if (j < cur) view_emit(adj, sel_row, b, cur, j, N, d < max_distance, bidirectional);
}


// ---------------------------------------------------------------------------
// EUCLID_CROSSBATCH on the matrix cores (B >= 32): the [N x B] block of squared distances of
// one graph's nodes against ALL current rows is |n|^2 + |c|^2 - 2 n.c^T, i.e. a [N x F] x [F x B]
// product (v_mfma_f32_32x32x2_f32) - the formulation torch.cdist itself switches to above 25
// rows.  One workgroup (8 waves) = one graph x 128 node rows.  The (scaled) current rows of a chunk
// of CB graphs are gathered straight from the state into LDS, transposed ([F][CB+1]: coalesced
// global reads along F, conflict-free LDS writes and MFMA operand reads), their squared norms
// computed in place - no separate gather kernel.  Wave w owns node rows 32 (w & 3) and the column
// half (w >> 2) of every chunk; sqrt and the mean over b' run on the accumulators, the two halves
// meet in LDS in fixed order.  Rows >= cur are skipped (distance.py:31-33).
// ---------------------------------------------------------------------------
template <int FT, int RB, int CB>   // F padded to 32 FT; RB node rows per workgroup; CB graphs per LDS chunk
__global__ __launch_bounds__(512) void k_euclid_mfma(
    View vw, const float* __restrict__ dist_param, float* __restrict__ adj,
    float* __restrict__ sel_row, float* __restrict__ dist_out, float max_distance, int bidirectional,
    int Bc, int N, int F) {
  const float* __restrict__ nodes = vw.nodes;
  constexpr int FP = 32 * FT, NS = FP + 1, CS = CB + 1;
  constexpr int RWN = RB / 32;              // 32-row blocks of the workgroup
  constexpr int CG = 8 / RWN;               // column groups: wave = (row block, column group)
  constexpr int HT = CB / (32 * CG);        // 32-column tiles per wave and chunk
  static_assert(RB % 32 == 0 && 8 % RWN == 0 && HT >= 1 && CB % (32 * CG) == 0, "tiling");
  const int b = blockIdx.y, j0 = blockIdx.x * RB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int rw = wave % RWN, ch = wave / RWN;
  int sh;
  const int cur = view_cur(vw, b, N, sh);
  if (j0 >= cur && dist_out == nullptr) return;   // whole block beyond the live rows

  extern __shared__ float smem[];
  float* sN = smem;                 // [RB][NS]   node rows (scaled)
  float* sC = sN + RB * NS;         // [FP][CS]   current rows, transposed, one chunk of CB graphs
  float* sNn = sC + FP * CS;        // [RB] |n|^2
  float* sCn = sNn + RB;            // [CB] |c|^2
  float* sPart = sCn + CB;          // [CG - 1][RB] row sums of column groups 1 ..

  const float inv_scale_den = dist_param ? dist_param[0] : 1.f;
  DSTAMP(0);
  // node rows of this block: every load in flight before the first LDS store (a load -> store loop
  // exposes one memory round trip per element)
  {
    constexpr int PER = RB * FP / 512;
    float v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 512 * i, r = e / FP, f = e % FP;
      const int j = j0 + r + sh;   // stored row of image row j0 + r
      v[i] = nodes[((size_t)b * N + (j < N ? j : N - 1)) * F + (f < F ? f : F - 1)];
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 512 * i, r = e / FP, f = e % FP;
      const float t = v[i];
      sN[r * NS + f] = (j0 + r < N && f < F) ? (dist_param ? t / inv_scale_den : t) : 0.f;
    }
  }
  __syncthreads();
  DSTAMP(1);
  if (tid < RB) {
    float s = 0.f;
    for (int f = 0; f < FP; ++f) s = fmaf(sN[tid * NS + f], sN[tid * NS + f], s);
    sNn[tid] = s;
  }
  DSTAMP(2);
  const int r_base = rw * 32;
  const bool wave_live = (j0 + r_base < cur) || dist_out != nullptr;
  float rowsum[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rowsum[r] = 0.f;

  for (int c0 = 0; c0 < Bc; c0 += CB) {   // Bc current rows: the local batch, or every rank's (sharded)
    __syncthreads();
    {   // current rows of this chunk of graphs -> [F][CS]; a batch of loads in flight before its stores
      constexpr int PER = FP * CB / 512, STEP = PER <= 16 ? PER : (PER % 16 == 0 ? 16 : 8);
      static_assert(PER % STEP == 0, "chunking");
#pragma unroll 1
      for (int i0 = 0; i0 < PER; i0 += STEP) {
        float v[STEP];
        if (vw.count || vw.cur_rows) {   // (uniform) the current nodes are the observations / the gathered rows
          const float* rows = vw.cur_rows ? vw.cur_rows : vw.obs;
#pragma unroll
          for (int i = 0; i < STEP; ++i) {
            const int e = tid + 512 * (i0 + i), c = e / FP, f = e % FP;
            const int g = c0 + c < Bc ? c0 + c : Bc - 1;
            v[i] = rows[(size_t)g * F + (f < F ? f : F - 1)];
          }
        } else {
#pragma unroll
          for (int i = 0; i < STEP; ++i) {
            const int e = tid + 512 * (i0 + i), c = e / FP, f = e % FP;
            const int g = c0 + c < Bc ? c0 + c : Bc - 1;
            int64_t cg = vw.cur_idx[g];
            cg = cg < 0 ? 0 : (cg > N - 1 ? N - 1 : cg);
            v[i] = nodes[((size_t)g * N + cg) * F + (f < F ? f : F - 1)];
          }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < STEP; ++i) {
          const int e = tid + 512 * (i0 + i), c = e / FP, f = e % FP;
          const float t = dist_param ? v[i] / inv_scale_den : v[i];
          sC[f * CS + c] = (f < F && c0 + c < Bc) ? t : 0.f;
        }
      }
    }
    __syncthreads();
    DSTAMP(3);
    if (tid < CB) {   // |c|^2, features in ascending order
      float s = 0.f;
      for (int f = 0; f < FP; ++f) s = fmaf(sC[f * CS + tid], sC[f * CS + tid], s);
      sCn[tid] = s;
    }
    __syncthreads();
    DSTAMP(4);
    if (wave_live) {
      float nn[16];   // |n|^2 of this lane's 16 accumulator rows
#pragma unroll
      for (int r = 0; r < 16; ++r) nn[r] = sNn[r_base + (r & 3) + 8 * (r >> 2) + 4 * lh];
      const int n_t = min(HT, (Bc - c0 - ch * HT * 32 + 31) / 32);   // column tiles of this wave in this chunk
      auto finish = [&](const f32x16& acc, int t) {
        const int ct = ch * HT + t;
        const int col = c0 + ct * 32 + li;
        const float cn = sCn[ct * 32 + li];
        const float keep = col < Bc ? 1.f : 0.f;
        // v_sqrt_f32 (1 ulp) instead of the correctly rounded library routine (a dozen instructions):
        // the epilogue is otherwise as long as the MFMA chain
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d2 = fmaf(-2.f, acc[r], nn[r] + cn);
          rowsum[r] = fmaf(keep, __builtin_amdgcn_sqrtf(fmaxf(d2, 0.f)), rowsum[r]);
        }
      };
      if constexpr (FT <= 2) {
        // operands in registers before the MFMA chain starts: A (this wave's 32 node rows) once per
        // chunk, B per column tile; with two tiles per chunk their chains are interleaved
        float av[FP / 2];
        {
          const float* ap = sN + (r_base + li) * NS + lh;        // A(i=row, k=f)
#pragma unroll
          for (int q = 0; q < FP / 2; ++q) av[q] = ap[2 * q];
        }
        auto load_b = [&](int t, float (&dst)[FP / 2]) {
          const float* bp = sC + lh * CS + (ch * HT + t) * 32 + li;   // B(k=f, j=b')
#pragma unroll
          for (int q = 0; q < FP / 2; ++q) dst[q] = bp[2 * q * CS];
        };
        if constexpr (HT >= 2) {
#pragma unroll
          for (int t = 0; t < HT; t += 2) {
            if (t < n_t) {
              float b0[FP / 2], b1[FP / 2];
              const bool two = t + 1 < n_t;
              load_b(t, b0);
              load_b(two ? t + 1 : t, b1);
              f32x16 acc0, acc1;
#pragma unroll
              for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
              for (int q = 0; q < FP / 2; ++q) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], b0[q], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], b1[q], acc1, 0, 0, 0);
              }
              finish(acc0, t);
              if (two) finish(acc1, t + 1);
            }
          }
        } else if (n_t > 0) {
          float b0[FP / 2];
          load_b(0, b0);
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int q = 0; q < FP / 2; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], b0[q], acc, 0, 0, 0);
          finish(acc, 0);
        }
      } else {
        // wide features: operands stay in LDS (the register-resident form would spill)
        for (int t = 0; t < n_t; ++t) {
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          const float* ap = sN + (r_base + li) * NS + lh;
          const float* bp = sC + lh * CS + (ch * HT + t) * 32 + li;
#pragma unroll 8
          for (int k = 0; k < FP; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], bp[k * CS], acc, 0, 0, 0);
          finish(acc, t);
        }
      }
    }
  }
  DSTAMP(5);
  // sum over the 32 columns held by the lanes of each half-wave, then the column groups in fixed order
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = rowsum[r];
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    rowsum[r] = v;
  }
  if (ch > 0 && li == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sPart[(ch - 1) * RB + r_base + (r & 3) + 8 * (r >> 2) + 4 * lh] = rowsum[r];
  }
  __syncthreads();
  if (!wave_live || ch > 0) return;
  if (li == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = r_base + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int j = j0 + rl;
      if (j >= N) continue;
      float tot = rowsum[r];
#pragma unroll
      for (int g = 0; g < CG - 1; ++g) tot += sPart[g * RB + rl];
      const float d = tot / (float)Bc;
      if (dist_out) dist_out[(size_t)b * N + j] = d;
      if (j < cur) view_emit(adj, sel_row, b, cur, j, N, d < max_distance, bidirectional);
    }
  }
  DSTAMP(6);
}

// The cached live-row step (rows_cached.hip: k_step_rows_cached_img's arithmetic, H1, H2 <= 32) riding on the distance
// kernel as the work of wave 0 behind the decisions - like the GNN tail of the LearnedEdge selection kernel: a
// chain from empty graphs whose selector is EuclideanEdge runs ONE launch per step instead of two (the second
// one's launch, ramp-up, weight loads and decision-row round trip: ~3 us of 21 at cfg3).  The four weight matrices
// are staged into LDS, k-major, with the kernel's other staging loads; the state is advanced in place.
struct StepTail {
  const float* params;      // packed GNN parameters (the biases)
  const float* image;       // gcm_dense_rows_cached_weight_image: [4][64][64], image[m][k][out]
  float *nodes, *adj;       // the donated state
  int64_t* count;
  float *cH, *cA, *cX;      // the chain's caches
  float* saved;             // the step's record (gcm_dense_rows_cached_layout), mx at its head
  size_t o_v, o_hdr, o_coef, o_live, total;   // total = 0: mx only
  uint32_t* flags;
  int act1, act2, H1, H2;
  int cur_host;             // >= 0: the row every graph's new node lands in, known to the host (a chain from empty
                            // graphs: the number of steps made so far) - nothing waits for the count; -1: read it
  uint32_t* abits;          // the chain's adjacency as bits, [B][N][4] (row i: bit j = adj[i, j]), or NULL: kept by the
                            // cached steps for the steady-state step below, which reads its live rows' adjacency from it
  size_t o_rows;            // TAIL = 2: the general record's rows section (rows_common.h: SavedLayout), rw floats a row
  int rw;
};

// ---------------------------------------------------------------------------
// The same product for F <= 64 (FT <= 2), laid out for a graph that is only partly filled: the work of a
// step is (live 32-row blocks) x (column tiles), and the block rows >= cur are skipped - so the tiles must be
// dealt so that every SIMD gets its share of the LIVE ones.  Wave w owns column tile (w & 3) of every
// 128-graph chunk (its B operand is read from LDS once per chunk) and the row blocks of parity (w >> 2);
// waves w and w + 4 share a SIMD, so each SIMD multiplies exactly `nb` tiles per chunk whatever nb is (the
// wave = (row block, column half) layout above keeps two SIMDs idle while a graph holds < 64 nodes and runs
// at the full graph's time from the first node on).  The chunks of current rows are double-buffered: the
// loads of chunk c + 1 are in flight during the products of chunk c.  Row sums meet in LDS in fixed order
// (chunks in sequence inside a wave, then the four column tiles): same decisions on every run.
// ---------------------------------------------------------------------------
template <int FT, int TAIL>   // TAIL: 0 the selector alone; 1 + the cached step (a chain that has not rolled yet); 2 + the
                               // steady-state step (every graph full: roll, live rows re-evaluated) - see behind the decisions
__global__ __launch_bounds__(1024) void k_euclid_mfma2(
    View vw, const float* __restrict__ dist_param, float* __restrict__ adj,
    float* __restrict__ sel_row, float* __restrict__ dist_out, float max_distance, int bidirectional,
    int Bc, int N, int F, StepTail tl) {
  const float* __restrict__ nodes = vw.nodes;
  constexpr int FP = 32 * FT, NS = FP + 1, RB = 128, CB = 128, CS = CB + 1;
  constexpr int NT = 1024;
  const int b = blockIdx.y, j0 = blockIdx.x * RB;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int ct = wave & 3, rb = wave >> 2;   // 16 waves: SIMD = column tile, its four waves = the row blocks
  int sh = 0;
  int cur;
  if (TAIL == 2) { cur = N - 1; sh = 1; }   // every graph is full: image row j is stored row j + 1 (gcm.py:323-355)
  else if (TAIL && tl.cur_host >= 0) cur = tl.cur_host < N ? tl.cur_host : N - 1;   // (uniform; no load behind it)
  else cur = view_cur(vw, b, N, sh);
  const int nb = dist_out ? RB / 32 : max(0, min(RB / 32, (cur - j0 + 31) / 32));   // live 32-row blocks
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sN = smem;                   // [RB][NS]      node rows (scaled)
  float* sC = sN + RB * NS;           // [2][FP][CS]   current rows, transposed, two chunks of CB graphs
  float* sNn = sC + 2 * FP * CS;      // [RB]   |n|^2
  float* sCn = sNn + RB;              // [2][CB] |c|^2
  float* sPart = sCn + 2 * CB;        // [4][2][RB] sums over b' per (column tile, lane half)
  float* sW = sPart + 8 * RB;         // TAIL: [2 FP + 64][32] W_rel1 | W_root1 (k < FP each) | W_rel2 | W_root2 (k < 32 each), k-major
  float* sDec = sW + (2 * FP + 64) * 32;   // TAIL: [RB] this step's decisions
  float* sHc = sDec + RB;                  // TAIL: [N][H1] the h1 cache of this graph (flat copy; H1 % 4 == 0); TAIL = 2: the live rows' h1
  uint32_t* sBits = reinterpret_cast<uint32_t*>(sHc + 128 * 32);   // TAIL = 2: [RB][4] the adjacency bits of this graph

  const float inv_scale_den = dist_param ? dist_param[0] : 1.f;
  const float* crows = vw.cur_rows ? vw.cur_rows : vw.obs;
  const bool from_rows = vw.count || vw.cur_rows;   // (uniform) the current nodes are the observations / gathered rows
  // Staging: thread t holds SEG = FP / 8 consecutive features (segment t & 7) of row t >> 3 of a 128-row tile -
  // 16-byte loads when F allows, conflict-free LDS stores in both layouts, and the row's squared norm is the
  // sum over the eight lanes of a row (two quad permutes and a half-row mirror on the DPP path - no LDS pass,
  // no second barrier).
  constexpr int SEG = FP / 8;
  static_assert(RB * FP == NT * SEG && CB * FP == NT * SEG, "one segment per thread");
  const int srow = tid >> 3, sf0 = (tid & 7) * SEG;
  const bool vec4 = (F & 3) == 0;
  auto load_seg = [&](const float* __restrict__ row, float (&v)[SEG]) __attribute__((always_inline)) {
    if (vec4) {
#pragma unroll
      for (int k = 0; k < SEG; k += 4) {
        const int f = sf0 + k < F ? sf0 + k : F - 4;
        // (the compiler's own 4-vector: ONE global_load_dwordx4 - HIP's float4 struct is split into scalars and
        //  re-merged only sometimes, and the staging phase is bound by the number of load instructions)
        const f32x4 t = *reinterpret_cast<const f32x4*>(row + f);
        v[k] = t[0]; v[k + 1] = t[1]; v[k + 2] = t[2]; v[k + 3] = t[3];
      }
    } else {
#pragma unroll
      for (int k = 0; k < SEG; ++k) v[k] = row[sf0 + k < F ? sf0 + k : F - 1];
    }
  };
  auto seg_norm = [&](const float (&v)[SEG]) __attribute__((always_inline)) {
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < SEG; ++k) q = fmaf(v[k], v[k], q);
    q += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(q), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    q += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(q), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    q += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(q), 0x141, 0xF, 0xF, true));   // row_half_mirror
    return q;
  };
  auto load_chunk = [&](int c0, float (&v)[SEG]) __attribute__((always_inline)) {
    const int g = c0 + srow < Bc ? c0 + srow : Bc - 1;
    if (from_rows) {
      load_seg(crows + (size_t)g * F, v);
    } else {
      int64_t cg = vw.cur_idx[g];
      cg = cg < 0 ? 0 : (cg > N - 1 ? N - 1 : cg);
      load_seg(nodes + ((size_t)g * N + cg) * F, v);
    }
  };
  auto store_chunk = [&](float* dst, float* dst_n, int c0, float (&v)[SEG]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < SEG; ++k) {
      const float t = dist_param ? v[k] / inv_scale_den : v[k];
      v[k] = (sf0 + k < F && c0 + srow < Bc) ? t : 0.f;
      dst[(sf0 + k) * CS + srow] = v[k];
    }
    const float q = seg_norm(v);
    if ((tid & 7) == 0) dst_n[srow] = q;
  };
  // (Tried in round 4: both chunks of current rows staged before the one barrier - they fit the two LDS buffers at
  //  Bc <= 256 - and no barrier per chunk, hoping the four waves of a SIMD would drift apart and overlap one's sqrt
  //  epilogue with another's MFMA chain: they start in lockstep and stay there, and the longer staging cost 1 us.)
  // TAIL: the observation of this graph, for the cached step behind the decisions (requested now: one register)
  float pf_xc = 0.f;
  if (TAIL && wave <= 1) pf_xc = vw.obs[(unsigned)b * (unsigned)F + (unsigned)(lane < F ? lane : F - 1)];
  DSTAMP(0);
  // TAIL = 2 - the overflow roll (gcm.py:323-355) without reading the state back.  The adjacency of such a chain is
  // 0 / 1 and the chain keeps it as bits: the rolled rows 0 .. N - 2 (new[i, j] = old[i + 1, j + 1], last column empty)
  // do not depend on this step's decisions, so they are WRITTEN from the bit image behind the staging barrier - 16.8 MB
  // of stores per step at cfg3 that drain under the distance phase, no loads of the old matrix - instead of moving
  // 2 x 16.8 MB in place behind it.  The node rows are in this kernel's registers anyway (staged for the distances):
  // stored one row up at the same point.
  constexpr int WN = (2 * FP + 64) * 32, NIW = WN / NT;
  static_assert(WN % NT == 0, "whole rounds of the workgroup over the weight image");
  const bool hc_lds = TAIL == 1 && (tl.H1 & 3) == 0 && N * tl.H1 <= 4 * NT;   // h1 cache staged in LDS
  constexpr int RA = 128 * 128 / 4 / NT;
  uint4 wb[RA], wa[RA];   // old rows row + 1 (what moves in) and row (what the fp32 matrix holds there now) of the bit image
  const unsigned n_magic = 0xFFFFFFFFu / (unsigned)N + 1u;   // e / N = umulhi(e, n_magic) for e < 2^16
  float vn[SEG];
  {
    // the first chunk of current rows (its addresses do not depend on this graph's fill level: in flight while
    // `cur` arrives) and the node rows of the live blocks: every load in flight before the first LDS store
    float vc[SEG];
    load_chunk(0, vc);
    if (!TAIL && j0 >= cur && dist_out == nullptr) return;   // whole block beyond the live rows (uniform)
    const int j = j0 + srow + sh;   // stored row of image row j0 + srow
    const bool row_live = srow < 32 * nb && j0 + srow < N;
    if (row_live) {
      load_seg(nodes + ((size_t)b * N + (j < N ? j : N - 1)) * F, vn);
    } else {
#pragma unroll
      for (int k = 0; k < SEG; ++k) vn[k] = 0.f;
    }
    if (TAIL == 2) {   // (requested behind the staging loads: one round trip for all)
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const int row = (int)__umulhi((unsigned)(4 * (tid + NT * i)), n_magic);
        wb[i] = reinterpret_cast<const uint4*>(tl.abits + ((size_t)b * N + (row + 1 < N ? row + 1 : N - 1)) * 4)[0];
        wa[i] = reinterpret_cast<const uint4*>(tl.abits + ((size_t)b * N + (row < N ? row : N - 1)) * 4)[0];   // (N < 128: items beyond the matrix)
      }
    }
    // TAIL: what the cached step behind the decisions reads that does not depend on them - the four weight matrices
    // (lane-major image) and the h1 cache of this graph - rides the staging round trip and is in LDS behind the staging
    // barrier (their LDS regions are not touched by the distance phase).  Until round 6 these loads were issued behind
    // the MFMA loop and stored behind the row-sum barrier (wave 0's stamps: 2.0 k -> 1.0 k cycles between the last
    // chunk and the decisions; the kernel as a whole 13.01 -> 12.99 us - the other waves' skew covered most of it);
    // before that the tail started with them.
    float pfw[NIW];
    float4 pfh;
    if (TAIL) {
#pragma unroll
      for (int i = 0; i < NIW; ++i) {
        const int e = tid + i * NT;
        const int k = e >> 5, h = e & 31;                     // row k of the k-major image, output h
        const int m = k < FP ? 0 : (k < 2 * FP ? 1 : (k < 2 * FP + 32 ? 2 : 3));
        const int kk = m == 0 ? k : (m == 1 ? k - FP : (m == 2 ? k - 2 * FP : k - 2 * FP - 32));
        pfw[i] = tl.image[m * 4096 + kk * 64 + h];
      }
      if (hc_lds) {
        const int n4 = N * tl.H1 / 4;
        pfh = reinterpret_cast<const float4*>(tl.cH + (size_t)b * N * tl.H1)[tid < n4 ? tid : n4 - 1];
      }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < SEG; ++k) {
      const float t = dist_param ? vn[k] / inv_scale_den : vn[k];
      vn[k] = (row_live && sf0 + k < F) ? t : 0.f;
      sN[srow * NS + sf0 + k] = vn[k];
    }
    const float q = seg_norm(vn);
    if ((tid & 7) == 0) sNn[srow] = q;
    store_chunk(sC, sCn, 0, vc);
    if (TAIL) {
#pragma unroll
      for (int i = 0; i < NIW; ++i) sW[tid + i * NT] = pfw[i];
      if (hc_lds && tid < N * tl.H1 / 4) reinterpret_cast<float4*>(sHc)[tid] = pfh;
    }
  }
  __syncthreads();
  if (TAIL == 2) {
    if (srow < N - 1) {   // stored row srow + 1 -> row srow (every thread's load of its row has landed)
      float* dst = tl.nodes + ((size_t)b * N + srow) * F + sf0;
#pragma unroll
      for (int k = 0; k < SEG; k += 4)
        if (sf0 + k < F) *reinterpret_cast<f32x4*>(dst + k) = f32x4{vn[k], vn[k + 1], vn[k + 2], vn[k + 3]};
    }
    float* ga = tl.adj + (size_t)b * N * N;
    const int a_end = (N - 1) * N;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int e = 4 * (tid + NT * i);
      if (e < a_end) {
        const int row = (int)__umulhi((unsigned)e, n_magic), c1 = e - row * N + 1;   // new row, columns c .. c + 3 <- old row + 1, columns c + 1 .. c + 4
        const uint4 w = wb[i];
        const int wi = c1 >> 5;
        const uint32_t lo = wi == 0 ? w.x : (wi == 1 ? w.y : (wi == 2 ? w.z : w.w));
        const uint32_t hi = wi == 0 ? w.y : (wi == 1 ? w.z : (wi == 2 ? w.w : 0u));
        uint32_t b4 = (uint32_t)((((unsigned long long)hi << 32) | lo) >> (c1 & 31)) & 15u;
        const int n_ok = N - c1;                               // columns c1 .. N - 1 exist (the last column stays empty)
        b4 &= n_ok >= 4 ? 15u : ((1u << n_ok) - 1u);
        // what the matrix holds in this piece now: the same four columns of old row `row` (the chain's bit image mirrors
        // the fp32 adjacency) - a piece whose bits do not change is not stored (the band / cluster structure of a
        // selector's decisions moves onto itself under the roll: most of the 16.8 MB a step stayed what it was)
        const uint4 wo = wa[i];
        const int c0 = c1 - 1, wj = c0 >> 5;
        const uint32_t ow = wj == 0 ? wo.x : (wj == 1 ? wo.y : (wj == 2 ? wo.z : wo.w));
        const uint32_t o4 = (ow >> (c0 & 31)) & 15u;          // (c0 is a multiple of 4: the piece lies inside one word)
        if (b4 != o4) {
          f32x4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = ((b4 >> q) & 1u) ? 1.f : 0.f;
          *reinterpret_cast<f32x4*>(ga + e) = v;
        }
      }
    }
  }
  DSTAMP(1);
  DSTAMP(2);

  // euclid_chain.h: acc(i = current row, j = node) = C' N' with the squared norms as one more k step - the accumulator
  // IS |c|^2 + |n|^2 - 2 c.n, a lane holds one node and sixteen current rows: sqrt and the sum over b' stay in the lane
  // (no epilogue as long as the MFMA chain, no butterfly over the columns).  Same arithmetic, value for value, as the
  // time-parallel form (euclid_tp.hip: k_euclid_tp).
  constexpr int KQ = FP / 2;
  float part = 0.f;
  float nv[KQ + 1];   // N'(k, j = node): this wave's 32 node rows (times -2) | (1, |n|^2), in registers for every chunk
  if (rb < nb) {
    const float* np = sN + (rb * 32 + li) * NS + lh;
#pragma unroll
    for (int q = 0; q < KQ; ++q) nv[q] = -2.f * np[2 * q];
    nv[KQ] = lh ? sNn[rb * 32 + li] : 1.f;
  }

  int buf = 0;
  for (int c0 = 0; c0 < Bc; c0 += CB, buf ^= 1) {
    const bool has_next = c0 + CB < Bc;
    float vnext[SEG];
    if (has_next) load_chunk(c0 + CB, vnext);
    const float* sCb = sC + buf * FP * CS;
    if (c0 + ct * 32 < Bc && rb < nb) {   // this wave's column tile holds graphs, and its row block is live
      const float c_last = lh ? 1.f : sCn[buf * CB + ct * 32 + li];
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      float unused;
      gcm_dist_chain<KQ, (GCM_CHAIN_W < KQ ? GCM_CHAIN_W : KQ), false>(acc, nv, sCb + lh * CS + ct * 32 + li, CS, c_last, acc, unused);   // C'(i = b', k)
      part += gcm_dist_tile_sum(acc, lh, Bc - (c0 + ct * 32));
    }
    if (c0 == 0) DSTAMP(3);
    if (has_next) store_chunk(sC + (buf ^ 1) * FP * CS, sCn + (buf ^ 1) * CB, c0 + CB, vnext);
    __syncthreads();   // chunk c0 is consumed, chunk c0 + CB (rows and norms) is in LDS
    if (c0 == 0) DSTAMP(4);
  }
  DSTAMP(5);
  uint32_t ob0 = 0, ob1 = 0, ob2 = 0, ob3 = 0;   // TAIL = 2: old row tid + 1 of the bits -> (shifted below) image row tid
  if (TAIL == 2 && tid < N - 1) {
    const uint4 t = reinterpret_cast<const uint4*>(tl.abits + ((size_t)b * N + tid + 1) * 4)[0];
    ob0 = t.x; ob1 = t.y; ob2 = t.z; ob3 = t.w;
  }
  // the lane's sum over b' of its node, per (column tile, lane half); met in fixed order below
  if (rb < nb) sPart[(2 * ct + lh) * RB + rb * 32 + li] = part;
  __syncthreads();
  if (tid < RB && tid < 32 * nb) {
    const int j = j0 + tid;
    if (j < N) {
      const int tiles = min(4, (Bc + 31) / 32);   // (column tiles that ever held graphs)
      const float d = gcm_dist_total(sPart, RB, tid, tiles) / (float)Bc;
      if (dist_out) dist_out[(size_t)b * N + j] = d;
      if (TAIL) sDec[tid] = (j < cur && d < max_distance) ? 1.f : 0.f;
      else if (j < cur) view_emit(adj, sel_row, b, cur, j, N, d < max_distance, bidirectional);
    }
  }
  DSTAMP(6);
  if (TAIL == 2) {
    // ---- the steady-state step: every graph is full, the step drops the oldest node (gcm.py:263-271, 323-355).  The
    //      layer-1 rows of older nodes are no longer final (a row loses the sources the roll drops), so the live rows -
    //      the selected ones - are re-evaluated here: their adjacency from the chain's bit image (one 16-byte row each,
    //      fetched at kernel start: no round trip behind the decisions), their sources from the node image staged for
    //      the distances, one live row per wave at a time; wave 0 then row cur.  The record is the GENERAL live-row
    //      record (rows_common.h: the rows h1 | agg1 | x travel in it).  dist_param == NULL (the image is unscaled).
    if (tid < RB && !(tid < 32 * nb && j0 + tid < N)) sDec[tid] = 0.f;   // rows beyond the live blocks
    float* sXc = reinterpret_cast<float*>(sBits + RB * 4);    // [64] the observation of this graph
    if (wave == 0 && lane < FP) sXc[lane] = lane < F ? pf_xc : 0.f;
    if (tid < RB) {   // the adjacency bits after the roll: row i <- (old row i + 1) >> 1 (column 0 falls off)
      uint32_t* o = sBits + tid * 4;
      o[0] = (ob0 >> 1) | (ob1 << 31); o[1] = (ob1 >> 1) | (ob2 << 31); o[2] = (ob2 >> 1) | (ob3 << 31); o[3] = ob3 >> 1;
    }
    __syncthreads();   // decisions, weights and bits are in LDS
    DSTAMP(12);
    const int H1 = tl.H1, H2 = tl.H2;
    const unsigned gb = (unsigned)b;
    // the selected rows (image rows j < N - 1): every wave takes the same two ballots
    const unsigned long long m0 = __ballot(lane < N - 1 && sDec[lane] != 0.f);
    const unsigned long long m1 = __ballot(lane + 64 < N - 1 && sDec[(lane + 64) & (RB - 1)] != 0.f);
    const int n0 = __popcll(m0), n_sel = n0 + __popcll(m1);
    {   // the state: row N - 1 (the new node, its decisions); the bit image (the rolled rows went out behind the staging)
      float* ga = tl.adj + (size_t)b * N * N;
      float* gn = tl.nodes + (size_t)b * N * F;
      const int a_end = (N - 1) * N, n_end = (N - 1) * F;
      if (tid < N) ga[a_end + tid] = tid < N - 1 ? sDec[tid] : 0.f;
      if (wave == 1 && lane < F) gn[n_end + lane] = vw.obs[gb * (unsigned)F + (unsigned)lane];
      if (tid < N) {
        uint4 t;
        if (tid < N - 1) { const uint32_t* q = sBits + tid * 4; t = make_uint4(q[0], q[1], q[2], q[3]); }
        else t = make_uint4((uint32_t)m0, (uint32_t)(m0 >> 32), (uint32_t)m1, (uint32_t)(m1 >> 32));
        reinterpret_cast<uint4*>(tl.abits + ((size_t)b * N + tid) * 4)[0] = t;
      }
    }
    DSTAMP(13);
    // this wave's weights of layer 1, once: lanes 0-31 k < FP / 2, lanes 32-63 the rest (met by one cross-half add)
    constexpr int KH = FP / 2;
    const int hl = lane & 31, kh = lane >> 5, fl = lane < FP ? lane : FP - 1;
    float wr[KH], wt[KH];
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      wr[k] = sW[(kh * KH + k) * 32 + hl];
      wt[k] = sW[(FP + kh * KH + k) * 32 + hl];
    }
    const float* b1p = tl.params + 2 * (size_t)H1 * F;
    const float* b2p = b1p + H1 + 2 * (size_t)H2 * H1;
    const float bias1 = b1p[hl < H1 ? hl : H1 - 1], bias2 = b2p[hl < H2 ? hl : H2 - 1];
    const int act1_v = gcm_vgpr(tl.act1), act2_v = gcm_vgpr(tl.act2);
    float* svw = sC + wave * 2 * FP;                          // (the chunk buffers are free) this wave's agg1 | x
    float* sH1 = sHc;                                         // [n_sel][32] the live rows' h1, list order
    const bool rec = tl.total != 0;
    float* rows = tl.saved + tl.o_rows + (size_t)gb * N * tl.rw;
    auto layer1 = [&](float agg, float x) __attribute__((always_inline)) {   // lane f holds agg1[f], x[f] -> h1[lane & 31]
      if (lane < FP) { svw[lane] = agg; svw[FP + lane] = x; }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float pa = 0.f, pb = 0.f;
#pragma unroll
      for (int k4 = 0; k4 < KH / 4; ++k4) {
        const float4 a = *reinterpret_cast<const float4*>(svw + kh * KH + 4 * k4);
        const float4 xx = *reinterpret_cast<const float4*>(svw + FP + kh * KH + 4 * k4);
        pa = fmaf(wr[4 * k4], a.x, pa); pb = fmaf(wt[4 * k4], xx.x, pb);
        pa = fmaf(wr[4 * k4 + 1], a.y, pa); pb = fmaf(wt[4 * k4 + 1], xx.y, pb);
        pa = fmaf(wr[4 * k4 + 2], a.z, pa); pb = fmaf(wt[4 * k4 + 2], xx.z, pb);
        pa = fmaf(wr[4 * k4 + 3], a.w, pa); pb = fmaf(wt[4 * k4 + 3], xx.w, pb);
      }
      float p1 = pa + pb;
      p1 += __shfl_xor(p1, 32);
      p1 += bias1;
      __builtin_amdgcn_wave_barrier();                        // (the next row overwrites svw)
      return hl < H1 ? gcm_act_sel(p1, act1_v) : 0.f;
    };
    DSTAMP(14);
    // live rows: list position l (ascending j) -> wave l mod 16.  No scalar bit loops (each find-first-bit step
    // compiled to a branch tree with a VALU round trip): a selected row's position is its rank among the mask bits
    // below it, its sources become a compact ascending list in this wave's LDS scratch, gathered eight per trip.
    {
      const unsigned long long below = (1ull << lane) - 1ull;
      const int rank0 = __popcll(m0 & below), rank1 = n0 + __popcll(m1 & below);
      int* widx = reinterpret_cast<int*>(sC + 16 * 2 * FP) + wave * 128;     // [128] this wave's source list
      // (list position n_sel is row cur itself: its sources are the selected rows, its features the observation - its
      //  layer 1 runs beside the others instead of behind the barrier on wave 0)
      for (int l = wave; l <= n_sel; l += 16) {
        const bool is_cur = l == n_sel;
        const unsigned long long h0 = __ballot(((m0 >> lane) & 1ull) && rank0 == l);
        const unsigned long long h1 = __ballot(((m1 >> lane) & 1ull) && rank1 == l);
        const int j = is_cur ? N - 1 : (h0 ? __builtin_ctzll(h0) : 64 + __builtin_ctzll(h1));     // (uniform)
        const uint4 wq = *reinterpret_cast<const uint4*>(sBits + (is_cur ? 0 : j) * 4);
        const unsigned long long s0 = is_cur ? m0 : (((unsigned long long)wq.y << 32) | wq.x);
        const unsigned long long s1 = is_cur ? m1 : (((unsigned long long)wq.w << 32) | wq.z);
        const int c0 = __popcll(s0), n_src = __builtin_amdgcn_readfirstlane(c0 + __popcll(s1));
        if ((s0 >> lane) & 1ull) widx[__popcll(s0 & below)] = lane;
        if ((s1 >> lane) & 1ull) widx[c0 + __popcll(s1 & below)] = lane + 64;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float x = is_cur ? sXc[fl] : sN[j * NS + fl];
        float agg = 0.f;
#pragma unroll 1
        for (int q0 = 0; q0 < n_src; q0 += 8) {                  // sources ascending (adj[j, k] = 1, k < j), eight per trip
          const int4 ia = *reinterpret_cast<const int4*>(widx + q0), ib = *reinterpret_cast<const int4*>(widx + q0 + 4);
          const int ks[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
          float bx[8];
#pragma unroll
          for (int qq = 0; qq < 8; ++qq) {
            const bool any = q0 + qq < n_src;
            const float tx = sN[(any ? ks[qq] : 0) * NS + fl];
            bx[qq] = any ? tx : 0.f;
          }
#pragma unroll
          for (int qq = 0; qq < 8; ++qq) agg += bx[qq];
        }
        __builtin_amdgcn_wave_barrier();                          // (the next row rewrites widx)
        const float h = layer1(lane < F ? agg : 0.f, lane < F ? x : 0.f);
        if (lane < 32) sH1[l * 32 + lane] = h;
        if (rec) {
          float* row = rows + (size_t)l * tl.rw;
          if (lane < H1) row[lane] = h;
          if (lane < F) { row[H1 + lane] = agg; row[H1 + F + lane] = x; }
        }
      }
    }
    DSTAMP(15);
    __syncthreads();
    if (wave != 0) return;
    DSTAMP(16);
    // ---- row cur = N - 1 on wave 0: agg2 over the live rows' h1 (ascending), layer 2
    float agg2 = 0.f;
#pragma unroll 1
    for (int l0 = 0; l0 < n_sel; l0 += 8) {
      float bh[8];
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) {
        const float th = sH1[((l0 + qq) & 127) * 32 + hl];
        bh[qq] = l0 + qq < n_sel ? th : 0.f;
      }
#pragma unroll
      for (int qq = 0; qq < 8; ++qq) agg2 += bh[qq];
    }
    agg2 = hl < H1 ? agg2 : 0.f;
    DSTAMP(17);
    const float h1c = sH1[(n_sel & 127) * 32 + hl];
    DSTAMP(18);
    if (lane < 32) { svw[lane] = agg2; svw[32 + lane] = h1c; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float p2;
    {   // layer 2: lanes 0-31 W_rel2 . agg2, lanes 32-63 W_root2 . h1cur
      float w2r[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) w2r[k] = sW[(2 * FP + kh * 32 + k) * 32 + hl];
      float pa = 0.f;
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) {
        const float4 a = *reinterpret_cast<const float4*>(svw + kh * 32 + 4 * k4);
        pa = fmaf(w2r[4 * k4], a.x, pa);
        pa = fmaf(w2r[4 * k4 + 1], a.y, pa);
        pa = fmaf(w2r[4 * k4 + 2], a.z, pa);
        pa = fmaf(w2r[4 * k4 + 3], a.w, pa);
      }
      p2 = pa + __shfl_xor(pa, 32) + bias2;
    }
    const float v = gcm_act_sel(p2, act2_v);
    if (lane < H2) tl.saved[gb * H2 + lane] = v;
    if (rec) {
      if (lane < H1) {
        tl.saved[tl.o_v + gb * 2 * H1 + lane] = agg2;
        tl.saved[tl.o_v + gb * 2 * H1 + H1 + lane] = h1c;
      }
      float* coef = tl.saved + tl.o_coef + (size_t)gb * N;    // (row cur - last in the list - was written with the live rows)
      for (int l = lane; l <= n_sel; l += 64) coef[l] = l < n_sel ? 1.f : 0.f;
      if (lane == 0) {
        int* hdr = reinterpret_cast<int*>(tl.saved + tl.o_hdr) + 4 * gb;
        hdr[0] = n_sel + 1; hdr[1] = n_sel; hdr[2] = N - 1; hdr[3] = 1;
      }
    }
    const bool nonfinite = __any(lane < H2 && !isfinite(v));
    if (lane == 0 && (nonfinite || gb == 0))
      atomicOr(tl.flags, (nonfinite ? GCM_FLAG_NONFINITE : 0u) | (gb == 0 ? GCM_FLAG_WRAPPED : 0u));   // gcm.py:264-266
    DSTAMP(19);
    return;
  }
  if (TAIL == 1) {
    if (tid < RB && !(tid < 32 * nb && j0 + tid < N)) sDec[tid] = 0.f;   // rows beyond the live blocks
    __syncthreads();
    // ---- the cached live-row step on row cur (rows_cached.hip).  Wave 0 gathers the selected rows while the other
    //      fifteen waves, done with their tiles, bring the four weight matrices into LDS (k-major: one round trip,
    //      under the gather's); then wave 0 alone: two matrix-vector products, the state, the caches, the record.
    const int H1 = tl.H1, H2 = tl.H2;
    const unsigned gb = (unsigned)b;
    const int fl = lane < F ? lane : F - 1, hl = lane & 31, kh = lane >> 5;
    const int64_t n64 = tl.cur_host >= 0 ? (int64_t)tl.cur_host : vw.count[b];
    const bool bad = n64 < 0 || n64 >= N;                     // (a chain from empty graphs never rolls)
    // The gather of the selected rows from the LDS images (node rows staged for the distances, the h1 cache staged at
    // kernel start) is FOUR waves' work, one 32-row block each (round 6): wave 0 alone walked a rank-compacted list of all
    // of them, eight per trip - 2.4 k cycles of the tail on a full graph with its ~16 near neighbours.  Waves 0, 2, 3, 4
    // sum their block's selected rows (a scalar loop over the block's mask bits), the partial sums meet in LDS behind one
    // barrier that wave 1 passes too before its bookkeeping (a live wave that never arrives would hold it); the other
    // waves are gone by then.
    const bool lds_gather = hc_lds && !dist_param && sh == 0;   // (uniform)
    if (wave > (lds_gather ? 4 : 1)) return;                  // (their share was done behind the MFMA loop)
    DSTAMP(7);
    unsigned long long m0 = 0, m1 = 0;
    float agg1 = 0.f, agg2 = 0.f;
    const float xc = pf_xc;
    float* sv = sPart;                                        // (free: every row sum has been read)
    float* sGp = sPart + 4 * RB;                              // [4][128] the blocks' partial sums: x (64) | h1 (32)
    m0 = __ballot(!bad && lane < cur && sDec[lane] != 0.f);
    m1 = __ballot(!bad && lane + 64 < cur && lane + 64 < RB && sDec[(lane + 64) & (RB - 1)] != 0.f);
    const int hc = hl < H1 ? hl : H1 - 1;
    // (round 6) What of the two matrix-vector products does not wait for the gather is taken out of wave 0's chain behind
    // it: wave 1 - idle until the barrier - forms the root term W_root1 . x_cur (the same fma chain per lane as wave 0 ran: same bits)
    // and leaves it in LDS (the chunk norms' words: free behind the MFMA loop).
    constexpr int KH = FP / 2;
    float* sXc = sCn;                                         // [FP] x_cur   (wave 1 -> wave 1)
    float* sRoot = sCn + FP;                                  // [64] the root term's half sums, lane-major (wave 1 -> wave 0)
    if (lds_gather && wave == 1) {
      if (lane < FP) sXc[lane] = lane < F ? xc : 0.f;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float pb = 0.f;
#pragma unroll
      for (int k4 = 0; k4 < KH / 4; ++k4) {
        const float4 x = *reinterpret_cast<const float4*>(sXc + kh * KH + 4 * k4);
        pb = fmaf(sW[(FP + kh * KH + 4 * k4) * 32 + hl], x.x, pb);
        pb = fmaf(sW[(FP + kh * KH + 4 * k4 + 1) * 32 + hl], x.y, pb);
        pb = fmaf(sW[(FP + kh * KH + 4 * k4 + 2) * 32 + hl], x.z, pb);
        pb = fmaf(sW[(FP + kh * KH + 4 * k4 + 3) * 32 + hl], x.w, pb);
      }
      sRoot[lane] = pb;
    }
    if (lds_gather) {
      if (wave != 1) {
        const int kb = wave == 0 ? 0 : wave - 1;              // this wave's 32-row block
        uint32_t mk = (uint32_t)((kb < 2 ? m0 : m1) >> (32 * (kb & 1)));
        float ax0 = 0.f, ax1 = 0.f, ah0 = 0.f, ah1 = 0.f;
        while (mk) {                                          // (uniform) two rows a trip, ascending
          const int ja = __builtin_ctz(mk);
          mk &= mk - 1;
          const bool two = mk != 0;
          const int jb = two ? __builtin_ctz(mk) : ja;
          mk &= two ? mk - 1 : mk;
          const float xa = sN[(32 * kb + ja) * NS + fl], ha = sHc[(32 * kb + ja) * H1 + hc];
          const float xb = sN[(32 * kb + jb) * NS + fl], hb = sHc[(32 * kb + jb) * H1 + hc];
          ax0 += xa; ah0 += ha;
          ax1 += two ? xb : 0.f; ah1 += two ? hb : 0.f;
        }
        sGp[kb * 128 + lane] = ax0 + ax1;
        if (lane < 32) sGp[kb * 128 + 64 + lane] = ah0 + ah1;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (wave > 1) return;
    }
    if (wave == 1) {
      // ---- wave 1: what of the step depends on the decisions alone - the state's row cur (observation, adjacency
      //      row, bit image, count) and the record's live list - beside wave 0's gather and products, not behind them
      //      (it was a quarter of the tail)
      const unsigned rc = gb * (unsigned)N + (unsigned)cur;
      if (!bad) {
        if (lane < F) {
          tl.nodes[rc * F + lane] = xc;
          tl.cX[rc * F + lane] = xc;
        }
        float* arow = tl.adj + (size_t)rc * N;
        if (lane < N && ((m0 >> lane) & 1ull)) arow[lane] = 1.f;
        if (lane + 64 < N && ((m1 >> lane) & 1ull)) arow[lane + 64] = 1.f;
        if (tl.abits && lane == 0)      // (the bit image the steady-state step reads its live rows' adjacency from)
          reinterpret_cast<uint4*>(tl.abits + (size_t)rc * 4)[0] =
              make_uint4((uint32_t)m0, (uint32_t)(m0 >> 32), (uint32_t)m1, (uint32_t)(m1 >> 32));
        if (lane == 0) tl.count[gb] = cur + 1;
      }
      if (tl.total) {
        const unsigned long long l0 = m0 | (cur < 64 ? 1ull << cur : 0ull), l1 = m1 | (cur >= 64 ? 1ull << (cur - 64) : 0ull);
        int* live = reinterpret_cast<int*>(tl.saved + tl.o_live) + gb * N;
        float* coef = tl.saved + tl.o_coef + gb * N;
        const bool in0 = (l0 >> lane) & 1ull, in1 = (l1 >> lane) & 1ull;
        const int pos0 = __popcll(l0 & ((1ull << lane) - 1ull));
        const int pos1 = __popcll(l0) + __popcll(l1 & ((1ull << lane) - 1ull));
        if (in0) { live[pos0] = lane; coef[pos0] = lane == cur ? 0.f : 1.f; }
        if (in1) { live[pos1] = lane + 64; coef[pos1] = lane + 64 == cur ? 0.f : 1.f; }
        if (lane == 0) {
          int* hdr = reinterpret_cast<int*>(tl.saved + tl.o_hdr) + 4 * gb;
          const int L = __popcll(l0) + __popcll(l1);
          const int l_cur = cur < 64 ? __popcll(l0 & ((1ull << cur) - 1ull))
                                     : __popcll(l0) + __popcll(l1 & ((1ull << (cur - 64)) - 1ull));
          hdr[0] = L; hdr[1] = l_cur; hdr[2] = cur; hdr[3] = 0;
        }
      }
      return;
    }
    if (lds_gather) {
      // (the four blocks' partial sums, in block order)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) { agg1 += sGp[kb * 128 + lane]; agg2 += sGp[kb * 128 + 64 + (lane & 31)]; }
    } else {
      unsigned long long a0 = m0, a1 = m1;
      while (a0 | a1) {          // eight rows per round trip, added in ascending order
        float bx[8], bh[8];
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) {
          const bool any = (a0 | a1) != 0;
          const int j = !any ? 0 : (a0 ? __builtin_ctzll(a0) : 64 + __builtin_ctzll(a1));
          const bool low = a0 != 0;
          a0 &= low ? a0 - 1 : a0;
          a1 &= (low || !any) ? a1 : a1 - 1;
          const unsigned rj = gb * (unsigned)N + (unsigned)j;
          const float tx = tl.nodes[rj * F + fl], th = tl.cH[rj * H1 + hc];
          bx[qq] = any ? tx : 0.f;
          bh[qq] = any ? th : 0.f;
        }
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) { agg1 += bx[qq]; agg2 += bh[qq]; }
      }
    }
    agg1 = lane < F ? agg1 : 0.f;
    agg2 = lane < H1 ? agg2 : 0.f;
    if (lane < FP) { sv[lane] = lane < F ? agg1 : 0.f; sv[FP + lane] = lane < F ? xc : 0.f; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    DSTAMP(8);
    const float* b1p = tl.params + 2 * (size_t)H1 * F;
    const float* b2p = b1p + H1 + 2 * (size_t)H2 * H1;
    const float bias1 = b1p[hl < H1 ? hl : H1 - 1], bias2 = b2p[hl < H2 ? hl : H2 - 1];
    const int act1_v = gcm_vgpr(tl.act1), act2_v = gcm_vgpr(tl.act2);
    // layer 1: the half-waves split k (lanes 0-31: k < FP / 2, lanes 32-63: the rest), met by one cross-half add
    float p1;
    float w2[32];   // layer 2's weight column of this lane: read with layer 1's where that has room (no wt[] below)
    {
      float pa = 0.f, pb = 0.f;
      float wr[KH];
      if (lds_gather) {   // (uniform) the root term from wave 1
        pb = sRoot[lane];
#pragma unroll
        for (int k = 0; k < KH; ++k) wr[k] = sW[(kh * KH + k) * 32 + hl];
#pragma unroll
        for (int k = 0; k < 32; ++k) w2[k] = sW[(2 * FP + kh * 32 + k) * 32 + hl];
#pragma unroll
        for (int k4 = 0; k4 < KH / 4; ++k4) {
          const float4 a = *reinterpret_cast<const float4*>(sv + kh * KH + 4 * k4);
          pa = fmaf(wr[4 * k4], a.x, pa);
          pa = fmaf(wr[4 * k4 + 1], a.y, pa);
          pa = fmaf(wr[4 * k4 + 2], a.z, pa);
          pa = fmaf(wr[4 * k4 + 3], a.w, pa);
        }
      } else {
        float wt[KH];
#pragma unroll
        for (int k = 0; k < KH; ++k) {
          wr[k] = sW[(kh * KH + k) * 32 + hl];
          wt[k] = sW[(FP + kh * KH + k) * 32 + hl];
        }
#pragma unroll
        for (int k4 = 0; k4 < KH / 4; ++k4) {
          const float4 a = *reinterpret_cast<const float4*>(sv + kh * KH + 4 * k4);
          const float4 x = *reinterpret_cast<const float4*>(sv + FP + kh * KH + 4 * k4);
          pa = fmaf(wr[4 * k4], a.x, pa); pb = fmaf(wt[4 * k4], x.x, pb);
          pa = fmaf(wr[4 * k4 + 1], a.y, pa); pb = fmaf(wt[4 * k4 + 1], x.y, pb);
          pa = fmaf(wr[4 * k4 + 2], a.z, pa); pb = fmaf(wt[4 * k4 + 2], x.z, pb);
          pa = fmaf(wr[4 * k4 + 3], a.w, pa); pb = fmaf(wt[4 * k4 + 3], x.w, pb);
        }
      }
      p1 = pa + pb;
      p1 += __shfl_xor(p1, 32);
      p1 += bias1;
    }
    const float h1c = hl < H1 ? gcm_act_sel(p1, act1_v) : 0.f;   // (both halves hold h1c[lane & 31])
    DSTAMP(9);
    // (one wave: its LDS operations execute in order - the reads above are done before these writes land)
    if (lane < 32) { sv[lane] = agg2; sv[32 + lane] = h1c; }
    asm volatile("" ::: "memory");   // (lanes exchange through LDS: keep the loads below behind these stores)
    // layer 2: lanes 0-31 W_rel2 . agg2, lanes 32-63 W_root2 . h1cur
    float p2;
    {
      if (!lds_gather) {
#pragma unroll
        for (int k = 0; k < 32; ++k) w2[k] = sW[(2 * FP + kh * 32 + k) * 32 + hl];
      }
      float pa = 0.f;
#pragma unroll
      for (int k4 = 0; k4 < 8; ++k4) {
        const float4 a = *reinterpret_cast<const float4*>(sv + kh * 32 + 4 * k4);
        pa = fmaf(w2[4 * k4], a.x, pa);
        pa = fmaf(w2[4 * k4 + 1], a.y, pa);
        pa = fmaf(w2[4 * k4 + 2], a.z, pa);
        pa = fmaf(w2[4 * k4 + 3], a.w, pa);
      }
      p2 = pa + __shfl_xor(pa, 32) + bias2;
    }
    const float v = gcm_act_sel(p2, act2_v);
    DSTAMP(10);
    const unsigned rc = gb * (unsigned)N + (unsigned)cur;
    if (!bad) {   // (the rest of row cur and the record's lists: wave 1, above)
      if (lane < F) tl.cA[rc * F + lane] = agg1;
      if (lane < H1) tl.cH[rc * H1 + lane] = h1c;
    }
    if (lane < H2) tl.saved[gb * H2 + lane] = v;
    if (tl.total && lane < H1) {
      tl.saved[tl.o_v + gb * 2 * H1 + lane] = agg2;
      tl.saved[tl.o_v + gb * 2 * H1 + H1 + lane] = h1c;
    }
    const bool nonfinite = __any(lane < H2 && !isfinite(v));
    if ((nonfinite || bad) && lane == 0)
      atomicOr(tl.flags, (nonfinite ? GCM_FLAG_NONFINITE : 0u) | (bad ? GCM_FLAG_BAD_COUNT : 0u));
    DSTAMP(11);
  }
}

// per-graph modes: one thread per (b, j)
__global__ void k_pergraph(View vw, const float* __restrict__ dist_param, float* __restrict__ adj,
                           float* __restrict__ sel_row, float* __restrict__ dist_out, int mode,
                           float max_distance, int a0, int a1, int b0, int bidirectional, int B, int N,
                           int F) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * N) return;
  const int b = i / N, j = i - b * N;
  int sh;
  const int cur = view_cur(vw, b, N, sh);
  if (j >= cur && dist_out == nullptr) return;
  const float* c = view_cur_row(vw, b, cur, N, F);
  const float* n = vw.nodes + ((size_t)b * N + (j + sh < N ? j + sh : N - 1)) * F;
  const float sc = dist_param ? dist_param[0] : 1.f;
  float d;
  if (mode == GCM_DIST_L2_PERGRAPH) {
    float s = 0.f;
    for (int f = 0; f < a1 - a0; ++f) {
      const float t = c[a0 + f] / sc - n[b0 + f] / sc;
      s = fmaf(t, t, s);
    }
    d = sqrtf(s);
  } else {  // cosine similarity, torch semantics: normalise each side by max(norm, eps) first
    float cc = 0.f, nn = 0.f;
    for (int f = 0; f < F; ++f) {
      const float cv = c[f] / sc, nv = n[f] / sc;
      cc = fmaf(cv, cv, cc);
      nn = fmaf(nv, nv, nn);
    }
    const float ic = 1.f / fmaxf(sqrtf(cc), 1e-8f), in = 1.f / fmaxf(sqrtf(nn), 1e-8f);
    float s = 0.f;
    for (int f = 0; f < F; ++f) s = fmaf((c[f] / sc) * ic, (n[f] / sc) * in, s);
    d = s;
  }
  if (dist_out) dist_out[i] = d;
  if (j < cur) view_emit(adj, sel_row, b, cur, j, N, d < max_distance, bidirectional);
}

}  // namespace

extern "C" size_t gcm_edge_distance_workspace_bytes(int mode, int B, int N, int F) {
  (void)N;   // B: the number of current rows (the local batch, or n_cur_rows of a sharded selector)
  if (mode != GCM_DIST_EUCLID_CROSSBATCH || B <= 0 || F <= 0) return 0;
  return ((size_t)B * F + B) * sizeof(float);
}

static int run_distance(const View& vw, float* adj, float* sel_row, int mode, float max_distance,
                        const float* dist_param, int a0, int a1, int b0, int b1, int bidirectional,
                        float* dist_out, void* workspace, size_t workspace_bytes, int B, int N, int F,
                        gcm_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (vw.cur_rows && mode != GCM_DIST_EUCLID_CROSSBATCH) return GCM_EINVAL;   // only the cross-batch mean couples graphs
  if (mode == GCM_DIST_EUCLID_CROSSBATCH) {
    if (F > 128 || B > 65535) return GCM_EUNSUPPORTED;
    const int Bc = vw.cur_rows ? vw.n_cur : B;   // rows the mean runs over
    if (Bc <= 0) return GCM_EINVAL;
    if (Bc >= 32) {   // matrix-core path (torch.cdist's own switch to the mm formulation is at 25)
      // 128 node rows x 256-graph chunks per workgroup (one workgroup per CU at cfg3: B = 256 graphs of
      // 128 nodes).  Measured against 64 rows x 128-graph chunks (two to three workgroups per CU, one's
      // staging under another's MFMA phase): 21.3 us vs 23.7 us - the extra staging traffic and the
      // single MFMA chain per wave cost more than the overlap returns.
      constexpr int RB = 128;
      dim3 grid((N + RB - 1) / RB, B);
      const int FT = (F + 31) / 32;
      const int CBv = FT >= 4 ? 128 : 256;   // (fits 160 KB of LDS)
      const size_t lds = sizeof(float) * ((size_t)RB * (32 * FT + 1) + (size_t)32 * FT * (CBv + 1) + RB + CBv +
                                          (size_t)(8 / (RB / 32) - 1) * RB);
#define GCM_EUCLID_MFMA(FTv)                                                                     \
  {                                                                                              \
    auto kern = k_euclid_mfma<FTv, RB, (FTv >= 4 ? 128 : 256)>;                                  \
    gcm_allow_dynamic_lds((const void*)kern, lds);                                               \
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, vw, dist_param, adj, sel_row, dist_out,    \
                       max_distance, bidirectional, Bc, N, F);                                   \
  }
      // F <= 64: tiles dealt by liveness, double-buffered chunks of 128 graphs (k_euclid_mfma2)
      const size_t lds2 = sizeof(float) * ((size_t)RB * (32 * FT + 1) + (size_t)2 * 32 * FT * 129 + RB + 2 * 128 +
                                           (size_t)8 * RB);
#define GCM_EUCLID_MFMA2(FTv)                                                                    \
  {                                                                                              \
    auto kern = k_euclid_mfma2<FTv, 0>;                                                          \
    gcm_allow_dynamic_lds((const void*)kern, lds2);                                              \
    hipLaunchKernelGGL(kern, grid, dim3(1024), lds2, s, vw, dist_param, adj, sel_row, dist_out,  \
                       max_distance, bidirectional, Bc, N, F, StepTail{});                       \
  }
      switch (FT) {
        case 1: GCM_EUCLID_MFMA2(1) break;
        case 2: GCM_EUCLID_MFMA2(2) break;
        case 3: GCM_EUCLID_MFMA(3) break;
        default: GCM_EUCLID_MFMA(4) break;
      }
#undef GCM_EUCLID_MFMA2
#undef GCM_EUCLID_MFMA
      return gcm_launch_status();
    }
    GCM_REQUIRE(workspace);
    if (workspace_bytes < ((size_t)Bc * F + Bc) * sizeof(float)) return GCM_EWORKSPACE;
    float* ws_cur = (float*)workspace;
    hipLaunchKernelGGL(k_gather_cur, dim3((Bc * F + 255) / 256), dim3(256), 0, s, vw, dist_param, ws_cur,
                       Bc, N, F);
    dim3 grid((N + 127) / 128, B);
#define GCM_EUCLID(FP)                                                                        \
  hipLaunchKernelGGL(k_euclid_crossbatch<FP>, grid, dim3(128), 0, s, vw, ws_cur, dist_param, adj, \
                     sel_row, dist_out, max_distance, bidirectional, Bc, N, F)
    if (F <= 16) GCM_EUCLID(16);
    else if (F <= 32) GCM_EUCLID(32);
    else if (F <= 64) GCM_EUCLID(64);
    else GCM_EUCLID(128);
#undef GCM_EUCLID
    return gcm_launch_status();
  }
  if (mode == GCM_DIST_L2_PERGRAPH) {
    GCM_REQUIRE(a0 >= 0 && a1 > a0 && a1 <= F && b0 >= 0 && b1 <= F && (b1 - b0) == (a1 - a0));
  } else if (mode != GCM_DIST_COSINE_SIM) {
    return GCM_EINVAL;
  }
  const int total = B * N;
  hipLaunchKernelGGL(k_pergraph, dim3((total + 255) / 256), dim3(256), 0, s, vw, dist_param, adj, sel_row,
                     dist_out, mode, max_distance, a0, a1, b0, bidirectional, B, N, F);
  return gcm_launch_status();
}

/* EuclideanEdge (cross-batch mean, distance.py:41-49) on the state as it comes in + the cached live-row step on row
 * cur (gcm_dense_rows_step_cached_ws's arithmetic) as ONE launch: k_euclid_mfma2 with the step as wave 0's tail.
 * For: Bc >= 32 current rows, F in {32, 64}, N <= 128, H1, H2 <= 32, no other selector.  GCM_EUNSUPPORTED otherwise
 * (the caller then runs the two launches).  lay5: gcm_dense_rows_cached_layout. */
extern "C" int gcm_edge_distance_step_cached_supported(int n_cur_rows, int B, int N, int F, int H1, int H2) {
  return !(n_cur_rows < 32 || (F != 32 && F != 64) || N > 128 || N < 1 || H1 > 32 || H1 < 1 || H2 > 32 || H2 < 1 ||
           B > 65535 || B < 1);
}

extern "C" int gcm_edge_distance_step_cached(const float* obs, float* nodes, float* adj, int64_t* count,
                                             float max_distance, const float* dist_param, const float* cur_rows,
                                             int n_cur_rows, const float* params, const float* weight_image, int act1,
                                             int act2, float* cache_h1, float* cache_agg1, float* cache_nodes,
                                             float* saved, const size_t* lay5, int record, int cur_host,
                                             uint32_t* adj_bits, uint32_t* flags, int B, int N, int F, int H1, int H2,
                                             gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes && adj && count && params && weight_image && cache_h1 && cache_agg1 && cache_nodes &&
              saved && lay5 && flags && B > 0);
  const int Bc = cur_rows ? n_cur_rows : B;
  if (!gcm_edge_distance_step_cached_supported(Bc, B, N, F, H1, H2)) return GCM_EUNSUPPORTED;
  View vw{nodes, nullptr, count, obs, cur_rows, n_cur_rows};
  StepTail tl{params, weight_image, nodes, adj, count, cache_h1, cache_agg1, cache_nodes, saved,
              lay5[1], lay5[2], lay5[3], lay5[4], record ? lay5[0] : 0, flags, act1, act2, H1, H2,
              cur_host >= 0 ? cur_host : -1, adj_bits, 0, 0};
  constexpr int RB = 128;
  const int FT = F / 32;
  const size_t lds = sizeof(float) * ((size_t)RB * (32 * FT + 1) + (size_t)2 * 32 * FT * 129 + RB + 2 * 128 +
                                      (size_t)8 * RB + (size_t)(2 * 32 * FT + 64) * 32 + RB + (size_t)128 * 32);
  hipStream_t s = (hipStream_t)stream;
  if (FT == 1) {
    auto kern = k_euclid_mfma2<1, 1>;
    gcm_allow_dynamic_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, dim3(1, B), dim3(1024), lds, s, vw, dist_param, adj, (float*)nullptr, (float*)nullptr,
                       max_distance, 0, Bc, N, F, tl);
  } else {
    auto kern = k_euclid_mfma2<2, 1>;
    gcm_allow_dynamic_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, dim3(1, B), dim3(1024), lds, s, vw, dist_param, adj, (float*)nullptr, (float*)nullptr,
                       max_distance, 0, Bc, N, F, tl);
  }
  return gcm_launch_status();
}

/* The same chain in the STEADY STATE (every graph holds N nodes: count[b] == N, which the caller guarantees - a chain
 * from empty graphs that has made >= N steps): the step drops each graph's oldest node (gcm.py:263-271, 323-355), so the
 * layer-1 rows of older nodes are no longer final and the live rows are re-evaluated - ONE launch still: distances on
 * the matrix cores, then the roll of the donated state IN PLACE (nodes, adj; count stays N), the live rows' layer 1
 * from the bit image of the adjacency the cached steps kept (adj_bits [B][N][4], rolled here too), row cur, the belief.
 * saved: the GENERAL live-row record (gcm_dense_rows_layout: lay6 = {total, v, hdr, coef, rows, rw}; mx at 0; in full
 * with record != 0), read by gcm_dense_rows_bptt.  No `learned` divisor, local batch only. */
extern "C" int gcm_edge_distance_step_ring_supported(int B, int N, int F, int H1, int H2) {
  return gcm_edge_distance_step_cached_supported(B, B, N, F, H1, H2) && N >= 8 && (N & 3) == 0;
}

extern "C" int gcm_edge_distance_step_ring(const float* obs, float* nodes, float* adj, int64_t* count,
                                           float max_distance, const float* params, const float* weight_image,
                                           int act1, int act2, uint32_t* adj_bits, float* saved, const size_t* lay6,
                                           int record, uint32_t* flags, int B, int N, int F, int H1, int H2,
                                           gcm_stream_t stream) {
  GCM_REQUIRE(obs && nodes && adj && count && params && weight_image && adj_bits && saved && lay6 && flags && B > 0);
  if (!gcm_edge_distance_step_ring_supported(B, N, F, H1, H2)) return GCM_EUNSUPPORTED;
  View vw{nodes, nullptr, count, obs, nullptr, 0};
  StepTail tl{params, weight_image, nodes, adj, count, nullptr, nullptr, nullptr, saved,
              lay6[1], lay6[2], lay6[3], 0, record ? lay6[0] : 0, flags, act1, act2, H1, H2, -1, adj_bits, lay6[4],
              (int)lay6[5]};
  constexpr int RB = 128;
  const int FT = F / 32;
  const size_t lds = sizeof(float) * ((size_t)RB * (32 * FT + 1) + (size_t)2 * 32 * FT * 129 + RB + 2 * 128 +
                                      (size_t)8 * RB + (size_t)(2 * 32 * FT + 64) * 32 + RB + (size_t)128 * 32 + 4 * RB + 64);
  hipStream_t s = (hipStream_t)stream;
  if (FT == 1) {
    auto kern = k_euclid_mfma2<1, 2>;
    gcm_allow_dynamic_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, dim3(1, B), dim3(1024), lds, s, vw, (const float*)nullptr, adj, (float*)nullptr,
                       (float*)nullptr, max_distance, 0, B, N, F, tl);
  } else {
    auto kern = k_euclid_mfma2<2, 2>;
    gcm_allow_dynamic_lds((const void*)kern, lds);
    hipLaunchKernelGGL(kern, dim3(1, B), dim3(1024), lds, s, vw, (const float*)nullptr, adj, (float*)nullptr,
                       (float*)nullptr, max_distance, 0, B, N, F, tl);
  }
  return gcm_launch_status();
}

extern "C" int gcm_edge_distance(const float* nodes, float* adj, const int64_t* cur_idx, int mode,
                                 float max_distance, const float* dist_param, int a0, int a1,
                                 int b0, int b1, int bidirectional, float* dist_out,
                                 void* workspace, size_t workspace_bytes, int B, int N, int F,
                                 gcm_stream_t stream) {
  return gcm_edge_distance_ex(nodes, adj, cur_idx, mode, max_distance, dist_param, a0, a1, b0, b1,
                              bidirectional, dist_out, nullptr, 0, workspace, workspace_bytes, B, N, F, stream);
}

extern "C" int gcm_edge_distance_ex(const float* nodes, float* adj, const int64_t* cur_idx, int mode,
                                    float max_distance, const float* dist_param, int a0, int a1,
                                    int b0, int b1, int bidirectional, float* dist_out,
                                    const float* cur_rows, int n_cur_rows, void* workspace,
                                    size_t workspace_bytes, int B, int N, int F, gcm_stream_t stream) {
  GCM_REQUIRE(nodes && adj && cur_idx && B > 0 && N > 0 && F > 0 && (!cur_rows || n_cur_rows > 0));
  const View vw{nodes, cur_idx, nullptr, nullptr, cur_rows, n_cur_rows};
  return run_distance(vw, adj, nullptr, mode, max_distance, dist_param, a0, a1, b0, b1, bidirectional,
                      dist_out, workspace, workspace_bytes, B, N, F, stream);
}

/* The same selectors on the state BEFORE the step (nodes_in, count_in = num_nodes going in, obs =
 * the nodes about to be inserted): sel_row [B, N] receives 1 / 0 for every image row j < cur_b (the
 * row the node lands in, after the overflow roll); entries j >= cur_b are left untouched.  This is
 * what gcm_dense_rows_step_fwd merges into row cur of the adjacency. */
extern "C" int gcm_edge_distance_pre(const float* nodes_in, const int64_t* count_in, const float* obs,
                                     float* sel_row, int mode, float max_distance,
                                     const float* dist_param, int a0, int a1, int b0, int b1,
                                     void* workspace, size_t workspace_bytes, int B, int N, int F,
                                     gcm_stream_t stream) {
  return gcm_edge_distance_pre_ex(nodes_in, count_in, obs, sel_row, mode, max_distance, dist_param, a0, a1, b0,
                                  b1, nullptr, 0, workspace, workspace_bytes, B, N, F, stream);
}

extern "C" int gcm_edge_distance_pre_ex(const float* nodes_in, const int64_t* count_in, const float* obs,
                                        float* sel_row, int mode, float max_distance,
                                        const float* dist_param, int a0, int a1, int b0, int b1,
                                        const float* cur_rows, int n_cur_rows, void* workspace,
                                        size_t workspace_bytes, int B, int N, int F, gcm_stream_t stream) {
  GCM_REQUIRE(nodes_in && count_in && obs && sel_row && B > 0 && N > 0 && F > 0 && (!cur_rows || n_cur_rows > 0));
  const View vw{nodes_in, nullptr, count_in, obs, cur_rows, n_cur_rows};
  return run_distance(vw, nullptr, sel_row, mode, max_distance, dist_param, a0, a1, b0, b1, 0, nullptr,
                      workspace, workspace_bytes, B, N, F, stream);
}
