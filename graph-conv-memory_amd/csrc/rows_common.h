// The per-step record the live-row forward (rows_step.hip) saves for the time-parallel backward
// (rows_bptt.hip).  One allocation per step, float offsets, 64-float aligned sections:
//
//   mx   [B, H2]          belief states (also the step's output tensor)
//   v    [B, 2*H1]        agg2 | h1[cur]               (the input of layer 2)
//   hdr  [B, 4] int32     L = number of live rows, l_cur = position of row cur in the list,
//                         cur, wrapped
//   coef [B, N]           adj[cur, j_l] for l < L      (compact, ascending j)
//   rows [B, N, rw]       per live row l < L: h1[j_l] [H1] | agg1[j_l] [F] | x[j_l] [F]; rw = H1 + 2F
//   deg  [B, N]           row sums of the live adjacency rows, l < L (written only when a Linear
//                         preprocessor's bias is folded into the step: GCM_GNN_HAS_DEG_TERM)
//   live [B, N] int32     (GCM_GNN_RECORD_DX only) the row j_l of live slot l
//   arows [B, N, N]       (GCM_GNN_RECORD_DX only) the adjacency row of live slot l as layer 1 aggregated it
//
// Capacity is N rows per graph (DenseEdge makes every row <= cur live); only L rows are touched.
#pragma once
#include <stddef.h>

#define GCM_ROWS_MAX_STEPS 128   /* steps per launch: two pointer tables in the kernel arguments (2 KB of the 4 KB) */

namespace gcm_rows {

// device pointers of up to 64 recorded steps, by value in the kernel arguments
struct StepTable {
  const float* saved[GCM_ROWS_MAX_STEPS];
  const float* gmx[GCM_ROWS_MAX_STEPS];
};

// Pass A of the learned-step backward (learned_step.hip: gcm_learned_bptt), defined in rows_bptt.hip:
// k_bptt_rows over the full-layer buffers of up to GCM_ROWS_MAX_STEPS steps.  Offsets in floats.
struct LearnedSrc {
  size_t o_adj, o_mx, o_h1, o_agg1, o_agg2, o_idx;
  const float* w_rel1;
  int* hdr;
  int* live;
  float* da;
  float* dagg2;
  int s0;
  int adj_compact;   // the step buffers hold row cur of the adjacency only ([B, N] at o_adj)
  // cached steps (learned_step.hip: gcm_learned_step_cached): node matrix / h1 / agg1 in the chain's caches
  const float *c_nodes, *c_h1, *c_agg1;
};
int launch_bptt_learned(void* stream, int grid, const StepTable& tab, int n_steps, long gmx_sb, long gmx_sh,
                        const float* w_rel2, const float* w_root2, int act1, int act2, float* slabs,
                        const LearnedSrc& src, int B, int N, int F, int H1, int H2);
// ... per graph (rows_bptt.hip: k_bptt_learned_graph): every step a cached step, F = H1 = 32, H2 <= 32, N = 128, T <= 128;
// B slabs; src.da receives G1 rows (not W_rel1^T G1)
int launch_bptt_learned_graph(void* stream, const StepTable& tab, int n_steps, long gmx_sb, long gmx_sh, const float* w_rel2,
                              const float* w_root2, int act1, int act2, float* slabs, const LearnedSrc& src, int B, int H2);

struct SavedLayout {
  size_t total, o_v, o_hdr, o_coef, o_rows, o_deg;
  int rw;
  // records that also serve the gradient w.r.t. the observations / nodes (GCM_GNN_RECORD_DX): the row
  // index of every live row, and its adjacency row as layer 1 aggregated it (0: absent)
  size_t o_live;    // [B, N] int32   j_l
  size_t o_arows;   // [B, N, N]      adj'[j_l, :]  (capacity N rows per graph; L written)
};

static inline size_t pad64(size_t n) { return (n + 63) & ~(size_t)63; }

static inline SavedLayout make_layout(int B, int N, int F, int H1, int H2, bool dx = false) {
  SavedLayout l;
  l.rw = H1 + 2 * F;
  l.o_v = pad64((size_t)B * H2);
  l.o_hdr = l.o_v + pad64((size_t)B * 2 * H1);
  l.o_coef = l.o_hdr + pad64((size_t)B * 4);
  l.o_rows = l.o_coef + pad64((size_t)B * N);
  l.o_deg = l.o_rows + pad64((size_t)B * N * l.rw);
  l.total = l.o_deg + pad64((size_t)B * N);
  l.o_live = l.o_arows = 0;
  if (dx) {
    l.o_live = l.total;
    l.o_arows = l.o_live + pad64((size_t)B * N);
    l.total = l.o_arows + pad64((size_t)B * N * N);
  }
  return l;
}

// record of a cached step (rows_cached.hip): mx | v | hdr | coef | live - the first four at the offsets of
// SavedLayout (k_bptt_rows reads them alike), no rows section: the rows are in the chain's caches
struct CachedLayout {
  size_t total, o_v, o_hdr, o_coef, o_live;
};
static inline CachedLayout make_cached_layout(int B, int N, int H1, int H2) {
  CachedLayout l;
  l.o_v = pad64((size_t)B * H2);
  l.o_hdr = l.o_v + pad64((size_t)B * 2 * H1);
  l.o_coef = l.o_hdr + pad64((size_t)B * 4);
  l.o_live = l.o_coef + pad64((size_t)B * N);
  l.total = l.o_live + pad64((size_t)B * N);
  return l;
}

}  // namespace gcm_rows
