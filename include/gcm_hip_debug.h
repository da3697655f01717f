/* gcm_hip_debug.h - measurement aids of bench.py and tools/, NOT part of the product ABI.
 *
 * These entry points exist only in libgcm_hip_debug.so (`make -C graph-conv-memory_amd/csrc debug`: the live-row step
 * sources compiled with -DGCM_DEBUG_ABI next to the objects they need), which nothing in the gcm package loads; the
 * product library libgcm_hip.so (include/gcm_hip.h) exports none of them and keeps no process-wide state
 * (SURVEY 8b).  They launch the very kernels of the product library - same sources, same flags - with the HIP runtime's
 * dispatch-recorded events around each launch (hipExtLaunchKernelGGL start / stop events = the kernel begin / end
 * timestamps rocprofv3 --kernel-trace reports), which the product entry points have no argument for. */
#ifndef GCM_HIP_DEBUG_H
#define GCM_HIP_DEBUG_H
#include "gcm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement aid (bench.py): the NEXT gcm_dense_rows_step_fwd launch of the calling thread is
 * bracketed by the two hipEvent_t given here, recorded by the dispatch itself
 * (hipExtLaunchKernelGGL start / stop events: the kernel's own begin / end timestamps, what
 * rocprofv3 --kernel-trace reports) instead of by marker packets around it.  One-shot. */
int gcm_debug_time_next_launch(void* start_event, void* stop_event);

/* Measurement aid (bench.py): T steps of gcm_dense_rows_step_fwd on the evolving donated state,
 * enqueued back to back from C (the launch cadence of a replayed HIP graph), launch t bracketed by
 * start_events[t] / stop_events[t] (hipEvent_t, recorded by the dispatch itself).  obs_all [T,B,F];
 * saved_per_step: host array of T record pointers (gcm_dense_rows_layout). */
int gcm_debug_time_rows_rollout(const float* obs_all, float* nodes, float* adj, int64_t* count,
                                const gcm_selector_desc* selectors, int n_selectors, const float* params,
                                int has_bias, int act1, int act2, float* const* saved_per_step,
                                uint32_t* flags, void* const* start_events, void* const* stop_events, int T,
                                int B, int N, int F, int H1, int H2, gcm_stream_t stream);

/* measurement aid (bench.py): T <= N cached steps of a rollout from empty graphs enqueued back to back from C, each
 * launch bracketed by the caller's HIP events recorded by the dispatch itself (cf. gcm_debug_time_rows_rollout) */
int gcm_debug_time_cached_rollout(const float* obs_all, float* nodes, float* adj, int64_t* count,
                                  const gcm_selector_desc* selectors, int n_selectors, const float* params,
                                  const float* weight_image, int has_bias, int act1, int act2, float* cache_h1,
                                  float* cache_agg1, float* cache_nodes, float* const* saved_per_step,
                                  uint32_t* flags, void* const* start_events, void* const* stop_events, int T, int B,
                                  int N, int F, int H1, int H2, gcm_stream_t stream);

/* The launch floor of this box: a HIP graph of `nodes` EMPTY kernels (grid x block threads each, one after the other
 * on one stream, as the captured per-step loop is), instantiated once and replayed `replays` times on a stream of its
 * own; -> microseconds per node (HIP events around the replays).  What a chain of dependent launches costs when the
 * kernels do nothing: the figure to read a 4-5 us step kernel against. */
int gcm_debug_empty_graph_cadence(int nodes, int grid, int block, int replays, float* us_per_node);
/* the same `nodes` empty kernels enqueued back to back without a graph, each bracketed by dispatch-recorded events:
 * -> mean kernel begin -> end duration in microseconds (the duration floor of one dispatch) and, in cadence_us, the
 * mean distance between consecutive kernels' begin timestamps */
int gcm_debug_empty_launch_duration(int nodes, int grid, int block, float* duration_us, float* cadence_us);

#ifdef __cplusplus
}
#endif
#endif
