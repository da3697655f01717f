// libgcm_hip_debug.so only (include/gcm_hip_debug.h): what a chain of dependent launches costs on this box when the
// kernels do nothing - the floor bench.py prints beside the step kernel's duration.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <vector>

#include "gcm_hip_debug.h"

namespace {
__global__ void k_empty() {}

struct Guard {   // releases whatever was created, in reverse order, on every return path
  hipStream_t s = nullptr;
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  std::vector<hipEvent_t> evs;
  ~Guard() {
    for (hipEvent_t e : evs) (void)hipEventDestroy(e);
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (ge) (void)hipGraphExecDestroy(ge);
    if (g) (void)hipGraphDestroy(g);
    if (s) (void)hipStreamDestroy(s);
  }
};
#define HIP_TRY(x)                    \
  do {                                \
    const hipError_t e_ = (x);        \
    if (e_ != hipSuccess) return (int)e_; \
  } while (0)
}  // namespace

extern "C" int gcm_debug_empty_graph_cadence(int nodes, int grid, int block, int replays, float* us_per_node) {
  if (nodes <= 0 || grid <= 0 || block <= 0 || block > 1024 || replays <= 0 || !us_per_node) return GCM_EINVAL;
  Guard G;
  HIP_TRY(hipStreamCreate(&G.s));
  HIP_TRY(hipStreamBeginCapture(G.s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), 0, G.s);
  HIP_TRY(hipStreamEndCapture(G.s, &G.g));
  HIP_TRY(hipGraphInstantiate(&G.ge, G.g, nullptr, nullptr, 0));
  HIP_TRY(hipEventCreate(&G.a));
  HIP_TRY(hipEventCreate(&G.b));
  for (int i = 0; i < 3; ++i) HIP_TRY(hipGraphLaunch(G.ge, G.s));
  HIP_TRY(hipStreamSynchronize(G.s));
  HIP_TRY(hipEventRecord(G.a, G.s));
  for (int i = 0; i < replays; ++i) HIP_TRY(hipGraphLaunch(G.ge, G.s));
  HIP_TRY(hipEventRecord(G.b, G.s));
  HIP_TRY(hipStreamSynchronize(G.s));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, G.a, G.b));
  *us_per_node = ms * 1e3f / ((float)replays * (float)nodes);
  return GCM_OK;
}

extern "C" int gcm_debug_empty_launch_duration(int nodes, int grid, int block, float* duration_us, float* cadence_us) {
  if (nodes <= 1 || grid <= 0 || block <= 0 || block > 1024 || !duration_us || !cadence_us) return GCM_EINVAL;
  Guard G;
  HIP_TRY(hipStreamCreate(&G.s));
  G.evs.resize(2 * (size_t)nodes, nullptr);
  for (auto& e : G.evs) HIP_TRY(hipEventCreate(&e));
  for (int rep = 0; rep < 2; ++rep) {   // (the first pass warms the code object and the clocks)
    for (int i = 0; i < nodes; ++i)
      hipExtLaunchKernelGGL(k_empty, dim3(grid), dim3(block), 0, G.s, G.evs[2 * i], G.evs[2 * i + 1], 0);
    HIP_TRY(hipStreamSynchronize(G.s));
  }
  double dur = 0.0;
  for (int i = 0; i < nodes; ++i) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, G.evs[2 * i], G.evs[2 * i + 1]));
    dur += ms;
  }
  float span = 0.f;
  HIP_TRY(hipEventElapsedTime(&span, G.evs[0], G.evs[2 * (nodes - 1)]));
  *duration_us = (float)(dur * 1e3 / nodes);
  *cadence_us = span * 1e3f / (float)(nodes - 1);
  return GCM_OK;
}
