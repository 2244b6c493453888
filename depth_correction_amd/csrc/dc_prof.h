// In-library kernel timer shared by the translation units that launch the hot kernels (dc_consistency.hip: kinds 0-2,
// dc_features.hip: kind 3); control functions dc_profiler_* in dc_consistency.hip.
#pragma once
#include <mutex>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

// ---- in-library kernel timer: HIP events recorded on the launch stream right around the main kernels ------
namespace dc {
constexpr int kProfKinds = 4;            // 0 points_fwd, 1 consistency_fwd, 2 consistency_bwd, 3 features_fwd
constexpr int kProfCap = 4096;
// Process-wide state behind a mutex: launches from different host threads (distinct streams) may time concurrently;
// the events of one ProfScope are only touched by the thread that owns it until its destructor publishes them.
struct ProfState {
  std::mutex mu;
  int every = 0;                           // 0 = off, N = time every N-th launch of each kind
  int64_t seen[kProfKinds] = {};
  int count[kProfKinds] = {};
  hipEvent_t start[kProfKinds][kProfCap];
  hipEvent_t stop[kProfKinds][kProfCap];
  int created[kProfKinds] = {};
  const char* last_kernel[kProfKinds] = {};     // instantiation launched last, per kind
};
inline ProfState g_prof;           // one instance for the library (C++17 inline variable): dc_consistency.hip and dc_features.hip share it

// One timed launch: the kernel is launched through hipExtLaunchKernelGGL, which stamps the two events with the
// dispatch's own start / end times (what rocprofv3 reports) instead of bracketing it with event packets.
struct ProfScope {
  int kind, slot;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  explicit ProfScope(int kind_) : kind(kind_), slot(-1) {
    std::lock_guard<std::mutex> lock(g_prof.mu);
    if (g_prof.every <= 0 || g_prof.count[kind] >= kProfCap) return;
    if ((g_prof.seen[kind]++ % g_prof.every) != 0) return;
    slot = g_prof.count[kind];
    if (slot >= g_prof.created[kind]) {
      if (hipEventCreate(&g_prof.start[kind][slot]) != hipSuccess || hipEventCreate(&g_prof.stop[kind][slot]) != hipSuccess) { slot = -1; return; }
      g_prof.created[kind] = slot + 1;
    }
    g_prof.count[kind] = slot + 1;         // reserved now, so that a concurrent scope takes the next slot
    ev0 = g_prof.start[kind][slot];
    ev1 = g_prof.stop[kind][slot];
  }
  hipEvent_t start() const { return ev0; }
  hipEvent_t stop() const { return ev1; }
  void name(const char* kernel) const { g_prof.last_kernel[kind] = kernel; }
};
// launch of a hot kernel inside a `ProfScope prof` block
#define DC_TIMED_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  (prof.name(#kernel), hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, prof.start(), prof.stop(), 0, __VA_ARGS__))
}  // namespace dc

