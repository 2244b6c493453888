// Device-side view of the per-point inputs of a sequence (local scans + poses + model) shared by the
// point, consistency and ICP kernels.
#pragma once
#include "dc_common.h"
#include "dc_device.h"
#include "dc_pointmath.h"

namespace dc {

struct PointInputs {
  const void* vps;        // [N,3]
  const void* dirs;       // [N,3]
  const void* depth;      // [N]
  const void* inc;        // [N]   (may be null when model kind is NONE)
  const uint8_t* lmask;   // [N]   (null = all true)
  const int32_t* scan_id; // [N]   (null = scan 0)
  const double* poses;    // [S,12] device, row-major [R|t] (null = identity)
  const double* w;        // [P] device
  const double* e;        // [P] device
  int model_kind, n_terms, n_scans;
};

__device__ __forceinline__ void load_model(const PointInputs& in, ModelParams& mp) {
  mp.kind = in.model_kind;
  mp.n_terms = in.n_terms;
#pragma unroll
  for (int k = 0; k < DC_MAX_MODEL_TERMS; ++k) {
    const bool on = k < in.n_terms && in.model_kind != DC_MODEL_NONE;
    mp.w[k] = on ? in.w[k] : 0.0;
    mp.e[k] = on ? in.e[k] : 0.0;
  }
}

__device__ __forceinline__ void load_pose(const PointInputs& in, int s, double* T) {
  if (in.poses) {
    const double* p = in.poses + (int64_t)s * 12;
#pragma unroll
    for (int q = 0; q < 12; ++q) T[q] = p[q];
  } else {
#pragma unroll
    for (int q = 0; q < 12; ++q) T[q] = (q == 0 || q == 5 || q == 10) ? 1.0 : 0.0;
  }
}

}  // namespace dc
