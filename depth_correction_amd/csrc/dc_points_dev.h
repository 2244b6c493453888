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
  // [blocks, 2 n_scans + 1] or null: the points of every 256-point block are grouped -- those inside the loss mask first, by scan
  // id, then those outside it, by scan id (the plan ordered them so) -- and segment v of block b (v < S: inside, scan v; v >= S:
  // outside, scan v - S) is its lanes seg_start[b (2 S + 1) + v] .. seg_start[b (2 S + 1) + v + 1]  (reduce_pose_grads_grouped)
  const uint16_t* seg_start = nullptr;
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

// Poses of a sequence are read per LANE (a block of Morton-ordered points holds points of most scans).  Every block
// copies the (few) poses into LDS once, coalesced, and the lanes read them from there instead of gathering 12 doubles
// each from global memory (6 x 64 L1 tag lookups per wavefront).
constexpr int kLdsScans = 32;              // 3 KB; sequences with more scans read their poses from global memory

struct PoseTile {
  const double* lds;                       // staged poses, or nullptr: read from global memory
};

// Call from all threads of the block; the caller's next __syncthreads() publishes the tile.
__device__ __forceinline__ PoseTile stage_poses(const PointInputs& in, double* s_pose) {
  if (!in.poses || in.n_scans > kLdsScans) return PoseTile{nullptr};
  for (int t = threadIdx.x; t < in.n_scans * 12; t += blockDim.x) s_pose[t] = in.poses[t];
  return PoseTile{s_pose};
}

__device__ __forceinline__ void load_pose(const PointInputs& in, const PoseTile& tile, int s, double* T) {
  if (in.poses) {
    const double* p = (tile.lds ? tile.lds : in.poses) + (int64_t)s * 12;
#pragma unroll
    for (int q = 0; q < 12; ++q) T[q] = p[q];
  } else {
#pragma unroll
    for (int q = 0; q < 12; ++q) T[q] = (q == 0 || q == 5 || q == 10) ? 1.0 : 0.0;
  }
}

__device__ __forceinline__ void load_pose(const PointInputs& in, int s, double* T) { load_pose(in, PoseTile{nullptr}, s, T); }

}  // namespace dc
