// Raw scan -> DepthCloud source fields on the device (gfx950), one pass over the uploaded file contents.
//
// The reference reads a scan on the host, filters it there and only then converts it to tensors:
//   KITTI-360 Velodyne .bin  float32 [N,4] (x, y, z, intensity), ego-vehicle box |x| <= d and |y| <= d removed
//                            (datasets/kitti360.py:96-109)
//   ASL-laser CSV / .npz     columns x, y, z (datasets/asl_laser.py:33-45), FEE corridor structured .npz with viewpoints
//                            (datasets/fee_corridor.py:35-38)
//   depth pre-filter         filters.filter_depth filters.py:116-141 (preproc.filtered_cloud preproc.py:25-28)
//   DepthCloud.from_points   depth_cloud.py:592-638: rays = pts - vps, depth = |rays|, dirs = rays / depth (rays with zero
//                            depth are left as they are, :626-627)
// Here the raw rows are uploaded once (pinned host buffer -> device) and ONE flag kernel + ONE stable compaction produce
// vps / dirs / depth of the kept points in their original order, plus the indices of the kept rows.
#include <cstring>
#include <cstdlib>
#include "dc_common.h"
#include "../../include/dc_hip.h"
#include "dc_device.h"
#include "dc_hostutil.h"
#include "dc_sort.h"

namespace dc {

template <typename TI, typename TO>
__device__ __forceinline__ void ray_of(const TI* __restrict__ pts, int stride, const TI* __restrict__ vps, int64_t i, TO* ray, TO* vp) {
  // the reference converts to the cloud dtype first (torch.as_tensor(pts, dtype=dtype)), then subtracts
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    vp[a] = vps ? (TO)vps[i * 3 + a] : (TO)0;
    ray[a] = (TO)pts[i * stride + a] - vp[a];
  }
}

// keep[i] = 1 when the raw row survives the ego-box crop and the depth bounds
template <typename TI, typename TO>
__global__ __launch_bounds__(kBlock) void scan_flags_kernel(const TI* __restrict__ pts, int stride, const TI* __restrict__ vps,
                                                            int64_t n, TI ego, TI dmin, TI dmax, int32_t* __restrict__ keep) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  bool k = true;
  if (ego > (TI)0) {               // kitti360.py:101-105, on the raw values
    const TI x = pts[i * stride], y = pts[i * stride + 1];
    k = (x < -ego) || (x > ego) || (y < -ego) || (y > ego);
  }
  if (dmin > -(TI)INFINITY || dmax < (TI)INFINITY) {        // filters.py:116-141 on the RAW array (its own dtype), inclusive, NaN fails
    TI ray[3], vp[3];
    ray_of<TI, TI>(pts, stride, vps, i, ray, vp);
    const TI d = sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
    if (dmin > -(TI)INFINITY) k = k && (d >= dmin);
    if (dmax < (TI)INFINITY) k = k && (d <= dmax);
  }
  keep[i] = k ? 1 : 0;
}

// pos = exclusive prefix of keep: kept row i goes to output row pos[i]
template <typename TI, typename TO>
__global__ __launch_bounds__(kBlock) void scan_compact_kernel(const TI* __restrict__ pts, int stride, const TI* __restrict__ vps,
                                                              int64_t n, const int32_t* __restrict__ keep,
                                                              const int32_t* __restrict__ pos, TO* __restrict__ dirs,
                                                              TO* __restrict__ depth, TO* __restrict__ vps_out,
                                                              int32_t* __restrict__ index_out, int64_t* __restrict__ count_out) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if (i == n - 1 && count_out) *count_out = (int64_t)pos[i] + keep[i];
  if (!keep[i]) return;
  const int64_t o = pos[i];
  TO ray[3], vp[3];
  ray_of<TI, TO>(pts, stride, vps, i, ray, vp);
  const TO d = sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
  depth[o] = d;
  const bool unit = d > (TO)0;                              // zero-depth rays are left un-normalised (depth_cloud.py:626-627)
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    dirs[o * 3 + a] = unit ? ray[a] / d : ray[a];
    if (vps_out) vps_out[o * 3 + a] = vp[a];
  }
  if (index_out) index_out[o] = (int32_t)i;
}

// no crop and no depth bounds: every row is kept -- one kernel, no flags, no prefix sums
template <typename TI, typename TO>
__global__ __launch_bounds__(kBlock) void scan_direct_kernel(const TI* __restrict__ pts, int stride, const TI* __restrict__ vps, int64_t n,
                                                             TO* __restrict__ dirs, TO* __restrict__ depth, TO* __restrict__ vps_out,
                                                             int32_t* __restrict__ index_out, int64_t* __restrict__ count_out) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if (i == 0) *count_out = n;
  TO ray[3], vp[3];
  ray_of<TI, TO>(pts, stride, vps, i, ray, vp);
  const TO d = sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
  depth[i] = d;
  const bool unit = d > (TO)0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    dirs[i * 3 + a] = unit ? ray[a] / d : ray[a];
    if (vps_out) vps_out[i * 3 + a] = vp[a];
  }
  if (index_out) index_out[i] = (int32_t)i;
}

}  // namespace dc

using namespace dc;

extern "C" {

size_t dc_cloud_from_points_workspace_bytes(int64_t n) {
  if (n < 0) return 0;
  const size_t ne = (size_t)(n > 0 ? n : 1);
  Carver c(nullptr);
  c.take<int32_t>(ne); c.take<int32_t>(ne);
  const size_t cb = scan_bytes(ne);
  c.take<char>(cb);
  return c.off + 256;
}

int dc_cloud_from_points(const void* points, int stride, int in_dtype, const void* vps, int64_t n, double ego_box,
                         double min_depth, double max_depth, int out_dtype, void* dirs_out, void* depth_out, void* vps_out,
                         int32_t* index_out, int64_t* count_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n < 0 || stride < 3 || !count_out || (n > 0 && (!points || !dirs_out || !depth_out || !ws))) return DC_ERR_ARG;
  if ((in_dtype != DC_F32 && in_dtype != DC_F64) || (out_dtype != DC_F32 && out_dtype != DC_F64)) return DC_ERR_DTYPE;
  if (n >= (int64_t)0x7fffffff) return DC_ERR_UNSUPPORTED;
  if (n == 0) return (int)hipMemsetAsync(count_out, 0, sizeof(int64_t), stream);
  if (ws_bytes < dc_cloud_from_points_workspace_bytes(n)) return DC_ERR_WORKSPACE;
  Carver c(ws);
  int32_t* keep = c.take<int32_t>((size_t)n);
  int32_t* pos = c.take<int32_t>((size_t)n);
  const size_t cb = scan_bytes((size_t)n);
  void* tmp = c.take<char>(cb);
  const double lo = (min_depth == min_depth) ? min_depth : -INFINITY, hi = (max_depth == max_depth) ? max_depth : INFINITY;   // NaN = unbounded
  const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
  const bool unfiltered = !(ego_box > 0.0) && lo == -INFINITY && hi == INFINITY;
#define RUN(TI, TO)                                                                                                          \
  do {                                                                                                                       \
    if (unfiltered) {                                                                                                        \
      hipLaunchKernelGGL((scan_direct_kernel<TI, TO>), grid, block, 0, stream, (const TI*)points, stride, (const TI*)vps, n,  \
                         (TO*)dirs_out, (TO*)depth_out, (TO*)vps_out, index_out, count_out);                                 \
      break;                                                                                                                 \
    }                                                                                                                        \
    hipLaunchKernelGGL((scan_flags_kernel<TI, TO>), grid, block, 0, stream, (const TI*)points, stride, (const TI*)vps, n,     \
                       (TI)ego_box, (TI)lo, (TI)hi, keep);                                                                   \
    DC_HIP(exclusive_scan_32(tmp, cb, keep, pos, (size_t)n, stream));                                                        \
    hipLaunchKernelGGL((scan_compact_kernel<TI, TO>), grid, block, 0, stream, (const TI*)points, stride, (const TI*)vps, n,   \
                       keep, pos, (TO*)dirs_out, (TO*)depth_out, (TO*)vps_out, index_out, count_out);                        \
  } while (0)
  if (in_dtype == DC_F32 && out_dtype == DC_F32) RUN(float, float);
  else if (in_dtype == DC_F32) RUN(float, double);
  else if (out_dtype == DC_F32) RUN(double, float);
  else RUN(double, double);
#undef RUN
  DC_HIP(hipGetLastError());
  return DC_OK;
}

}  // extern "C"
