// Mask / filter kernels of the map-consistency set-up phase (gfx950).  Reference: filters.py:85-113
// (within_bounds), :184-193 (valid neighbours), :196-254 (eigenvalue / ratio bounds),
// depth_cloud.py:314-326 (dir / vp dispersion), preproc.py:122-164 (global_cloud_mask).
// HBM-bound elementwise passes; one lane per point, coalesced.
#include "dc_common.h"
#include "../../include/dc_hip.h"
#include "dc_device.h"

namespace dc {

__device__ __forceinline__ bool in_bounds(double v, double lo, double hi) {
  // inclusive; -inf / +inf mean "unbounded" (filters.py:99-106); NaN fails any active bound
  bool keep = true;
  if (lo > -INFINITY) keep = keep && (v >= lo);
  if (hi < INFINITY) keep = keep && (v <= hi);
  return keep;
}

// mask[i] &= lo <= num[i*ns + ni] / den[i*ds + di] <= hi   (den == null: plain value)
template <typename T>
__global__ __launch_bounds__(kBlock) void mask_bounds_kernel(const T* __restrict__ num, int ns, int ni,
                                                             const T* __restrict__ den, int ds, int di, int64_t n, double lo,
                                                             double hi, uint8_t* __restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  T v = num[i * ns + ni];
  if (den) v = v / den[i * ds + di];          // in storage precision, like the reference's tensor division
  if (!in_bounds((double)v, lo, hi)) mask[i] = 0;
}

__global__ __launch_bounds__(kBlock) void valid_count_kernel(const int32_t* __restrict__ nbr, int64_t n, int k,
                                                             int32_t* __restrict__ cnt) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  int c = 0;
  for (int q = 0; q < k; ++q) c += nbr[i * k + q] >= 0;
  cnt[i] = c;
}

// Trace of the weighted covariance of vec[nbr[i]] (utils.covs + utils.trace), weights [N,K] or validity.
template <typename T>
__global__ __launch_bounds__(kBlock) void dispersion_kernel(const T* __restrict__ vec, const int32_t* __restrict__ nbr,
                                                            const T* __restrict__ weights, int64_t n, int k,
                                                            T* __restrict__ out) {
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  const int64_t blk = xcd_block(nblocks);
  if (blk < 0) return;
  const int64_t i = blk * kBlock + threadIdx.x;
  if (i >= n) return;
  const double a0 = (double)vec[i * 3], a1 = (double)vec[i * 3 + 1], a2 = (double)vec[i * 3 + 2];
  double W = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0, S = 0.0;
  for (int q = 0; q < k; ++q) {
    int32_t j = nbr[i * k + q];
    double w = weights ? (double)weights[i * k + q] : (j >= 0 ? 1.0 : 0.0);
    if (j < 0) { if (w == 0.0) continue; j += (int32_t)n; }       // torch wraps -1 to the last row
    const double d0 = (double)vec[(int64_t)j * 3] - a0, d1 = (double)vec[(int64_t)j * 3 + 1] - a1,
                 d2 = (double)vec[(int64_t)j * 3 + 2] - a2;
    W += w;
    s0 += w * d0; s1 += w * d1; s2 += w * d2;
    S += w * (d0 * d0 + d1 * d1 + d2 * d2);
  }
  double D = W - 1.0;
  D = D < 1e-6 ? 1e-6 : D;
  out[i] = (T)((S - (s0 * s0 + s1 * s1 + s2 * s2) / W) / D);
}

}  // namespace dc

using namespace dc;

extern "C" {

int dc_mask_bounds(const void* num, int num_stride, int num_index, const void* den, int den_stride, int den_index,
                   int dtype, int64_t n, double lo, double hi, uint8_t* mask, hipStream_t stream) {
  if (n == 0) return DC_OK;
  if (!num || !mask || n < 0 || num_stride < 1 || num_index < 0 || num_index >= num_stride) return DC_ERR_ARG;
  if (den && (den_stride < 1 || den_index < 0 || den_index >= den_stride)) return DC_ERR_ARG;
  if (n == 0) return DC_OK;
  const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
  if (dtype == DC_F32)
    hipLaunchKernelGGL((mask_bounds_kernel<float>), grid, block, 0, stream, (const float*)num, num_stride, num_index,
                       (const float*)den, den_stride, den_index, n, lo, hi, mask);
  else if (dtype == DC_F64)
    hipLaunchKernelGGL((mask_bounds_kernel<double>), grid, block, 0, stream, (const double*)num, num_stride, num_index,
                       (const double*)den, den_stride, den_index, n, lo, hi, mask);
  else return DC_ERR_DTYPE;
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DC_OK : (int)e;
}

int dc_valid_count(const int32_t* nbr, int64_t n, int k, int32_t* count_out, hipStream_t stream) {
  if (n == 0) return DC_OK;
  if (!nbr || !count_out || n < 0 || k < 1) return DC_ERR_ARG;
  if (n == 0) return DC_OK;
  hipLaunchKernelGGL(valid_count_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, nbr, n, k, count_out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DC_OK : (int)e;
}

int dc_dispersion(const void* vec, int dtype, const int32_t* nbr, const void* weights, int64_t n, int k, void* out,
                  hipStream_t stream) {
  if (n == 0) return DC_OK;
  if (!vec || !nbr || !out || n < 0 || k < 1) return DC_ERR_ARG;
  if (n == 0) return DC_OK;
  const dim3 grid((unsigned)xcd_grid((n + kBlock - 1) / kBlock)), block(kBlock);
  if (dtype == DC_F32)
    hipLaunchKernelGGL((dispersion_kernel<float>), grid, block, 0, stream, (const float*)vec, nbr, (const float*)weights, n, k, (float*)out);
  else if (dtype == DC_F64)
    hipLaunchKernelGGL((dispersion_kernel<double>), grid, block, 0, stream, (const double*)vec, nbr, (const double*)weights, n, k, (double*)out);
  else return DC_ERR_DTYPE;
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DC_OK : (int)e;
}

}  // extern "C"
