// Neighbourhood features of the un-fused API path (gfx950): DepthCloud.update_features depth_cloud.py:426-433
// (update_mean :291-295, update_cov :366-369 -> utils.covs utils.py:109-149, update_eig :376-399,
// update_normals :409-415, update_incidence_angles :417-424) in one launch, and the hand-derived backward of
// (mean, cov, eigvals) for the torch.autograd.Function behind it.  C ABI at the bottom (include/dc_hip.h).
#include <atomic>
#include <cstdlib>
#include "dc_common.h"
#include "dc_device.h"
#include "dc_pointmath.h"
#include "dc_prof.h"
#include "../../include/dc_hip.h"

namespace dc {

// ------------------------------------------------------------------------------------------------
// Gather one neighbourhood and accumulate mean / second moments about the centre point.
// ------------------------------------------------------------------------------------------------
template <typename T, typename PT, int STRIDE>
__device__ __forceinline__ void gather_neighbourhood(const PT* __restrict__ x, const int32_t* __restrict__ nbr,
                                                     const T* __restrict__ wmean, int64_t i, int k, const double* xi,
                                                     const QParams& qp, CovAcc& acc) {
  cov_init(acc);
  const int32_t* row = nbr + i * k;
  for (int q = 0; q < k; ++q) {
    const int32_t j = row[q];
    if (j < 0) continue;
    double xj[3];
    Row3<PT, STRIDE>::load(x, j, xj, qp);
    const double wm = wmean ? (double)wmean[i * k + q] : 1.0;
    cov_add(acc, xj[0] - xi[0], xj[1] - xi[1], xj[2] - xi[2], wm);
  }
}

// ------------------------------------------------------------------------------------------------
// Full neighbourhood features (DepthCloud.update_features): mean, cov, eigvals, eigvecs, normals,
// incidence angles, valid-neighbour count, weights.
// ------------------------------------------------------------------------------------------------
template <typename T, int STRIDE>
__global__ __launch_bounds__(kBlock) void features_fwd_kernel(
    const T* __restrict__ x, const int32_t* __restrict__ nbr, const T* __restrict__ wmean, const T* __restrict__ dirs,
    int64_t n, int k, double scale, T* __restrict__ mean, T* __restrict__ cov, T* __restrict__ eigvals,
    T* __restrict__ eigvecs, T* __restrict__ normals, T* __restrict__ inc, int32_t* __restrict__ nvalid,
    T* __restrict__ weights_out, T* __restrict__ cmean_out, T* __restrict__ invd_out) {
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  const int64_t blk = xcd_block(nblocks);
  if (blk < 0) return;
  const int64_t i = blk * kBlock + threadIdx.x;
  if (i >= n) return;
  const QParams qp{};
  double xi[3];
  Row3<T, STRIDE>::load(x, i, xi, qp);
  CovAcc acc;
  gather_neighbourhood<T, T, STRIDE>(x, nbr, wmean, i, k, xi, qp, acc);
  double moff[3], cm[3], C[6], D, omega;
  cov_finish(acc, scale, moff, cm, C, &D, &omega);
  if (mean) { mean[i * 3] = (T)(xi[0] + moff[0]); mean[i * 3 + 1] = (T)(xi[1] + moff[1]); mean[i * 3 + 2] = (T)(xi[2] + moff[2]); }
  if (cmean_out) { cmean_out[i * 3] = (T)(xi[0] + cm[0]); cmean_out[i * 3 + 1] = (T)(xi[1] + cm[1]); cmean_out[i * 3 + 2] = (T)(xi[2] + cm[2]); }
  if (invd_out) invd_out[i] = (T)(omega / D);
  if (cov) {
    T* c = cov + i * 9;
    c[0] = (T)C[0]; c[1] = (T)C[1]; c[2] = (T)C[2];
    c[3] = (T)C[1]; c[4] = (T)C[3]; c[5] = (T)C[4];
    c[6] = (T)C[2]; c[7] = (T)C[4]; c[8] = (T)C[5];
  }
  if (nvalid) nvalid[i] = (int32_t)acc.W;
  if (weights_out) {
    const int32_t* row = nbr + i * k;
    for (int q = 0; q < k; ++q) weights_out[i * k + q] = row[q] >= 0 ? (T)omega : (T)0;
  }
  if (eigvals || eigvecs || normals || inc) {
    double lam[3], V[3][3];
    eig3_sym<double>(C[0], C[1], C[2], C[3], C[4], C[5], lam, V);
    if (eigvals) { eigvals[i * 3] = (T)lam[0]; eigvals[i * 3 + 1] = (T)lam[1]; eigvals[i * 3 + 2] = (T)lam[2]; }
    if (eigvecs) {   // torch layout: eigvecs[i, :, k] = k-th eigenvector
      T* e = eigvecs + i * 9;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) e[r * 3 + c] = (T)V[c][r];
    }
    if (normals || inc) {
      double dr[3], nrm[3], a;
      Row3<T, 3>::load(dirs, i, dr, qp);
      normal_and_incidence(dr, V[0], nrm, &a);
      if (normals) Row3<T, 3>::store(normals, i, nrm, qp);
      if (inc) inc[i] = (T)a;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same features, tiled (round 4): what dc_features_fwd launches for the neighbour counts the reference's configurations
// use (K = 4 / 8 / 10 / 16 compiled in), validity weights in the mean and 16-B aligned arrays.  features_fwd_kernel above walks a run-time
// slot loop -- one dependent {index, row} round trip per neighbour, 75 % of its wave cycles waiting -- reads the [N, K] index
// table with a lane stride of 4 K bytes and writes 28 output words per lane with strides of 12 / 36 bytes.  Here a WAVEFRONT
// owns 64 consecutive centres and its own LDS region -- one wavefront per workgroup: no block barrier, and the dispatcher balances
// 3 125 small workgroups over the CUs where 782 large ones left some CUs a third more work than the average (17.0 vs 18.9 us):
//   * its 64 x K index words are one contiguous piece of the table: read as 16-B words, lane-contiguous, handed over through
//     LDS, and every lane picks its own row up with 8-B reads (K even: no bank conflict inside a 32-lane group);
//   * the K neighbour rows are K independent gathers in flight at once (compile-time K: straight-line code, one dwordx3 /
//     dwordx4 per neighbour); the centre and its direction are lane-contiguous loads issued before the indices arrive;
//   * the full decomposition is eig3_sym_v2 (dc_eig3.h: ~270 instructions instead of ~700);
//   * every output array goes through the wavefront's LDS region (row-major, odd word stride: conflict-free) and leaves as
//     16-B words that are contiguous across the lanes -- full 64-B segments instead of 4- to 16-B pieces of them.
// Results: the moments are summed in the same order as above, the eigen-solver differs by round-off (<= 1e-14 relative to the
// covariance's norm for both, tests/test_hostcheck.py).  Wavefronts of the last, partial block of 64 write lane by lane.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // a wavefront's LDS operations execute in order: only the compiler
  __builtin_amdgcn_wave_barrier();                           // has to be kept from moving them across the hand-over
}

// C values per lane -> out[(i0 + lane) * C + c]; `full`: all 64 lanes hold a centre and out + i0 * C is 16-B aligned
template <typename T, int C>
__device__ __forceinline__ void wave_store_rows(T* __restrict__ out, int64_t i0, int lane, bool full, bool active,
                                                const double* v, T* stage) {
  if (full) {
#pragma unroll
    for (int c = 0; c < C; ++c) stage[lane * C + c] = (T)v[c];
    wave_lds_sync();
    constexpr int kVec = 64 * C * (int)sizeof(T) / 16;
    const int4* s4 = reinterpret_cast<const int4*>(stage);
    int4* o4 = reinterpret_cast<int4*>(out + i0 * C);
#pragma unroll
    for (int t = 0; t < (kVec + 63) / 64; ++t) {
      const int e = t * 64 + lane;
      if (e < kVec) o4[e] = s4[e];
    }
    wave_lds_sync();
  } else if (active) {
#pragma unroll
    for (int c = 0; c < C; ++c) out[(i0 + lane) * C + c] = (T)v[c];
  }
}

template <typename T, int STRIDE, int K>
__global__ __launch_bounds__(kWave) void features_fwd_tile_kernel(
    const T* __restrict__ x, const int32_t* __restrict__ nbr, const T* __restrict__ dirs, int64_t n, double scale,
    T* __restrict__ mean, T* __restrict__ cov, T* __restrict__ eigvals, T* __restrict__ eigvecs, T* __restrict__ normals,
    T* __restrict__ inc, int32_t* __restrict__ nvalid, T* __restrict__ cmean_out, T* __restrict__ invd_out, T* __restrict__ weights_out) {
  constexpr int kIdxBytes = 64 * K * 4, kStageBytes = 64 * (K > 9 ? K : 9) * (int)sizeof(T);
  constexpr int kRegion16 = (kIdxBytes > kStageBytes ? kIdxBytes : kStageBytes) / 16;
  __shared__ int4 region[kRegion16];
  const int64_t nblocks = (n + kWave - 1) / kWave;
  const int64_t blk = xcd_block(nblocks);
  if (blk < 0) return;
  const int lane = threadIdx.x;
  const int64_t i0 = blk * kWave;
  const int64_t left = n - i0;
  const bool full = left >= kWave;
  const bool active = lane < left;
  const int64_t i = active ? i0 + lane : i0;               // idle lanes of the last wavefront shadow its first centre
  const QParams qp{};
  // ---- the wavefront's piece of the index table: 64 K words, 16 K of them 16-B words ----
  constexpr int kIdx16 = 16 * K, kTrips = (kIdx16 + 63) / 64;
  // centre and direction: lane-contiguous, in flight together with the indices
  double xi[3], dr[3] = {0.0, 0.0, 0.0};
  Row3<T, STRIDE>::load(x, i, xi, qp);
  const bool want_dir = normals || inc;
  if (want_dir) Row3<T, 3>::load(dirs, i, dr, qp);
  int32_t row[K];
  if (full) {
    const int4* tile = reinterpret_cast<const int4*>(nbr + i0 * K);
    int4 t0 = make_int4(0, 0, 0, 0), t1 = t0, t2 = t0, t3 = t0;
    t0 = tile[lane];
    if (kTrips > 1 && 64 + lane < kIdx16) t1 = tile[64 + lane];
    if (kTrips > 2 && 128 + lane < kIdx16) t2 = tile[128 + lane];
    if (kTrips > 3 && 192 + lane < kIdx16) t3 = tile[192 + lane];
    region[lane] = t0;
    if (kTrips > 1 && 64 + lane < kIdx16) region[64 + lane] = t1;
    if (kTrips > 2 && 128 + lane < kIdx16) region[128 + lane] = t2;
    if (kTrips > 3 && 192 + lane < kIdx16) region[192 + lane] = t3;
    wave_lds_sync();
    const int2* r2 = reinterpret_cast<const int2*>(region) + lane * (K / 2);
#pragma unroll
    for (int q = 0; q < K / 2; ++q) { const int2 v = r2[q]; row[2 * q] = v.x; row[2 * q + 1] = v.y; }
    wave_lds_sync();
  } else {
#pragma unroll
    for (int q = 0; q < K; ++q) row[q] = nbr[i * K + q];
  }
  // ---- K gathers in flight ----
  // (a k-NN table lists every point first among its own neighbours: when that holds for the whole wavefront the first gather is
  // the centre the lane already has -- a tenth of the kernel's random line look-ups)
  double xj[K][3];
  bool any_missing = false;
  const bool self_first = __builtin_amdgcn_ballot_w64((int64_t)row[0] != i) == 0;
#pragma unroll
  for (int q = 0; q < K; ++q) {
    any_missing |= row[q] < 0;
    if (q == 0 && self_first) { xj[0][0] = xi[0]; xj[0][1] = xi[1]; xj[0][2] = xi[2]; }
    else Row3<T, STRIDE>::load(x, row[q] < 0 ? i : (int64_t)row[q], xj[q], qp);
  }
  CovAcc acc;
  cov_init(acc);
  if (__builtin_amdgcn_ballot_w64(any_missing) == 0) {     // a full k-NN table: no selects
#pragma unroll
    for (int q = 0; q < K; ++q) cov_add(acc, xj[q][0] - xi[0], xj[q][1] - xi[1], xj[q][2] - xi[2], 1.0);
  } else {
#pragma unroll
    for (int q = 0; q < K; ++q)
      if (row[q] >= 0) cov_add(acc, xj[q][0] - xi[0], xj[q][1] - xi[1], xj[q][2] - xi[2], 1.0);
  }
  double moff[3], cm[3], C[6], D, omega;
  cov_finish(acc, scale, moff, cm, C, &D, &omega);
  T* stage = reinterpret_cast<T*>(region);
  if (mean) {
    const double v[3] = {xi[0] + moff[0], xi[1] + moff[1], xi[2] + moff[2]};
    wave_store_rows<T, 3>(mean, i0, lane, full, active, v, stage);
  }
  if (cmean_out) {
    const double v[3] = {xi[0] + cm[0], xi[1] + cm[1], xi[2] + cm[2]};
    wave_store_rows<T, 3>(cmean_out, i0, lane, full, active, v, stage);
  }
  if (invd_out && active) invd_out[i] = (T)(omega / D);
  if (weights_out) {                       // update_weights (depth_cloud.py:356-364): the centre's weight on its valid neighbours
    double v[K];
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] = row[q] >= 0 ? omega : 0.0;
    wave_store_rows<T, K>(weights_out, i0, lane, full, active, v, stage);
  }
  if (nvalid && active) nvalid[i] = (int32_t)acc.W;
  if (cov) {
    const double v[9] = {C[0], C[1], C[2], C[1], C[3], C[4], C[2], C[4], C[5]};
    wave_store_rows<T, 9>(cov, i0, lane, full, active, v, stage);
  }
  if (eigvals || eigvecs || want_dir) {
    double lam[3], V[3][3];
    eig3_sym_v2(C[0], C[1], C[2], C[3], C[4], C[5], lam, V);
    if (eigvals) wave_store_rows<T, 3>(eigvals, i0, lane, full, active, lam, stage);
    if (eigvecs) {   // torch layout: eigvecs[i, :, k] = k-th eigenvector
      const double v[9] = {V[0][0], V[1][0], V[2][0], V[0][1], V[1][1], V[2][1], V[0][2], V[1][2], V[2][2]};
      wave_store_rows<T, 9>(eigvecs, i0, lane, full, active, v, stage);
    }
    if (want_dir) {
      double nrm[3], a;
      normal_and_incidence(dr, V[0], nrm, &a);
      if (normals) wave_store_rows<T, 3>(normals, i0, lane, full, active, nrm, stage);
      if (inc && active) inc[i] = (T)a;
    }
  }
}

// Generic neighbourhood-features backward for the un-fused API path:
//   dL/dx_j = sum_{i -> j} [ Gs_i (x_j - cmean_i) + gm_i ],  grec[N,12] = {cmean.xyz, Gs(xx xy xz yy yz zz), gm.xyz}
template <typename T, int STRIDE>
__global__ __launch_bounds__(kBlock) void features_bwd_kernel(const T* __restrict__ x, const T* __restrict__ grec,
                                                              const int32_t* __restrict__ csr_ptr,
                                                              const int32_t* __restrict__ csr_src, int64_t n,
                                                              T* __restrict__ grad_points) {
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  const int64_t blk = xcd_block(nblocks);
  if (blk < 0) return;
  const int64_t j = blk * kBlock + threadIdx.x;
  if (j >= n) return;
  double xj[3], g[3] = {0.0, 0.0, 0.0};
  Row3<T, STRIDE>::load(x, j, xj, QParams{});
  const int32_t beg = csr_ptr[j], end = csr_ptr[j + 1];
  for (int32_t e = beg; e < end; ++e) {
    const T* r = grec + (int64_t)csr_src[e] * 12;
    const double d0 = xj[0] - (double)r[0], d1 = xj[1] - (double)r[1], d2 = xj[2] - (double)r[2];
    const double xx = r[3], xy = r[4], xz = r[5], yy = r[6], yz = r[7], zz = r[8];
    g[0] += xx * d0 + xy * d1 + xz * d2 + (double)r[9];
    g[1] += xy * d0 + yy * d1 + yz * d2 + (double)r[10];
    g[2] += xz * d0 + yz * d1 + zz * d2 + (double)r[11];
  }
  Row3<T, STRIDE>::store(grad_points, j, g, QParams{});
}

// Build the generic backward record from upstream gradients of (mean, cov, eigvals):
//   G = V diag(ge) V^T + sym(gcov);  Gs = 2 (omega / D) G;  gm = gmean / W.
template <typename T>
__global__ __launch_bounds__(kBlock) void features_grec_kernel(const T* __restrict__ cmean, const T* __restrict__ invd,
                                                               const int32_t* __restrict__ nvalid,
                                                               const T* __restrict__ eigvecs, const T* __restrict__ g_mean,
                                                               const T* __restrict__ g_cov, const T* __restrict__ g_eig,
                                                               int64_t n, T* __restrict__ grec) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  double G[6] = {0, 0, 0, 0, 0, 0};
  if (g_eig) {
    const T* e = eigvecs + i * 9;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double ge = (double)g_eig[i * 3 + k];
      const double v0 = e[0 * 3 + k], v1 = e[1 * 3 + k], v2 = e[2 * 3 + k];
      G[0] += ge * v0 * v0; G[1] += ge * v0 * v1; G[2] += ge * v0 * v2;
      G[3] += ge * v1 * v1; G[4] += ge * v1 * v2; G[5] += ge * v2 * v2;
    }
  }
  if (g_cov) {
    const T* c = g_cov + i * 9;
    G[0] += (double)c[0]; G[1] += 0.5 * ((double)c[1] + (double)c[3]); G[2] += 0.5 * ((double)c[2] + (double)c[6]);
    G[3] += (double)c[4]; G[4] += 0.5 * ((double)c[5] + (double)c[7]); G[5] += (double)c[8];
  }
  const double f = 2.0 * (double)invd[i];
  T* r = grec + i * 12;
  r[0] = cmean[i * 3]; r[1] = cmean[i * 3 + 1]; r[2] = cmean[i * 3 + 2];
#pragma unroll
  for (int q = 0; q < 6; ++q) r[3 + q] = (T)(f * G[q]);
  const double w = (double)nvalid[i];
#pragma unroll
  for (int q = 0; q < 3; ++q) r[9 + q] = g_mean ? (T)((double)g_mean[i * 3 + q] / w) : (T)0;
}

}  // namespace dc

using namespace dc;

#define DC_CHECK_LAUNCH()                                  \
  do {                                                     \
    hipError_t err__ = hipGetLastError();                  \
    if (err__ != hipSuccess) return (int)err__;            \
  } while (0)

static inline int64_t n_blocks(int64_t n) { return (n + kBlock - 1) / kBlock; }

// A-B switch for measurements (tools/): 0 sends every call to the general kernel.  Not a product option.
static std::atomic<int> g_features_tiled{1};

extern "C" {

// (an A-B switch like dc_set_option: refused -- the setting reported, nothing changed -- unless the process asked for the switches)
int dc_features_set_tiled(int on) {
  static const bool enabled = [] { const char* e = getenv("DC_ENABLE_ABLATIONS"); return e && atoi(e) != 0; }();
  if (!enabled) return g_features_tiled.load(std::memory_order_relaxed);
  return g_features_tiled.exchange(on ? 1 : 0, std::memory_order_relaxed);
}

int dc_features_fwd(const void* points, int stride, int dtype, const int32_t* nbr, int64_t n, int k,
                    const void* mean_weights, double scale, const void* dirs, void* mean, void* cov, void* eigvals,
                    void* eigvecs, void* normals, void* inc_angles, int32_t* nvalid, void* weights_out,
                    void* cmean_out, void* invd_out, hipStream_t stream) {
  if (n == 0) return DC_OK;
  if (n < 0 || k < 1 || !points || !nbr || (stride != 3 && stride != 4)) return DC_ERR_ARG;
  if ((normals || inc_angles) && !dirs) return DC_ERR_ARG;
  dim3 grid((unsigned)xcd_grid(n_blocks(n))), block(kBlock);
  ProfScope prof(3);                       // dc_profiler_*: kind 3
  // the tiled kernel: compiled-in neighbour counts, validity weights, every array 16-B aligned (torch allocations are; a
  // sliced view may not be) -- anything else takes the general kernel
  {
    const void* arrs[] = {points, nbr, dirs, mean, cov, eigvals, eigvecs, normals, inc_angles, nvalid, cmean_out, invd_out, weights_out};
    bool aligned = true;
    for (const void* a : arrs) aligned = aligned && (((uintptr_t)a & 15u) == 0);
    const bool tiled = aligned && !mean_weights && (k == 4 || k == 8 || k == 10 || k == 16) &&
                       (dtype == DC_F32 || dtype == DC_F64) && g_features_tiled.load(std::memory_order_relaxed);
    if (tiled) {
#define TILE(T, S, KK) \
  DC_TIMED_LAUNCH((features_fwd_tile_kernel<T, S, KK>), dim3((unsigned)xcd_grid((n + kWave - 1) / kWave)), dim3(kWave), 0, stream, \
                  (const T*)points, nbr, (const T*)dirs, n, scale, (T*)mean, (T*)cov, (T*)eigvals, (T*)eigvecs, (T*)normals, \
                  (T*)inc_angles, nvalid, (T*)cmean_out, (T*)invd_out, (T*)weights_out)
#define TILE_K(T, S) \
  do { if (k == 4) TILE(T, S, 4); else if (k == 8) TILE(T, S, 8); else if (k == 10) TILE(T, S, 10); else TILE(T, S, 16); } while (0)
      if (dtype == DC_F32) { if (stride == 3) TILE_K(float, 3); else TILE_K(float, 4); }
      else { if (stride == 3) TILE_K(double, 3); else TILE_K(double, 4); }
#undef TILE_K
#undef TILE
      DC_CHECK_LAUNCH();
      return DC_OK;
    }
  }
#define LAUNCH(T, S) \
  DC_TIMED_LAUNCH((features_fwd_kernel<T, S>), grid, block, 0, stream, (const T*)points, nbr, (const T*)mean_weights, \
                     (const T*)dirs, n, k, scale, (T*)mean, (T*)cov, (T*)eigvals, (T*)eigvecs, (T*)normals, \
                     (T*)inc_angles, nvalid, (T*)weights_out, (T*)cmean_out, (T*)invd_out)
  if (dtype == DC_F32) { if (stride == 3) LAUNCH(float, 3); else LAUNCH(float, 4); }
  else if (dtype == DC_F64) { if (stride == 3) LAUNCH(double, 3); else LAUNCH(double, 4); }
  else return DC_ERR_DTYPE;
#undef LAUNCH
  DC_CHECK_LAUNCH();
  return DC_OK;
}

int dc_features_bwd(const void* points, int stride, int dtype, const int32_t* csr_ptr, const int32_t* csr_src,
                    int64_t n, const void* cmean, const void* invd, const int32_t* nvalid, const void* eigvecs,
                    const void* grad_mean, const void* grad_cov, const void* grad_eigvals, void* grec_ws,
                    void* grad_points, hipStream_t stream) {
  if (n == 0) return DC_OK;
  if (n < 0 || !points || !csr_ptr || !csr_src || !cmean || !invd || !nvalid || !grec_ws || !grad_points) return DC_ERR_ARG;
  if (stride != 3 && stride != 4) return DC_ERR_ARG;
  if (grad_eigvals && !eigvecs) return DC_ERR_ARG;
  if (n == 0) return DC_OK;
  dim3 block(kBlock);
#define LAUNCH(T, S) \
  do { \
    hipLaunchKernelGGL((features_grec_kernel<T>), dim3((unsigned)n_blocks(n)), block, 0, stream, (const T*)cmean, \
                       (const T*)invd, nvalid, (const T*)eigvecs, (const T*)grad_mean, (const T*)grad_cov, \
                       (const T*)grad_eigvals, n, (T*)grec_ws); \
    hipLaunchKernelGGL((features_bwd_kernel<T, S>), dim3((unsigned)xcd_grid(n_blocks(n))), block, 0, stream, \
                       (const T*)points, (const T*)grec_ws, csr_ptr, csr_src, n, (T*)grad_points); \
  } while (0)
  if (dtype == DC_F32) { if (stride == 3) LAUNCH(float, 3); else LAUNCH(float, 4); }
  else if (dtype == DC_F64) { if (stride == 3) LAUNCH(double, 3); else LAUNCH(double, 4); }
  else return DC_ERR_DTYPE;
#undef LAUNCH
  DC_CHECK_LAUNCH();
  return DC_OK;
}

}  // extern "C"
