// ICP consistency between two consecutive scans with precomputed correspondences (gfx950): point-to-plane
// (loss.point_to_plane_dist loss.py:406-488) and point-to-point (loss.point_to_point_dist :491-565), both called from
// icp_loss :373-403 after model(c) and c.transform(pose) (:381-386), correspondences from train.py:178-210.
// Forward and hand-derived backward are one kernel: the loss is a sum of |n . (x2 - x1)| (or |x2 - x1|) terms, so every
// correspondence contributes its gradient straight to the model weights and to the two scan poses (block partial
// sums, fixed-order reduction).
#include "dc_common.h"
#include "../../include/dc_hip.h"
#include "dc_device.h"
#include "dc_pointmath.h"
#include "dc_points_dev.h"

namespace dc {

struct ScanView {
  const void* vps; const void* dirs; const void* depth; const void* inc; const uint8_t* lmask; const void* normals;
};

template <typename T>
struct ScanPoint {
  double xl[3], dr[3], nl[3], x[3], n[3], d, inc;
  bool lm;
};

template <typename T>
__device__ __forceinline__ void load_scan_point(const ScanView& s, const ModelParams& mp, const double* T12, int64_t i,
                                                ScanPoint<T>& p) {
  double vp[3];
  if (s.vps) Row3<T, 3>::load((const T*)s.vps, i, vp, QParams{});
  else { vp[0] = vp[1] = vp[2] = 0.0; }
  Row3<T, 3>::load((const T*)s.dirs, i, p.dr, QParams{});
  if (s.normals) Row3<T, 3>::load((const T*)s.normals, i, p.nl, QParams{});      // NULL: point-to-point, no normals
  else { p.nl[0] = p.nl[1] = p.nl[2] = 0.0; }
  p.d = (double)((const T*)s.depth)[i];
  p.lm = s.lmask ? s.lmask[i] != 0 : true;
  p.inc = (mp.kind != DC_MODEL_NONE && p.lm) ? (double)((const T*)s.inc)[i] : 0.0;
  const double dc_ = model_depth(mp, p.d, p.inc, p.lm);
#pragma unroll
  for (int a = 0; a < 3; ++a) p.xl[a] = vp[a] + dc_ * p.dr[a];
  rot3(T12, p.xl, p.x);
  p.x[0] += T12[3]; p.x[1] += T12[7]; p.x[2] += T12[11];
  // loss.py:436-437: points are cast to fp32 before the distances are formed
#pragma unroll
  for (int a = 0; a < 3; ++a) p.x[a] = (double)(float)p.x[a];
  rot3(T12, p.nl, p.n);
}

// gx = dL/dx, gn = dL/dn (world frame) -> model and pose gradients of this scan point.
template <typename T>
__device__ __forceinline__ void scan_point_bwd(const ModelParams& mp, const double* T12, const ScanPoint<T>& p,
                                               const double* gx, const double* gn, double* gw, double* ge, double* gT) {
  const double rg0 = T12[0] * gx[0] + T12[4] * gx[1] + T12[8] * gx[2];
  const double rg1 = T12[1] * gx[0] + T12[5] * gx[1] + T12[9] * gx[2];
  const double rg2 = T12[2] * gx[0] + T12[6] * gx[1] + T12[10] * gx[2];
  const double gd = p.dr[0] * rg0 + p.dr[1] * rg1 + p.dr[2] * rg2;
  if (mp.kind != DC_MODEL_NONE && p.lm) {
    const double base = mp.kind == DC_MODEL_SCALED_POLYNOMIAL ? -p.d * gd : -gd;
#pragma unroll
    for (int k = 0; k < DC_MAX_MODEL_TERMS; ++k) {
      if (k < mp.n_terms) {
        const double pk = pow_term(p.inc, mp.e[k]);
        gw[k] += base * pk;
        ge[k] += (p.inc > 0.0) ? base * mp.w[k] * pk * log(p.inc) : 0.0;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) gT[r * 4 + c] += gx[r] * p.xl[c] + gn[r] * p.nl[c];
    gT[r * 4 + 3] += gx[r];
  }
}

constexpr int kIcpAcc = 2 + 2 * DC_MAX_MODEL_TERMS + 24;

template <typename T, bool PLANE>
__global__ __launch_bounds__(kBlock) void p2plane_pair_kernel(ScanView A, ScanView B, const double* __restrict__ poseA,
                                                              const double* __restrict__ poseB, int model_kind, int n_terms,
                                                              const double* __restrict__ w, const double* __restrict__ e,
                                                              const int32_t* __restrict__ idxA,
                                                              const int32_t* __restrict__ idxB, int64_t m,
                                                              double* __restrict__ partials) {
  __shared__ double lds[(kBlock / kWave) * kIcpAcc];
  ModelParams mp;
  mp.kind = model_kind; mp.n_terms = n_terms;
#pragma unroll
  for (int k = 0; k < DC_MAX_MODEL_TERMS; ++k) {
    const bool on = k < n_terms && model_kind != DC_MODEL_NONE;
    mp.w[k] = on ? w[k] : 0.0; mp.e[k] = on ? e[k] : 0.0;
  }
  double TA[12], TB[12];
#pragma unroll
  for (int q = 0; q < 12; ++q) { TA[q] = poseA[q]; TB[q] = poseB[q]; }
  double acc[kIcpAcc];
#pragma unroll
  for (int q = 0; q < kIcpAcc; ++q) acc[q] = 0.0;
  double* gw = acc + 2;
  double* ge = acc + 2 + DC_MAX_MODEL_TERMS;
  double* gTA = acc + 2 + 2 * DC_MAX_MODEL_TERMS;
  double* gTB = gTA + 12;
  const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (c < m) {
    ScanPoint<T> a, b;
    load_scan_point<T>(A, mp, TA, idxA[c], a);
    load_scan_point<T>(B, mp, TB, idxB[c], b);
    const double dx[3] = {b.x[0] - a.x[0], b.x[1] - a.x[1], b.x[2] - a.x[2]};
    double gxa[3], gxb[3], gna[3], gnb[3];
    if (PLANE) {
      // 1 -> 2: | n1 . (x2 - x1) | |n1|
      const double k12 = a.n[0] * dx[0] + a.n[1] * dx[1] + a.n[2] * dx[2];
      const double na = sqrt(a.n[0] * a.n[0] + a.n[1] * a.n[1] + a.n[2] * a.n[2]);
      // 2 -> 1: | n2 . (x1 - x2) | |n2|
      const double k21 = -(b.n[0] * dx[0] + b.n[1] * dx[1] + b.n[2] * dx[2]);
      const double nb = sqrt(b.n[0] * b.n[0] + b.n[1] * b.n[1] + b.n[2] * b.n[2]);
      acc[0] = fabs(k12) * na;
      acc[1] = fabs(k21) * nb;
      const double s12 = k12 > 0.0 ? 1.0 : (k12 < 0.0 ? -1.0 : 0.0);
      const double s21 = k21 > 0.0 ? 1.0 : (k21 < 0.0 ? -1.0 : 0.0);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const double t = s12 * na * a.n[q] - s21 * nb * b.n[q];      // d/dx2 ; d/dx1 is its negative
        gxb[q] = t;
        gxa[q] = -t;
        gna[q] = s12 * na * dx[q] + (na > 0.0 ? fabs(k12) * a.n[q] / na : 0.0);
        gnb[q] = -s21 * nb * dx[q] + (nb > 0.0 ? fabs(k21) * b.n[q] / nb : 0.0);
      }
    } else {
      // point to point: | x2 - x1 | (loss.py:552-553); the norm's subgradient at zero is zero, as torch's
      const double len = sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]);
      acc[0] = len;
      const double inv = len > 0.0 ? 1.0 / len : 0.0;
#pragma unroll
      for (int q = 0; q < 3; ++q) { gxb[q] = dx[q] * inv; gxa[q] = -gxb[q]; gna[q] = gnb[q] = 0.0; }
    }
    scan_point_bwd<T>(mp, TA, a, gxa, gna, gw, ge, gTA);
    scan_point_bwd<T>(mp, TB, b, gxb, gnb, gw, ge, gTB);
  }
  block_sum<kIcpAcc>(acc, lds);
  if (threadIdx.x == 0) {
    double* row = partials + (int64_t)blockIdx.x * kIcpAcc;
#pragma unroll
    for (int q = 0; q < kIcpAcc; ++q) row[q] = acc[q];
  }
}

// Sum rows [n_rows, kIcpAcc] and compact to out = {sum12, sum21, gw[P], ge[P], gTA[12], gTB[12]}.
__global__ __launch_bounds__(kBlock) void p2plane_reduce_kernel(const double* __restrict__ partials, int64_t n_rows,
                                                                int n_terms, double* __restrict__ out) {
  __shared__ double lds[kBlock / kWave];
  const int a = blockIdx.x;     // source slot
  double s = 0.0;
  for (int64_t r = threadIdx.x; r < n_rows; r += kBlock) s += partials[r * kIcpAcc + a];
  double v[1] = {s};
  block_sum<1>(v, lds);
  if (threadIdx.x != 0) return;
  int dst = -1;
  if (a < 2) dst = a;
  else if (a < 2 + DC_MAX_MODEL_TERMS) { if (a - 2 < n_terms) dst = a; }
  else if (a < 2 + 2 * DC_MAX_MODEL_TERMS) { if (a - 2 - DC_MAX_MODEL_TERMS < n_terms) dst = 2 + n_terms + (a - 2 - DC_MAX_MODEL_TERMS); }
  else dst = 2 + 2 * n_terms + (a - 2 - 2 * DC_MAX_MODEL_TERMS);
  if (dst >= 0) out[dst] = v[0];
}

// Sequence layout: out = { loss, dw[P], dexponent[P], d[R|t][S,12] }; every pair ADDS weight * its sums (pairs run
// one after the other on the stream, one thread per slot => fixed order).  Block 0 folds both directed sums into the
// loss slot; block 1 is idle.
__global__ __launch_bounds__(kBlock) void p2plane_reduce_seq_kernel(const double* __restrict__ partials, int64_t n_rows,
                                                                    int n_terms, double weight, int scan_a, int scan_b,
                                                                    double* __restrict__ out) {
  __shared__ double lds[kBlock / kWave];
  const int a = blockIdx.x;
  if (a == 1) return;
  double s = 0.0;
  for (int64_t r = threadIdx.x; r < n_rows; r += kBlock)
    s += partials[r * kIcpAcc + a] + (a == 0 ? partials[r * kIcpAcc + 1] : 0.0);
  double v[1] = {s};
  block_sum<1>(v, lds);
  if (threadIdx.x != 0) return;
  int dst = -1;
  const int pw = 2, pe = 2 + DC_MAX_MODEL_TERMS, pa = 2 + 2 * DC_MAX_MODEL_TERMS, pb = pa + 12;
  if (a == 0) dst = 0;
  else if (a < pe) { if (a - pw < n_terms) dst = 1 + (a - pw); }
  else if (a < pa) { if (a - pe < n_terms) dst = 1 + n_terms + (a - pe); }
  else if (a < pb) dst = 1 + 2 * n_terms + 12 * scan_a + (a - pa);
  else dst = 1 + 2 * n_terms + 12 * scan_b + (a - pb);
  if (dst >= 0) out[dst] += weight * v[0];
}

static void launch_pair(bool plane, int dtype, int64_t rows, hipStream_t stream, const ScanView& A, const ScanView& B,
                        const double* poseA, const double* poseB, int model_kind, int n_terms, const double* w, const double* e,
                        const int32_t* idxA, const int32_t* idxB, int64_t m, double* partials_ws) {
  const dim3 grid((unsigned)rows), block(kBlock);
#define ICP_LAUNCH(T, PLANE) \
  hipLaunchKernelGGL((p2plane_pair_kernel<T, PLANE>), grid, block, 0, stream, A, B, poseA, poseB, model_kind, n_terms, w, e, idxA, idxB, m, partials_ws)
  if (dtype == DC_F32) { if (plane) ICP_LAUNCH(float, true); else ICP_LAUNCH(float, false); }
  else { if (plane) ICP_LAUNCH(double, true); else ICP_LAUNCH(double, false); }
#undef ICP_LAUNCH
}

}  // namespace dc

using namespace dc;

extern "C" {

int64_t dc_p2plane_partial_count(int64_t m) { return (m <= 0 ? 1 : (m + kBlock - 1) / kBlock) * kIcpAcc; }

static int icp_pair_impl(bool plane, const void* vpsA, const void* dirsA, const void* depthA, const void* incA, const uint8_t* lmaskA,
                    const void* normalsA, const void* vpsB, const void* dirsB, const void* depthB, const void* incB,
                    const uint8_t* lmaskB, const void* normalsB, int dtype, const double* poseA, const double* poseB,
                    int model_kind, int n_terms, const double* w, const double* e, const int32_t* idxA,
                    const int32_t* idxB, int64_t m, int want_exponent_grad, int want_pose_grad, double* partials_ws,
                    double* out, hipStream_t stream) {
  (void)want_exponent_grad; (void)want_pose_grad;      // always produced: the kernel is tiny next to the k-NN set-up
  if (!dirsA || !depthA || !dirsB || !depthB || (plane && (!normalsA || !normalsB))) return DC_ERR_ARG;
  if (!plane) normalsA = normalsB = nullptr;
  if (!poseA || !poseB || !idxA || !idxB || m < 0 || !partials_ws || !out) return DC_ERR_ARG;
  if (model_kind < DC_MODEL_NONE || model_kind > DC_MODEL_SCALED_POLYNOMIAL) return DC_ERR_ARG;
  if (model_kind != DC_MODEL_NONE && (n_terms < 1 || n_terms > DC_MAX_MODEL_TERMS || !incA || !incB || !w || !e)) return DC_ERR_ARG;
  if (model_kind == DC_MODEL_NONE) n_terms = 0;
  const int n_out = 2 + 2 * n_terms + 24;
  hipError_t err = hipMemsetAsync(out, 0, n_out * sizeof(double), stream);
  if (err != hipSuccess) return (int)err;
  if (m == 0) return DC_OK;
  ScanView A{vpsA, dirsA, depthA, incA, lmaskA, normalsA}, B{vpsB, dirsB, depthB, incB, lmaskB, normalsB};
  const int64_t rows = (m + kBlock - 1) / kBlock;
  if (dtype != DC_F32 && dtype != DC_F64) return DC_ERR_DTYPE;
  launch_pair(plane, dtype, rows, stream, A, B, poseA, poseB, model_kind, n_terms, w, e, idxA, idxB, m, partials_ws);
  hipLaunchKernelGGL(p2plane_reduce_kernel, dim3(kIcpAcc), dim3(kBlock), 0, stream, partials_ws, rows, n_terms, out);
  err = hipGetLastError();
  return err == hipSuccess ? DC_OK : (int)err;
}

int dc_p2plane_pair(const void* vpsA, const void* dirsA, const void* depthA, const void* incA, const uint8_t* lmaskA,
                    const void* normalsA, const void* vpsB, const void* dirsB, const void* depthB, const void* incB,
                    const uint8_t* lmaskB, const void* normalsB, int dtype, const double* poseA, const double* poseB,
                    int model_kind, int n_terms, const double* w, const double* e, const int32_t* idxA,
                    const int32_t* idxB, int64_t m, int want_exponent_grad, int want_pose_grad, double* partials_ws,
                    double* out, hipStream_t stream) {
  return icp_pair_impl(true, vpsA, dirsA, depthA, incA, lmaskA, normalsA, vpsB, dirsB, depthB, incB, lmaskB, normalsB, dtype, poseA,
                       poseB, model_kind, n_terms, w, e, idxA, idxB, m, want_exponent_grad, want_pose_grad, partials_ws, out, stream);
}

int dc_p2point_pair(const void* vpsA, const void* dirsA, const void* depthA, const void* incA, const uint8_t* lmaskA,
                    const void* vpsB, const void* dirsB, const void* depthB, const void* incB, const uint8_t* lmaskB, int dtype,
                    const double* poseA, const double* poseB, int model_kind, int n_terms, const double* w, const double* e,
                    const int32_t* idxA, const int32_t* idxB, int64_t m, double* partials_ws, double* out, hipStream_t stream) {
  return icp_pair_impl(false, vpsA, dirsA, depthA, incA, lmaskA, nullptr, vpsB, dirsB, depthB, incB, lmaskB, nullptr, dtype, poseA,
                       poseB, model_kind, n_terms, w, e, idxA, idxB, m, 1, 1, partials_ws, out, stream);
}

static int icp_sequence_impl(bool plane, const dcIcpScan* scans, int n_scans, const dcIcpPair* pairs, int n_pairs, int dtype,
                        const double* poses, int model_kind, int n_terms, const double* w, const double* e,
                        double* partials_ws, double* out, hipStream_t stream) {
  if (n_scans < 0 || n_pairs < 0 || (n_scans > 0 && !scans) || (n_pairs > 0 && !pairs) || !out) return DC_ERR_ARG;
  if (dtype != DC_F32 && dtype != DC_F64) return DC_ERR_DTYPE;
  if (model_kind < DC_MODEL_NONE || model_kind > DC_MODEL_SCALED_POLYNOMIAL) return DC_ERR_ARG;
  if (model_kind != DC_MODEL_NONE && (n_terms < 1 || n_terms > DC_MAX_MODEL_TERMS || !w || !e)) return DC_ERR_ARG;
  if (model_kind == DC_MODEL_NONE) n_terms = 0;
  for (int p = 0; p < n_pairs; ++p) {
    const dcIcpPair& q = pairs[p];
    if (q.scan_a < 0 || q.scan_a >= n_scans || q.scan_b < 0 || q.scan_b >= n_scans || q.scan_a == q.scan_b) return DC_ERR_ARG;
    if (q.m < 0 || (q.m > 0 && (!q.idx_a || !q.idx_b || !poses || !partials_ws))) return DC_ERR_ARG;
    for (int side = 0; side < 2; ++side) {
      const dcIcpScan& c = scans[side ? q.scan_b : q.scan_a];
      if (!c.dirs || !c.depth || (plane && !c.normals) || (model_kind != DC_MODEL_NONE && !c.inc)) return DC_ERR_ARG;
    }
  }
  hipError_t err = hipMemsetAsync(out, 0, (size_t)(1 + 2 * n_terms + 12 * n_scans) * sizeof(double), stream);
  if (err != hipSuccess) return (int)err;
  for (int p = 0; p < n_pairs; ++p) {
    const dcIcpPair& q = pairs[p];
    if (q.m == 0) continue;
    const dcIcpScan &a = scans[q.scan_a], &b = scans[q.scan_b];
    ScanView A{a.vps, a.dirs, a.depth, a.inc, a.lmask, plane ? a.normals : nullptr},
        B{b.vps, b.dirs, b.depth, b.inc, b.lmask, plane ? b.normals : nullptr};
    const double *poseA = poses + 12 * q.scan_a, *poseB = poses + 12 * q.scan_b;
    const int64_t rows = (q.m + kBlock - 1) / kBlock;
    launch_pair(plane, dtype, rows, stream, A, B, poseA, poseB, model_kind, n_terms, w, e, q.idx_a, q.idx_b, q.m, partials_ws);
    hipLaunchKernelGGL(p2plane_reduce_seq_kernel, dim3(kIcpAcc), dim3(kBlock), 0, stream, partials_ws, rows, n_terms,
                       q.weight, q.scan_a, q.scan_b, out);
  }
  err = hipGetLastError();
  return err == hipSuccess ? DC_OK : (int)err;
}

int dc_p2plane_sequence(const dcIcpScan* scans, int n_scans, const dcIcpPair* pairs, int n_pairs, int dtype,
                        const double* poses, int model_kind, int n_terms, const double* w, const double* e,
                        double* partials_ws, double* out, hipStream_t stream) {
  return icp_sequence_impl(true, scans, n_scans, pairs, n_pairs, dtype, poses, model_kind, n_terms, w, e, partials_ws, out, stream);
}

int dc_p2point_sequence(const dcIcpScan* scans, int n_scans, const dcIcpPair* pairs, int n_pairs, int dtype,
                        const double* poses, int model_kind, int n_terms, const double* w, const double* e,
                        double* partials_ws, double* out, hipStream_t stream) {
  return icp_sequence_impl(false, scans, n_scans, pairs, n_pairs, dtype, poses, model_kind, n_terms, w, e, partials_ws, out, stream);
}

}  // extern "C"
