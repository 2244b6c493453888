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
  if (mp.kind > DC_MODEL_SCALED_POLYNOMIAL && p.lm) {        // Linear / InvCos / ScaledInvCos
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k < mp.n_terms) gw[k] += gd * model_dw_other(mp, k, p.d, p.inc);
  } else if (mp.kind != DC_MODEL_NONE && p.lm) {
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

// kIcpAcc sums of a block through packed wavefront reductions (eight values per butterfly: 10 exchange steps instead of 48),
// then one LDS row per wavefront; thread q < kIcpAcc ends with the block's total of accumulator q.  Fixed order.
constexpr int kIcpGroups = (kIcpAcc + 7) / 8;
__device__ __forceinline__ double icp_block_sums(double* acc, double* lds /* [kWavesPerBlock][kIcpGroups * 8] */) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
  for (int g = 0; g < kIcpGroups; ++g) {
    double v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (g * 8 + q < kIcpAcc) ? acc[g * 8 + q] : 0.0;
    const double tot = wave_sum_packed<8>(v);
    if (lane < 8) lds[wave * (kIcpGroups * 8) + g * 8 + packed_value_of_lane<8>(lane)] = tot;
  }
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x < kIcpAcc) {
    const int q = threadIdx.x;
    r = (lds[q] + lds[kIcpGroups * 8 + q]) + (lds[2 * kIcpGroups * 8 + q] + lds[3 * kIcpGroups * 8 + q]);
  }
  return r;
}

// One block of one scan pair: correspondences [256 blk, 256 blk + 256) of (idxA, idxB) -> row[kIcpAcc] of block sums.
template <typename T, bool PLANE>
__device__ __forceinline__ void icp_pair_block(const ScanView& A, const ScanView& B, const double* __restrict__ poseA,
                                               const double* __restrict__ poseB, int model_kind, int n_terms,
                                               const double* __restrict__ w, const double* __restrict__ e,
                                               const int32_t* __restrict__ idxA, const int32_t* __restrict__ idxB, int64_t m,
                                               int64_t blk, double* __restrict__ row, double* lds) {
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
  const int64_t c = blk * kBlock + threadIdx.x;
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
  const double tot = icp_block_sums(acc, lds);
  if (threadIdx.x < kIcpAcc) row[threadIdx.x] = tot;
}

template <typename T, bool PLANE>
__global__ __launch_bounds__(kBlock) void p2plane_pair_kernel(ScanView A, ScanView B, const double* __restrict__ poseA,
                                                              const double* __restrict__ poseB, int model_kind, int n_terms,
                                                              const double* __restrict__ w, const double* __restrict__ e,
                                                              const int32_t* __restrict__ idxA,
                                                              const int32_t* __restrict__ idxB, int64_t m,
                                                              double* __restrict__ partials) {
  __shared__ double lds[kWavesPerBlock * kIcpGroups * 8];
  icp_pair_block<T, PLANE>(A, B, poseA, poseB, model_kind, n_terms, w, e, idxA, idxB, m, blockIdx.x,
                           partials + (int64_t)blockIdx.x * kIcpAcc, lds);
}

// All (up to kIcpSeqPairs) scan pairs of a sequence in ONE launch: the pairs' descriptors travel as kernel arguments, block b
// belongs to the pair whose row range holds it.  (One launch per pair left a C4-shaped iteration -- nine pairs of ~15 k
// correspondences -- with eighteen dependent launches of ~50 blocks each: 0.22 of its 0.29 ms.)
constexpr int kIcpSeqPairs = 16;
constexpr int kIcpDstScans = 1024;      // scans whose pose-gradient entries p2plane_reduce_all_kernel keeps in LDS (more: global adds)
struct IcpSeqPair {
  ScanView A, B;
  const int32_t* idxA; const int32_t* idxB;
  int64_t m;
  double weight;
  int32_t scan_a, scan_b, row0, rows;
};
struct IcpSeqArgs {
  IcpSeqPair pairs[kIcpSeqPairs];
  int n_pairs;
};

template <typename T, bool PLANE>
__global__ __launch_bounds__(kBlock) void p2plane_seq_kernel(IcpSeqArgs args, const double* __restrict__ poses, int model_kind, int n_terms,
                                                             const double* __restrict__ w, const double* __restrict__ e,
                                                             double* __restrict__ partials) {
  __shared__ double lds[kWavesPerBlock * kIcpGroups * 8];
  const int b = blockIdx.x;
  int p = 0;
#pragma unroll 1
  for (int q = 1; q < args.n_pairs; ++q)
    if (b >= args.pairs[q].row0) p = q;
  const IcpSeqPair& pr = args.pairs[p];
  icp_pair_block<T, PLANE>(pr.A, pr.B, poses + 12 * pr.scan_a, poses + 12 * pr.scan_b, model_kind, n_terms, w, e, pr.idxA, pr.idxB, pr.m,
                           b - pr.row0, partials + (int64_t)b * kIcpAcc, lds);
}

// out = { loss, dw[P], dexponent[P], d[R|t][S,12] } += weight * (sums of every pair), pair after pair in ONE launch.  Block 0:
// the loss (both directed sums), blocks 1 .. 2 MAX: the model gradients, the last twelve: entry j of BOTH poses of every pair --
// one block per destination, so the additions to a scan's slot happen in the order consecutive launches would make them.
struct IcpSeqReduce {
  double weight[kIcpSeqPairs];
  int32_t scan_a[kIcpSeqPairs], scan_b[kIcpSeqPairs], row0[kIcpSeqPairs], rows[kIcpSeqPairs];
  int n_pairs;
};
__global__ __launch_bounds__(kBlock) void p2plane_reduce_all_kernel(const double* __restrict__ partials, IcpSeqReduce rq, int n_terms,
                                                                    double* __restrict__ out, int first, int n_scans) {
  __shared__ double lds[2 * (kBlock / kWave)];
  const int a = blockIdx.x;
  const int pw = 2, pe = 2 + DC_MAX_MODEL_TERMS, pa = 2 + 2 * DC_MAX_MODEL_TERMS, pb = pa + 12;
  const bool pose = a >= 1 + 2 * DC_MAX_MODEL_TERMS;
  const int j = a - (1 + 2 * DC_MAX_MODEL_TERMS);                       // pose entry
  const int col0 = a == 0 ? 0 : (pose ? pa + j : a + 1), col1 = a == 0 ? 1 : (pose ? pb + j : -1);
  if (!pose && a >= 1) {                                                // model gradient slots beyond n_terms: nothing to do
    const int k = (a - 1) % DC_MAX_MODEL_TERMS;
    if (k >= n_terms) return;
  }
  // the first launch of a sequence starts its slots from zero (instead of a memset in front: two launches fewer in an iteration
  // that is a chain of short ones).  A pose block keeps its entry of every scan in LDS while the pairs are added and writes all of
  // them once at the end; the other blocks have one destination, kept in a register: the additions happen in the order they
  // always did, but none of them is a dependent read-modify-write of global memory any more (C4 train() iteration 48 -> 45.8 us)
  __shared__ double s_dst[kIcpDstScans];
  const bool dst_lds = pose && n_scans <= kIcpDstScans;
  if (dst_lds)
    for (int sc = threadIdx.x; sc < n_scans; sc += kBlock) s_dst[sc] = first ? 0.0 : out[1 + 2 * n_terms + 12 * sc + j];
  else if (pose && first && threadIdx.x == 0)
    for (int sc = 0; sc < n_scans; ++sc) out[1 + 2 * n_terms + 12 * sc + j] = 0.0;
  // every pair's rows are requested before anything is added (a pair after the other cost a round trip to the fabric and two
  // barriers each: 14 us for nine pairs); the sums of a pair are formed in the same order as before and added pair after pair
  __shared__ double s_pair[kBlock / kWave][2 * kIcpSeqPairs];
  double v[kIcpSeqPairs][2];
#pragma unroll
  for (int p = 0; p < kIcpSeqPairs; ++p) {
    v[p][0] = v[p][1] = 0.0;
    if (p < rq.n_pairs) {
      const double* rows = partials + (int64_t)rq.row0[p] * kIcpAcc;
      for (int r = threadIdx.x; r < rq.rows[p]; r += kBlock) {
        v[p][0] += rows[(int64_t)r * kIcpAcc + col0];
        if (col1 >= 0) v[p][1] += rows[(int64_t)r * kIcpAcc + col1];
      }
    }
  }
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
  for (int p = 0; p < kIcpSeqPairs; ++p) {
    if (p < rq.n_pairs) {                                                // (uniform)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const double t = wave_sum(v[p][c]);                              // (as block_sum: the same order of additions as before)
        if (lane == 0) s_pair[wave][2 * p + c] = t;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int slot = a == 0 ? 0 : 1 + (a - 1 < DC_MAX_MODEL_TERMS ? 0 : n_terms) + (a - 1) % DC_MAX_MODEL_TERMS;     // (not a pose block)
    double x = (pose || first) ? 0.0 : out[slot];
    for (int p = 0; p < rq.n_pairs; ++p) {
      double t0 = 0.0, t1 = 0.0;
      for (int wv = 0; wv < kBlock / kWave; ++wv) { t0 += s_pair[wv][2 * p]; t1 += s_pair[wv][2 * p + 1]; }
      const double wt = rq.weight[p];
      if (a == 0) x += wt * (t0 + t1);
      else if (!pose) x += wt * t0;
      else if (dst_lds) {
        s_dst[rq.scan_a[p]] += wt * t0;
        s_dst[rq.scan_b[p]] += wt * t1;
      } else {
        out[1 + 2 * n_terms + 12 * rq.scan_a[p] + j] += wt * t0;
        out[1 + 2 * n_terms + 12 * rq.scan_b[p] + j] += wt * t1;
      }
    }
    if (!pose) out[slot] = x;
  }
  if (dst_lds) {
    __syncthreads();
    for (int sc = threadIdx.x; sc < n_scans; sc += kBlock) out[1 + 2 * n_terms + 12 * sc + j] = s_dst[sc];
  }
  (void)lds;
  (void)pw; (void)pe;
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
  if (model_kind < DC_MODEL_NONE || model_kind > DC_MODEL_LAST) return DC_ERR_ARG;
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
  if (model_kind < DC_MODEL_NONE || model_kind > DC_MODEL_LAST) return DC_ERR_ARG;
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
  hipError_t err = hipSuccess;
  bool launched = false;
  if (err != hipSuccess) return (int)err;
  // chunks of up to kIcpSeqPairs pairs: one pair launch + one reduction launch per chunk (the workspace holds the rows of the
  // largest chunk: dc_p2plane_sequence_partial_count)
  int p = 0;
  while (p < n_pairs) {
    IcpSeqArgs args{};
    IcpSeqReduce rq{};
    int64_t row0 = 0;
    int np = 0;
    for (; p < n_pairs && np < kIcpSeqPairs; ++p) {
      const dcIcpPair& q = pairs[p];
      if (q.m == 0) continue;
      const dcIcpScan &a = scans[q.scan_a], &b = scans[q.scan_b];
      IcpSeqPair& t = args.pairs[np];
      t.A = ScanView{a.vps, a.dirs, a.depth, a.inc, a.lmask, plane ? a.normals : nullptr};
      t.B = ScanView{b.vps, b.dirs, b.depth, b.inc, b.lmask, plane ? b.normals : nullptr};
      t.idxA = q.idx_a; t.idxB = q.idx_b; t.m = q.m; t.weight = q.weight; t.scan_a = q.scan_a; t.scan_b = q.scan_b;
      const int64_t rows = (q.m + kBlock - 1) / kBlock;
      if (row0 + rows > INT32_MAX) return DC_ERR_UNSUPPORTED;
      t.row0 = (int32_t)row0; t.rows = (int32_t)rows;
      rq.weight[np] = q.weight; rq.scan_a[np] = q.scan_a; rq.scan_b[np] = q.scan_b; rq.row0[np] = t.row0; rq.rows[np] = t.rows;
      row0 += rows;
      ++np;
    }
    if (np == 0) continue;
    args.n_pairs = rq.n_pairs = np;
    const dim3 grid((unsigned)row0), block(kBlock);
#define ICP_SEQ(T, PLANE) hipLaunchKernelGGL((p2plane_seq_kernel<T, PLANE>), grid, block, 0, stream, args, poses, model_kind, n_terms, w, e, partials_ws)
    if (dtype == DC_F32) { if (plane) ICP_SEQ(float, true); else ICP_SEQ(float, false); }
    else { if (plane) ICP_SEQ(double, true); else ICP_SEQ(double, false); }
#undef ICP_SEQ
    hipLaunchKernelGGL(p2plane_reduce_all_kernel, dim3(1 + 2 * DC_MAX_MODEL_TERMS + 12), dim3(kBlock), 0, stream, partials_ws, rq, n_terms, out,
                       launched ? 0 : 1, n_scans);
    launched = true;
  }
  if (!launched) {                                                      // no correspondences at all: the sums are zero
    err = hipMemsetAsync(out, 0, (size_t)(1 + 2 * n_terms + 12 * n_scans) * sizeof(double), stream);
    if (err != hipSuccess) return (int)err;
  }
  err = hipGetLastError();
  return err == hipSuccess ? DC_OK : (int)err;
}

int64_t dc_p2plane_sequence_partial_count(const dcIcpPair* pairs, int n_pairs) {
  int64_t best = 1, rows = 0;
  int np = 0;
  for (int p = 0; pairs && p < n_pairs; ++p) {
    if (pairs[p].m <= 0) continue;
    if (np == kIcpSeqPairs) { rows = 0; np = 0; }
    rows += (pairs[p].m + kBlock - 1) / kBlock;
    ++np;
    best = rows > best ? rows : best;
  }
  return (best + kIcpSeqPairs) * kIcpAcc;      // (+ one row per pair: dc_icp_sequence_step's second level)
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

// ================================================================================================
// Pose-correction chain of the joint model + pose optimisation (eval.create_corrected_poses eval.py:68-82):
//   T_s = T0_s * [ Exp(axis-angle_s) | xyz_s ]      (transform.xyz_axis_angle_to_matrix transform.py:68-78, rotation by
//   pytorch3d's axis_angle_to_matrix: axis-angle -> quaternion with the small-angle series -> matrix)
// and its backward dL/d(xyz, axis-angle) from dL/dT.  As torch ops this is ~50 tiny kernels forward and as many backward
// per iteration (the largest host-side cost of a config-4 training iteration); here it is one launch each way.
// One 6-vector per pose (PoseCorrection.pose) or one shared by all poses (common / sequence: its gradient is the sum).
// The backward is the adjoint of the forward operation by operation, including the conventions autograd applies to it:
// the norm's subgradient at zero is zero, and of the two branches of the small-angle switch only the taken one
// receives a gradient.
// ================================================================================================
namespace dc {

struct PoseChain {
  double q[4], s, k, theta, sh, ch;      // sh, ch: sin / cos of theta / 2 (one sincos; the adjoint needs them again)
  bool small;
};

__device__ __forceinline__ void pose_chain_fwd(const double* d6, double* R, PoseChain& c) {
  const double a0 = d6[3], a1 = d6[4], a2 = d6[5];
  c.theta = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
  c.small = fabs(c.theta) < 1e-6;
  sincos(0.5 * c.theta, &c.sh, &c.ch);
  c.k = c.small ? 0.5 - c.theta * c.theta / 48.0 : c.sh / c.theta;
  c.q[0] = c.ch; c.q[1] = a0 * c.k; c.q[2] = a1 * c.k; c.q[3] = a2 * c.k;
  const double r = c.q[0], i = c.q[1], j = c.q[2], k = c.q[3];
  c.s = 2.0 / (r * r + i * i + j * j + k * k);
  const double s = c.s;
  R[0] = 1.0 - s * (j * j + k * k); R[1] = s * (i * j - k * r); R[2] = s * (i * k + j * r);
  R[3] = s * (i * j + k * r); R[4] = 1.0 - s * (i * i + k * k); R[5] = s * (j * k - i * r);
  R[6] = s * (i * k - j * r); R[7] = s * (j * k + i * r); R[8] = 1.0 - s * (i * i + j * j);
}

// gR = dL/dR (row major 3x3) -> g6[3..5] += dL/d(axis-angle)
__device__ __forceinline__ void pose_chain_bwd(const double* d6, const PoseChain& c, const double* gR, double* g_aa) {
  const double r = c.q[0], i = c.q[1], j = c.q[2], k = c.q[3], s = c.s;
  double ds = 0.0, dr = 0.0, di = 0.0, dj = 0.0, dk = 0.0;
  ds += -(j * j + k * k) * gR[0]; dj += -2.0 * s * j * gR[0]; dk += -2.0 * s * k * gR[0];
  ds += (i * j - k * r) * gR[1]; di += s * j * gR[1]; dj += s * i * gR[1]; dk += -s * r * gR[1]; dr += -s * k * gR[1];
  ds += (i * k + j * r) * gR[2]; di += s * k * gR[2]; dk += s * i * gR[2]; dj += s * r * gR[2]; dr += s * j * gR[2];
  ds += (i * j + k * r) * gR[3]; di += s * j * gR[3]; dj += s * i * gR[3]; dk += s * r * gR[3]; dr += s * k * gR[3];
  ds += -(i * i + k * k) * gR[4]; di += -2.0 * s * i * gR[4]; dk += -2.0 * s * k * gR[4];
  ds += (j * k - i * r) * gR[5]; dj += s * k * gR[5]; dk += s * j * gR[5]; di += -s * r * gR[5]; dr += -s * i * gR[5];
  ds += (i * k - j * r) * gR[6]; di += s * k * gR[6]; dk += s * i * gR[6]; dj += -s * r * gR[6]; dr += -s * j * gR[6];
  ds += (j * k + i * r) * gR[7]; dj += s * k * gR[7]; dk += s * j * gR[7]; di += s * r * gR[7]; dr += s * i * gR[7];
  ds += -(i * i + j * j) * gR[8]; di += -2.0 * s * i * gR[8]; dj += -2.0 * s * j * gR[8];
  const double dn = -0.5 * s * s * ds;                       // s = 2 / n
  dr += 2.0 * r * dn; di += 2.0 * i * dn; dj += 2.0 * j * dn; dk += 2.0 * k * dn;
  const double a0 = d6[3], a1 = d6[4], a2 = d6[5];
  const double dkk = a0 * di + a1 * dj + a2 * dk;            // through q_vec = a * k
  double dtheta = -0.5 * c.sh * dr;                          // through q0 = cos(theta / 2)
  dtheta += c.small ? -(c.theta / 24.0) * dkk
                    : (0.5 * c.ch / c.theta - c.sh / (c.theta * c.theta)) * dkk;
  const double inv = c.theta > 0.0 ? 1.0 / c.theta : 0.0;     // d|a|/da = a / |a|, zero at the origin
  g_aa[0] = c.k * di + a0 * inv * dtheta;
  g_aa[1] = c.k * dj + a1 * inv * dtheta;
  g_aa[2] = c.k * dk + a2 * inv * dtheta;
}

// One block; thread t handles poses t, t + 256, ...  Forward (grad_T == nullptr): T_out.  Backward: grad_delta.
__global__ __launch_bounds__(kBlock) void pose_correct_kernel(const double* __restrict__ T0, const double* __restrict__ delta,
                                                              int n_poses, int n_delta, double* __restrict__ T_out,
                                                              const double* __restrict__ grad_T, double* __restrict__ grad_delta) {
  __shared__ double lds[(kBlock / kWave) * 6];
  double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int p = threadIdx.x; p < n_poses; p += kBlock) {
    const double* d6 = delta + (n_delta == 1 ? 0 : (int64_t)p * 6);
    const double* A = T0 + (int64_t)p * 16;
    double R[9];
    PoseChain c;
    pose_chain_fwd(d6, R, c);
    if (!grad_T) {
      double* o = T_out + (int64_t)p * 16;
#pragma unroll
      for (int a = 0; a < 4; ++a) {                            // every row of T0, also the last: T = T0 X in full
#pragma unroll
        for (int b = 0; b < 3; ++b) o[a * 4 + b] = A[a * 4] * R[b] + A[a * 4 + 1] * R[3 + b] + A[a * 4 + 2] * R[6 + b];
        o[a * 4 + 3] = A[a * 4] * d6[0] + A[a * 4 + 1] * d6[1] + A[a * 4 + 2] * d6[2] + A[a * 4 + 3];
      }
    } else {
      const double* G = grad_T + (int64_t)p * 16;
      double gR[9], g6[6];
#pragma unroll
      for (int a = 0; a < 3; ++a) {                            // dL/dX = T0^T G restricted to X's free entries
#pragma unroll
        for (int b = 0; b < 3; ++b) gR[a * 3 + b] = A[a] * G[b] + A[4 + a] * G[4 + b] + A[8 + a] * G[8 + b] + A[12 + a] * G[12 + b];
        g6[a] = A[a] * G[3] + A[4 + a] * G[7] + A[8 + a] * G[11] + A[12 + a] * G[15];
      }
      pose_chain_bwd(d6, c, gR, g6 + 3);
      if (n_delta == 1) {
#pragma unroll
        for (int q = 0; q < 6; ++q) acc[q] += g6[q];
      } else {
#pragma unroll
        for (int q = 0; q < 6; ++q) grad_delta[(int64_t)p * 6 + q] = g6[q];
      }
    }
  }
  if (grad_T && n_delta == 1) {
    block_sum<6>(acc, lds);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < 6; ++q) grad_delta[q] = acc[q];
    }
  }
}


// ---- one training iteration with pose corrections, finished in one launch (train.py:300-322 after the loss: backward through
// eval.create_corrected_poses, the first pose kept fixed, optimizer.step() on the model weights and on the pose corrections) --------
// `sums` = {sum loss, count, dL/dw [P], dL/de [P], dL/d[R|t] of every pose [12 S]} of dc_sequence_eval with the corrected poses
// `T_used`.  The mean loss sum / count is what train() back-propagates: every gradient is scaled by 1 / count.  One block; thread p
// takes pose p: the adjoint of T = T0 X(delta) (pose_chain_bwd), Adam on delta_p -- torch.optim.Adam's single-tensor update,
// step t = *step + 1 -- and the corrected pose of the NEXT iteration from the updated correction; threads 0 .. P - 1 take the
// weights.  `record` (optional) <- {sums, the weights, corrections and corrected poses [12 S] this iteration USED, then the
// caller's `rec_extra` doubles}: what a checkpoint of the iteration holds.
struct PoseTrainArgs {
  const double* sums;
  int n_sums, count_index, grad_w_off, grad_T_off;      // layout of `sums`; count_index < 0: the gradients are those of the loss itself
  const double* totals;                                 // or nullptr: {loss, divisor, dL/dw [P]} over ALL sequences of the loss (dc_pose_train_combine)
  int n_terms, n_scans, n_deltas, zero_first;
  double *w, *w_m, *w_v;                 // w == nullptr: the model is not optimised (validation sequences)
  const double* T0;                      // [S, 16]
  double *delta, *d_m, *d_v;             // [n_deltas, 6]
  int64_t* step;
  double lr_w, lr_d, b1, b2, eps;
  const double* T_used;                  // [S, 16]
  double* record;                        // or nullptr: a ring of ring_rows records; this iteration's is row *step mod ring_rows
  int ring_rows;
  double* T_next;                        // [S, 16]
  double* P12_next;                      // [S, 12]
  const double* rec_extra;               // or nullptr: n_rec_extra doubles appended to the record (the joint sums of every loss of the
  int n_rec_extra;                       // iteration: a rank's log then holds the training AND the validation loss of all ranks)
};

// {loss, divisor, dL/dw} of a loss over several sequences (eval.py:85-112 sums the sequences' sums and counts; icp_loss averages the
// sequences' losses, loss.py:403): layout 0: {sum of sums, sum of counts, ...}; layout 1: {sum of losses, number of sequences, ...}.
constexpr int kTrainSeqs = 16;
struct PoseCombineArgs {
  const double* outs[2 * kTrainSeqs];            // group 0 (n_seq of them), then group 1 (n_seq_b)
  int n_seq, n_seq_b, layout, n_terms;
  double* totals;                                // [2 + P], or [2][2 + P] with two groups (block g takes group g)
};
__global__ void pose_train_combine_kernel(PoseCombineArgs a) {
  const int q = threadIdx.x;                     // 0: loss, 1: divisor, 2 + k: dL/dw_k
  if (q >= 2 + a.n_terms) return;
  const int g = blockIdx.x, first = g == 0 ? 0 : a.n_seq, count = g == 0 ? a.n_seq : a.n_seq_b;
  const int head = a.layout == 0 ? 2 : 1;
  double s = 0.0;
  for (int i = first; i < first + count; ++i) {  // fixed order
    if (q == 0) s += a.outs[i][0];
    else if (q == 1) s += a.layout == 0 ? a.outs[i][1] : 1.0;
    else s += a.outs[i][head + (q - 2)];
  }
  a.totals[g * (2 + a.n_terms) + q] = s;
}

__device__ __forceinline__ double adam_one(double p0, double& m, double& v, double g, double lr, double b1, double b2, double eps,
                                           double bias1, double bias2_sqrt) {
  m = m + (g - m) * (1.0 - b1);                         // exp_avg.lerp_(grad, 1 - beta1)
  v = v * b2 + (1.0 - b2) * g * g;                      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  const double denom = sqrt(v) / bias2_sqrt + eps;
  return p0 + (-(lr / bias1)) * (m / denom);
}

// (one block of kBlock threads; lds: (kBlock / kWave) * 6 doubles)
__device__ __forceinline__ void pose_train_finish_block(const PoseTrainArgs& a, double* lds) {
  const int S = a.n_scans, P = a.n_terms, tid = threadIdx.x;
  const int n_sums = a.n_sums;
  // Everything a thread needs is requested up front -- its pose's correction, reference pose, gradient rows and Adam moments, the
  // step counter, the weights' state -- and held in registers: the block is ONE wavefront's worth of work at the end of an
  // iteration, and every dependent round trip to memory (there were six: the counter for the record's row, the record, the inputs,
  // the moments, the corrections after the update, T_next read back for P12_next) is a microsecond of the loop's critical path.
  const int64_t step0 = *a.step;
  const bool per_pose = a.n_deltas != 1;
  const bool mine = tid < S;                                   // (S <= kBlock poses take the register path; more: the loops below)
  double dl[6] = {0, 0, 0, 0, 0, 0}, Av[16], Gv[12], mv[6], vv[6];
  const double* gT = a.sums + a.grad_T_off;
  if (mine) {
    const double* d6 = a.delta + (per_pose ? (int64_t)tid * 6 : 0);
#pragma unroll
    for (int q = 0; q < 6; ++q) dl[q] = d6[q];
#pragma unroll
    for (int q = 0; q < 16; ++q) Av[q] = a.T0[(int64_t)tid * 16 + q];
#pragma unroll
    for (int q = 0; q < 12; ++q) Gv[q] = gT[(int64_t)tid * 12 + q];
    if (per_pose) {
#pragma unroll
      for (int q = 0; q < 6; ++q) { mv[q] = a.d_m[(int64_t)tid * 6 + q]; vv[q] = a.d_v[(int64_t)tid * 6 + q]; }
    }
  }
  const bool step_w = a.w && tid < P && a.lr_w != 0.0;         // (lr_w = 0: weights that are recorded, not optimised)
  double w0 = 0.0, wm = 0.0, wv = 0.0, gw = 0.0;
  if (step_w) { w0 = a.w[tid]; wm = a.w_m[tid]; wv = a.w_v[tid]; gw = a.totals ? a.totals[2 + tid] : a.sums[a.grad_w_off + tid]; }
  const double divisor = a.totals ? a.totals[1] : (a.count_index >= 0 ? a.sums[a.count_index] : 1.0);
  if (a.record) {
    double* r = a.record + (step0 % a.ring_rows) * (int64_t)(n_sums + P + 6 * a.n_deltas + 12 * S + a.n_rec_extra);
    for (int q = tid; q < n_sums; q += kBlock) r[q] = a.sums[q];
    r += n_sums;
    for (int q = tid; q < P; q += kBlock) r[q] = a.w ? a.w[q] : 0.0;
    r += P;
    for (int q = tid; q < 6 * a.n_deltas; q += kBlock) r[q] = a.delta[q];
    r += 6 * a.n_deltas;
    for (int q = tid; q < 12 * S; q += kBlock) r[q] = a.T_used[(q / 12) * 16 + q % 12];
    r += 12 * S;
    for (int q = tid; q < a.n_rec_extra; q += kBlock) r[q] = a.rec_extra[q];
  }
  __syncthreads();                                             // the record holds what this iteration USED: copied before any update
  const double t = (double)(step0 + 1);
  const double bias1 = 1.0 - pow(a.b1, t), bias2_sqrt = sqrt(1.0 - pow(a.b2, t));
  const double gscale = (a.totals || a.count_index >= 0) ? 1.0 / divisor : 1.0;
  double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int p = tid; p < S; p += kBlock) {
    if (p != tid) {                                            // (a pose beyond the first kBlock: fetched here)
      const double* d6 = a.delta + (per_pose ? (int64_t)p * 6 : 0);
#pragma unroll
      for (int q = 0; q < 6; ++q) dl[q] = d6[q];
#pragma unroll
      for (int q = 0; q < 16; ++q) Av[q] = a.T0[(int64_t)p * 16 + q];
#pragma unroll
      for (int q = 0; q < 12; ++q) Gv[q] = gT[(int64_t)p * 12 + q];
      if (per_pose) {
#pragma unroll
        for (int q = 0; q < 6; ++q) { mv[q] = a.d_m[(int64_t)p * 6 + q]; vv[q] = a.d_v[(int64_t)p * 6 + q]; }
      }
    }
    const double* A = Av;
    const double* G = Gv;                                      // rows 0..2 of dL/dT (its last row is zero)
    double R[9], gR[9], g6[6];
    PoseChain c;
    pose_chain_fwd(dl, R, c);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int b = 0; b < 3; ++b) gR[i * 3 + b] = (A[i] * G[b] + A[4 + i] * G[4 + b] + A[8 + i] * G[8 + b]) * gscale;
      g6[i] = (A[i] * G[3] + A[4 + i] * G[7] + A[8 + i] * G[11]) * gscale;
    }
    pose_chain_bwd(dl, c, gR, g6 + 3);
    if (a.zero_first && p == 0) {
#pragma unroll
      for (int q = 0; q < 6; ++q) g6[q] = 0.0;
    }
    if (!per_pose) {
#pragma unroll
      for (int q = 0; q < 6; ++q) acc[q] += g6[q];
    } else {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int64_t i = (int64_t)p * 6 + q;
        dl[q] = adam_one(dl[q], mv[q], vv[q], g6[q], a.lr_d, a.b1, a.b2, a.eps, bias1, bias2_sqrt);
        a.delta[i] = dl[q];
        a.d_m[i] = mv[q]; a.d_v[i] = vv[q];
      }
    }
  }
  if (!per_pose) {                                             // one correction for the whole sequence: its gradient is the sum
    block_sum<6>(acc, lds);                                      // (the totals are thread 0's)
    if (tid == 0) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        double m = a.d_m[q], v = a.d_v[q];
        a.delta[q] = adam_one(a.delta[q], m, v, acc[q], a.lr_d, a.b1, a.b2, a.eps, bias1, bias2_sqrt);
        a.d_m[q] = m; a.d_v[q] = v;
      }
    }
  }
  if (step_w) {
    a.w[tid] = adam_one(w0, wm, wv, gw * gscale, a.lr_w, a.b1, a.b2, a.eps, bias1, bias2_sqrt);
    a.w_m[tid] = wm; a.w_v[tid] = wv;
  }
  __syncthreads();                                             // the corrections are updated; the step counter and T_used have been read
  if (tid == 0) *a.step = (int64_t)t;
  for (int p = tid; p < S; p += kBlock) {
    // (the registers still hold this pose's updated correction and reference pose when every thread had one pose at most)
    if (!per_pose || S > kBlock) {                             // the shared correction, or more poses than threads: as stored
      const double* d6 = a.delta + (per_pose ? (int64_t)p * 6 : 0);
#pragma unroll
      for (int q = 0; q < 6; ++q) dl[q] = d6[q];
    }
    if (S > kBlock) {
#pragma unroll
      for (int q = 0; q < 16; ++q) Av[q] = a.T0[(int64_t)p * 16 + q];
    }
    const double* A = Av;
    double R[9], o[16];
    PoseChain c;
    pose_chain_fwd(dl, R, c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int b = 0; b < 3; ++b) o[i * 4 + b] = A[i * 4] * R[b] + A[i * 4 + 1] * R[3 + b] + A[i * 4 + 2] * R[6 + b];
      o[i * 4 + 3] = A[i * 4] * dl[0] + A[i * 4 + 1] * dl[1] + A[i * 4 + 2] * dl[2] + A[i * 4 + 3];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) a.T_next[(int64_t)p * 16 + q] = o[q];
#pragma unroll
    for (int q = 0; q < 12; ++q) a.P12_next[(int64_t)p * 12 + q] = o[q];
  }
}

__global__ __launch_bounds__(kBlock) void pose_train_finish_kernel(PoseTrainArgs a) {
  __shared__ double lds[(kBlock / kWave) * 6];
  pose_train_finish_block(a, lds);
}

// ---- a whole ICP iteration in ONE launch (round 5; train.py:300-312 with icp_loss, loss.py:373-488) ----------------------------
// p2plane_seq_kernel, p2plane_reduce_all_kernel and pose_train_finish_kernel were three dependent launches of 17 + 16 + 12 us for ~500
// blocks of work: launch and dependency latency, not arithmetic.  Here every block writes its row, makes it visible and takes a
// ticket; the block that draws the LAST one sums the rows (pair by pair, fixed order: each thread owns one of the 42 sums of a
// sixth of the rows, the sixths joined in order), writes `out` as dc_p2plane_sequence does, and -- when the iteration has nothing
// to wait for (one sequence in the loss, one rank) -- goes straight on with pose_train_finish_block: adjoint of the pose chain,
// both Adam updates, next poses, record.  A few hundred tickets cost nothing here (the grid is resident at once; the C2 step
// kernel's 7 800 blocks would queue on them).
constexpr int kIcpPhases = 6;                 // kBlock / kIcpAcc row phases of a pair's sum
static_assert(kIcpPhases * kIcpAcc <= kBlock, "one thread per (phase, sum)");
constexpr int kIcpInFlight = 16;              // rows a thread requests before it adds any (a pair of ~15 k correspondences: 59 rows, 10 per phase)
template <typename T, bool PLANE>
__global__ __launch_bounds__(kBlock) void p2plane_seq_fused_kernel(IcpSeqArgs args, const double* __restrict__ poses, int model_kind, int n_terms,
                                                                   const double* __restrict__ w, const double* __restrict__ e,
                                                                   double* __restrict__ partials, IcpSeqReduce rq, double* __restrict__ out,
                                                                   int n_scans, int32_t* __restrict__ ticket, PoseTrainArgs fin, int with_finish) {
  __shared__ double lds[kWavesPerBlock * kIcpGroups * 8];
  __shared__ double s_part[kIcpPhases][kIcpAcc];
  __shared__ double s_pair[kIcpSeqPairs][kIcpAcc];
  constexpr int kOutLds = 1 + 2 * DC_MAX_MODEL_TERMS + 12 * 32;      // `out` of up to 32 scans, kept for the finishing step
  __shared__ double s_out[kOutLds];
  __shared__ int s_last;
  const int b = blockIdx.x, tid = threadIdx.x;
  int p = 0;
#pragma unroll 1
  for (int q = 1; q < args.n_pairs; ++q)
    if (b >= args.pairs[q].row0) p = q;
  {
    const IcpSeqPair& pr = args.pairs[p];
    icp_pair_block<T, PLANE>(pr.A, pr.B, poses + 12 * pr.scan_a, poses + 12 * pr.scan_b, model_kind, n_terms, w, e, pr.idxA, pr.idxB, pr.m,
                             b - pr.row0, partials + (int64_t)b * kIcpAcc, lds);
  }
  // Two levels of tickets: the last block of a PAIR sums that pair's rows (one round of loads: every thread owns one of the 42 sums
  // of a sixth of the rows), the last pair to finish joins the pairs and goes on.  A row is written back to memory before its
  // ticket says so -- a RELEASE only: an acquire here would invalidate this XCD's L2 under the blocks still gathering scan points
  // through it (530 of them: the launch took 54 us).
  double* pair_rows = partials + (int64_t)gridDim.x * kIcpAcc;          // [n_pairs][kIcpAcc] behind the block rows
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (tid == 0) {
    const int t = __hip_atomic_fetch_add(ticket + 1 + p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == rq.rows[p] - 1;
    if (s_last) __hip_atomic_store(ticket + 1 + p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
  }
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // the rows of the pair's other blocks, wherever they ran
  const int c = tid % kIcpAcc, f = tid / kIcpAcc;
  if (f < kIcpPhases) {
    const double* rows = partials + (int64_t)rq.row0[p] * kIcpAcc;
    const int n_r = rq.rows[p];
    double v = 0.0;
    for (int r0 = f; r0 < n_r; r0 += kIcpPhases * kIcpInFlight) {
      double x[kIcpInFlight];
#pragma unroll
      for (int u = 0; u < kIcpInFlight; ++u) {
        const int r = r0 + u * kIcpPhases;
        x[u] = r < n_r ? rows[(int64_t)r * kIcpAcc + c] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < kIcpInFlight; ++u) v += x[u];
    }
    s_part[f][c] = v;
  }
  __syncthreads();
  if (tid < kIcpAcc) {
    double t = 0.0;
#pragma unroll
    for (int ff = 0; ff < kIcpPhases; ++ff) t += s_part[ff][tid];
    pair_rows[p * kIcpAcc + tid] = t * rq.weight[p];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (tid == 0) {
    const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = t == rq.n_pairs - 1;
    if (s_last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  for (int i = tid; i < rq.n_pairs * kIcpAcc; i += kBlock) s_pair[i / kIcpAcc][i % kIcpAcc] = pair_rows[i];
  __syncthreads();
  // out = { loss, dw[P], dexponent[P], d[R|t][S,12] }: one thread per destination, the pairs added in their order
  const int pw = 2, pe = 2 + DC_MAX_MODEL_TERMS, pa = 2 + 2 * DC_MAX_MODEL_TERMS, pb = pa + 12;
  const int n_out = 1 + 2 * n_terms + 12 * n_scans;
  for (int dst = tid; dst < n_out; dst += kBlock) {
    double x = 0.0;
    if (dst == 0) {
      for (int q = 0; q < rq.n_pairs; ++q) x += s_pair[q][0] + s_pair[q][1];
    } else if (dst < 1 + n_terms) {
      for (int q = 0; q < rq.n_pairs; ++q) x += s_pair[q][pw + dst - 1];
    } else if (dst < 1 + 2 * n_terms) {
      for (int q = 0; q < rq.n_pairs; ++q) x += s_pair[q][pe + dst - 1 - n_terms];
    } else {
      const int sc = (dst - 1 - 2 * n_terms) / 12, j = (dst - 1 - 2 * n_terms) % 12;
      for (int q = 0; q < rq.n_pairs; ++q) {
        if (rq.scan_a[q] == sc) x += s_pair[q][pa + j];
        if (rq.scan_b[q] == sc) x += s_pair[q][pb + j];
      }
    }
    out[dst] = x;
    if (n_out <= kOutLds) s_out[dst] = x;
  }
  if (!with_finish) return;
  __threadfence_block();
  __syncthreads();                              // `out` is what the finishing step reads as its sums -- from LDS where it fits: reading
  if (n_out <= kOutLds) fin.sums = s_out;       // back what this block has just stored is a round trip to memory on the critical path
  pose_train_finish_block(fin, lds);
}

}  // namespace dc

extern "C" {

// dc_p2plane_sequence / dc_p2point_sequence as ONE launch, optionally followed IN THE SAME LAUNCH by dc_pose_train_finish (fin != NULL):
// for sequences of at most kIcpSeqPairs pairs with at least one correspondence; DC_ERR_UNSUPPORTED otherwise (the caller then issues the
// separate calls).  ticket: device int32, zero before the first call (the launch leaves it zero).
int dc_icp_sequence_step(int plane, const dcIcpScan* scans, int n_scans, const dcIcpPair* pairs, int n_pairs, int dtype,
                         const double* poses, int model_kind, int n_terms, const double* w, const double* e, double* partials_ws,
                         double* out, int32_t* ticket, const dcPoseTrainStep* fin, hipStream_t stream) {
  if (n_scans < 1 || n_pairs < 1 || !scans || !pairs || !out || !ticket || !poses || !partials_ws) return DC_ERR_ARG;
  if (dtype != DC_F32 && dtype != DC_F64) return DC_ERR_DTYPE;
  if (model_kind < DC_MODEL_NONE || model_kind > DC_MODEL_LAST) return DC_ERR_ARG;
  if (model_kind != DC_MODEL_NONE && (n_terms < 1 || n_terms > DC_MAX_MODEL_TERMS || !w || !e)) return DC_ERR_ARG;
  const int nt_model = model_kind == DC_MODEL_NONE ? 0 : n_terms;
  dc::IcpSeqArgs args{};
  dc::IcpSeqReduce rq{};
  int64_t row0 = 0;
  int np = 0;
  for (int p = 0; p < n_pairs; ++p) {
    const dcIcpPair& q = pairs[p];
    if (q.scan_a < 0 || q.scan_a >= n_scans || q.scan_b < 0 || q.scan_b >= n_scans || q.scan_a == q.scan_b) return DC_ERR_ARG;
    if (q.m < 0 || (q.m > 0 && (!q.idx_a || !q.idx_b))) return DC_ERR_ARG;
    if (q.m == 0) continue;
    if (np == dc::kIcpSeqPairs) return DC_ERR_UNSUPPORTED;
    const dcIcpScan &a = scans[q.scan_a], &b = scans[q.scan_b];
    for (const dcIcpScan* c : {&a, &b})
      if (!c->dirs || !c->depth || (plane && !c->normals) || (model_kind != DC_MODEL_NONE && !c->inc)) return DC_ERR_ARG;
    dc::IcpSeqPair& t = args.pairs[np];
    t.A = dc::ScanView{a.vps, a.dirs, a.depth, a.inc, a.lmask, plane ? a.normals : nullptr};
    t.B = dc::ScanView{b.vps, b.dirs, b.depth, b.inc, b.lmask, plane ? b.normals : nullptr};
    t.idxA = q.idx_a; t.idxB = q.idx_b; t.m = q.m; t.weight = q.weight; t.scan_a = q.scan_a; t.scan_b = q.scan_b;
    const int64_t rows = (q.m + dc::kBlock - 1) / dc::kBlock;
    if (row0 + rows > INT32_MAX) return DC_ERR_UNSUPPORTED;
    t.row0 = (int32_t)row0; t.rows = (int32_t)rows;
    rq.weight[np] = q.weight; rq.scan_a[np] = q.scan_a; rq.scan_b[np] = q.scan_b; rq.row0[np] = t.row0; rq.rows[np] = t.rows;
    row0 += rows;
    ++np;
  }
  if (np == 0) return DC_ERR_UNSUPPORTED;
  args.n_pairs = rq.n_pairs = np;
  dc::PoseTrainArgs fa{};
  if (fin) {
    if (nt_model < 1 || (fin->n_deltas != 1 && fin->n_deltas != n_scans) || !fin->poses0 || !fin->deltas || !fin->d_m || !fin->d_v || !fin->step ||
        !fin->poses_used || !fin->poses_next || !fin->poses12_next || (fin->w && (!fin->w_m || !fin->w_v)) || (fin->record && fin->ring_rows < 1) ||
        fin->n_record_extra < 0 || (fin->n_record_extra > 0 && !fin->record_extra))
      return DC_ERR_ARG;
    if (!(fin->lr_w >= 0.0) || !(fin->lr_d >= 0.0) || !(fin->beta1 >= 0.0 && fin->beta1 < 1.0) || !(fin->beta2 >= 0.0 && fin->beta2 < 1.0) ||
        !(fin->eps >= 0.0))
      return DC_ERR_ARG;
    fa = dc::PoseTrainArgs{out, 1 + 2 * nt_model + 12 * n_scans, -1, 1, 1 + 2 * nt_model, nullptr, nt_model, n_scans, fin->n_deltas, fin->zero_first,
                       fin->w, fin->w_m, fin->w_v, fin->poses0, fin->deltas, fin->d_m, fin->d_v, fin->step, fin->lr_w, fin->lr_d, fin->beta1,
                       fin->beta2, fin->eps, fin->poses_used, fin->record, fin->ring_rows, fin->poses_next, fin->poses12_next, fin->record_extra,
                       fin->n_record_extra};
  }
  const dim3 grid((unsigned)row0), block(dc::kBlock);
#define ICP_FUSED(T, PLANE) hipLaunchKernelGGL((dc::p2plane_seq_fused_kernel<T, PLANE>), grid, block, 0, stream, args, poses, model_kind, nt_model, w, e, \
                                              partials_ws, rq, out, n_scans, ticket, fa, fin ? 1 : 0)
  if (dtype == DC_F32) { if (plane) ICP_FUSED(float, true); else ICP_FUSED(float, false); }
  else { if (plane) ICP_FUSED(double, true); else ICP_FUSED(double, false); }
#undef ICP_FUSED
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? DC_OK : (int)err;
}

int dc_pose_correct_fwd(const double* poses, const double* deltas, int n_poses, int n_deltas, double* poses_out,
                        hipStream_t stream) {
  if (n_poses < 0 || (n_deltas != 1 && n_deltas != n_poses) || (n_poses > 0 && (!poses || !deltas || !poses_out))) return DC_ERR_ARG;
  if (n_poses == 0) return DC_OK;
  hipLaunchKernelGGL(dc::pose_correct_kernel, dim3(1), dim3(dc::kBlock), 0, stream, poses, deltas, n_poses, n_deltas, poses_out,
                     (const double*)nullptr, (double*)nullptr);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? DC_OK : (int)err;
}

int dc_pose_correct_bwd(const double* poses, const double* deltas, int n_poses, int n_deltas, const double* grad_poses,
                        double* grad_deltas, hipStream_t stream) {
  if (n_poses < 0 || (n_deltas != 1 && n_deltas != n_poses) || !grad_deltas || (n_poses > 0 && (!poses || !deltas || !grad_poses)))
    return DC_ERR_ARG;
  if (n_poses == 0) return (int)hipMemsetAsync(grad_deltas, 0, (size_t)n_deltas * 6 * sizeof(double), stream);
  hipLaunchKernelGGL(dc::pose_correct_kernel, dim3(1), dim3(dc::kBlock), 0, stream, poses, deltas, n_poses, n_deltas,
                     (double*)nullptr, grad_poses, grad_deltas);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? DC_OK : (int)err;
}

int dc_pose_train_combine(const double* const* outs, int n_seq, int layout, int n_terms, double* totals, hipStream_t stream) {
  if (n_seq < 1) return DC_ERR_ARG;
  return dc_pose_train_combine2(outs, n_seq, nullptr, -1, layout, n_terms, totals, stream);
}

int dc_pose_train_combine2(const double* const* outs_a, int n_a, const double* const* outs_b, int n_b, int layout, int n_terms,
                           double* totals, hipStream_t stream) {
  // n_b < 0: one group, totals [2 + P]; otherwise two groups (either may be empty: zeros), totals [2][2 + P]
  if (n_a < 0 || n_a > dc::kTrainSeqs || n_b > dc::kTrainSeqs || (n_a > 0 && !outs_a) || (n_b > 0 && !outs_b) || (layout != 0 && layout != 1) ||
      n_terms < 1 || n_terms > DC_MAX_MODEL_TERMS || !totals)
    return DC_ERR_ARG;
  dc::PoseCombineArgs a{};
  for (int i = 0; i < n_a; ++i) {
    if (!outs_a[i]) return DC_ERR_ARG;
    a.outs[i] = outs_a[i];
  }
  for (int i = 0; i < n_b; ++i) {
    if (!outs_b[i]) return DC_ERR_ARG;
    a.outs[n_a + i] = outs_b[i];
  }
  a.n_seq = n_a; a.n_seq_b = n_b < 0 ? 0 : n_b; a.layout = layout; a.n_terms = n_terms; a.totals = totals;
  hipLaunchKernelGGL(dc::pose_train_combine_kernel, dim3(n_b < 0 ? 1 : 2), dim3(64), 0, stream, a);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? DC_OK : (int)err;
}

int dc_pose_train_finish(const double* sums, int layout, int n_terms, int n_scans, double* w, double* w_m, double* w_v, const double* poses0,
                         double* deltas, double* d_m, double* d_v, int n_deltas, int zero_first, int64_t* step, double lr_w, double lr_d,
                         double beta1, double beta2, double eps, const double* poses_used, double* record, int ring_rows, double* poses_next,
                         double* poses12_next, const double* totals, const double* record_extra, int n_record_extra, hipStream_t stream) {
  if (n_record_extra < 0 || (n_record_extra > 0 && !record_extra)) return DC_ERR_ARG;
  if (!sums || n_terms < 1 || n_terms > DC_MAX_MODEL_TERMS || n_scans < 1 || (n_deltas != 1 && n_deltas != n_scans) || !poses0 || !deltas ||
      !d_m || !d_v || !step || !poses_used || !poses_next || !poses12_next || (w && (!w_m || !w_v)) || (record && ring_rows < 1))
    return DC_ERR_ARG;
  if (!(lr_w >= 0.0) || !(lr_d >= 0.0) || !(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0)) return DC_ERR_ARG;
  if (layout != 0 && layout != 1) return DC_ERR_ARG;
  // layout 0: dc_sequence_eval {sum, count, d/dw, d/de, d/d[R|t]}; 1: dc_p2plane_sequence / dc_p2point_sequence {loss, d/dw, d/de, d/d[R|t]}
  const int head = layout == 0 ? 2 : 1;
  dc::PoseTrainArgs a{sums, head + 2 * n_terms + 12 * n_scans, layout == 0 ? 1 : -1, head, head + 2 * n_terms, totals,
                      n_terms, n_scans, n_deltas, zero_first, w, w_m, w_v, poses0, deltas, d_m, d_v, step, lr_w, lr_d, beta1, beta2, eps,
                      poses_used, record, ring_rows, poses_next, poses12_next, record_extra, n_record_extra};
  hipLaunchKernelGGL(dc::pose_train_finish_kernel, dim3(1), dim3(dc::kBlock), 0, stream, a);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? DC_OK : (int)err;
}

}  // extern "C"
