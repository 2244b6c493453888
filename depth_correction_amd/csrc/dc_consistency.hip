// Map-consistency hot path kernels (gfx950): corrected points, neighbourhood features + loss forward,
// hand-derived backward.  C ABI at the bottom; declarations and reference citations in include/dc_hip.h.
//
// Data layout in HBM (all arrays owned by the caller, i.e. torch's allocator):
//   per point (SoA): vps[N,3], dirs[N,3], depth[N], inc[N] (T = f32|f64), lmask u8[N], scan_id i32[N]
//   points          x[N,S]  S = 3 (API layout) or 4 (padded: one 16-B gather per neighbour for f32 / q32);
//                   point format PT = f32 | f64 | q32 (DC_Q32: int32 fixed point, x = origin + q * scale)
//   neighbours      nbr i32[N,K] row major, -1 = missing
//   backward record rec[N,8] = {cmean.xyz, c1, v0.xyz, c2} in the point format (32 B: two 16-B loads per edge)
//   incoming edges  csr_ptr i32[N+1], csr_src i32[E]  (transpose of nbr, built once per neighbourhood set)
//   block tables    per 256-row block the distinct rows it references + u16 block-local positions, slot-major
//                   (dc_blocktab.hip): what the hot ("staged") kernels read instead of nbr / csr_*
// Arithmetic on chip is fp64 (differences against the centre point are exact, covariances and the eigen-solve
// keep LAPACK-level accuracy; the backward's per-edge term for q32 records is formed in fp32 from fp32-exact
// inputs); only storage is T.  No atomics: block partial sums are
// written to a workspace and reduced in a fixed order, so every result is bitwise reproducible.
#include <type_traits>
#include <atomic>
#include <mutex>
#include <cstdio>
#include <hip/hip_ext.h>
#include "dc_common.h"
#include "dc_device.h"
#include "dc_pointmath.h"
#include "dc_prof.h"
#include "dc_points_dev.h"
#include "../../include/dc_hip.h"

namespace dc {

// ------------------------------------------------------------------------------------------------
// K1+K2+K3: d' = model(d, inc) on masked points, (vps, dirs) -> pose frame, x = vps' + d' dirs'.
// ------------------------------------------------------------------------------------------------
template <typename T, typename PT, int STRIDE>
__global__ __launch_bounds__(kBlock) void points_fwd_kernel(PointInputs in, int64_t n, QParams qp, PT* __restrict__ x_out,
                                                            T* __restrict__ vps_out, T* __restrict__ dirs_out,
                                                            T* __restrict__ depth_out) {
  __shared__ double s_pose[kLdsScans * 12];
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t ic = i < n ? i : n - 1;                                // lanes past the end read the last point, store nothing
  ModelParams mp;
  load_model(in, mp);
  // the point's own loads are issued (raw, unconverted: no wait yet) BEFORE the poses are staged, so the two global
  // round trips overlap
  T vp_r[3] = {T(0), T(0), T(0)}, dr_r[3];
  if (in.vps) { const T* q = (const T*)in.vps + ic * 3; vp_r[0] = q[0]; vp_r[1] = q[1]; vp_r[2] = q[2]; }   // NULL: sensor origin
  { const T* q = (const T*)in.dirs + ic * 3; dr_r[0] = q[0]; dr_r[1] = q[1]; dr_r[2] = q[2]; }
  const T d_r = ((const T*)in.depth)[ic];
  const bool lm = in.lmask ? in.lmask[ic] != 0 : true;
  const T inc_r = (mp.kind != DC_MODEL_NONE) ? ((const T*)in.inc)[ic] : T(0);
  const int sid = in.scan_id ? in.scan_id[ic] : 0;
  const PoseTile poses = stage_poses(in, s_pose);
  __syncthreads();
  if (i >= n) return;
  double vp[3] = {(double)vp_r[0], (double)vp_r[1], (double)vp_r[2]}, dr[3] = {(double)dr_r[0], (double)dr_r[1], (double)dr_r[2]};
  double T12[12];
  const double dc_ = model_depth(mp, (double)d_r, lm ? (double)inc_r : 0.0, lm);
  load_pose(in, poses, sid, T12);
  double vr[3], drr[3], x[3];
  rot3(T12, vp, vr);
  vr[0] += T12[3]; vr[1] += T12[7]; vr[2] += T12[11];
  rot3(T12, dr, drr);
  x[0] = vr[0] + dc_ * drr[0]; x[1] = vr[1] + dc_ * drr[1]; x[2] = vr[2] + dc_ * drr[2];
  Row3<PT, STRIDE>::store(x_out, i, x, qp);
  if (vps_out) Row3<T, 3>::store(vps_out, i, vr, qp);
  if (dirs_out) Row3<T, 3>::store(dirs_out, i, drr, qp);
  if (depth_out) depth_out[i] = (T)dc_;
}

// Everything of the forward after the neighbourhood moments are gathered: covariance -> smallest eigenpair -> loss,
// backward record, masked loss / count of this lane (acc2).  Shared by the gather and the LDS-staged kernel.
template <typename T, typename PT, bool FULL_EIG>
__device__ __forceinline__ void consistency_point(CovAcc& acc, const typename Pt<PT>::Raw& ci, int64_t i,
                                                  const uint8_t* __restrict__ mask, const T* __restrict__ offset,
                                                  const LossParams& lp, const QParams& qp, PT* __restrict__ rec,
                                                  T* __restrict__ pointwise, T* __restrict__ eigvals, double* acc2) {
  cov_same_weights(acc);
  // moments are in raw units (q32: multiples of the resolution); scale once
  const double u = Pt<PT>::unit(qp), u2 = u * u;
  double moff[3], cm[3], C[6], D, omega;
  cov_finish(acc, 0.0, moff, cm, C, &D, &omega, u2);
  const bool m = mask ? mask[i] != 0 : true;
  const double off = offset ? (double)offset[i] : 0.0;
  double lam0, v0[3], tr, c1, c2, l;
  if (FULL_EIG) {
    double lam[3], V[3][3];
    eig3_sym<double>(C[0], C[1], C[2], C[3], C[4], C[5], lam, V);
    lam0 = lam[0]; v0[0] = V[0][0]; v0[1] = V[0][1]; v0[2] = V[0][2];
    tr = lam[0] + lam[1] + lam[2];
    eigvals[i * 3] = (T)lam[0]; eigvals[i * 3 + 1] = (T)lam[1]; eigvals[i * 3 + 2] = (T)lam[2];
  } else {
    eig3_smallest(C[0], C[1], C[2], C[3], C[4], C[5], &lam0, v0, &tr);
  }
  double raw;
  l = loss_and_coeffs(lp, lam0, tr, D, off, m, &c1, &c2, &raw);
  const bool drop = loss_dropped(lp, l);
  if (drop) c1 = c2 = 0.0;
  if (m && !drop) { acc2[0] = l; acc2[1] = 1.0; }
  // record: covariance mean in the point format; coefficients act on differences in metres
  if (rec) RecRaw<PT>::store(rec, i, Pt<PT>::offset(ci, cm), c1, v0, c2);
  if (pointwise) pointwise[i] = (T)(lp.raw_pointwise ? raw : l);
}

// ------------------------------------------------------------------------------------------------
// Hot-path forward: neighbourhood covariance -> smallest eigenpair -> pointwise loss + backward
// record; block partial sums of (masked loss, mask count).
// ------------------------------------------------------------------------------------------------
template <typename T, typename PT, int STRIDE, bool FULL_EIG>
__global__ __launch_bounds__(kBlock) void consistency_fwd_kernel(
    const PT* __restrict__ x, const int32_t* __restrict__ nbr, const int32_t* __restrict__ centre_idx, int64_t n, int k,
    const uint8_t* __restrict__ mask,
    const T* __restrict__ offset, LossParams lp, QParams qp, PT* __restrict__ rec, T* __restrict__ pointwise,
    T* __restrict__ eigvals, double* __restrict__ partials) {
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  const int64_t blk = xcd_block(nblocks);
  double acc2[2] = {0.0, 0.0};
  if (blk >= 0) {
    const int64_t i = blk * kBlock + threadIdx.x;          // row of nbr / rec / pointwise
    if (i < n) {
      // with a centre list only the listed points are centres (e.g. the masked ones); rows stay compact
      const int64_t ip = centre_idx ? (int64_t)centre_idx[i] : i;
      const typename Pt<PT>::Raw ci = Pt<PT>::template load<STRIDE>(x, ip, qp);
      CovAcc acc;
      cov_init(acc);
      const int32_t* row = nbr + i * k;
      for (int q0 = 0; q0 < k; q0 += 4) {
        // four independent gathers in flight per trip (a missing neighbour re-reads the centre row: always valid)
        int32_t j[4];
        typename Pt<PT>::Raw cj[4];
#pragma unroll
        for (int u_ = 0; u_ < 4; ++u_) j[u_] = (q0 + u_ < k) ? row[q0 + u_] : -1;
#pragma unroll
        for (int u_ = 0; u_ < 4; ++u_) cj[u_] = Pt<PT>::template load<STRIDE>(x, j[u_] >= 0 ? (int64_t)j[u_] : ip, qp);
#pragma unroll
        for (int u_ = 0; u_ < 4; ++u_) {
          // a missing neighbour re-read the centre row: its difference is exactly zero, only the count is masked
          double d[3];
          Pt<PT>::delta(cj[u_], ci, d);
          cov_add1(acc, d[0], d[1], d[2]);
          acc.W -= (j[u_] >= 0) ? 0.0 : 1.0;
        }
      }
      consistency_point<T, PT, FULL_EIG>(acc, ci, i, mask, offset, lp, qp, rec, pointwise, eigvals, acc2);
    }
  }
  wave_partials<2>(acc2, partials);
}

// ------------------------------------------------------------------------------------------------
// Backward epilogue per point: dL/dx_j -> dL/dw, dL/dexponent, dL/d[R|t] of the point's scan.
// acc layout: [0,P) grad w, [P,2P) grad exponent, then 12 per-scan slots handled by the caller.
// ------------------------------------------------------------------------------------------------
// the per-point inputs of the epilogue in their storage type, so they can be requested early (before a gather loop)
template <typename T>
struct PointRaw {
  T dr[3], d, inc;
  bool lm;
  int s;
};
template <typename T>
__device__ __forceinline__ PointRaw<T> load_point_raw(const PointInputs& in, const ModelParams& mp, int64_t j) {
  PointRaw<T> r;
  const T* dirs = (const T*)in.dirs;
  r.dr[0] = dirs[j * 3]; r.dr[1] = dirs[j * 3 + 1]; r.dr[2] = dirs[j * 3 + 2];
  r.d = ((const T*)in.depth)[j];
  r.lm = in.lmask ? in.lmask[j] != 0 : true;
  r.s = in.scan_id ? in.scan_id[j] : 0;
  r.inc = (mp.kind != DC_MODEL_NONE) ? ((const T*)in.inc)[j] : (T)0;
  return r;
}

template <typename T>
__device__ __forceinline__ void points_bwd_point(const PointInputs& in, const PoseTile& poses, const ModelParams& mp, int64_t j,
                                                 const PointRaw<T>& raw, const double* g, double* gw, double* ge, double* gT,
                                                 bool want_e, bool want_pose, int* scan) {
  double vp[3], dr[3], T12[12];
  const QParams qp0{};
  if (in.vps) Row3<T, 3>::load((const T*)in.vps, j, vp, qp0);
  else { vp[0] = vp[1] = vp[2] = 0.0; }
  dr[0] = (double)raw.dr[0]; dr[1] = (double)raw.dr[1]; dr[2] = (double)raw.dr[2];
  const double d = (double)raw.d;
  const bool lm = raw.lm;
  const int s = raw.s;
  *scan = s;
  load_pose(in, poses, s, T12);
  // dL/dd' = (R dir) . g = dir . (R^T g)
  const double rg0 = T12[0] * g[0] + T12[4] * g[1] + T12[8] * g[2];
  const double rg1 = T12[1] * g[0] + T12[5] * g[1] + T12[9] * g[2];
  const double rg2 = T12[2] * g[0] + T12[6] * g[1] + T12[10] * g[2];
  const double gd = dr[0] * rg0 + dr[1] * rg1 + dr[2] * rg2;
  double dcorr = d;
  if (mp.kind > DC_MODEL_SCALED_POLYNOMIAL && lm) {          // Linear / InvCos / ScaledInvCos: no exponents
    const double inc = (double)raw.inc;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k < mp.n_terms) gw[k] += gd * model_dw_other(mp, k, d, inc);
    dcorr = model_depth(mp, d, inc, true);
  } else if (mp.kind != DC_MODEL_NONE && lm) {
    const double inc = (double)raw.inc;
    const double base = mp.kind == DC_MODEL_SCALED_POLYNOMIAL ? -d * gd : -gd;
    double bias = 0.0;
#pragma unroll
    for (int k = 0; k < DC_MAX_MODEL_TERMS; ++k) {
      if (k < mp.n_terms) {
        const double pk = pow_term(inc, mp.e[k]);
        bias += pk * mp.w[k];
        gw[k] += base * pk;
        if (want_e) ge[k] += (inc > 0.0) ? base * mp.w[k] * pk * log(inc) : 0.0;
      }
    }
    dcorr = mp.kind == DC_MODEL_SCALED_POLYNOMIAL ? d * (1.0 - bias) : d - bias;
  }
  if (want_pose) {
    // x = R xl + t, xl = vps + d' dirs:  dL/dR = g xl^T, dL/dt = g
    const double xl0 = vp[0] + dcorr * dr[0], xl1 = vp[1] + dcorr * dr[1], xl2 = vp[2] + dcorr * dr[2];
    // kept factored (6 values, not 12) until the block reduction: gT = [g, xl], dL/d[R|t]_{a,b} = g_a * [xl, 1]_b
    gT[0] = g[0]; gT[1] = g[1]; gT[2] = g[2]; gT[3] = xl0; gT[4] = xl1; gT[5] = xl2;
  }
}

template <typename T>
__device__ __forceinline__ void points_bwd_point(const PointInputs& in, const PoseTile& poses, const ModelParams& mp, int64_t j,
                                                 const double* g, double* gw, double* ge, double* gT, bool want_e,
                                                 bool want_pose, int* scan) {
  points_bwd_point<T>(in, poses, mp, j, load_point_raw<T>(in, mp, j), g, gw, ge, gT, want_e, want_pose, scan);
}

// Per-scan sums of the 12 pose-gradient values g_a * [xl, 1]_b of the 256 points of a block (gx = [g, xl] of this
// lane; pcol0 = first pose slot of this block's partial column).  The lanes are counting-sorted by scan id (wave ballots + a tiny prefix), staged in LDS in that
// order (14 KB) and every (scan, value) pair is summed by one thread over its contiguous segment: ~25 additions per pair
// instead of a 256-lane tree per scan, and a fixed order (wave, lane) => bitwise reproducible.  Scans absent from the
// block keep the zeros the caller put into the workspace.
constexpr int kPoseRow = 7;              // doubles per staged point: the factors [g, xl] + 1 pad (conflict-free 8-B LDS stores)
constexpr int kMaxBlockScans = 64;       // more distinct scans in one block: per-scan tree reduction instead

__device__ __forceinline__ void reduce_pose_grads(const PointInputs& in, bool active, const double* gx, int scan,
                                                  double* lds, double* __restrict__ pcol0, double* s_val) {
  constexpr int NW = kBlock / kWave;
  __shared__ int s_cnt[NW][kMaxBlockScans];
  __shared__ int s_start[kMaxBlockScans + 1];
  __shared__ int s_range[2];
  const int64_t rs = (int64_t)gridDim.x * kWavesPerBlock;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  if (tid == 0) { s_range[0] = 0x7fffffff; s_range[1] = -1; }
  __syncthreads();
  const bool ok = active && scan >= 0 && scan < in.n_scans;
  {
    // range of the scans present: wave-level min / max first, one LDS atomic per wavefront
    int lo = ok ? scan : 0x7fffffff, hi = ok ? scan : -1;
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
      lo = min(lo, __shfl_xor(lo, off, kWave));
      hi = max(hi, __shfl_xor(hi, off, kWave));
    }
    if (lane == 0 && hi >= 0) { atomicMin(&s_range[0], lo); atomicMax(&s_range[1], hi); }
  }
  __syncthreads();
  const int s_lo = s_range[0], s_hi = s_range[1];
  if (s_hi < s_lo) return;                                  // block-uniform: nothing to add
  const int ns = s_hi - s_lo + 1;
  if (ns > kMaxBlockScans) {
    for (int s = s_lo; s <= s_hi; ++s) {
      double t[12];
      const bool mine = ok && scan == s;
#pragma unroll
      for (int q = 0; q < 12; ++q) t[q] = mine ? gx[q >> 2] * ((q & 3) == 3 ? 1.0 : gx[3 + (q & 3)]) : 0.0;
      block_sum<12>(t, lds);
      if (tid == 0) {
#pragma unroll
        for (int q = 0; q < 12; ++q) pcol0[(s * 12 + q) * rs] = t[q];
      }
    }
    return;
  }
  const int r = ok ? scan - s_lo : -1;
  int rank_in_wave = 0;
  for (int q = 0; q < ns; ++q) {
    const unsigned long long m = __ballot(r == q);
    if (lane == 0) s_cnt[wave][q] = __popcll(m);
    if (r == q) rank_in_wave = __popcll(m & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  if (wave == 0) {
    // exclusive prefix of the per-scan totals over the (at most 64) scans of the range, inside one wavefront
    int tot = 0;
    if (lane < ns) {
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += s_cnt[w][lane];
    }
    int incl = tot;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int up = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += up;
    }
    if (lane < ns) s_start[lane] = incl - tot;
    if (lane == ns - 1) s_start[ns] = incl;                 // number of contributing lanes
  }
  __syncthreads();
  if (ok) {
    int pos = s_start[r] + rank_in_wave;
    for (int w = 0; w < wave; ++w) pos += s_cnt[w][r];
#pragma unroll
    for (int q = 0; q < 6; ++q) s_val[pos * kPoseRow + q] = gx[q];
  }
  __syncthreads();
  // the 12 products g_a * [xl, 1]_b are formed while summing (each rounded, then added: no contraction, so the sums do
  // not depend on how many factors were staged)
  // Two lanes per (scan, entry): even / odd elements of the segment, four independent partial sums each so that the LDS
  // reads pipeline (a single running sum made this loop a chain of ~100 dependent LDS round trips per block).
  for (int item = tid; item < ns * 24; item += kBlock) {
    const int pair = item >> 1, part = item & 1;
    const int rr = pair / 12, q = pair - rr * 12, a = q >> 2, b = q & 3;
    const int beg = s_start[rr] + part, end = s_start[rr + 1];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    int p = beg;
    for (; p + 6 < end; p += 8) {
#pragma unroll
      for (int u_ = 0; u_ < 4; ++u_) {
        const double ga = s_val[(p + 2 * u_) * kPoseRow + a];
        acc[u_] = __dadd_rn(acc[u_], b == 3 ? ga : __dmul_rn(ga, s_val[(p + 2 * u_) * kPoseRow + 3 + b]));
      }
    }
    for (; p < end; p += 2) {
      const double ga = s_val[p * kPoseRow + a];
      acc[0] = __dadd_rn(acc[0], b == 3 ? ga : __dmul_rn(ga, s_val[p * kPoseRow + 3 + b]));
    }
    double sum = __dadd_rn(__dadd_rn(acc[0], acc[1]), __dadd_rn(acc[2], acc[3]));
    sum = __dadd_rn(sum, __shfl_xor(sum, 1, kWave));         // even + odd half (the two lanes are neighbours)
    if (part == 0) pcol0[((s_lo + rr) * 12 + q) * rs] = sum;
  }
}

// The same sums when the plan has grouped the points of every block by scan id (PointInputs.seg_start; round 4): the segments
// are contiguous lane ranges known since set-up, so nothing is counted, ranked or sorted at run time, and NO BARRIER is left
// at the block's tail.  Every wavefront sums its own 64 lanes: the six factors go to its part of an LDS array where they stand,
// (scan, entry, half) items of the two to four scans its lanes belong to walk their stretch of the segment, and the sums go to
// the wavefront's row of s_part.  A ticket in LDS tells the wavefront that finishes last to add the four rows in fixed order
// (bitwise reproducible) and to store the block's 12 S sums as ONE contiguous row of the row-major pose partials
// [blocks][12 S]; the others have long retired.  Before: a counting sort of the block's lanes by scan -- three more barriers,
// a ballot loop over the scans, a prefix pass -- one running block-wide barrier before the sums, 12 S pieces of 32 B stored to
// as many pages per block, and a 30 MB memset of the pose columns per evaluation (113 us at C2, 65 % of the wave cycles
// waiting).
constexpr int kTicketScans = 16;          // the four partial rows cost 96 B of LDS per scan: sequences of up to 16 scans take the ticket form
// (module-scope LDS word: the backward kernels zero it before their staging barrier, pose_ticket_init)
__shared__ int s_pose_ticket;
__device__ __forceinline__ void pose_ticket_init() { if (threadIdx.x == 0) s_pose_ticket = 0; }
__device__ __forceinline__ void reduce_pose_grads_grouped(const PointInputs& in, int64_t blk, bool active, const double* gx,
                                                          int scan, double* __restrict__ prow, double* s_val) {
  constexpr int NW = kBlock / kWave;
  __shared__ double s_part[NW][12 * kTicketScans];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int S = in.n_scans, V = 2 * S;                      // V segments per block: (inside the loss mask, scan) then (outside, scan)
  if (blk < 0) {                                            // padding block of the last round: its row holds zeros
    for (int item = tid; item < S * 12; item += kBlock) prow[item] = 0.0;
    return;
  }
  const uint16_t* seg = in.seg_start + blk * (V + 1);
  // the segments that reach into this wavefront's 64 lanes: lane v < V looks at segment v
  const int w_beg = wave * kWave, w_end = w_beg + kWave;
  int sb_l = 0, se_l = 0;
  if (lane < V) { sb_l = seg[lane]; se_l = seg[lane + 1]; }
  const unsigned long long present = __ballot(lane < V && sb_l < w_end && se_l > w_beg && se_l > sb_l);
  const int v_lo = present ? __builtin_ctzll(present) : 0, v_hi = present ? 63 - __builtin_clzll(present) : -1;
  const int n_items = (v_hi - v_lo + 1) * 24;
  for (int item = lane; item < S * 12; item += kWave) s_part[wave][item] = 0.0;
#pragma unroll
  for (int q = 0; q < 6; ++q) s_val[tid * kPoseRow + q] = active ? gx[q] : 0.0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // own lanes' values: the wavefront's LDS operations are in order
  __builtin_amdgcn_wave_barrier();
  for (int item0 = 0; item0 < n_items; item0 += kWave) {
    const int item = item0 + lane;
    const int pair = item >> 1, part = item & 1;
    const int rs_ = pair / 12, q = pair - rs_ * 12, a = q >> 2, b = q & 3;
    const int v = min(v_lo + rs_, V - 1);
    // (the shuffles run with every lane active: a masked-off source lane would hand back zero)
    const int sb = __shfl(sb_l, v, kWave), se = __shfl(se_l, v, kWave);
    if (item >= n_items) continue;
    const int beg = max(sb, w_beg) + part, end = min(se, w_end);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    int p = beg;
    for (; p + 6 < end; p += 8) {
#pragma unroll
      for (int u_ = 0; u_ < 4; ++u_) {
        const double ga = s_val[(p + 2 * u_) * kPoseRow + a];
        acc[u_] = __dadd_rn(acc[u_], b == 3 ? ga : __dmul_rn(ga, s_val[(p + 2 * u_) * kPoseRow + 3 + b]));
      }
    }
    for (; p < end; p += 2) {
      const double ga = s_val[p * kPoseRow + a];
      acc[0] = __dadd_rn(acc[0], b == 3 ? ga : __dmul_rn(ga, s_val[p * kPoseRow + 3 + b]));
    }
    double sum = __dadd_rn(__dadd_rn(acc[0], acc[1]), __dadd_rn(acc[2], acc[3]));
    sum = __dadd_rn(sum, __shfl_xor(sum, 1, kWave));         // even + odd half (the two lanes are neighbours)
    // the two segments of a scan (inside / outside the mask) can both reach into one wavefront: an add, not a store (two
    // addends on a zeroed slot: the same sum in either order)
    if (part == 0) atomicAdd(&s_part[wave][(v >= S ? v - S : v) * 12 + q], sum);
  }
  // ticket: the wavefront that arrives last adds the rows up and stores the block's row
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  int old = 0;
  if (lane == 0) old = atomicAdd(&s_pose_ticket, 1);
  old = __builtin_amdgcn_readfirstlane(old);
  if (old != NW - 1) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int item = lane; item < S * 12; item += kWave) {
    double t = s_part[0][item];
#pragma unroll
    for (int w = 1; w < NW; ++w) t = __dadd_rn(t, s_part[w][item]);
    prow[item] = t;
  }
}

// Sequences of 17 .. 64 scans: the same static segments, summed by the whole block behind one barrier (the partial rows of the
// ticket form would take 24 KB of LDS).
__device__ __forceinline__ void reduce_pose_grads_grouped_wide(const PointInputs& in, int64_t blk, bool active, const double* gx,
                                                               double* __restrict__ prow, double* s_val) {
  const int tid = threadIdx.x;
  const int S = in.n_scans, V = 2 * S;
  if (blk < 0) {
    for (int item = tid; item < S * 12; item += kBlock) prow[item] = 0.0;
    return;
  }
  const uint16_t* seg = in.seg_start + blk * (V + 1);
#pragma unroll
  for (int q = 0; q < 6; ++q) s_val[tid * kPoseRow + q] = active ? gx[q] : 0.0;
  __syncthreads();
  for (int item = tid; item < S * 24; item += kBlock) {
    const int pair = item >> 1, part = item & 1;
    const int rr = pair / 12, q = pair - rr * 12, a = q >> 2, b = q & 3;
    double sum = 0.0;
    for (int half = 0; half < 2; ++half) {                  // the scan's points inside the loss mask, then those outside
      const int beg = (int)seg[rr + half * S] + part, end = (int)seg[rr + half * S + 1];
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      int p = beg;
      for (; p + 6 < end; p += 8) {
#pragma unroll
        for (int u_ = 0; u_ < 4; ++u_) {
          const double ga = s_val[(p + 2 * u_) * kPoseRow + a];
          acc[u_] = __dadd_rn(acc[u_], b == 3 ? ga : __dmul_rn(ga, s_val[(p + 2 * u_) * kPoseRow + 3 + b]));
        }
      }
      for (; p < end; p += 2) {
        const double ga = s_val[p * kPoseRow + a];
        acc[0] = __dadd_rn(acc[0], b == 3 ? ga : __dmul_rn(ga, s_val[p * kPoseRow + 3 + b]));
      }
      sum = __dadd_rn(sum, __dadd_rn(__dadd_rn(acc[0], acc[1]), __dadd_rn(acc[2], acc[3])));
    }
    sum = __dadd_rn(sum, __shfl_xor(sum, 1, kWave));
    if (part == 0) prow[pair] = sum;
  }
}

// Parameter gradients of one block into the partial rows of its wavefronts (row = 4 * block + wave; no LDS, no barrier:
// every wavefront leaves as soon as its lane 0 has written its sums):
//   [0,P) w, [P,2P) exponent, [2P, 2P + 12 S) poses.  Only the slots in use are reduced.  The pose slots are summed per
// block (counting sort by scan in LDS) into the row of the block's first wavefront; the others keep the caller's zeros.
template <typename T>
__device__ __forceinline__ void reduce_param_grads(const PointInputs& in, bool active, bool want_e, bool want_pose,
                                                   double* gw, double* ge, double* gT, int scan, double* lds,
                                                   double* __restrict__ partials, int64_t blk = -2) {
  const int64_t rs = (int64_t)gridDim.x * kWavesPerBlock;              // slot a lives at partials[a * rs + row]
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  double* prow = partials + (int64_t)blockIdx.x * kWavesPerBlock + wave;
  const int P = in.n_terms;
#pragma unroll
  for (int k = 0; k < DC_MAX_MODEL_TERMS; ++k) {
    if (k < P) {
      const double sw = wave_sum(gw[k]);
      const double se = want_e ? wave_sum(ge[k]) : 0.0;
      if (lane == 0) { prow[k * rs] = sw; prow[(P + k) * rs] = se; }
    }
  }
  // blk: the block's logical index where the caller's lanes are the plan's points in order (the grouped layout applies), -2 otherwise
  if (!want_pose) return;
  __shared__ double s_val[kBlock * kPoseRow];      // ONE staging array for whichever form runs (each function's own would add up: 14 KB apiece)
  if (in.seg_start && blk != -2) {
    double* prow = partials + 2 * P * rs + (int64_t)blockIdx.x * 12 * in.n_scans;
    if (in.n_scans <= kTicketScans) reduce_pose_grads_grouped(in, blk, active, gT, scan, prow, s_val);
    else reduce_pose_grads_grouped_wide(in, blk, active, gT, prow, s_val);
  }
  else reduce_pose_grads(in, active, gT, scan, lds, partials + (int64_t)blockIdx.x * kWavesPerBlock + 2 * P * rs, s_val);
}

// ------------------------------------------------------------------------------------------------
// Hot-path backward: dL/dx_j = sum over incoming edges (i -> j) of c1_i (v0_i . d) v0_i - c2_i d,
// d = x_j - cmean_i, gathered through the transposed neighbour list; fused with the point epilogue.
// ------------------------------------------------------------------------------------------------
#ifndef DC_BWD_F32
#define DC_BWD_F32 1
#endif
template <typename PT> constexpr bool kEdgeF32 = DC_BWD_F32 && std::is_same<PT, q32>::value;

// g += sum over four incoming edges of c1 (v0 . d) v0 - c2 d, d = x_j - cmean_i, from the raw 16-B pieces of the four
// records (an all-zero record contributes exactly nothing).  Shared by the gather and the LDS-staged kernel.
//   q32: the records hold c1, v0, c2 in fp32 and the integer differences of neighbouring points are exact in fp32
//   (< 2^24 grid steps), so the terms are formed in fp32 and the trip's partial sum is folded into the fp64
//   accumulators (error ~1e-7 of the largest term of the trip);
//   float / double points: fp64 throughout, separate multiplies and adds (measured faster than the dependent FMA
//   chains contraction produces: 90 vs 98 us at N = 2 M).
template <typename PT>
__device__ __forceinline__ void edge_terms4(const typename Pt<PT>::Raw& cj, const int4 (*q)[RecRaw<PT>::kRow16], double* g) {
  if constexpr (kEdgeF32<PT>) {
#pragma clang fp contract(fast)
    float gf[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int u_ = 0; u_ < 4; ++u_) {
      const int4 ra = q[u_][0], rb = q[u_][1];
      const float d0 = (float)(cj.v[0] - ra.x), d1 = (float)(cj.v[1] - ra.y), d2 = (float)(cj.v[2] - ra.z);
      const float c1 = __int_as_float(ra.w), c2 = __int_as_float(rb.w);
      const float v0 = __int_as_float(rb.x), v1 = __int_as_float(rb.y), v2 = __int_as_float(rb.z);
      const float t = c1 * (v0 * d0 + v1 * d1 + v2 * d2);
      gf[0] += t * v0 - c2 * d0;
      gf[1] += t * v1 - c2 * d1;
      gf[2] += t * v2 - c2 * d2;
    }
    g[0] += (double)gf[0]; g[1] += (double)gf[1]; g[2] += (double)gf[2];
  } else {
#pragma clang fp contract(off)
#pragma unroll
    for (int u_ = 0; u_ < 4; ++u_) {
      typename Pt<PT>::Raw m;
      double v[3], c1, c2, d[3];
      RecRaw<PT>::from_row(q[u_], m, &c1, v, &c2);
      Pt<PT>::delta(cj, m, d);
      const double t = c1 * (v[0] * d[0] + v[1] * d[1] + v[2] * d[2]);
      g[0] += t * v[0] - c2 * d[0];
      g[1] += t * v[1] - c2 * d[1];
      g[2] += t * v[2] - c2 * d[2];
    }
  }
}

template <typename T, typename PT, int STRIDE, bool WANT_E, bool WANT_POSE>
__global__ __launch_bounds__(kBlock) void consistency_bwd_kernel(
    const PT* __restrict__ x, const PT* __restrict__ rec, const int32_t* __restrict__ csr_ptr,
    const int32_t* __restrict__ csr_src, const uint8_t* __restrict__ lane_perm, int64_t n, PointInputs in, QParams qp,
    T* __restrict__ grad_points, double* __restrict__ partials, int n_acc, uint32_t rec_bytes) {
  // measured: the edge loop is faster with separate multiplies and adds (more independent work per trip) than with
  // the dependent FMA chains contraction produces (90 vs 98 us at N = 2 M)
#pragma clang fp contract(off)
  constexpr int want_e = WANT_E, want_pose = WANT_POSE;
  __shared__ double lds[(kBlock / kWave) * 2 * DC_MAX_MODEL_TERMS];
  __shared__ double s_pose[kLdsScans * 12];
  const PoseTile poses = in.dirs ? stage_poses(in, s_pose) : PoseTile{nullptr};
  if (WANT_POSE) pose_ticket_init();
  __syncthreads();
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  const int64_t blk = xcd_block(nblocks);
  ModelParams mp;
  load_model(in, mp);
  double gw[DC_MAX_MODEL_TERMS], ge[DC_MAX_MODEL_TERMS], gT[6];
#pragma unroll
  for (int k = 0; k < DC_MAX_MODEL_TERMS; ++k) gw[k] = ge[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) gT[k] = 0.0;
  int scan = -1;
  bool active = false;
  if (blk >= 0) {
    // lane_perm orders the 256 points of a block by in-degree, so the lanes of a wavefront run similar trip counts
    const int64_t base = blk * kBlock;
    const int64_t j = base + (lane_perm ? (int64_t)lane_perm[base + threadIdx.x] : (int64_t)threadIdx.x);
    if (j < n) {
      active = true;
      double g[3] = {0.0, 0.0, 0.0};
      const typename Pt<PT>::Raw cj = Pt<PT>::template load<STRIDE>(x, j, qp);
      const double u = Pt<PT>::unit(qp);
      const int32_t beg = csr_ptr[j], end = csr_ptr[j + 1];
      // four edges per trip; the indices of the NEXT trip are requested before the records of this one are used,
      // so each trip exposes one memory latency instead of two.  Records come through a buffer resource: 32-bit
      // offsets, and an empty slot (index -1) is out of range and reads an all-zero record (c1 = c2 = 0).
      const BufRsrc rrec = make_rsrc(rec, rec_bytes);
      int32_t nxt[4];
#pragma unroll
      for (int u_ = 0; u_ < 4; ++u_) nxt[u_] = (beg + u_ < end) ? csr_src[beg + u_] : -1;
      for (int32_t e0 = beg; e0 < end; e0 += 4) {
        int32_t src[4];
        int4 q[4][RecRaw<PT>::kRow16];
#pragma unroll
        for (int u_ = 0; u_ < 4; ++u_) src[u_] = nxt[u_];
#pragma unroll
        for (int u_ = 0; u_ < 4; ++u_) {
#pragma unroll
          for (int a = 0; a < RecRaw<PT>::kRow16; ++a) q[u_][a] = buf_load16(rrec, (uint32_t)src[u_] * RowBytes<PT>::rec + 16u * a);
        }
#pragma unroll
        for (int u_ = 0; u_ < 4; ++u_) nxt[u_] = (e0 + 4 + u_ < end) ? csr_src[e0 + 4 + u_] : -1;
        edge_terms4<PT>(cj, q, g);
      }
      g[0] *= u; g[1] *= u; g[2] *= u;
      if (grad_points) Row3<T, STRIDE>::store(grad_points, j, g, QParams{});
      if (in.dirs) points_bwd_point<T>(in, poses, mp, j, g, gw, ge, gT, want_e != 0, want_pose != 0, &scan);
    }
  }
  if (in.dirs) reduce_param_grads<T>(in, active, want_e, want_pose, gw, ge, gT, scan, lds, partials, lane_perm ? -2 : blk);
}

// ------------------------------------------------------------------------------------------------
// LDS-staged gathers through a block table (dc_blocktab.hip).  A per-lane gather from global memory costs one L1 tag
// lookup per reference and the L1 serves one lookup per cycle: at K = 10 the 2560 references of a 256-point block kept
// the L1 of its CU busy for ~2560 cycles, which bounded both hot kernels (rocprofv3 TCP_TOTAL_CACHE_ACCESSES ~ 1.06 per
// CU cycle).  In Morton order those references hit only ~375 distinct rows, so every block first copies its distinct
// rows into LDS (one lookup each), then gathers from LDS through 16-bit block-local positions that are stored
// slot-major (a wavefront reads one slot with a single coalesced 128-B load).  Same arithmetic in the same order as the
// gather kernels => bit-identical results.
// ------------------------------------------------------------------------------------------------
struct BlockTab {
  const int32_t* __restrict__ blk_ptr;
  const int32_t* __restrict__ blk_ids;
  const int32_t* __restrict__ slot_ptr;
  const uint16_t* __restrict__ loc;
};
constexpr uint32_t kNoLoc = 0xFFFFu;

// Copy the distinct rows of block `blk` (ROW16 16-B pieces each) into `tile`; returns their number.  Piece a of row t
// lives at tile[a * cap + t]: a ds_read_b128 serves 16 lanes per cycle from 16 four-bank groups, and with whole rows
// back to back (32-B records) the group would be (2 t + a) mod 16 -- only 8 of the 16 for each piece, measured as 2/3
// of all LDS cycles lost to bank conflicts.  Piece-major, the group is t mod 16.
template <int ROW16>
__device__ __forceinline__ int stage_rows(const int32_t* __restrict__ blk_ptr, const int32_t* __restrict__ blk_ids, int64_t blk,
                                          const int4* __restrict__ src, int4* tile, int cap) {
  const int32_t base = blk_ptr[blk], nd = blk_ptr[blk + 1] - base;
  for (int t = threadIdx.x; t < nd; t += kBlock) {
    const int64_t id = blk_ids[base + t];
#pragma unroll
    for (int a = 0; a < ROW16; ++a) tile[a * cap + t] = src[id * ROW16 + a];
  }
  return nd;
}

// `off` = 16 x (row in the block's distinct list), exactly what the table stores: the LDS byte address needs no shift
template <int ROW16>
__device__ __forceinline__ void read_row(const int4* tile, int cap, uint32_t off, int4* q) {
  const char* p = reinterpret_cast<const char*>(tile) + off;
#pragma unroll
  for (int a = 0; a < ROW16; ++a) q[a] = *reinterpret_cast<const int4*>(p + (size_t)a * cap * 16);
}

constexpr int kPreSlots = 16;

// one slot from LDS; returns 1 if it held a neighbour.  MISS = false: the caller knows the slot is filled.
template <typename PT, bool MISS>
__device__ __forceinline__ int slot_add(const int4* tile, int cap, const typename Pt<PT>::Raw& ci, uint32_t l, CovAcc& acc) {
  constexpr int XR = Pt<PT>::kRow16;
  const bool have = !MISS || l != kNoLoc;
  int4 piece[XR];
  read_row<XR>(tile, cap, have ? l : 0u, piece);
  const typename Pt<PT>::Raw cj = Pt<PT>::from_row(piece);
  double d[3];
  Pt<PT>::delta(have ? cj : ci, ci, d);
  cov_add_d(acc, d[0], d[1], d[2]);
  return have ? 1 : 0;
}

// the first min(nslots, 16) slots from registers: whole trips of four with the LDS reads batched, then the remainder
template <typename PT, bool MISS>
__device__ __forceinline__ int gather_slots(const int4* tile, int cap, const typename Pt<PT>::Raw& ci, const uint32_t* pre,
                                            int nslots, CovAcc& acc) {
  constexpr int XR = Pt<PT>::kRow16;
  int n_have = 0;
#pragma unroll
  for (int t = 0; t < kPreSlots / 4; ++t) {
    if (4 * t + 4 <= nslots) {
      typename Pt<PT>::Raw cj[4];
      bool have[4];
#pragma unroll
      for (int u_ = 0; u_ < 4; ++u_) {
        have[u_] = !MISS || pre[4 * t + u_] != kNoLoc;
        int4 piece[XR];
        read_row<XR>(tile, cap, have[u_] ? pre[4 * t + u_] : 0u, piece);
        cj[u_] = Pt<PT>::from_row(piece);
      }
#pragma unroll
      for (int u_ = 0; u_ < 4; ++u_) {
        double d[3];
        Pt<PT>::delta(have[u_] ? cj[u_] : ci, ci, d);
        cov_add_d(acc, d[0], d[1], d[2]);
        n_have += have[u_] ? 1 : 0;
      }
    } else {
#pragma unroll
      for (int u_ = 0; u_ < 4; ++u_)
        if (4 * t + u_ < nslots) n_have += slot_add<PT, MISS>(tile, cap, ci, pre[4 * t + u_], acc);
    }
  }
  return n_have;
}

// The slots beyond the first 16 (radius neighbourhoods: a hundred or two per point), in trips of eight: the positions of the
// NEXT trip are requested before the rows of this one are read, and the eight row reads of a trip are in flight together.  One
// slot per iteration -- a dependent {position load, LDS read} pair each -- left the kernel waiting on memory for most of a
// 200-slot row.  `packed` (dcBlockTable.packed): rows fill their slots from 0 upwards, so a wavefront stops at the first trip
// that is empty for all its lanes -- its own longest row, not the block's.
constexpr int kTrip = 8;
template <typename PT>
__device__ __forceinline__ int gather_tail(const int4* tile, int cap, const typename Pt<PT>::Raw& ci, const uint16_t* lrow,
                                           int nslots, bool packed, CovAcc& acc) {
  constexpr int XR = Pt<PT>::kRow16;
  int n_have = 0;
  uint32_t nxt[kTrip];
#pragma unroll
  for (int u_ = 0; u_ < kTrip; ++u_) nxt[u_] = (kPreSlots + u_ < nslots) ? (uint32_t)lrow[(kPreSlots + u_) * kBlock] : kNoLoc;
  for (int q0 = kPreSlots; q0 < nslots; q0 += kTrip) {
    uint32_t l[kTrip];
#pragma unroll
    for (int u_ = 0; u_ < kTrip; ++u_) l[u_] = nxt[u_];
    if (packed && __all((int)(l[0] == kNoLoc))) break;
#pragma unroll
    for (int u_ = 0; u_ < kTrip; ++u_) nxt[u_] = (q0 + kTrip + u_ < nslots) ? (uint32_t)lrow[(q0 + kTrip + u_) * kBlock] : kNoLoc;
    typename Pt<PT>::Raw cj[kTrip];
#pragma unroll
    for (int u_ = 0; u_ < kTrip; ++u_) {
      int4 piece[XR];
      read_row<XR>(tile, cap, l[u_] != kNoLoc ? l[u_] : 0u, piece);
      cj[u_] = Pt<PT>::from_row(piece);
    }
#pragma unroll
    for (int u_ = 0; u_ < kTrip; ++u_) {
      const bool have = l[u_] != kNoLoc;
      double d[3];
      Pt<PT>::delta(have ? cj[u_] : ci, ci, d);
      cov_add_d(acc, d[0], d[1], d[2]);
      n_have += have ? 1 : 0;
    }
  }
  return n_have;
}

template <typename T, typename PT, bool FULL_EIG>
__global__ __launch_bounds__(kBlock) void consistency_fwd_staged_kernel(
    const PT* __restrict__ x, BlockTab tab, int cap, const int32_t* __restrict__ centre_idx, int64_t n,
    const uint8_t* __restrict__ mask, const T* __restrict__ offset, LossParams lp, QParams qp, PT* __restrict__ rec,
    T* __restrict__ pointwise, T* __restrict__ eigvals, double* __restrict__ partials) {
  constexpr int XR = Pt<PT>::kRow16;
  extern __shared__ int4 tile[];
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  const int64_t blk = xcd_block(nblocks);
  double acc2[2] = {0.0, 0.0};
  const int4* xg = reinterpret_cast<const int4*>(x);
  const int64_t i = blk * kBlock + threadIdx.x;
  const bool live = blk >= 0 && i < n;
  typename Pt<PT>::Raw ci;
  int32_t nslots = 0;
  const uint16_t* lrow = tab.loc;
  uint32_t pre[kPreSlots];                 // the lane's first 16 block-local positions
  if (blk >= 0) {
    // the lane's own requests go out before the staging loop, so their latency hides behind it
    const int32_t s0 = tab.slot_ptr[blk];
    nslots = tab.slot_ptr[blk + 1] - s0;
    lrow = tab.loc + (int64_t)s0 * kBlock + threadIdx.x;
    if (live) ci = Pt<PT>::from_row(xg + (centre_idx ? (int64_t)centre_idx[i] : i) * XR);
#pragma unroll
    for (int q = 0; q < kPreSlots; ++q) pre[q] = (live && q < nslots) ? (uint32_t)lrow[q * kBlock] : kNoLoc;
    stage_rows<XR>(tab.blk_ptr, tab.blk_ids, blk, xg, tile, cap);
  }
  __syncthreads();
  if (live) {
    CovAcc acc;
    cov_init(acc);
    // a missing neighbour (empty slot) contributes a zero difference and is not counted; only wavefronts that hold one
    // pay for the selects
    bool miss = false;
#pragma unroll
    for (int q = 0; q < kPreSlots; ++q) miss |= (q < nslots) && pre[q] == kNoLoc;
    int n_have = 0;
    if (__any((int)miss)) n_have = gather_slots<PT, true>(tile, cap, ci, pre, nslots, acc);
    else n_have = gather_slots<PT, false>(tile, cap, ci, pre, nslots, acc);
    for (int q = kPreSlots; q < nslots; ++q) {             // K > 16: one slot at a time
      const uint32_t l = lrow[q * kBlock];
      n_have += slot_add<PT, true>(tile, cap, ci, l, acc);
    }
    acc.W = (double)n_have;
    consistency_point<T, PT, FULL_EIG>(acc, ci, i, mask, offset, lp, qp, rec, pointwise, eigvals, acc2);
  }
  wave_partials<2>(acc2, partials);
}

// Fixed slot count: a forward table built from a neighbour table [rows, K] has exactly K slots in every block, so the
// slot loop, the position loads and the validity handling are resolved at compile time (the run-time variant above
// spends ~60 VALU instructions per point on them).  Same arithmetic in the same order => bit-identical results.
template <typename PT, int NS, bool MISS>
__device__ __forceinline__ int gather_fixed(const int4* tile, int cap, const typename Pt<PT>::Raw& ci, const uint32_t* pre,
                                            CovAcc& acc) {
  constexpr int XR = Pt<PT>::kRow16;
  int n_have = 0;
#pragma unroll
  for (int q0 = 0; q0 < NS; q0 += 4) {
    typename Pt<PT>::Raw cj[4];
    bool have[4];
#pragma unroll
    for (int u_ = 0; u_ < 4; ++u_) {
      if (q0 + u_ < NS) {
        have[u_] = !MISS || pre[q0 + u_] != kNoLoc;
        int4 piece[XR];
        read_row<XR>(tile, cap, have[u_] ? pre[q0 + u_] : 0u, piece);
        cj[u_] = Pt<PT>::from_row(piece);
      }
    }
#pragma unroll
    for (int u_ = 0; u_ < 4; ++u_) {
      if (q0 + u_ < NS) {
        double d[3];
        Pt<PT>::delta(have[u_] ? cj[u_] : ci, ci, d);
        cov_add_d(acc, d[0], d[1], d[2]);
        n_have += have[u_] ? 1 : 0;
      }
    }
  }
  return n_have;
}

template <typename T, typename PT, bool FULL_EIG, int NS>
__global__ __launch_bounds__(kBlock) void consistency_fwd_fixed_kernel(
    const PT* __restrict__ x, BlockTab tab, int cap, const int32_t* __restrict__ centre_idx, int64_t n,
    const uint8_t* __restrict__ mask, const T* __restrict__ offset, LossParams lp, QParams qp, PT* __restrict__ rec,
    T* __restrict__ pointwise, T* __restrict__ eigvals, double* __restrict__ partials) {
  constexpr int XR = Pt<PT>::kRow16;
  extern __shared__ int4 tile[];
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  const int64_t blk = xcd_block(nblocks);
  double acc2[2] = {0.0, 0.0};
  const int32_t s0 = blk >= 0 ? tab.slot_ptr[blk] : 0;
  // a table with another slot count than the launch was specialised for (not a table of [rows, NS]): fail loudly
  const bool bad = blk >= 0 && tab.slot_ptr[blk + 1] - s0 != NS;
  if (blk >= 0 && !bad) {
    const int4* xg = reinterpret_cast<const int4*>(x);
    const int64_t i = blk * kBlock + threadIdx.x;
    const bool live = i < n;
    // the table holds all 256 lanes of every block (0xFFFF beyond the last row): unconditional, immediate-offset loads
    const uint16_t* lrow = tab.loc + (int64_t)s0 * kBlock + threadIdx.x;
    uint32_t pre[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) pre[q] = (uint32_t)lrow[q * kBlock];
    typename Pt<PT>::Raw ci = Pt<PT>::from_row(xg + (live ? (centre_idx ? (int64_t)centre_idx[i] : i) : 0) * XR);
    stage_rows<XR>(tab.blk_ptr, tab.blk_ids, blk, xg, tile, cap);
    __syncthreads();
    if (live) {
      CovAcc acc;
      cov_init(acc);
      uint32_t mx = pre[0];
#pragma unroll
      for (int q = 1; q < NS; ++q) mx = max(mx, pre[q]);
      int n_have;
      if (__any((int)(mx == kNoLoc))) n_have = gather_fixed<PT, NS, true>(tile, cap, ci, pre, acc);
      else n_have = gather_fixed<PT, NS, false>(tile, cap, ci, pre, acc);
      acc.W = (double)n_have;
      consistency_point<T, PT, FULL_EIG>(acc, ci, i, mask, offset, lp, qp, rec, pointwise, eigvals, acc2);
    }
  } else {
    __syncthreads();
  }
  if (bad) acc2[0] = acc2[1] = __longlong_as_double(0x7ff8000000000000ll);
  wave_partials<2>(acc2, partials);
}

template <typename T, typename PT, bool WANT_E, bool WANT_POSE>
__global__ __launch_bounds__(kBlock) void consistency_bwd_staged_kernel(
    const PT* __restrict__ x, const PT* __restrict__ rec, BlockTab tab, int cap, int64_t n, PointInputs in, QParams qp,
    T* __restrict__ grad_points, double* __restrict__ partials, int n_acc) {
  constexpr int want_e = WANT_E, want_pose = WANT_POSE;
  constexpr int RR = RecRaw<PT>::kRow16, XR = Pt<PT>::kRow16;
  extern __shared__ int4 tile[];
  __shared__ double lds[(kBlock / kWave) * 2 * DC_MAX_MODEL_TERMS];
  __shared__ double s_pose[kLdsScans * 12];
  const PoseTile poses = in.dirs ? stage_poses(in, s_pose) : PoseTile{nullptr};      // published by the staging barrier
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  const int64_t blk = xcd_block(nblocks);
  ModelParams mp;
  load_model(in, mp);
  double gw[DC_MAX_MODEL_TERMS], ge[DC_MAX_MODEL_TERMS], gT[6];
#pragma unroll
  for (int k = 0; k < DC_MAX_MODEL_TERMS; ++k) gw[k] = ge[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) gT[k] = 0.0;
  int scan = -1;
  const int64_t j = blk * kBlock + threadIdx.x;
  const bool active = blk >= 0 && j < n;
  typename Pt<PT>::Raw cj;
  PointRaw<T> raw;
  uint32_t pre[kPreSlots];
  int32_t nslots = 0;
  uint32_t nd = 0;
  const uint16_t* lrow = tab.loc;
  if (blk >= 0) {
    const int32_t s0 = tab.slot_ptr[blk];
    nslots = tab.slot_ptr[blk + 1] - s0;
    lrow = tab.loc + (int64_t)s0 * kBlock + threadIdx.x;
    if (active) {
      // everything this lane needs later is requested before the staging loop: its latency hides behind it
      cj = Pt<PT>::from_row(reinterpret_cast<const int4*>(x) + j * XR);
      if (in.dirs) raw = load_point_raw<T>(in, mp, j);
    }
#pragma unroll
    for (int q = 0; q < kPreSlots; ++q) pre[q] = (active && q < nslots) ? (uint32_t)lrow[q * kBlock] : kNoLoc;
    nd = (uint32_t)stage_rows<RR>(tab.blk_ptr, tab.blk_ids, blk, reinterpret_cast<const int4*>(rec), tile, cap);
    // row nd is an all-zero record: empty slots point there and contribute exactly nothing
    if (threadIdx.x < RR) tile[threadIdx.x * cap + nd] = make_int4(0, 0, 0, 0);
  }
  const uint32_t nd16 = nd * 16u;                           // positions are byte offsets; 0xFFFF (empty) clamps to the zero record
  if (WANT_POSE) pose_ticket_init();
  __syncthreads();
  if (active) {
    double g[3] = {0.0, 0.0, 0.0};
    const double u = Pt<PT>::unit(qp);
    // a lane's slots fill from 0 upwards: once a whole trip is empty for every lane of the wavefront, so are the rest
    bool more = true;
#pragma unroll
    for (int t = 0; t < kPreSlots / 4; ++t) {
      if (more && 4 * t < nslots) {
        if (__all((int)(pre[4 * t] == kNoLoc))) {
          more = false;
        } else {
          int4 q[4][RR];
#pragma unroll
          for (int u_ = 0; u_ < 4; ++u_) read_row<RR>(tile, cap, min(pre[4 * t + u_], nd16), q[u_]);
          edge_terms4<PT>(cj, q, g);
        }
      }
    }
    if (more && nslots > kPreSlots) {                       // in-degrees above 16: positions fetched trip by trip
      uint32_t nxt[4];
#pragma unroll
      for (int u_ = 0; u_ < 4; ++u_) nxt[u_] = (kPreSlots + u_ < nslots) ? (uint32_t)lrow[(kPreSlots + u_) * kBlock] : kNoLoc;
      for (int q0 = kPreSlots; q0 < nslots; q0 += 4) {
        uint32_t l[4];
        int4 q[4][RR];
#pragma unroll
        for (int u_ = 0; u_ < 4; ++u_) l[u_] = nxt[u_];
        if (__all((int)(l[0] == kNoLoc))) break;
#pragma unroll
        for (int u_ = 0; u_ < 4; ++u_) nxt[u_] = (q0 + 4 + u_ < nslots) ? (uint32_t)lrow[(q0 + 4 + u_) * kBlock] : kNoLoc;
#pragma unroll
        for (int u_ = 0; u_ < 4; ++u_) read_row<RR>(tile, cap, min(l[u_], nd16), q[u_]);
        edge_terms4<PT>(cj, q, g);
      }
    }
    g[0] *= u; g[1] *= u; g[2] *= u;
    if (grad_points) Row3<T, 4>::store(grad_points, j, g, QParams{});
    if (in.dirs) points_bwd_point<T>(in, poses, mp, j, raw, g, gw, ge, gT, want_e != 0, want_pose != 0, &scan);
  }
  if (in.dirs) reduce_param_grads<T>(in, active, want_e, want_pose, gw, ge, gT, scan, lds, partials, blk);
}

// Backward over a "lane run" table (dc_block_table_build_runs): the positions of a point's incoming edges are stored
// per point, contiguous and padded to a multiple of four (one 8-B load = one trip of four edges) instead of slot-major
// padded to the block's largest in-degree -- at K = 10 that is 12 instead of 16.2 stored positions per point (48 + 8 MB
// of run pointers instead of 105 MB at C2), and every lane stops at its own in-degree.  Same edge order and the same
// trips of four as the other backward kernels => bit-identical gradients.
struct RunTab {
  const int32_t* __restrict__ blk_ptr;
  const int32_t* __restrict__ blk_ids;
  const int32_t* __restrict__ run_ptr;     // [n + 1] in units of runs (4 positions = 8 B)
  const uint16_t* __restrict__ loc;        // 16 x position, 0xFFFF = padding
};
constexpr int kPreRuns = 4;

template <typename PT>
__device__ __forceinline__ void run_edges(const int4* tile, int cap, uint2 r, uint32_t nd16, const typename Pt<PT>::Raw& cj, double* g) {
  constexpr int RR = RecRaw<PT>::kRow16;
  int4 q[4][RR];
  read_row<RR>(tile, cap, min(r.x & 0xFFFFu, nd16), q[0]);
  read_row<RR>(tile, cap, min(r.x >> 16, nd16), q[1]);
  read_row<RR>(tile, cap, min(r.y & 0xFFFFu, nd16), q[2]);
  read_row<RR>(tile, cap, min(r.y >> 16, nd16), q[3]);
  edge_terms4<PT>(cj, q, g);
}

template <typename T, typename PT, bool WANT_E, bool WANT_POSE>
__global__ __launch_bounds__(kBlock) void consistency_bwd_runs_kernel(
    const PT* __restrict__ x, const PT* __restrict__ rec, RunTab tab, int cap, int64_t n, PointInputs in, QParams qp,
    T* __restrict__ grad_points, double* __restrict__ partials, int n_acc) {
  constexpr int want_e = WANT_E, want_pose = WANT_POSE;
  constexpr int RR = RecRaw<PT>::kRow16, XR = Pt<PT>::kRow16;
  extern __shared__ int4 tile[];
  __shared__ double lds[(kBlock / kWave) * 2 * DC_MAX_MODEL_TERMS];
  __shared__ double s_pose[kLdsScans * 12];
  const PoseTile poses = in.dirs ? stage_poses(in, s_pose) : PoseTile{nullptr};      // published by the staging barrier
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  const int64_t blk = xcd_block(nblocks);
  ModelParams mp;
  load_model(in, mp);
  double gw[DC_MAX_MODEL_TERMS], ge[DC_MAX_MODEL_TERMS], gT[6];
#pragma unroll
  for (int k = 0; k < DC_MAX_MODEL_TERMS; ++k) gw[k] = ge[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) gT[k] = 0.0;
  int scan = -1;
  const int64_t j = blk * kBlock + threadIdx.x;
  const bool active = blk >= 0 && j < n;
  typename Pt<PT>::Raw cj;
  PointRaw<T> raw;
  uint2 pre[kPreRuns];
  int32_t nruns = 0;
  uint32_t nd = 0;
  const uint2* runs = reinterpret_cast<const uint2*>(tab.loc);
  if (blk >= 0) {
    if (active) {
      // everything this lane needs later is requested before the staging loop: its latency hides behind it
      const int32_t r0 = tab.run_ptr[j];
      nruns = tab.run_ptr[j + 1] - r0;
      runs += r0;
      cj = Pt<PT>::from_row(reinterpret_cast<const int4*>(x) + j * XR);
      if (in.dirs) raw = load_point_raw<T>(in, mp, j);
    }
#pragma unroll
    for (int t = 0; t < kPreRuns; ++t) pre[t] = t < nruns ? runs[t] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    nd = (uint32_t)stage_rows<RR>(tab.blk_ptr, tab.blk_ids, blk, reinterpret_cast<const int4*>(rec), tile, cap);
    // row nd is an all-zero record: padding positions clamp to it and contribute exactly nothing
    if (threadIdx.x < RR) tile[threadIdx.x * cap + nd] = make_int4(0, 0, 0, 0);
  }
  const uint32_t nd16 = nd * 16u;
  if (WANT_POSE) pose_ticket_init();
  __syncthreads();
  if (active) {
    double g[3] = {0.0, 0.0, 0.0};
    const double u = Pt<PT>::unit(qp);
#pragma unroll
    for (int t = 0; t < kPreRuns; ++t)
      if (__any((int)(t < nruns))) run_edges<PT>(tile, cap, pre[t], nd16, cj, g);
    if (__any((int)(nruns > kPreRuns))) {                   // in-degrees above 16: runs fetched trip by trip, one ahead
      uint2 nxt = kPreRuns < nruns ? runs[kPreRuns] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
      for (int t = kPreRuns; __any((int)(t < nruns)); ++t) {
        const uint2 r = nxt;
        nxt = t + 1 < nruns ? runs[t + 1] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        run_edges<PT>(tile, cap, r, nd16, cj, g);
      }
    }
    g[0] *= u; g[1] *= u; g[2] *= u;
    if (grad_points) Row3<T, 4>::store(grad_points, j, g, QParams{});
    if (in.dirs) points_bwd_point<T>(in, poses, mp, j, raw, g, gw, ge, gT, want_e != 0, want_pose != 0, &scan);
  }
  if (in.dirs) reduce_param_grads<T>(in, active, want_e, want_pose, gw, ge, gT, scan, lds, partials, blk);
}

// ================================================================================================
// Basis form of the iteration (fixed poses, fixed exponents): every model of the reference is affine in its weights
// (Polynomial d' = d - sum w_k g^e_k, ScaledPolynomial d' = d (1 - sum w_k g^e_k), Linear, InvCos, ScaledInvCos:
// model.py:113-349), so the world point of ray j is
//     x_j(w) = X0_j + (sum_k w_k c_kj) u_j,     X0_j = R (vp + d0 dir) + t,   u_j = R dir,   c_kj = dd'/dw_k   (zero outside the local mask)
// with X0, u and c constant while the poses do not move.  They are computed once (points_basis_kernel); an iteration then
// needs no pass over the points to refresh x: the forward forms the rows it stages (and its centre) from the basis rows on
// the fly, the backward its own point, and the chain to the weights is dL/dw_k = sum_j (g_j . u_j) c_kj -- no model, no
// pose, no incidence angles in the loop.  X0 lives on the q32 grid, u and c in float32 (the correction sum w c is
// centimetres, so its fp32 rounding is ~1e-9 m, far below the grid).  A coordinate is the grid value
// X0 + rint((sum w_k c_k) u / step): the same integer for every block that forms it, rounded twice (X0 and the increment)
// instead of once.  A row is 24 + 4 P bytes: 32 for the two-term models, one aligned sector per gathered point.
// ================================================================================================
struct PointBasis {
  const void* __restrict__ rows;       // [n, 6 + P] words: X0, u, c_0 .. c_{P-1} (Basis<PT>: int32 / float32 bits for q32, fp64 for double)
  const double* __restrict__ w;        // [P] device weights of this evaluation
  int n_terms;
  double w_scale;                      // weights are staged as w_k * w_scale: 1 / grid step for q32, 1 for fp64 points
};

// s_w[k] = w_k * w_scale for the lanes of the block (call before a barrier); coherent: the weights were written by other
// blocks of this very launch (chained steps), so the load must not be served by this XCD's L2
__device__ __forceinline__ void stage_weights(const PointBasis& pb, double* s_w, bool coherent = false) {
  if ((int)threadIdx.x < pb.n_terms) {
    const double w = coherent ? __hip_atomic_load(pb.w + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : pb.w[threadIdx.x];
    s_w[threadIdx.x] = w * pb.w_scale;
  }
}

// Row layout and arithmetic of the basis per point format.  q32 (float32 clouds): X0 on the fixed-point grid, u and c in
// float32 (the correction sum w c is centimetres, so its fp32 rounding is ~1e-9 m, far below the grid); a coordinate is
// X0 + rint((sum w_k c_k) u / step).  double (float64 clouds, the reference's default float_type): everything fp64,
// x = X0 + (sum w_k c_k) u -- the same point as R (vp + d' dir) + t up to the order of the fp64 operations.
template <typename PT> struct Basis;
template <> struct Basis<q32> {
  using T = float;                                       // dtype of the cloud's arrays
  // P > 0: term count known at compile time (one contiguous row, loads issued together), P = 0: run-time count
  template <int P>
  static __device__ __forceinline__ Pt<q32>::Raw point(const PointBasis& pb, const double* wq, int64_t row) {
    const int np = P > 0 ? P : pb.n_terms;
    const int32_t* r = static_cast<const int32_t*>(pb.rows) + row * (6 + np);
    int32_t q[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) q[c] = r[c];
    // float32 throughout (one formula for every kernel that forms a point from its basis row, so that all of them form the
    // same integer): the correction sum is < 2^20 grid steps, its float32 rounding a few hundredths of a step
    float sc = 0.0f;
    if constexpr (P > 0) {
      float c[P];
#pragma unroll
      for (int k = 0; k < P; ++k) c[k] = __int_as_float(r[6 + k]);
#pragma unroll
      for (int k = 0; k < P; ++k) sc = fmaf((float)wq[k], c[k], sc);
    } else {
      for (int k = 0; k < np; ++k) sc = fmaf((float)wq[k], __int_as_float(r[6 + k]), sc);
    }
    Pt<q32>::Raw o;
#pragma unroll
    for (int a = 0; a < 3; ++a) o.v[a] = q[a] + (int32_t)rintf(sc * __int_as_float(q[3 + a]));
    return o;
  }
  static __device__ __forceinline__ void stage(int4* tile, int, int t, const Pt<q32>::Raw& r) { tile[t] = make_int4(r.v[0], r.v[1], r.v[2], 0); }
  // gw[k] += (g . u_j) c_kj for the point's own row (g in metres^-1 units of the loss)
  template <int NP>
  static __device__ __forceinline__ void chain(const PointBasis& pb, int np, int64_t j, const double* g, double* gw) {
    const int32_t* r = static_cast<const int32_t*>(pb.rows) + j * (6 + np);
    const double gu = g[0] * (double)__int_as_float(r[3]) + g[1] * (double)__int_as_float(r[4]) + g[2] * (double)__int_as_float(r[5]);
#pragma unroll
    for (int k = 0; k < NP; ++k)
      if (k < np) gw[k] = gu * (double)__int_as_float(r[6 + k]);
  }
  static __device__ __forceinline__ void write(void* rows, int64_t i, int nt, const double* x0, const double* u, const QParams& qp) {
    int32_t* r = static_cast<int32_t*>(rows) + i * (6 + nt);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      r[a] = quantize(x0[a], qp.origin[a], qp.inv_scale, qp.flag);
      r[3 + a] = __float_as_int((float)u[a]);
    }
  }
  static __device__ __forceinline__ void write_term(void* rows, int64_t i, int nt, int k, double c) {
    static_cast<int32_t*>(rows)[i * (6 + nt) + 6 + k] = __float_as_int((float)c);
  }
};
template <> struct Basis<double> {
  using T = double;
  template <int P>
  static __device__ __forceinline__ Pt<double>::Raw point(const PointBasis& pb, const double* wq, int64_t row) {
    const int np = P > 0 ? P : pb.n_terms;
    const double* r = static_cast<const double*>(pb.rows) + row * (6 + np);
    double q[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) q[c] = r[c];
    double sc = 0.0;
    if constexpr (P > 0) {
      double c[P];
#pragma unroll
      for (int k = 0; k < P; ++k) c[k] = r[6 + k];
#pragma unroll
      for (int k = 0; k < P; ++k) sc += wq[k] * c[k];
    } else {
      for (int k = 0; k < np; ++k) sc += wq[k] * r[6 + k];
    }
    Pt<double>::Raw o;
#pragma unroll
    for (int a = 0; a < 3; ++a) o.v[a] = q[a] + sc * q[3 + a];
    return o;
  }
  // 32-B rows, piece-major like stage_rows<2>: (x, y) at tile[t], (z, -) at tile[cap + t]
  static __device__ __forceinline__ void stage(int4* tile, int cap, int t, const Pt<double>::Raw& r) {
    tile[t] = make_int4(__double2loint(r.v[0]), __double2hiint(r.v[0]), __double2loint(r.v[1]), __double2hiint(r.v[1]));
    tile[cap + t] = make_int4(__double2loint(r.v[2]), __double2hiint(r.v[2]), 0, 0);
  }
  template <int NP>
  static __device__ __forceinline__ void chain(const PointBasis& pb, int np, int64_t j, const double* g, double* gw) {
    const double* r = static_cast<const double*>(pb.rows) + j * (6 + np);
    const double gu = g[0] * r[3] + g[1] * r[4] + g[2] * r[5];
#pragma unroll
    for (int k = 0; k < NP; ++k)
      if (k < np) gw[k] = gu * r[6 + k];
  }
  static __device__ __forceinline__ void write(void* rows, int64_t i, int nt, const double* x0, const double* u, const QParams&) {
    double* r = static_cast<double*>(rows) + i * (6 + nt);
#pragma unroll
    for (int a = 0; a < 3; ++a) { r[a] = x0[a]; r[3 + a] = u[a]; }
  }
  static __device__ __forceinline__ void write_term(void* rows, int64_t i, int nt, int k, double c) {
    static_cast<double*>(rows)[i * (6 + nt) + 6 + k] = c;
  }
};

// the lane's centre from the staged rows (row `t` of the block's distinct list)
template <typename PT>
__device__ __forceinline__ typename Pt<PT>::Raw staged_point(const int4* tile, int cap, int t) {
  int4 piece[Pt<PT>::kRow16];
  read_row<Pt<PT>::kRow16>(tile, cap, (uint32_t)t * 16u, piece);
  return Pt<PT>::from_row(piece);
}

// X0, u and c of every point (once per pose set): the same inputs and arithmetic as points_fwd_kernel.
template <typename PT>
__global__ __launch_bounds__(kBlock) void points_basis_kernel(PointInputs in, int64_t n, QParams qp, void* __restrict__ rows) {
  using T = typename Basis<PT>::T;
  __shared__ double s_pose[kLdsScans * 12];
  const PoseTile poses = stage_poses(in, s_pose);
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  ModelParams mp;
  load_model(in, mp);
  double vp[3], dr[3], T12[12];
  if (in.vps) Row3<T, 3>::load((const T*)in.vps, i, vp, qp);
  else { vp[0] = vp[1] = vp[2] = 0.0; }
  Row3<T, 3>::load((const T*)in.dirs, i, dr, qp);
  const double d = (double)((const T*)in.depth)[i];
  const bool lm = in.lmask ? in.lmask[i] != 0 : true;
  const double inc = (mp.kind != DC_MODEL_NONE && lm) ? (double)((const T*)in.inc)[i] : 0.0;
  load_pose(in, poses, in.scan_id ? in.scan_id[i] : 0, T12);
  double vr[3], drr[3], x0[3];
  rot3(T12, vp, vr);
  vr[0] += T12[3]; vr[1] += T12[7]; vr[2] += T12[11];
  rot3(T12, dr, drr);
  const bool on = mp.kind != DC_MODEL_NONE && lm;
  const double d0 = (on && mp.kind == DC_MODEL_LINEAR) ? 0.0 : d;      // d' at w = 0
#pragma unroll
  for (int a = 0; a < 3; ++a) x0[a] = vr[a] + d0 * drr[a];
  Basis<PT>::write(rows, i, mp.n_terms, x0, drr, qp);
#pragma unroll
  for (int k = 0; k < DC_MAX_MODEL_TERMS; ++k) {
    if (k < mp.n_terms) {
      double dk = 0.0;                                                  // dd'/dw_k
      if (on) {
        if (mp.kind > DC_MODEL_SCALED_POLYNOMIAL) dk = model_dw_other(mp, k, d, inc);
        else dk = (mp.kind == DC_MODEL_SCALED_POLYNOMIAL ? -d : -1.0) * pow_term(inc, mp.e[k]);
      }
      Basis<PT>::write_term(rows, i, mp.n_terms, k, dk);
    }
  }
}

template <typename PT, bool FULL_EIG, int NS, int P>
__global__ __launch_bounds__(kBlock) void consistency_fwd_basis_kernel(
    PointBasis pb, BlockTab tab, const int32_t* __restrict__ own_base, int cap, const int32_t* __restrict__ centre_idx, int64_t n,
    const uint8_t* __restrict__ mask, const typename Basis<PT>::T* __restrict__ offset, LossParams lp, QParams qp, PT* __restrict__ rec,
    typename Basis<PT>::T* __restrict__ pointwise, typename Basis<PT>::T* __restrict__ eigvals, double* __restrict__ partials) {
  using T = typename Basis<PT>::T;
  extern __shared__ int4 tile[];
  __shared__ double s_w[DC_MAX_MODEL_TERMS];
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  const int64_t blk = xcd_block(nblocks);
  double acc2[2] = {0.0, 0.0};
  const int32_t s0 = blk >= 0 ? tab.slot_ptr[blk] : 0;
  // a table with another slot count than the launch was specialised for (not a table of [rows, NS]): fail loudly
  const bool bad = blk >= 0 && tab.slot_ptr[blk + 1] - s0 != NS;
  if (blk >= 0 && !bad) {
    const int64_t i = blk * kBlock + threadIdx.x;
    const bool live = i < n;
    const uint16_t* lrow = tab.loc + (int64_t)s0 * kBlock + threadIdx.x;
    uint32_t pre[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) pre[q] = (uint32_t)lrow[q * kBlock];
    stage_weights(pb, s_w);
    const int32_t base = tab.blk_ptr[blk], nd = tab.blk_ptr[blk + 1] - base;
    // the block's own rows sit contiguously in its list (k-NN: every point is its own neighbour): the centre comes from LDS
    const int32_t own = (own_base && !centre_idx) ? own_base[blk] : -1;
    __syncthreads();
    double wq[P > 0 ? P : DC_MAX_MODEL_TERMS];
#pragma unroll
    for (int k = 0; k < (P > 0 ? P : DC_MAX_MODEL_TERMS); ++k) wq[k] = (P > 0 || k < pb.n_terms) ? s_w[k] : 0.0;
    for (int t = threadIdx.x; t < nd; t += kBlock)
      Basis<PT>::stage(tile, cap, t, Basis<PT>::template point<P>(pb, wq, tab.blk_ids[base + t]));
    typename Pt<PT>::Raw ci;
    if (own < 0) ci = Basis<PT>::template point<P>(pb, wq, live ? (centre_idx ? (int64_t)centre_idx[i] : i) : 0);
    __syncthreads();
    if (own >= 0) ci = staged_point<PT>(tile, cap, own + (live ? (int)threadIdx.x : 0));
    if (live) {
      CovAcc acc;
      cov_init(acc);
      uint32_t mx = pre[0];
#pragma unroll
      for (int q = 1; q < NS; ++q) mx = max(mx, pre[q]);
      int n_have;
      if (__any((int)(mx == kNoLoc))) n_have = gather_fixed<PT, NS, true>(tile, cap, ci, pre, acc);
      else n_have = gather_fixed<PT, NS, false>(tile, cap, ci, pre, acc);
      acc.W = (double)n_have;
      consistency_point<T, PT, FULL_EIG>(acc, ci, i, mask, offset, lp, qp, rec, pointwise, eigvals, acc2);
    }
  } else {
    __syncthreads();
    __syncthreads();
  }
  if (bad) acc2[0] = acc2[1] = __longlong_as_double(0x7ff8000000000000ll);
  wave_partials<2>(acc2, partials);
}

// The same for any slot count (radius neighbourhoods: the reference's default nn_r = 0.25, K = the largest count; or the
// run-time-slot ablation): the slot loop of consistency_fwd_staged_kernel over rows formed from the basis.
template <typename PT, bool FULL_EIG, int P>
__global__ __launch_bounds__(kBlock) void consistency_fwd_basis_slots_kernel(
    PointBasis pb, BlockTab tab, const int32_t* __restrict__ own_base, int cap, const int32_t* __restrict__ centre_idx, int64_t n,
    const uint8_t* __restrict__ mask, const typename Basis<PT>::T* __restrict__ offset, LossParams lp, QParams qp, PT* __restrict__ rec,
    typename Basis<PT>::T* __restrict__ pointwise, typename Basis<PT>::T* __restrict__ eigvals, double* __restrict__ partials) {
  using T = typename Basis<PT>::T;
  extern __shared__ int4 tile[];
  __shared__ double s_w[DC_MAX_MODEL_TERMS];
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  const int64_t blk = xcd_block(nblocks);
  double acc2[2] = {0.0, 0.0};
  const int64_t i = blk * kBlock + threadIdx.x;
  const bool live = blk >= 0 && i < n;
  int32_t nslots = 0, own = -1;
  const uint16_t* lrow = tab.loc;
  uint32_t pre[kPreSlots];
  stage_weights(pb, s_w);
  if (blk >= 0) {
    const int32_t s0 = tab.slot_ptr[blk];
    nslots = tab.slot_ptr[blk + 1] - s0;
    lrow = tab.loc + (int64_t)s0 * kBlock + threadIdx.x;
#pragma unroll
    for (int q = 0; q < kPreSlots; ++q) pre[q] = (live && q < nslots) ? (uint32_t)lrow[q * kBlock] : kNoLoc;
    own = (own_base && !centre_idx) ? own_base[blk] : -1;
  }
  __syncthreads();
  double wq[P > 0 ? P : DC_MAX_MODEL_TERMS];
#pragma unroll
  for (int k = 0; k < (P > 0 ? P : DC_MAX_MODEL_TERMS); ++k) wq[k] = (P > 0 || k < pb.n_terms) ? s_w[k] : 0.0;
  typename Pt<PT>::Raw ci;
  if (blk >= 0) {
    const int32_t base = tab.blk_ptr[blk], nd = tab.blk_ptr[blk + 1] - base;
    for (int t = threadIdx.x; t < nd; t += kBlock)
      Basis<PT>::stage(tile, cap, t, Basis<PT>::template point<P>(pb, wq, tab.blk_ids[base + t]));
    if (own < 0) ci = Basis<PT>::template point<P>(pb, wq, live ? (centre_idx ? (int64_t)centre_idx[i] : i) : 0);
  }
  __syncthreads();
  if (live) {
    if (own >= 0) ci = staged_point<PT>(tile, cap, own + (int)threadIdx.x);
    CovAcc acc;
    cov_init(acc);
    bool miss = false;
#pragma unroll
    for (int q = 0; q < kPreSlots; ++q) miss |= (q < nslots) && pre[q] == kNoLoc;
    int n_have = 0;
    if (__any((int)miss)) n_have = gather_slots<PT, true>(tile, cap, ci, pre, nslots, acc);
    else n_have = gather_slots<PT, false>(tile, cap, ci, pre, nslots, acc);
    for (int q = kPreSlots; q < nslots; ++q) {             // K > 16: one slot at a time
      const uint32_t l = lrow[q * kBlock];
      n_have += slot_add<PT, true>(tile, cap, ci, l, acc);
    }
    acc.W = (double)n_have;
    consistency_point<T, PT, FULL_EIG>(acc, ci, i, mask, offset, lp, qp, rec, pointwise, eigvals, acc2);
  }
  wave_partials<2>(acc2, partials);
}

// torch.optim.Adam (single-tensor path, no amsgrad) for one fp64 parameter; grad is scaled first.
struct AdamArgs {
  double* p; double* m; double* v;     // parameters, exp_avg, exp_avg_sq (p == nullptr: no update)
  int n;
  double grad_scale, lr, b1, b2, eps, weight_decay, bias1, bias2_sqrt;
};
// (p0, m0, v0: the parameter's state, loaded by the caller -- early, so the trip hides behind its own work)
__device__ __forceinline__ void adam_apply(const AdamArgs& a, int i, double grad, double p0, double m0, double v0) {
  double g = grad * a.grad_scale;
  if (a.weight_decay != 0.0) g += a.weight_decay * p0;
  const double mi = m0 + (g - m0) * (1.0 - a.b1);                   // exp_avg.lerp_(grad, 1 - beta1)
  const double vi = v0 * a.b2 + (1.0 - a.b2) * g * g;               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  a.m[i] = mi; a.v[i] = vi;
  const double denom = sqrt(vi) / a.bias2_sqrt + a.eps;
  a.p[i] = p0 + (-(a.lr / a.bias1)) * (mi / denom);
}
__device__ __forceinline__ void adam_update(const AdamArgs& a, int i, double grad) { adam_apply(a, i, grad, a.p[i], a.m[i], a.v[i]); }

// ---- chained steps: the previous evaluation's final sums inside the next evaluation's launch -----------------------------
// A dependent reduction launch after a kernel that filled every L2 costs ~9 us, an eighth of a C2 step.  In a chain
// of steps the launch of step t + 1 therefore starts with `n_front` leading blocks that finish step t: block a < 2 + P sums
// column a of step t's partial rows (one per BLOCK in this mode: 7.8 k rows, one trip for 256 lanes), writes out_prev[a],
// takes weight (a - 2)'s Adam step and raises ready[parity]; the other blocks fetch everything that does not depend on the
// weights, then wait for ready[parity] == P.  The leading blocks are dispatched first and need nothing from the waiting ones,
// so they always finish; the wait is bounded all the same (a grid must drain).  parity alternates per launch: this launch
// clears the other flag and writes its own partial rows to the other buffer.  The last step of a chain is finished by the
// ordinary reduction launch (flush).
struct StepChain {
  int32_t* ready;            // 64 bytes, zero before the first launch of a chain (the published weights, chain_publish); nullptr: ordinary
                             // launch (per-wavefront rows)
  uint32_t stamp;            // this launch's number: what marks a published word as belonging to it
  int parity, has_prev, n_front, n_out;
  int reverse;               // this launch walks every XCD's share of the blocks backwards (chain_block_of)
  const double* prev;        // the previous launch's rows [(2 + P)][prev_rows]
  int64_t prev_rows;
  const double* grad_sum;    // or: the previous evaluation's dL/dw already summed (over the ranks: an all-reduce ran in between);
                             // the leading blocks then only take the Adam update (out_prev is not written)
  double* out_prev;          // [n_out] <- sums of the previous evaluation (slots beyond 2 + P: 0)
  double* w_prev_out;        // [P] or nullptr <- the weights the previous evaluation used (a training log records them: train.py)
  int32_t* status;           // bit 0: a point left the q32 extent (read); bit 1: a wait for the weights ran out (raised here)
  int spin_limit;            // polls a waiting block makes before it gives up (dc_set_option(5, n); 0: gives up at once)
  const double* w_now;       // the caller's weights (what a launch with nothing to finish publishes)
  const int32_t* prev_status;  // linked chains: the status word of the sequence whose rows are finished (nullptr: `status`)
  const double* acc_in;      // [2 + P] or nullptr: added to the previous launch's sums (the sequences of ONE loss evaluated launch after
                             // launch: the running sums of the step's earlier sequences)
  AdamArgs adam;             // the update the previous evaluation's gradient feeds (bias corrections of ITS step)
  const uint8_t* blk_skip;   // [blocks] or nullptr: blocks none of whose centres is inside the loss mask (dcSequenceDesc.blk_skip): they add
                             // nothing to the loss, the count or dL/dw and are treated like the padding blocks of the last round
};
constexpr int32_t kStatusChainTimeout = 2;     // (bit 0: q32 overflow, raised by quantize())
constexpr int kChainFront = 8;             // leading blocks of a chained launch (a multiple of the XCD count)

// The logical block of this workgroup (xcd_block_of); -1: padding.  Every other launch of a chain of fixed-K steps walks its XCD's
// share of the blocks BACKWARDS: the basis rows are the same in every step, an XCD's 4 MB of L2 still holds the rows of the last
// ~260 blocks it finished, and a launch's first round -- 1 536 blocks that all start by fetching rows -- is its slowest
// (profiles/r04_block_trace.md: 12 us per block against 7.7 later on).  Walking back, the first round finds its rows in L2:
// C2 step 42.8 -> 41.4 us.  Block -> row of the partial sums is by grid position either way, so the order of the final
// additions differs between the two directions by rounding only (deterministic: the direction is the launch's parity).
__device__ __forceinline__ int64_t chain_block_of(const StepChain& ch, bool chained, int64_t nblocks) {
  const int64_t b = (int64_t)blockIdx.x - (chained ? ch.n_front : 0);
  if (!(chained && ch.reverse)) return xcd_block_of(b, nblocks);
  const int64_t per = (nblocks + kXcds - 1) / kXcds;
  const int64_t logical = (b % kXcds) * per + (per - 1 - b / kXcds);
  return logical < nblocks ? logical : -1;
}

// Weight k of a chained launch, published by the leading block that finished it and picked up by every other block in ONE
// memory trip: the double travels as two 64-bit words {low half | stamp}, {high half | stamp} (8-byte accesses are single-copy
// atomic), the stamp being the launch's number -- a word carrying it can only be this launch's.  (Rounds 2-4: a counter raised
// after a fence, polled, and then the weights loaded: two dependent trips through the fabric in front of every block's staging.)
__device__ __forceinline__ void chain_publish(const StepChain& ch, int k, double w) {
  unsigned long long* pub = reinterpret_cast<unsigned long long*>(ch.ready);
  const unsigned long long st = (unsigned long long)ch.stamp << 32;
  __hip_atomic_store(pub + 2 * k, st | (unsigned)__double2loint(w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(pub + 2 * k + 1, st | (unsigned)__double2hiint(w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int P>
__device__ __forceinline__ void chain_front_block(const StepChain& ch, double* lds /* [kBlock / kWave] in LDS */) {
  const int a = blockIdx.x;
  if (a == 0 && ch.has_prev && !ch.grad_sum)
    for (int z = 2 + P + threadIdx.x; z < ch.n_out; z += kBlock) ch.out_prev[z] = 0.0;
  if (a >= 2 + P) return;
  if (ch.grad_sum) {                     // the sums exist already: weight a - 2's update, then publish
    if (a >= 2 && threadIdx.x == 0) {
      if (ch.has_prev && ch.adam.p) adam_update(ch.adam, a - 2, ch.grad_sum[a - 2]);
      const double wk = ch.adam.p ? ch.adam.p[a - 2] : ch.w_now[a - 2];
      chain_publish(ch, a - 2, wk);
      if (ch.w_prev_out) ch.w_prev_out[a - 2] = wk;      // (this form records the weights THIS evaluation uses: its sums are current)
    }
    return;
  }
  const bool step = ch.has_prev && ch.adam.p && a >= 2 && threadIdx.x == 0;
  const int32_t* st_prev = ch.prev_status ? ch.prev_status : ch.status;
  const bool flagged = ch.has_prev && a == 0 && threadIdx.x == 0 && st_prev &&
                       __hip_atomic_load(st_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  double p0 = 0.0, m0 = 0.0, v0 = 0.0;
  if (step) {
    p0 = ch.adam.p[a - 2]; m0 = ch.adam.m[a - 2]; v0 = ch.adam.v[a - 2];
    if (ch.w_prev_out) ch.w_prev_out[a - 2] = p0;
  }
  double s = 0.0;
  if (ch.has_prev) {
    const double* p = ch.prev + (int64_t)a * ch.prev_rows;
    constexpr int U = 32;
    for (int64_t r0 = threadIdx.x; r0 < ch.prev_rows; r0 += (int64_t)U * kBlock) {
      double v[U];
#pragma unroll
      for (int u_ = 0; u_ < U; ++u_) v[u_] = (r0 + (int64_t)u_ * kBlock < ch.prev_rows) ? p[r0 + (int64_t)u_ * kBlock] : 0.0;
#pragma unroll
      for (int w_ = U / 2; w_ > 0; w_ >>= 1) {
#pragma unroll
        for (int u_ = 0; u_ < w_; ++u_) v[u_] += v[u_ + w_];
      }
      s += v[0];
    }
    s = wave_sum(s);
  }
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == 0) lds[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (ch.has_prev) {
      double t = 0.0;
      for (int wv = 0; wv < kBlock / kWave; ++wv) t += lds[wv];
      if (ch.acc_in) t += ch.acc_in[a];                      // (read before out_prev is written: the two may be one buffer)
      if (flagged) t = __longlong_as_double(0x7ff8000000000000ll);
      ch.out_prev[a] = t;
      if (step) adam_apply(ch.adam, a - 2, t, p0, m0, v0);
    }
    // weight a - 2 is final for this launch (the value just stored, or the caller's when there was nothing to finish)
    if (a >= 2) chain_publish(ch, a - 2, step ? ch.adam.p[a - 2] : ch.w_now[a - 2]);
  }
}

// chain_front_block on its own: finishes the last launch of a linked chain (dc_sequence_chain_flush_linked)
template <int P>
__global__ __launch_bounds__(kBlock) void chain_front_only_kernel(StepChain ch) {
  __shared__ double s_front[kBlock / kWave];
  chain_front_block<P>(ch, s_front);
}

// s_w[k] <- w_k * w_scale of this launch for the lanes of the block, as soon as its leading blocks have published them
// (bounded wait); *s_ok <- 0 when a wait ran out.  Every thread of the block calls it; the caller's barrier follows.
template <int P>
__device__ __forceinline__ void chain_weights(const StepChain& ch, double w_scale, double* s_w, int* s_ok) {
  const int tid = threadIdx.x;
  if (tid == 0) *s_ok = 1;
  if (tid < 2 * P) {
    const unsigned long long* pub = reinterpret_cast<const unsigned long long*>(ch.ready) + tid;
    unsigned long long word = 0;
    bool ok = false;
    for (int spin = 0; spin < ch.spin_limit; ++spin) {
      // (relaxed: an agent-scope acquire would invalidate this XCD's L2 on every poll -- nothing else is read through it)
      word = __hip_atomic_load(pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((uint32_t)(word >> 32) == ch.stamp) { ok = true; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    const int half = (int)(uint32_t)word;
    const int other = __shfl_xor(half, 1, kWave);                       // lanes 2k / 2k + 1: low / high half of weight k
    if ((tid & 1) == 0) s_w[tid >> 1] = (ok ? __hiloint2double(other, half) : __longlong_as_double(0x7ff8000000000000ll)) * w_scale;
    // a wait that ran out poisons this launch's sums (NaN) AND says so: the status word tells it apart from a q32 overflow
    if (!ok) {
      *s_ok = 0;
      if (ch.status) atomicOr(ch.status, kStatusChainTimeout);
    }
  }
}

// ---- loss AND dL/dw in one pass (forward-mode accumulation) -------------------------------------------------------------
// With only the P model weights to differentiate, the reverse pass over the transposed table is not needed:
//     dL/dw_k = sum_i sum_{j in N(i)} (dl_i/dx_j) . (dx_j/dw_k),   dl_i/dx_j = c1_i (v0_i . d) v0_i - c2_i d,  d = x_j - cmean_i,
//     dx_j/dw_k = c_kj u_j
// is a second sweep of centre i over its OWN neighbours, whose rows (x_j and now also u_j, c_kj) already sit in LDS.  No
// backward record is written or read (64 + 64 MB per iteration at C2), no transposed table, no second launch; the terms
// are the ones the backward kernel adds up, grouped by centre instead of by point, with c1 / c2 / cmean in fp64.
// Staged row (piece-major, 16-B pieces; piece 0 starts with the point in its usual row format, so the first sweep and the
// centre read it as before):  q32: {x0, x1, x2, u0 | u1, u2, c0, c1 | c2}, x on the grid, u / c float32 bits;
// double: {x0, x1 | x2, u0' u1' | u2', c0', c1', c2'} (u', c': float32 copies for the second sweep).
template <typename PT, int P> struct StepRow;
template <int P> struct StepRow<q32, P> {
  static constexpr int kPieces = (6 + P + 3) / 4;
  struct Raw { int32_t q[6 + P]; };                       // a basis row as fetched (before the weights are known)
  static __device__ __forceinline__ Raw fetch(const PointBasis& pb, int64_t row) {
    const int32_t* r = static_cast<const int32_t*>(pb.rows) + row * (6 + P);
    Raw o;
#pragma unroll
    for (int c = 0; c < 6 + P; ++c) o.q[c] = r[c];
    return o;
  }
  static __device__ __forceinline__ void stage(const PointBasis& pb, const double* wq, int64_t row, int4* tile, int cap, int t) {
    place(fetch(pb, row), wq, tile, cap, t);
  }
  static __device__ __forceinline__ void place(const Raw& raw, const double* wq, int4* tile, int cap, int t) {
    const int32_t* q = raw.q;
    float sc = 0.0f;                                      // exactly Basis<q32>::point's arithmetic
#pragma unroll
    for (int k = 0; k < P; ++k) sc = fmaf((float)wq[k], __int_as_float(q[6 + k]), sc);
    int32_t x[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) x[a] = q[a] + (int32_t)rintf(sc * __int_as_float(q[3 + a]));
    tile[t] = make_int4(x[0], x[1], x[2], q[3]);
    tile[cap + t] = make_int4(q[4], q[5], q[6], P > 1 ? q[P > 1 ? 7 : 6] : 0);
    if constexpr (P > 2) tile[2 * cap + t] = make_int4(q[8], 0, 0, 0);
  }
  // the neighbourhood mean in the staged rows' units: x_i + cm (grid steps; exact in fp64)
  static __device__ __forceinline__ void mean_of(const Pt<q32>::Raw& ci, const double* cm, double* mean) {
#pragma unroll
    for (int a = 0; a < 3; ++a) mean[a] = (double)ci.v[a] + cm[a];
  }
  // e = x_j - mean (grid steps), u_j, c_kj of the staged row at byte offset `off`
  static __device__ __forceinline__ void load(const int4* tile, int cap, uint32_t off, const double* mean, double* e, double* u, double* c) {
    const char* row = reinterpret_cast<const char*>(tile) + off;
    const int4 p0 = *reinterpret_cast<const int4*>(row);
    const int4 p1 = *reinterpret_cast<const int4*>(row + (size_t)cap * 16);
    e[0] = (double)p0.x - mean[0]; e[1] = (double)p0.y - mean[1]; e[2] = (double)p0.z - mean[2];
    u[0] = (double)__int_as_float(p0.w); u[1] = (double)__int_as_float(p1.x); u[2] = (double)__int_as_float(p1.y);
    c[0] = (double)__int_as_float(p1.z);
    if constexpr (P > 1) c[1] = (double)__int_as_float(p1.w);
    if constexpr (P > 2) c[2] = (double)__int_as_float(reinterpret_cast<const int4*>(row + (size_t)cap * 32)->x);
  }
};
template <int P> struct StepRow<double, P> {
  // Staged row of a float64 cloud (round 5): {x0, x1 | x2, u0' u1' | u2', c0', c1', c2'} -- the point in fp64 as before (the first
  // sweep and the centre read pieces 0 and 1: the loss is what it was, bit for bit), u and c as float32 COPIES for the second sweep:
  // 48 B instead of 64, three LDS reads per neighbour there instead of four.  The kernel is bound by LDS reads at random rows
  // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.62, LDS busy two thirds of the launch).  The point itself is formed from the
  // fp64 basis row; the float32 copies enter dL/dw only: a relative rounding of 6e-8 per term, of random sign over 2e7 terms.
  static constexpr int kPieces = 3;
  static_assert(P <= 3, "three float32 weights' terms fit the third piece");
  static __device__ __forceinline__ int4 pack(double a, double b) {
    return make_int4(__double2loint(a), __double2hiint(a), __double2loint(b), __double2hiint(b));
  }
  struct Raw { double q[6 + P]; };
  static __device__ __forceinline__ Raw fetch(const PointBasis& pb, int64_t row) {
    const double* r = static_cast<const double*>(pb.rows) + row * (6 + P);
    Raw o;
#pragma unroll
    for (int c = 0; c < 6 + P; ++c) o.q[c] = r[c];
    return o;
  }
  static __device__ __forceinline__ void stage(const PointBasis& pb, const double* wq, int64_t row, int4* tile, int cap, int t) {
    place(fetch(pb, row), wq, tile, cap, t);
  }
  static __device__ __forceinline__ void place(const Raw& raw, const double* wq, int4* tile, int cap, int t) {
    double q[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) q[c] = c < 6 + P ? raw.q[c] : 0.0;
    double sc = 0.0;
#pragma unroll
    for (int k = 0; k < P; ++k) sc += wq[k] * q[6 + k];
#pragma unroll
    for (int a = 0; a < 3; ++a) q[a] += sc * q[3 + a];
    tile[t] = pack(q[0], q[1]);
    tile[cap + t] = make_int4(__double2loint(q[2]), __double2hiint(q[2]), __float_as_int((float)q[3]), __float_as_int((float)q[4]));
    tile[2 * cap + t] = make_int4(__float_as_int((float)q[5]), __float_as_int((float)q[6]), __float_as_int((float)q[7]), __float_as_int((float)q[8]));
  }
  static __device__ __forceinline__ void mean_of(const Pt<double>::Raw& ci, const double* cm, double* mean) {
#pragma unroll
    for (int a = 0; a < 3; ++a) mean[a] = ci.v[a] + cm[a];
  }
  static __device__ __forceinline__ void load(const int4* tile, int cap, uint32_t off, const double* mean, double* e, double* u, double* c) {
    const char* row = reinterpret_cast<const char*>(tile) + off;
    const int4 p0 = *reinterpret_cast<const int4*>(row);
    const int4 p1 = *reinterpret_cast<const int4*>(row + (size_t)cap * 16);
    const int4 p2 = *reinterpret_cast<const int4*>(row + (size_t)cap * 32);
    e[0] = __hiloint2double(p0.y, p0.x) - mean[0]; e[1] = __hiloint2double(p0.w, p0.z) - mean[1]; e[2] = __hiloint2double(p1.y, p1.x) - mean[2];
    u[0] = (double)__int_as_float(p1.z); u[1] = (double)__int_as_float(p1.w); u[2] = (double)__int_as_float(p2.x);
    c[0] = (double)__int_as_float(p2.y);
    if constexpr (P > 1) c[1] = (double)__int_as_float(p2.z);
    if constexpr (P > 2) c[2] = (double)__int_as_float(p2.w);
  }
};

// one neighbour's share of dL/dw: gw[k] += t c_kj, t = c1 (v . e)(v . u_j) - c2 (e . u_j); have = false: nothing
template <typename PT, int P>
__device__ __forceinline__ void chain_term(const int4* tile, int cap, uint32_t off, bool have, const double* mean, const double* v,
                                           double c1, double c2, double* gw) {
  double e[3], u[3], c[P];
  StepRow<PT, P>::load(tile, cap, have ? off : 0u, mean, e, u, c);
  const double al = v[0] * e[0] + v[1] * e[1] + v[2] * e[2];
  const double be = v[0] * u[0] + v[1] * u[1] + v[2] * u[2];
  const double ga = e[0] * u[0] + e[1] * u[1] + e[2] * u[2];
  double tj = c1 * al * be - c2 * ga;
  if (!have) tj = 0.0;
#pragma unroll
  for (int k = 0; k < P; ++k) gw[k] = fma(tj, c[k], gw[k]);
}

// everything of a centre after its moments are gathered: covariance -> smallest eigenpair -> loss (acc2) and the
// coefficients of its neighbours' terms; an empty neighbourhood (NaN mean, zero coefficients) contributes exactly nothing
template <typename PT>
__device__ __forceinline__ void step_point(CovAcc& acc, bool m, const LossParams& lp, const QParams& qp, double* acc2, double* cm,
                                           double* v0, double* c1, double* c2) {
  cov_same_weights(acc);
  const double u = Pt<PT>::unit(qp);
  double moff[3], C[6], D, omega, lam0, tr;
  cov_finish(acc, 0.0, moff, cm, C, &D, &omega, u * u);
  eig3_smallest_r2(C[0], C[1], C[2], C[3], C[4], C[5], &lam0, v0, &tr);      // (the A-B baseline form, dc_set_option(6, 0))
  const double l = loss_and_coeffs(lp, lam0, tr, D, 0.0, m, c1, c2);
  const bool drop = loss_dropped(lp, l);
  if (drop) *c1 = *c2 = 0.0;
  if (m && !drop) { acc2[0] = l; acc2[1] = 1.0; }
  if (!(*c1 != 0.0 || *c2 != 0.0)) { cm[0] = cm[1] = cm[2] = 0.0; v0[0] = v0[1] = v0[2] = 0.0; }
}

// ---- the slimmer forms of the one-pass kernel's per-centre work (VAR bits of consistency_step_basis_kernel) ----------------
constexpr int kVarF32Sweep = 1;     // second sweep in float32 (q32 points: differences, u and c are float32-exact already)
constexpr int kVarSlimTail = 2;     // covariance -> eigenpair -> loss without the intermediate normalisations (eig3_smallest_unit)
constexpr int kVarDppSums = 4;      // wavefront sums through DPP row operations instead of ds_bpermute shuffles
constexpr int kStepVar = kVarF32Sweep | kVarSlimTail | kVarDppSums;     // what every instantiation but the A-B baseline (0) uses

// Covariance, smallest eigenpair, loss and the coefficients c1, c2 of a centre from its moments about the centre point.
// The covariance is only ever needed divided by its trace (eig3_smallest_unit), so the Bessel / unit factor f = unit^2 / D
// multiplies the trace alone; `full` (wave-uniform): every lane of the wavefront has all NS neighbours, W and D are constants.
template <typename PT, int NS>
__device__ __forceinline__ void step_point2(const CovAcc& acc, int n_have, bool full, bool m, const LossParams& lp, const QParams& qp,
                                            double* acc2, double* cm, double* v0, double* c1, double* c2, double* sum_e2 = nullptr) {
  const double u = Pt<PT>::unit(qp);
  double invW, D, invD;
  if (full) {
    invW = 1.0 / NS; D = NS - 1.0; invD = 1.0 / (NS - 1.0);
  } else {
    const double W = (double)n_have;
    invW = recip1_(W);                                  // W = 0: inf * 0 -> NaN mean, like the reference's 0 / 0
    D = W - 1.0;
    D = D < 1e-6 ? 1e-6 : D;
    invD = recip1_(D);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) cm[a] = acc.s[a] * invW;
  double Cp[6];
  Cp[0] = fma(-acc.s[0], cm[0], acc.S[0]); Cp[1] = fma(-acc.s[0], cm[1], acc.S[1]); Cp[2] = fma(-acc.s[0], cm[2], acc.S[2]);
  Cp[3] = fma(-acc.s[1], cm[1], acc.S[3]); Cp[4] = fma(-acc.s[1], cm[2], acc.S[4]); Cp[5] = fma(-acc.s[2], cm[2], acc.S[5]);
  const double mp = (Cp[0] + Cp[3]) + Cp[5];            // trace in the units of the differences
  if (sum_e2) *sum_e2 = mp;                             // = sum_j |x_j - mean|^2
  const double f = (u * u) * invD;
  const double tr = mp * f;
  double lam_rel, inv_tr;
  if (!(mp > 0.0) || !(mp < (double)INFINITY)) {
    const bool zero = (mp == 0.0) && Cp[1] == 0.0 && Cp[2] == 0.0 && Cp[4] == 0.0;
    lam_rel = zero ? 0.0 : (double)NAN;
    inv_tr = 1e6;                                       // tr is 0 (or NaN): the clamp of loss.py:253 applies
    v0[0] = 1.0; v0[1] = 0.0; v0[2] = 0.0;
  } else {
    const double inv_m = recip1_(mp);
    eig3_smallest_unit<!std::is_same<PT, q32>::value>(Cp[0] * inv_m, Cp[1] * inv_m, Cp[2] * inv_m, Cp[3] * inv_m, Cp[4] * inv_m, Cp[5] * inv_m,
                                                       &lam_rel, v0);
    const double inv_u2 = std::is_same<PT, q32>::value ? qp.inv_scale * qp.inv_scale : 1.0;
    inv_tr = inv_m * (D * inv_u2);                      // 1 / tr = 1 / (mp f)
  }
  const double lam0 = lam_rel * tr;
  // loss_and_coeffs with the reciprocals at hand
  double raw, g_vv = 0.0, g_eye = 0.0;
  if (lp.kind == DC_LOSS_MIN_EIGVAL) {
    if (lp.normalization) {
      const double inv = tr < 1e-6 ? 1e6 : inv_tr;      // 1 / clamp(tr, 1e-6); NaN compares false
      raw = lam0 * inv;
      g_vv = inv;
      g_eye = (tr > 1e-6) ? -raw * inv : 0.0;
    } else {
      raw = lam0;
      g_vv = 1.0;
    }
  } else {
    raw = tr;
    g_eye = 1.0;
  }
  double l = raw;
  double a = (m && l > 0.0) ? 1.0 : 0.0;
  l = l > 0.0 ? l : (l != l ? l : 0.0);
  if (lp.sqrt_) {
    const double sq = sqrt(l);
    a = (l > 0.0) ? a * 0.5 / sq : 0.0;
    l = sq;
  }
  const bool drop = loss_dropped(lp, l);                    // (skip_nans / only_finite: not part of the reduction at all)
  const double fd = drop ? 0.0 : 2.0 * a * invD;
  *c1 = fd * g_vv;
  *c2 = -fd * g_eye;
  if (m && !drop) { acc2[0] = l; acc2[1] = 1.0; }
  // an empty neighbourhood (NaN mean) must contribute nothing to the second sweep: only possible when slots are missing
  if (!full && !(*c1 != 0.0 || *c2 != 0.0)) { cm[0] = cm[1] = cm[2] = 0.0; v0[0] = v0[1] = v0[2] = 0.0; }
}

// One neighbour's share of dL/dw in float32 (q32 rows): the difference to the centre is an exact int32, u and c are float32
// words already, and the per-centre factors are rounded once; the lane's sums stay float32 over its K neighbours and join the
// fp64 reduction afterwards.  vs = c1 v0, vu = v0, both float32.
template <int P>
__device__ __forceinline__ void chain_term_f32(const int4* tile, int cap, uint32_t off, bool have, const Pt<q32>::Raw& ci, const float* cmf,
                                               const float* vs, const float* vu, float c2f, float* gw) {
  const char* row = reinterpret_cast<const char*>(tile) + (have ? off : 0u);
  const int4 p0 = *reinterpret_cast<const int4*>(row);
  const int4 p1 = *reinterpret_cast<const int4*>(row + (size_t)cap * 16);
  const float e0 = (float)(p0.x - ci.v[0]) - cmf[0], e1 = (float)(p0.y - ci.v[1]) - cmf[1], e2 = (float)(p0.z - ci.v[2]) - cmf[2];
  const float u0 = __int_as_float(p0.w), u1 = __int_as_float(p1.x), u2 = __int_as_float(p1.y);
  const float al = fmaf(vs[2], e2, fmaf(vs[1], e1, vs[0] * e0));
  const float be = fmaf(vu[2], u2, fmaf(vu[1], u1, vu[0] * u0));
  const float ga = fmaf(e2, u2, fmaf(e1, u1, e0 * u0));
  float tj = fmaf(al, be, -(c2f * ga));
  if (!have) tj = 0.0f;
  gw[0] = fmaf(tj, __int_as_float(p1.z), gw[0]);
  if constexpr (P > 1) gw[1] = fmaf(tj, __int_as_float(p1.w), gw[1]);
  if constexpr (P > 2) gw[2] = fmaf(tj, __int_as_float(reinterpret_cast<const int4*>(row + (size_t)cap * 32)->x), gw[2]);
}

// the second sweep over the slots beyond the first 16, in the same trips of eight; a trip's float32 sums join the fp64 sums trip by
// trip (a row of two hundred neighbours is too long for one float32 running sum)
template <int P>
__device__ __forceinline__ void chain_tail_f32(const int4* tile, int cap, const uint16_t* lrow, int nslots, bool packed, const Pt<q32>::Raw& ci,
                                               const float* cmf, const float* vs, const float* vu, float c2f, double* gw) {
  uint32_t nxt[kTrip];
#pragma unroll
  for (int u_ = 0; u_ < kTrip; ++u_) nxt[u_] = (kPreSlots + u_ < nslots) ? (uint32_t)lrow[(kPreSlots + u_) * kBlock] : kNoLoc;
  for (int q0 = kPreSlots; q0 < nslots; q0 += kTrip) {
    uint32_t l[kTrip];
#pragma unroll
    for (int u_ = 0; u_ < kTrip; ++u_) l[u_] = nxt[u_];
    if (packed && __all((int)(l[0] == kNoLoc))) break;
#pragma unroll
    for (int u_ = 0; u_ < kTrip; ++u_) nxt[u_] = (q0 + kTrip + u_ < nslots) ? (uint32_t)lrow[(q0 + kTrip + u_) * kBlock] : kNoLoc;
    float g[P];
#pragma unroll
    for (int k = 0; k < P; ++k) g[k] = 0.0f;
#pragma unroll
    for (int u_ = 0; u_ < kTrip; ++u_) chain_term_f32<P>(tile, cap, l[u_], l[u_] != kNoLoc, ci, cmf, vs, vu, c2f, g);
#pragma unroll
    for (int k = 0; k < P; ++k) gw[k] += (double)g[k];
  }
}
template <typename PT, int P>
__device__ __forceinline__ void chain_tail(const int4* tile, int cap, const uint16_t* lrow, int nslots, bool packed, const double* mean,
                                           const double* v0, double c1, double c2, double* gw) {
  uint32_t nxt[kTrip];
#pragma unroll
  for (int u_ = 0; u_ < kTrip; ++u_) nxt[u_] = (kPreSlots + u_ < nslots) ? (uint32_t)lrow[(kPreSlots + u_) * kBlock] : kNoLoc;
  for (int q0 = kPreSlots; q0 < nslots; q0 += kTrip) {
    uint32_t l[kTrip];
#pragma unroll
    for (int u_ = 0; u_ < kTrip; ++u_) l[u_] = nxt[u_];
    if (packed && __all((int)(l[0] == kNoLoc))) break;
#pragma unroll
    for (int u_ = 0; u_ < kTrip; ++u_) nxt[u_] = (q0 + kTrip + u_ < nslots) ? (uint32_t)lrow[(q0 + kTrip + u_) * kBlock] : kNoLoc;
#pragma unroll
    for (int u_ = 0; u_ < kTrip; ++u_) chain_term<PT, P>(tile, cap, l[u_], l[u_] != kNoLoc, mean, v0, c1, c2, gw);
  }
}

// (Round 5, measured and dropped for float64 rows: loss AND dL/dw from ONE sweep.  t_j is bilinear in (1, cm) x d_j, so with
//  A_k = sum c_kj u_j, B_k = sum c_kj (d_j . u_j), M_k = sum c_kj u_j d_j^T the second sweep collapses to
//  c1 (v^T M_k v - (v . cm)(v . A_k)) - c2 (B_k - cm . A_k): every staged row read once, 64 B per neighbour instead of 96.  But the 26
//  fp64 accumulators beside the moments cost the occupancy the LDS saving was meant to buy: 246 registers = two wavefronts per SIMD and
//  101 us; held to 168 registers (three per SIMD) 117 us, against 69 us for the two sweeps at 126 registers.)
// ---- wavefront sums through DPP -------------------------------------------------------------------------------------------
// One 32-bit half of a double moved by a DPP row operation (quad permutes, rotations inside a row of 16 lanes)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  // (every lane has a source under these controls; bound_ctrl only spares the `old` operand its initialisation)
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
constexpr int kDppXor1 = 0xB1;      // quad_perm:[1,0,3,2]
constexpr int kDppXor2 = 0x4E;      // quad_perm:[2,3,0,1]
constexpr int kDppRor4 = 0x124;     // row_ror:4
constexpr int kDppRor8 = 0x128;     // row_ror:8
constexpr int kDppShl4 = 0x104;     // row_shl:4 (lane l reads lane l + 4 of its row)
constexpr int kDppQuad3 = 0xFF;     // quad_perm:[3,3,3,3]
// Four values per lane -> lane l < 4 of the wavefront holds the total of value bitrev2(l) (as wave_sum_packed<4>): two quad
// steps that also halve what a lane carries, two rotations inside the rows, two cross-row exchanges.
__device__ __forceinline__ double wave_sum4_dpp(double* v) {
  const int lane = threadIdx.x & (kWave - 1);
  const bool up1 = (lane & 1) != 0, up2 = (lane & 2) != 0;
  // bit 0: this lane keeps values {0, 1} (bit clear) or {2, 3} (bit set), the partner the others
  const double k0 = up1 ? v[2] : v[0], g0 = up1 ? v[0] : v[2];
  const double k1 = up1 ? v[3] : v[1], g1 = up1 ? v[1] : v[3];
  const double a0 = k0 + dpp_f64<kDppXor1>(g0);
  const double a1 = k1 + dpp_f64<kDppXor1>(g1);
  // bit 1: keeps the first of its two (bit clear) or the second
  const double kk = up2 ? a1 : a0, gg = up2 ? a0 : a1;
  double r = kk + dpp_f64<kDppXor2>(gg);
  r += dpp_f64<kDppRor4>(r);
  r += dpp_f64<kDppRor8>(r);
  r += __shfl_xor(r, 16, kWave);
  r += __shfl_xor(r, 32, kWave);
  return r;
}

// {sum loss, count} -> p_fwd columns, dL/dw -> p_bwd columns (same row stride: one row per wavefront), through one packed
// wavefront reduction of the 2 + P values
template <int P, bool DPP = false>
__device__ __forceinline__ void step_partials(const double* acc2, const double* gw, double* __restrict__ p_fwd, double* __restrict__ p_bwd,
                                              bool per_block = false, int n_front = 0) {
  constexpr int NV = 2 + P, NP2 = NV <= 4 ? 4 : 8;
  __shared__ double s_comb[kWavesPerBlock][NP2];
  double v[NP2];
  v[0] = acc2[0]; v[1] = acc2[1];
#pragma unroll
  for (int k = 0; k < NP2 - 2; ++k) v[2 + k] = k < P ? gw[k] : 0.0;
  double tot;
  if constexpr (DPP && NP2 == 4) tot = wave_sum4_dpp(v);
  else tot = wave_sum_packed<NP2>(v);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  int64_t rs = (int64_t)gridDim.x * kWavesPerBlock, row = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (per_block) {
    // chained steps: one row per block (a quarter of the rows for the next launch's leading blocks to sum)
    if (lane < NP2) s_comb[wave][lane] = tot;
    __syncthreads();
    if (wave != 0) return;
    if (lane < NP2) tot = (s_comb[0][lane] + s_comb[1][lane]) + (s_comb[2][lane] + s_comb[3][lane]);
    rs = (int64_t)gridDim.x - n_front;
    row = (int64_t)blockIdx.x - n_front;
  }
  if (lane < NP2) {
    const int q = packed_value_of_lane<NP2>(lane);
    if (q < 2) p_fwd[q * rs + row] = tot;
    else if (q < NV) p_bwd[(q - 2) * rs + row] = tot;
  }
}

// partial rows: columns {sum loss, count} at p_fwd (stride gridDim * 4) and [0, P) dL/dw at p_bwd (same stride)
template <typename PT, int NS, int P, int VAR = 0>
__global__ __launch_bounds__(kBlock) void consistency_step_basis_kernel(
    PointBasis pb, BlockTab tab, const int32_t* __restrict__ own_base, int cap, const int32_t* __restrict__ centre_idx, int64_t n,
    const uint8_t* __restrict__ mask, LossParams lp, QParams qp, double* __restrict__ p_fwd, double* __restrict__ p_bwd,
    StepChain ch) {
  extern __shared__ int4 tile[];
  __shared__ double s_w[DC_MAX_MODEL_TERMS];
  __shared__ int s_ok;
  __shared__ double s_front[kBlock / kWave];
  const bool chained = ch.ready != nullptr;
  if (chained && (int)blockIdx.x < ch.n_front) { chain_front_block<P>(ch, s_front); return; }
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  int64_t blk = chain_block_of(ch, chained, nblocks);
  if (blk >= 0 && ch.blk_skip && ch.blk_skip[blk]) blk = -1;            // no centre of this block is inside the loss mask
  double acc2[2] = {0.0, 0.0}, gw[P];
#pragma unroll
  for (int k = 0; k < P; ++k) gw[k] = 0.0;
  const int32_t s0 = blk >= 0 ? tab.slot_ptr[blk] : 0;
  bool bad = blk >= 0 && tab.slot_ptr[blk + 1] - s0 != NS;
  if (blk >= 0 && !bad) {
    const int64_t i = blk * kBlock + threadIdx.x;
    const bool live = i < n;
    const bool in_mask = live && (mask ? mask[i] != 0 : true);
    const uint16_t* lrow = tab.loc + (int64_t)s0 * kBlock + threadIdx.x;
    uint32_t pre[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) pre[q] = (uint32_t)lrow[q * kBlock];
    const int32_t base = tab.blk_ptr[blk], nd = tab.blk_ptr[blk + 1] - base;
    const int32_t own = (own_base && !centre_idx) ? own_base[blk] : -1;
    double wq[P];
    if (chained) {
      // the weights of this launch come from its leading blocks: fetch the rows this lane stages (the first two: a block
      // lists ~1.5 distinct rows per lane) BEFORE waiting for them, so that the wait hides behind the fetch or vice versa
      // (a wait that never ends poisons the sums instead of hanging)
      typename StepRow<PT, P>::Raw r0, r1;
      const int t0 = threadIdx.x, t1 = threadIdx.x + kBlock;
      if (t0 < nd) r0 = StepRow<PT, P>::fetch(pb, tab.blk_ids[base + t0]);
      if (t1 < nd) r1 = StepRow<PT, P>::fetch(pb, tab.blk_ids[base + t1]);
      chain_weights<P>(ch, pb.w_scale, s_w, &s_ok);
      __syncthreads();
      if (!s_ok) bad = true;
#pragma unroll
      for (int k = 0; k < P; ++k) wq[k] = s_w[k];
      if (t0 < nd) StepRow<PT, P>::place(r0, wq, tile, cap, t0);
      if (t1 < nd) StepRow<PT, P>::place(r1, wq, tile, cap, t1);
      for (int t = threadIdx.x + 2 * kBlock; t < nd; t += kBlock) StepRow<PT, P>::stage(pb, wq, tab.blk_ids[base + t], tile, cap, t);
    } else {
      stage_weights(pb, s_w);
      __syncthreads();
#pragma unroll
      for (int k = 0; k < P; ++k) wq[k] = s_w[k];
      for (int t = threadIdx.x; t < nd; t += kBlock) StepRow<PT, P>::stage(pb, wq, tab.blk_ids[base + t], tile, cap, t);
    }
    typename Pt<PT>::Raw ci;
    if (own < 0) ci = Basis<PT>::template point<P>(pb, wq, live ? (centre_idx ? (int64_t)centre_idx[i] : i) : 0);
    __syncthreads();
    if (own >= 0) ci = staged_point<PT>(tile, cap, own + (live ? (int)threadIdx.x : 0));
    if (live && (!mask || __any((int)in_mask))) {        // (a wavefront of masked-out centres only: nothing to add, see consistency_step_q32_kernel)
      CovAcc acc;
      cov_init(acc);
      uint32_t mx = pre[0];
#pragma unroll
      for (int q = 1; q < NS; ++q) mx = max(mx, pre[q]);
      const bool any_miss = __any((int)(mx == kNoLoc)) != 0;
      int n_have;
      if (any_miss) n_have = gather_fixed<PT, NS, true>(tile, cap, ci, pre, acc);
      else n_have = gather_fixed<PT, NS, false>(tile, cap, ci, pre, acc);
      acc.W = (double)n_have;
      double cm[3], v0[3], c1, c2;
      if constexpr ((VAR & kVarSlimTail) != 0) step_point2<PT, NS>(acc, n_have, !any_miss, in_mask, lp, qp, acc2, cm, v0, &c1, &c2);
      else step_point<PT>(acc, in_mask, lp, qp, acc2, cm, v0, &c1, &c2);
      const double u = Pt<PT>::unit(qp);
      if constexpr ((VAR & kVarF32Sweep) != 0 && std::is_same<PT, q32>::value) {
        // second sweep in float32 (see chain_term_f32); the lane's sums join the fp64 reduction
        float cmf[3], vs[3], vu[3], gwf[P];
#pragma unroll
        for (int a = 0; a < 3; ++a) { cmf[a] = (float)cm[a]; vs[a] = (float)(c1 * v0[a]); vu[a] = (float)v0[a]; }
        const float c2f = (float)c2;
#pragma unroll
        for (int k = 0; k < P; ++k) gwf[k] = 0.0f;
        // (groups of four with a scheduling fence between them: left alone, the compiler requests all 2 NS row pieces up
        // front -- 80 registers -- and the kernel drops from 6 to 4 wavefronts per SIMD)
        if (any_miss) {
#pragma unroll
          for (int q = 0; q < NS; ++q) {
            if (q % 4 == 0 && q > 0) __builtin_amdgcn_sched_barrier(0);
            chain_term_f32<P>(tile, cap, pre[q], pre[q] != kNoLoc, ci, cmf, vs, vu, c2f, gwf);
          }
        } else {
#pragma unroll
          for (int q = 0; q < NS; ++q) {
            if (q % 4 == 0 && q > 0) __builtin_amdgcn_sched_barrier(0);
            chain_term_f32<P>(tile, cap, pre[q], true, ci, cmf, vs, vu, c2f, gwf);
          }
        }
#pragma unroll
        for (int k = 0; k < P; ++k) gw[k] = (double)gwf[k] * u;
      } else {
        double mean[3];
        StepRow<PT, P>::mean_of(ci, cm, mean);
        // second sweep over the same slots (full wavefronts skip the validity selects)
        if (any_miss) {
#pragma unroll
          for (int q = 0; q < NS; ++q) chain_term<PT, P>(tile, cap, pre[q], pre[q] != kNoLoc, mean, v0, c1, c2, gw);
        } else {
#pragma unroll
          for (int q = 0; q < NS; ++q) chain_term<PT, P>(tile, cap, pre[q], true, mean, v0, c1, c2, gw);
        }
#pragma unroll
        for (int k = 0; k < P; ++k) gw[k] *= u;            // differences were in grid steps
      }
    }
  }
  if (bad) acc2[0] = acc2[1] = __longlong_as_double(0x7ff8000000000000ll);
  step_partials<P, (VAR & kVarDppSums) != 0>(acc2, gw, p_fwd, p_bwd, chained, chained ? ch.n_front : 0);
}

// ---- the one-pass kernel for q32 points and a fixed slot count: what a C2 step runs ----------------------------------------
// Same table, same basis rows, same staged rows {x0 x1 x2 u0 | u1 u2 c0 c1 | c2} (StepRow<q32, P>) and same sums as
// consistency_step_basis_kernel<q32, NS, P>; what differs is where the instructions go (the kernel issues VALU instructions
// ~98 % of its time -- rocprofv3 SQ_ACTIVE_INST_VALU -- so its duration IS its instruction count, at one wave64 VALU instruction
// per four cycles per SIMD whatever the type: tools/ubench/valu_rates.hip):
//  * the tile is STATIC LDS of kStepQ32Cap rows: its address and the piece stride are immediates of the ds_read instructions and
//    the table's 16-bit byte offset is the address register as it stands (dynamic LDS costs a v_add per row piece: 30 per centre);
//  * the per-centre tail is step_point2 (eig3_smallest_unit: adjugate eigenvector, reciprocals with one Newton step, the
//    deflation path only for needles);
//  * the second sweep is float32 (the difference to the centre is an exact int32, u and c are float32 words, the per-centre
//    factors are rounded once);
//  * the wavefront sums go through DPP row operations.
// A fp64-difference row format ({x - ref} as doubles: no int -> fp conversions in the sweeps, 80 instructions fewer) was measured
// and dropped: 48-B rows make the kernel LDS-bound (SQ_LDS_IDX_ACTIVE 87 % of its duration, 56 % of it bank conflicts of the
// random row reads: 61 us against 52).
constexpr int kStepQ32Cap = 512;          // rows of the static LDS tile (16 KB + 8 KB for a third piece: six blocks per CU); a second
                                          // instantiation takes 768 rows (24 KB: still six blocks per CU for one or two weights); tables with
                                          // more distinct rows per block take consistency_step_basis_kernel

// second sweep, one neighbour (float32): gw[k] += c_kj (c1 (v . e_j)(v . u_j) - c2 (e_j . u_j)); vs = c1 v0, vu = v0
typedef float float2v __attribute__((ext_vector_type(2)));
template <int P, int CAP>
__device__ __forceinline__ void chain_term_q32(const int4* tile, uint32_t off, bool have, const Pt<q32>::Raw& ci, const float* cmf, const float* vs,
                                               const float* vu, float c2f, float* gw) {
  const char* row = reinterpret_cast<const char*>(tile) + (have ? off : 0u);
  const int4 p0 = *reinterpret_cast<const int4*>(row);
  const int4 p1 = *reinterpret_cast<const int4*>(row + (size_t)CAP * 16);
  // (two-wide float operations where the operands already sit in neighbouring registers: v_pk_add_f32 / v_pk_fma_f32 issue two
  //  operations in one slot)
  const float2v e01 = float2v{(float)(p0.x - ci.v[0]), (float)(p0.y - ci.v[1])} - float2v{cmf[0], cmf[1]};
  const float e0 = e01.x, e1 = e01.y, e2 = (float)(p0.z - ci.v[2]) - cmf[2];
  const float u0 = __int_as_float(p0.w), u1 = __int_as_float(p1.x), u2 = __int_as_float(p1.y);
  const float al = fmaf(vs[2], e2, fmaf(vs[1], e1, vs[0] * e0));                       // c1 (v . e_j)
  const float be = fmaf(vu[2], u2, fmaf(vu[1], u1, vu[0] * u0));                       // v . u_j
  const float ga = fmaf(e2, u2, fmaf(e1, u1, e0 * u0));                                // e_j . u_j
  float tj = fmaf(al, be, -(c2f * ga));
  if (!have) tj = 0.0f;
  if constexpr (P == 2) {
    float2v g = float2v{gw[0], gw[1]};
    g = __builtin_elementwise_fma(float2v{tj, tj}, float2v{__int_as_float(p1.z), __int_as_float(p1.w)}, g);
    gw[0] = g.x; gw[1] = g.y;
  } else {
    gw[0] = fmaf(tj, __int_as_float(p1.z), gw[0]);
    if constexpr (P > 1) gw[1] = fmaf(tj, __int_as_float(p1.w), gw[1]);
    if constexpr (P > 2) gw[2] = fmaf(tj, __int_as_float(reinterpret_cast<const int4*>(row + (size_t)CAP * 32)->x), gw[2]);
  }
}

// (six wavefronts per SIMD: left alone the kernel takes 81 VGPRs -- one allocation granule over the 80 of six wavefronts, i.e. FIVE per SIMD;
//  at 79 + 12 B of scratch a step takes 44.4 instead of 47.2 us.  Seven -- 71 VGPRs, 44 B of scratch -- take 50.9 us.)
// Diagnostic build only (-DDC_BLOCK_TRACE, tools/block_trace.py): every block of the two one-pass step kernels records where and
// when it ran -- {XCC | HW_ID, start, end} on the 100 MHz constant clock -- so that the schedule of a launch can be drawn (which
// CU got how many blocks, when each CU ran dry).  The product library is built without it: the macros expand to nothing.
#ifdef DC_BLOCK_TRACE
__device__ unsigned long long* g_block_trace = nullptr;
#define DC_TRACE_BEGIN() const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime()
#define DC_TRACE_END() do { if (threadIdx.x == 0 && g_block_trace) { \
    unsigned long long* tr_ = g_block_trace + 4 * (size_t)blockIdx.x; \
    tr_[0] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4); \
    tr_[1] = trace_t0; tr_[2] = __builtin_amdgcn_s_memrealtime(); tr_[3] = 1; } } while (0)
#else
#define DC_TRACE_BEGIN() do {} while (0)
#define DC_TRACE_END() do {} while (0)
#endif

template <int NS, int P, int CAP>
__global__ __launch_bounds__(kBlock, (StepRow<q32, P>::kPieces * CAP * 16 <= 25 * 1024 ? 6 : 4)) void consistency_step_q32_kernel(
    PointBasis pb, BlockTab tab, const int32_t* __restrict__ own_base, const int32_t* __restrict__ centre_idx, int64_t n,
    const uint8_t* __restrict__ mask, LossParams lp, QParams qp, double* __restrict__ p_fwd, double* __restrict__ p_bwd,
    StepChain ch) {
  constexpr int NV = 2 + P, NP2 = NV <= 4 ? 4 : 8;
  constexpr int cap = CAP;
  __shared__ int4 tile[StepRow<q32, P>::kPieces * CAP];
  __shared__ double s_w[DC_MAX_MODEL_TERMS];
  __shared__ double s_front[kWavesPerBlock];
  __shared__ double s_comb[kWavesPerBlock * NP2];
  __shared__ int s_ok[2];
  DC_TRACE_BEGIN();
  const bool chained = ch.ready != nullptr;
  if (chained && (int)blockIdx.x < ch.n_front) { chain_front_block<P>(ch, s_front); return; }
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  int64_t blk = chain_block_of(ch, chained, nblocks);
  if (blk >= 0 && ch.blk_skip && ch.blk_skip[blk]) blk = -1;            // no centre of this block is inside the loss mask
  double acc2[2] = {0.0, 0.0}, gw[P];
#pragma unroll
  for (int k = 0; k < P; ++k) gw[k] = 0.0;
  const int32_t s0 = blk >= 0 ? tab.slot_ptr[blk] : 0;
  bool bad = blk >= 0 && tab.slot_ptr[blk + 1] - s0 != NS;
  if (blk >= 0 && !bad) {
    const int64_t i = blk * kBlock + threadIdx.x;
    const bool live = i < n;
    const bool in_mask = live && (mask ? mask[i] != 0 : true);
    const uint16_t* lrow = tab.loc + (int64_t)s0 * kBlock + threadIdx.x;
    uint32_t pre[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) pre[q] = (uint32_t)lrow[q * kBlock];
    const int32_t base = tab.blk_ptr[blk], nd = tab.blk_ptr[blk + 1] - base;
    const int32_t own = (own_base && !centre_idx) ? own_base[blk] : -1;
    double wq[P];
    if (chained) {
      // fetch the rows this lane stages before waiting for the weights of this launch (consistency_step_basis_kernel)
      typename StepRow<q32, P>::Raw r0, r1;
      const int t0 = threadIdx.x, t1 = threadIdx.x + kBlock;
      if (t0 < nd) r0 = StepRow<q32, P>::fetch(pb, tab.blk_ids[base + t0]);
      if (t1 < nd) r1 = StepRow<q32, P>::fetch(pb, tab.blk_ids[base + t1]);
      chain_weights<P>(ch, pb.w_scale, s_w, s_ok);
      __syncthreads();
      if (!s_ok[0]) bad = true;
#pragma unroll
      for (int k = 0; k < P; ++k) wq[k] = s_w[k];
      if (t0 < nd) StepRow<q32, P>::place(r0, wq, tile, cap, t0);
      if (t1 < nd) StepRow<q32, P>::place(r1, wq, tile, cap, t1);
      for (int t = threadIdx.x + 2 * kBlock; t < nd; t += kBlock) StepRow<q32, P>::stage(pb, wq, tab.blk_ids[base + t], tile, cap, t);
    } else {
      stage_weights(pb, s_w);
      __syncthreads();
#pragma unroll
      for (int k = 0; k < P; ++k) wq[k] = s_w[k];
      for (int t = threadIdx.x; t < nd; t += kBlock) StepRow<q32, P>::stage(pb, wq, tab.blk_ids[base + t], tile, cap, t);
    }
    Pt<q32>::Raw ci;
    if (own < 0) ci = Basis<q32>::template point<P>(pb, wq, live ? (centre_idx ? (int64_t)centre_idx[i] : i) : 0);
    __syncthreads();
    if (own >= 0) ci = staged_point<q32>(tile, cap, own + (live ? (int)threadIdx.x : 0));
    // a wavefront whose centres are ALL outside the loss mask adds nothing to the loss, the count or dL/dw (every term carries the
    // centre's mask): it has staged its rows and is done.  The plan groups masked-out points at the end of every block, so
    // these are whole wavefronts (bench.py reports their share).
    if (live && (!mask || __any((int)in_mask))) {
      CovAcc acc;
      cov_init(acc);
      // positions are multiples of 16, the empty-slot mark 0xFFFF is not: bit 0 of the OR of a lane's positions tells
      uint32_t mo = pre[0];
#pragma unroll
      for (int q = 1; q < NS; ++q) mo |= pre[q];
      const bool any_miss = __any((int)(mo & 1u)) != 0;
      int n_have;
      if (any_miss) n_have = gather_fixed<q32, NS, true>(tile, cap, ci, pre, acc);
      else n_have = gather_fixed<q32, NS, false>(tile, cap, ci, pre, acc);
      acc.W = (double)n_have;
      double cm[3], v0[3], c1, c2;
      step_point2<q32, NS>(acc, n_have, !any_miss, in_mask, lp, qp, acc2, cm, v0, &c1, &c2);
      float cmf[3], vs[3], vu[3], gwf[P];
#pragma unroll
      for (int a = 0; a < 3; ++a) { cmf[a] = (float)cm[a]; vs[a] = (float)(c1 * v0[a]); vu[a] = (float)v0[a]; }
      const float c2f = (float)c2;
#pragma unroll
      for (int k = 0; k < P; ++k) gwf[k] = 0.0f;
      // (groups of four with a scheduling fence between them: left alone, the compiler requests every row piece up front
      // and the kernel loses wavefronts per SIMD to the registers)
      if (any_miss) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          if (q % 4 == 0 && q > 0) __builtin_amdgcn_sched_barrier(0);
          chain_term_q32<P, CAP>(tile, pre[q], pre[q] != kNoLoc, ci, cmf, vs, vu, c2f, gwf);
        }
      } else {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          if (q % 4 == 0 && q > 0) __builtin_amdgcn_sched_barrier(0);
          chain_term_q32<P, CAP>(tile, pre[q], true, ci, cmf, vs, vu, c2f, gwf);
        }
      }
      const double u = qp.scale;
#pragma unroll
      for (int k = 0; k < P; ++k) gw[k] = (double)gwf[k] * u;          // differences were in grid steps
    }
  }
  if (bad) acc2[0] = acc2[1] = __longlong_as_double(0x7ff8000000000000ll);
  // ---- {sum loss, count, dL/dw} of the wavefront (one row per wavefront; chained: per block), as step_partials
  double v[NP2];
  v[0] = acc2[0]; v[1] = acc2[1];
#pragma unroll
  for (int k = 0; k < NP2 - 2; ++k) v[2 + k] = k < P ? gw[k] : 0.0;
  double tot;
  if constexpr (NP2 == 4) tot = wave_sum4_dpp(v);
  else tot = wave_sum_packed<NP2>(v);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  int64_t rs = (int64_t)gridDim.x * kWavesPerBlock, row = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (chained) {
    if (lane < NP2) s_comb[wave * NP2 + lane] = tot;
    __syncthreads();
    if (wave != 0) return;
    if (lane < NP2) tot = (s_comb[lane] + s_comb[NP2 + lane]) + (s_comb[2 * NP2 + lane] + s_comb[3 * NP2 + lane]);
    rs = (int64_t)gridDim.x - ch.n_front;
    row = (int64_t)blockIdx.x - ch.n_front;
  }
  if (lane < NP2) {
    const int q = packed_value_of_lane<NP2>(lane);
    if (q < 2) p_fwd[q * rs + row] = tot;
    else if (q < NV) p_bwd[(q - 2) * rs + row] = tot;
  }
  DC_TRACE_END();
}

// the same for any slot count (radius neighbourhoods): run-time slot loops, as consistency_fwd_basis_slots_kernel
template <typename PT, int P>
__global__ __launch_bounds__(kBlock) void consistency_step_basis_slots_kernel(
    PointBasis pb, BlockTab tab, const int32_t* __restrict__ own_base, int cap, const int32_t* __restrict__ centre_idx, int64_t n,
    const uint8_t* __restrict__ mask, LossParams lp, QParams qp, double* __restrict__ p_fwd, double* __restrict__ p_bwd,
    StepChain ch, int packed) {
  extern __shared__ int4 tile[];
  __shared__ double s_w[DC_MAX_MODEL_TERMS];
  __shared__ int s_ok;
  __shared__ double s_front[kBlock / kWave];
  const bool chained = ch.ready != nullptr;
  if (chained && (int)blockIdx.x < ch.n_front) { chain_front_block<P>(ch, s_front); return; }
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  int64_t blk = xcd_block_of((int64_t)blockIdx.x - (chained ? ch.n_front : 0), nblocks);
  if (blk >= 0 && ch.blk_skip && ch.blk_skip[blk]) blk = -1;            // no centre of this block is inside the loss mask
  double acc2[2] = {0.0, 0.0}, gw[P];
#pragma unroll
  for (int k = 0; k < P; ++k) gw[k] = 0.0;
  const int64_t i = blk * kBlock + threadIdx.x;
  const bool live = blk >= 0 && i < n;
  int32_t nslots = 0, own = -1, base = 0, nd = 0;
  const uint16_t* lrow = tab.loc;
  uint32_t pre[kPreSlots];
  if (blk >= 0) {
    const int32_t s0 = tab.slot_ptr[blk];
    nslots = tab.slot_ptr[blk + 1] - s0;
    lrow = tab.loc + (int64_t)s0 * kBlock + threadIdx.x;
#pragma unroll
    for (int q = 0; q < kPreSlots; ++q) pre[q] = (live && q < nslots) ? (uint32_t)lrow[q * kBlock] : kNoLoc;
    own = (own_base && !centre_idx) ? own_base[blk] : -1;
    base = tab.blk_ptr[blk];
    nd = tab.blk_ptr[blk + 1] - base;
  }
  double wq[P];
  bool timed_out = false;
  typename Pt<PT>::Raw ci;
  if (chained) {                           // as in consistency_step_basis_kernel: fetch, wait for the weights, place
    typename StepRow<PT, P>::Raw r0, r1;
    const int t0 = threadIdx.x, t1 = threadIdx.x + kBlock;
    if (t0 < nd) r0 = StepRow<PT, P>::fetch(pb, tab.blk_ids[base + t0]);
    if (t1 < nd) r1 = StepRow<PT, P>::fetch(pb, tab.blk_ids[base + t1]);
    chain_weights<P>(ch, pb.w_scale, s_w, &s_ok);
    __syncthreads();
    timed_out = !s_ok;
#pragma unroll
    for (int k = 0; k < P; ++k) wq[k] = s_w[k];
    if (t0 < nd) StepRow<PT, P>::place(r0, wq, tile, cap, t0);
    if (t1 < nd) StepRow<PT, P>::place(r1, wq, tile, cap, t1);
    for (int t = threadIdx.x + 2 * kBlock; t < nd; t += kBlock) StepRow<PT, P>::stage(pb, wq, tab.blk_ids[base + t], tile, cap, t);
  } else {
    stage_weights(pb, s_w);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < P; ++k) wq[k] = s_w[k];
    for (int t = threadIdx.x; t < nd; t += kBlock) StepRow<PT, P>::stage(pb, wq, tab.blk_ids[base + t], tile, cap, t);
  }
  if (blk >= 0 && own < 0) ci = Basis<PT>::template point<P>(pb, wq, live ? (centre_idx ? (int64_t)centre_idx[i] : i) : 0);
  __syncthreads();
  if (live) {
    if (own >= 0) ci = staged_point<PT>(tile, cap, own + (int)threadIdx.x);
    CovAcc acc;
    cov_init(acc);
    bool miss = false;
#pragma unroll
    for (int q = 0; q < kPreSlots; ++q) miss |= (q < nslots) && pre[q] == kNoLoc;
    int n_have = 0;
    if (__any((int)miss)) n_have = gather_slots<PT, true>(tile, cap, ci, pre, nslots, acc);
    else n_have = gather_slots<PT, false>(tile, cap, ci, pre, nslots, acc);
    if (nslots > kPreSlots) n_have += gather_tail<PT>(tile, cap, ci, lrow, nslots, packed != 0, acc);
    acc.W = (double)n_have;
    double cm[3], v0[3], c1, c2;
    step_point2<PT, 2>(acc, n_have, false, mask ? mask[i] != 0 : true, lp, qp, acc2, cm, v0, &c1, &c2);
    const double u = Pt<PT>::unit(qp);
    if constexpr (std::is_same<PT, q32>::value) {
      float cmf[3], vs[3], vu[3], gwf[P];
#pragma unroll
      for (int a = 0; a < 3; ++a) { cmf[a] = (float)cm[a]; vs[a] = (float)(c1 * v0[a]); vu[a] = (float)v0[a]; }
      const float c2f = (float)c2;
#pragma unroll
      for (int k = 0; k < P; ++k) gwf[k] = 0.0f;
#pragma unroll
      for (int q = 0; q < kPreSlots; ++q)
        if (q < nslots) chain_term_f32<P>(tile, cap, pre[q], pre[q] != kNoLoc, ci, cmf, vs, vu, c2f, gwf);
#pragma unroll
      for (int k = 0; k < P; ++k) gw[k] = (double)gwf[k];
      if (nslots > kPreSlots) chain_tail_f32<P>(tile, cap, lrow, nslots, packed != 0, ci, cmf, vs, vu, c2f, gw);
#pragma unroll
      for (int k = 0; k < P; ++k) gw[k] *= u;
    } else {
      double mean[3];
      StepRow<PT, P>::mean_of(ci, cm, mean);
#pragma unroll
      for (int q = 0; q < kPreSlots; ++q)
        if (q < nslots) chain_term<PT, P>(tile, cap, pre[q], pre[q] != kNoLoc, mean, v0, c1, c2, gw);
      if (nslots > kPreSlots) chain_tail<PT, P>(tile, cap, lrow, nslots, packed != 0, mean, v0, c1, c2, gw);
#pragma unroll
      for (int k = 0; k < P; ++k) gw[k] *= u;
    }
  }
  if (timed_out) acc2[0] = acc2[1] = __longlong_as_double(0x7ff8000000000000ll);
  step_partials<P, true>(acc2, gw, p_fwd, p_bwd, chained, chained ? ch.n_front : 0);
}

// ---- the one-pass kernel for ball neighbourhoods on float32 clouds (round 4) ------------------------------------------------
// The reference's default neighbourhood is a ball (nn_type = ball, nn_r = 0.25 m, config.py:187-189; 0.4 m in train_demo:61-63):
// on voxel-filtered scans a row has 70-200 neighbours, so the time is the two sweeps over the slots, not the per-centre tail.
// consistency_step_basis_slots_kernel spent ~58 VALU instructions per (centre, neighbour) pair at 0.65 of the issue peak; the
// arithmetic needs ~40.  What went:
//   * validity handling: an empty slot reads the lane's OWN row, whose difference to the centre is exactly zero -- one select on
//     the 16-bit position instead of selects on every coordinate and a count; the number of neighbours is the row's length
//     (dcBlockTable.row_ptr);
//   * address arithmetic: the tile's row capacity is a template argument, so the second piece of a row is an immediate off the
//     16-bit position;
//   * the dependent {position load, row read} pair per slot: trips of eight, the next trip's positions requested before the
//     rows of this one are read, and a wavefront stops at ITS longest row, not the block's;
//   * the first-sixteen-slots special case (registers kept across the per-centre tail).
// The second sweep accumulates float32 per trip and fp64 across trips.  Same sums as the slots kernel to the rounding of that
// order of additions.
template <int P, int CAP>
__global__ __launch_bounds__(kBlock, (CAP <= 1024 ? 5 : 4)) void consistency_step_ragged_q32_kernel(
    PointBasis pb, BlockTab tab, const int32_t* __restrict__ own_base, const int32_t* __restrict__ row_ptr, int64_t n,
    const uint8_t* __restrict__ mask, LossParams lp, QParams qp, double* __restrict__ p_fwd, double* __restrict__ p_bwd,
    StepChain ch) {
  using Row = StepRow<q32, P>;
  extern __shared__ int4 tile[];                   // Row::kPieces * CAP rows of 16 B (dynamic: up to 128 KB, see ragged_launch)
  __shared__ double s_w[DC_MAX_MODEL_TERMS];
  __shared__ int s_ok;
  __shared__ double s_front[kBlock / kWave];
  DC_TRACE_BEGIN();
  const bool chained = ch.ready != nullptr;
  if (chained && (int)blockIdx.x < ch.n_front) { chain_front_block<P>(ch, s_front); return; }
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  int64_t blk = xcd_block_of((int64_t)blockIdx.x - (chained ? ch.n_front : 0), nblocks);
  if (blk >= 0 && ch.blk_skip && ch.blk_skip[blk]) blk = -1;            // no centre of this block is inside the loss mask
  double acc2[2] = {0.0, 0.0}, gw[P];
#pragma unroll
  for (int k = 0; k < P; ++k) gw[k] = 0.0;
  const int64_t i = blk * kBlock + threadIdx.x;
  const bool live = blk >= 0 && i < n;
  int32_t nslots = 0, own = 0, base = 0, nd = 0, deg = 0;
  const uint16_t* lrow = tab.loc;
  uint32_t first[kTrip];
  if (blk >= 0) {
    const int32_t s0 = tab.slot_ptr[blk];
    nslots = tab.slot_ptr[blk + 1] - s0;
    lrow = tab.loc + (int64_t)s0 * kBlock + threadIdx.x;
    if (live) deg = row_ptr[i + 1] - row_ptr[i];
    own = own_base[blk];
    base = tab.blk_ptr[blk];
    nd = tab.blk_ptr[blk + 1] - base;
  }
  const uint32_t own_off = (uint32_t)(own + (int)threadIdx.x) * 16u;      // where padding slots point (and idle lanes read)
#pragma unroll
  for (int u_ = 0; u_ < kTrip; ++u_) first[u_] = (live && nslots > 0) ? (uint32_t)lrow[u_ * kBlock] : kNoLoc;     // (slot counts are multiples of 8)
  double wq[P];
  bool timed_out = false;
  if (chained) {                           // as in consistency_step_basis_kernel: fetch, wait for the weights, place
    typename Row::Raw r0, r1;
    const int t0 = threadIdx.x, t1 = threadIdx.x + kBlock;
    if (t0 < nd) r0 = Row::fetch(pb, tab.blk_ids[base + t0]);
    if (t1 < nd) r1 = Row::fetch(pb, tab.blk_ids[base + t1]);
    chain_weights<P>(ch, pb.w_scale, s_w, &s_ok);
    __syncthreads();
    timed_out = !s_ok;
#pragma unroll
    for (int k = 0; k < P; ++k) wq[k] = s_w[k];
    if (t0 < nd) Row::place(r0, wq, tile, CAP, t0);
    if (t1 < nd) Row::place(r1, wq, tile, CAP, t1);
    for (int t = threadIdx.x + 2 * kBlock; t < nd; t += kBlock) Row::stage(pb, wq, tab.blk_ids[base + t], tile, CAP, t);
  } else {
    stage_weights(pb, s_w);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < P; ++k) wq[k] = s_w[k];
    for (int t = threadIdx.x; t < nd; t += kBlock) Row::stage(pb, wq, tab.blk_ids[base + t], tile, CAP, t);
  }
  __syncthreads();
  const bool in_mask = live && (mask ? mask[i] != 0 : true);
  if (live && (!mask || __any((int)in_mask))) {          // (a wavefront of masked-out centres only: nothing to add, see consistency_step_q32_kernel)
    const char* tb = reinterpret_cast<const char*>(tile);
    const Pt<q32>::Raw ci = Pt<q32>::from_row(reinterpret_cast<const int4*>(tb + own_off));
    // the longest row among this wavefront's lanes bounds its trips
    int wmax = deg;
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) wmax = max(wmax, __shfl_xor(wmax, off, kWave));
    wmax = __builtin_amdgcn_readfirstlane(min(wmax, nslots));      // (uniform: scalar loop control)
    CovAcc acc;
    cov_init(acc);
    // one trip: the eight rows at positions l[] into the moments; the NEXT trip's positions (pn, immediates off one pointer) are
    // requested first.  Slot counts are multiples of eight and the table ends with eight rows of slack: no guards.
    auto sweep1 = [&](const uint32_t* l, uint32_t* nx, const uint16_t* pn) {
#pragma unroll
      for (int u_ = 0; u_ < kTrip; ++u_) nx[u_] = (uint32_t)pn[u_ * kBlock];
      int4 r[kTrip];
#pragma unroll
      for (int u_ = 0; u_ < kTrip; ++u_) r[u_] = *reinterpret_cast<const int4*>(tb + (l[u_] == kNoLoc ? own_off : l[u_]));
#pragma unroll
      for (int u_ = 0; u_ < kTrip; ++u_)
        cov_add_d(acc, (double)(r[u_].x - ci.v[0]), (double)(r[u_].y - ci.v[1]), (double)(r[u_].z - ci.v[2]));
    };
    {
      uint32_t la[kTrip], lb[kTrip];
#pragma unroll
      for (int u_ = 0; u_ < kTrip; ++u_) la[u_] = first[u_];
      const uint16_t* pn = lrow + kTrip * kBlock;
      for (int q0 = 0; q0 < wmax; q0 += 2 * kTrip) {        // two trips per iteration: the position registers alternate, no moves
        sweep1(la, lb, pn);
        pn += kTrip * kBlock;
        if (q0 + kTrip >= wmax) break;
        sweep1(lb, la, pn);
        pn += kTrip * kBlock;
      }
    }
    acc.W = (double)deg;
    double cm[3], v0[3], c1, c2;
    step_point2<q32, 2>(acc, deg, false, in_mask, lp, qp, acc2, cm, v0, &c1, &c2);
    const double u = Pt<q32>::unit(qp);
    float cmf[3], vs[3], vu[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { cmf[a] = (float)cm[a]; vs[a] = (float)(c1 * v0[a]); vu[a] = (float)v0[a]; }
    const float c2f = (float)c2;
    // (two neighbours per packed float32 instruction: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 do two lanes' worth of work in
    // one issue slot -- the arithmetic of a neighbour's term drops from ~24 to ~15 instructions)
    auto sweep2 = [&](const uint32_t* l, uint32_t* nx, const uint16_t* pn) {
#pragma unroll
      for (int u_ = 0; u_ < kTrip; ++u_) nx[u_] = (uint32_t)pn[u_ * kBlock];
      float2v g[P];
#pragma unroll
      for (int k = 0; k < P; ++k) g[k] = float2v{0.0f, 0.0f};
#pragma unroll
      for (int u_ = 0; u_ < kTrip; u_ += 2) {
        const bool ha = l[u_] != kNoLoc, hb = l[u_ + 1] != kNoLoc;      // (packed rows: the same as slot < deg)
        const char* ra = tb + (ha ? l[u_] : own_off);
        const char* rb = tb + (hb ? l[u_ + 1] : own_off);
        const int4 a0 = *reinterpret_cast<const int4*>(ra), a1 = *reinterpret_cast<const int4*>(ra + CAP * 16);
        const int4 b0 = *reinterpret_cast<const int4*>(rb), b1 = *reinterpret_cast<const int4*>(rb + CAP * 16);
        const float2v e0 = float2v{(float)(a0.x - ci.v[0]), (float)(b0.x - ci.v[0])} - float2v{cmf[0], cmf[0]};
        const float2v e1 = float2v{(float)(a0.y - ci.v[1]), (float)(b0.y - ci.v[1])} - float2v{cmf[1], cmf[1]};
        const float2v e2 = float2v{(float)(a0.z - ci.v[2]), (float)(b0.z - ci.v[2])} - float2v{cmf[2], cmf[2]};
        const float2v u0 = float2v{__int_as_float(a0.w), __int_as_float(b0.w)};
        const float2v u1 = float2v{__int_as_float(a1.x), __int_as_float(b1.x)};
        const float2v u2 = float2v{__int_as_float(a1.y), __int_as_float(b1.y)};
        const float2v al = __builtin_elementwise_fma(float2v{vs[2], vs[2]}, e2, __builtin_elementwise_fma(float2v{vs[1], vs[1]}, e1, float2v{vs[0], vs[0]} * e0));
        const float2v be = __builtin_elementwise_fma(float2v{vu[2], vu[2]}, u2, __builtin_elementwise_fma(float2v{vu[1], vu[1]}, u1, float2v{vu[0], vu[0]} * u0));
        const float2v ga = __builtin_elementwise_fma(e2, u2, __builtin_elementwise_fma(e1, u1, e0 * u0));
        float2v tj = __builtin_elementwise_fma(al, be, -(float2v{c2f, c2f} * ga));
        tj = float2v{ha ? tj.x : 0.0f, hb ? tj.y : 0.0f};           // an empty slot (the lane's own row) is not a neighbour
        g[0] = __builtin_elementwise_fma(tj, float2v{__int_as_float(a1.z), __int_as_float(b1.z)}, g[0]);
        if constexpr (P > 1) g[1] = __builtin_elementwise_fma(tj, float2v{__int_as_float(a1.w), __int_as_float(b1.w)}, g[1]);
        if constexpr (P > 2)
          g[2] = __builtin_elementwise_fma(tj, float2v{__int_as_float(reinterpret_cast<const int4*>(ra + CAP * 32)->x),
                                                       __int_as_float(reinterpret_cast<const int4*>(rb + CAP * 32)->x)}, g[2]);
      }
#pragma unroll
      for (int k = 0; k < P; ++k) gw[k] += (double)(g[k].x + g[k].y);
    };
    {
      uint32_t la[kTrip], lb[kTrip];
#pragma unroll
      for (int u_ = 0; u_ < kTrip; ++u_) la[u_] = first[u_];
      const uint16_t* pn = lrow + kTrip * kBlock;
      for (int q0 = 0; q0 < wmax; q0 += 2 * kTrip) {
        sweep2(la, lb, pn);
        pn += kTrip * kBlock;
        if (q0 + kTrip >= wmax) break;
        sweep2(lb, la, pn);
        pn += kTrip * kBlock;
      }
    }
#pragma unroll
    for (int k = 0; k < P; ++k) gw[k] *= u;
  }
  if (timed_out) acc2[0] = acc2[1] = __longlong_as_double(0x7ff8000000000000ll);
  step_partials<P, true>(acc2, gw, p_fwd, p_bwd, chained, chained ? ch.n_front : 0);
  DC_TRACE_END();
}

// ================================================================================================
// Pose mode in ONE pass (round 4): loss, dL/dw AND dL/d[R|t] of every scan from one launch
// ================================================================================================
// train() with pose corrections (train.py:300-312, eval.py:68-82; scripts/model_poses_learning:71) moves the poses every
// iteration, so the basis rows of the model-only step (X0 = R x + t) are stale after every step and the general path ran three
// full passes: dc_points_fwd (22 us) -> forward writing a record per centre (46) -> backward over the transposed table with
// per-scan sums (86; 113 in round 3).  Here one kernel does it:
//   * LOCAL basis rows (dc_points_local_basis, once per exponent set): {d0, dir, c_k, scan} in the SENSOR frame -- 32 B, nothing
//     in them depends on a pose.  Staging forms a row's world point with the CURRENT pose and weights: d' = d0 + sum w_k c_k,
//     x = R_s (d' dir) + t_s on the q32 grid, u = R_s dir; the sweeps then run as in consistency_step_q32_kernel.
//     The rows are stored PER BLOCK in the order of its list (24 B each, 1.75 x the points at C2; the scan of a row is a byte of
//     dcPoseTable.row_scan): staging reads them as one contiguous, coalesced stream -- as gathers through the id list they cost
//     two dependent memory latencies and ~2.3 cycles of the CU's address pipeline per row.
//   * reverse mode INSIDE the block for the poses: in the second sweep every centre ADDS its edges' gradients
//     g_ij = c1_i (v0_i . e) v0_i - c2_i e, e = x_j - mean_i, to the staged rows' sums in LDS (chain_term_pose: 64-bit integer
//     atomics under a per-block power-of-two scale, so the order they land in does not matter).  Rows shared by several blocks
//     get a partial sum in each; the sums over blocks are the reduction's.
//   * the block's distinct rows are listed BY SCAN (dcPoseTable.ids: (scan, id) order, row_seg = where each scan starts), so
//     dL/d[R|t]_s = (sum_j g_j (x_j - t_s)^T) R_s | sum_j g_j runs over a contiguous row range of the tile: eight lanes per scan,
//     fixed order, one row of the row-major pose partials per block.  Bitwise reproducible like everything else.
// 154 us of kernels in three launches -> one launch; see DESIGN 4 for the measured time.
struct PoseTab {
  const int32_t* __restrict__ blk_ptr;     // [blocks + 1], the forward table's
  const int32_t* __restrict__ ids;         // distinct rows of every block in (scan, id) order
  const uint16_t* __restrict__ loc;        // [blocks * K][256]: 16 x position in that order, 0xFFFF = empty slot
  const uint16_t* __restrict__ own_pos;    // [N]: 16 x position of the point's own row in its block's list
  const uint16_t* __restrict__ row_seg;    // [blocks][S + 1]: first row of every scan in the block's list; [S] = the row count
  const uint8_t* __restrict__ row_scan;    // the scan of every listed row (parallel to ids)
};
constexpr int kPoseCap = 512;              // rows of the static tile (the table builder refuses blocks with longer lists)

// Per block: (scan, id) order of its distinct rows, remapped positions, own positions.  info[0] <- 1 when a
// block cannot take the pose kernel (more than kPoseCap rows, or a block whose list misses one of its own rows).
template <int K>
__global__ __launch_bounds__(kBlock) void pose_table_kernel(BlockTab tab, const int32_t* __restrict__ own_base,
                                                            const int32_t* __restrict__ scan_id, int64_t n, int n_scans,
                                                            int32_t* __restrict__ ids_out, uint16_t* __restrict__ loc_out,
                                                            uint16_t* __restrict__ own_pos, uint16_t* __restrict__ row_seg,
                                                            uint8_t* __restrict__ row_scan, int32_t* __restrict__ info) {
  __shared__ int32_t s_id[kPoseCap];
  __shared__ uint8_t s_scan[kPoseCap];
  __shared__ uint16_t s_new[kPoseCap];
  __shared__ int s_start[kMaxBlockScans + 1];
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  const int32_t base = tab.blk_ptr[b], nd = tab.blk_ptr[b + 1] - base;
  const int32_t own = own_base[b];
  if (nd > kPoseCap || own < 0 || tab.slot_ptr[b + 1] - tab.slot_ptr[b] != K) {        // block-uniform
    if (tid == 0) atomicMax(info, 1);
    return;
  }
  if (tid <= n_scans) s_start[tid] = 0;
  __syncthreads();
  for (int t = tid; t < nd; t += kBlock) {
    const int32_t id = tab.blk_ids[base + t];
    const int sc = scan_id ? scan_id[id] : 0;
    s_id[t] = id;
    s_scan[t] = (uint8_t)sc;
    atomicAdd(&s_start[sc + 1], 1);                      // histogram, shifted by one for the prefix
  }
  __syncthreads();
  if (tid == 0) for (int q = 0; q < n_scans; ++q) s_start[q + 1] += s_start[q];
  __syncthreads();
  for (int t = tid; t < nd; t += kBlock) {
    const int sc = s_scan[t];
    int rank = 0;
    for (int t2 = 0; t2 < t; ++t2) rank += s_scan[t2] == sc ? 1 : 0;      // stable: ascending id inside a scan
    const int p = s_start[sc] + rank;
    s_new[t] = (uint16_t)p;
    ids_out[base + p] = s_id[t];
    row_scan[base + p] = (uint8_t)sc;
  }
  if (tid <= n_scans) row_seg[b * (n_scans + 1) + tid] = (uint16_t)s_start[tid];
  __syncthreads();
  const int64_t i = b * kBlock + tid;
  if (i < n) own_pos[i] = (uint16_t)(s_new[own + tid] << 4);
  const uint16_t* lrow = tab.loc + (int64_t)tab.slot_ptr[b] * kBlock + tid;
#pragma unroll
  for (int q = 0; q < K; ++q) {
    const uint16_t l = lrow[q * kBlock];
    loc_out[((int64_t)b * K + q) * kBlock + tid] = l == 0xFFFF ? (uint16_t)0xFFFF : (uint16_t)(s_new[l >> 4] << 4);
  }
}

// {d0, dir, c_0, c_1}: the pose-independent part of a ray (sensor frame; viewpoints at the sensor origin), 6 words.  One workgroup
// per block of the pose table writes the rows of the block's list, in its order, at rows [blk_ptr[b], blk_ptr[b + 1]).
template <typename T>
__global__ __launch_bounds__(kBlock) void points_local_basis_kernel(PointInputs in, PoseTab tab, int32_t* __restrict__ rows) {
  const int64_t b = blockIdx.x;
  const int32_t base = tab.blk_ptr[b], nd = tab.blk_ptr[b + 1] - base;
  ModelParams mp;
  load_model(in, mp);
  for (int t = threadIdx.x; t < nd; t += kBlock) {
    const int64_t i = tab.ids[base + t];
    const T* dp = (const T*)in.dirs + i * 3;
    const double d = (double)((const T*)in.depth)[i];
    const bool lm = in.lmask ? in.lmask[i] != 0 : true;
    const bool on = mp.kind != DC_MODEL_NONE && lm;
    const double inc = on ? (double)((const T*)in.inc)[i] : 0.0;
    const double d0 = (on && mp.kind == DC_MODEL_LINEAR) ? 0.0 : d;
    float c[2] = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k < mp.n_terms && on) {
        const double dk = mp.kind > DC_MODEL_SCALED_POLYNOMIAL ? model_dw_other(mp, k, d, inc)
                                                               : (mp.kind == DC_MODEL_SCALED_POLYNOMIAL ? -d : -1.0) * pow_term(inc, mp.e[k]);
        c[k] = (float)dk;
      }
    }
    int2* r = reinterpret_cast<int2*>(rows) + 3 * (int64_t)(base + t);
    r[0] = make_int2(__float_as_int((float)d0), __float_as_int((float)dp[0]));
    r[1] = make_int2(__float_as_int((float)dp[1]), __float_as_int((float)dp[2]));
    r[2] = make_int2(__float_as_int(c[0]), __float_as_int(c[1]));
  }
}

// second sweep of the pose kernel, one neighbour: chain_term_q32 plus the edge's gradient g_ij = al v0 - c2 e_j ADDED to the
// neighbour's row of the block's gradient planes.  The sums are 64-bit integers, so the order the wavefronts' LDS atomics land in
// does not matter (bit-reproducible); the coefficients arrive scaled by the block's power of two S with |g_ij| S < 2^50, and
// double(g) + 1.5 2^52 holds round(g) in its mantissa: the bit pattern minus that of 1.5 2^52 (low word zero) IS the integer.
template <int P, int CAP>
__device__ __forceinline__ void chain_term_pose(const int4* tile, unsigned long long* s_g, uint32_t off, bool have, const Pt<q32>::Raw& ci,
                                                const float* cmf, const float* vs, const float* vu, float c2f, float* gw) {
  const char* row = reinterpret_cast<const char*>(tile) + (have ? off : 0u);
  const int4 p0 = *reinterpret_cast<const int4*>(row);
  const int4 p1 = *reinterpret_cast<const int4*>(row + (size_t)CAP * 16);
  const float2v e01 = float2v{(float)(p0.x - ci.v[0]), (float)(p0.y - ci.v[1])} - float2v{cmf[0], cmf[1]};
  const float e0 = e01.x, e1 = e01.y, e2 = (float)(p0.z - ci.v[2]) - cmf[2];
  const float u0 = __int_as_float(p0.w), u1 = __int_as_float(p1.x), u2 = __int_as_float(p1.y);
  const float al = fmaf(vs[2], e2, fmaf(vs[1], e1, vs[0] * e0));                       // c1 (v . e_j)
  const float g0 = fmaf(al, vu[0], -(c2f * e0)), g1 = fmaf(al, vu[1], -(c2f * e1)), g2 = fmaf(al, vu[2], -(c2f * e2));
  float tj = fmaf(g2, u2, fmaf(g1, u1, g0 * u0));                                      // g_ij . u_j
  if (!have) tj = 0.0f;
  if constexpr (P == 2) {
    float2v g = float2v{gw[0], gw[1]};
    g = __builtin_elementwise_fma(float2v{tj, tj}, float2v{__int_as_float(p1.z), __int_as_float(p1.w)}, g);
    gw[0] = g.x; gw[1] = g.y;
  } else {
    gw[0] = fmaf(tj, __int_as_float(p1.z), gw[0]);
  }
  if (have) {
    constexpr double kMagic = 6755399441055744.0;                                      // 1.5 2^52
    unsigned long long* cell = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(s_g) + (off >> 1));
    const float gg[3] = {g0, g1, g2};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const unsigned long long bits = (unsigned long long)__double_as_longlong((double)gg[a] + kMagic) - 0x4338000000000000ull;
      atomicAdd(cell + a * CAP, bits);
    }
  }
}

// Pose-mode evaluation in ONE launch (float32 sequences, [rows, K] tables, no exponent gradients).  A block stages its distinct
// rows from the pose-independent local basis rows {d0, dir, c0, c1, scan} with the CURRENT poses and weights, runs the step kernel's
// two sweeps, and in the second sweep every centre adds its edges' gradients to the block's per-row gradient planes in LDS
// (chain_term_pose); the rows of one scan are contiguous in the block's list (dc_pose_table_build), so dL/d[R|t]_s of the block
// is a sum over a row range: one row [12 S] of the row-major pose partials per block, summed by reduce_eval_kernel.
template <int NS, int P>
__global__ __launch_bounds__(kBlock, 4) void consistency_step_pose_kernel(
    const int32_t* __restrict__ lrows, PoseTab tab, const double* __restrict__ poses, int n_scans, const double* __restrict__ w,
    int64_t n, const uint8_t* __restrict__ mask, LossParams lp, QParams qp, double* __restrict__ p_fwd, double* __restrict__ p_bwd) {
  constexpr int CAP = kPoseCap;
  __shared__ int4 tile[2 * CAP];                            // piece 0 {X, u0} | piece 1 {u1, u2, c0, c1}
  __shared__ unsigned long long s_g[3 * CAP];               // three planes: the rows' gradient sums (64-bit integers)
  __shared__ double s_pose[kLdsScans * 12];
  __shared__ float s_bound[kWavesPerBlock];
  const int tid = threadIdx.x;
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  const int64_t blk = xcd_block(nblocks);
  // one partial row per block in every column: {sum loss, count} at p_fwd, dL/dw, zeros for dL/de, the 12 S pose sums at p_bwd.  The
  // eight rows of a 64-byte line are blocks of ONE XCD (blockIdx & 7), so the line is completed in that XCD's L2, and the
  // reduction reads every column as one contiguous run.
  const int64_t rs = (int64_t)gridDim.x;
  double* pcol = p_bwd + 2 * P * rs + (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  double acc2[2] = {0.0, 0.0}, gw[P];
#pragma unroll
  for (int k = 0; k < P; ++k) gw[k] = 0.0;
  if (blk < 0) {                                            // padding block of the last round (block-uniform): zero rows
    for (int item = tid; item < 12 * n_scans; item += kBlock) pcol[item * rs] = 0.0;
  } else {
    for (int t = tid; t < n_scans * 12; t += kBlock) s_pose[t] = poses[t];
    double wq[P];
#pragma unroll
    for (int k = 0; k < P; ++k) wq[k] = w[k];
    const int64_t i = blk * kBlock + tid;
    const bool live = i < n;
    const bool in_mask = live && (mask ? mask[i] != 0 : true);
    const uint16_t* lrow = tab.loc + (blk * NS) * kBlock + tid;
    uint32_t pre[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) pre[q] = live ? (uint32_t)lrow[q * kBlock] : kNoLoc;
    const uint32_t own = live ? (uint32_t)tab.own_pos[i] : 0u;
    const int32_t base = tab.blk_ptr[blk], nd = tab.blk_ptr[blk + 1] - base;
    // the (at most two) rows this thread stages, in flight before anything else: the block's rows are one contiguous stream
    static_assert(CAP == 2 * kBlock, "two staged rows per thread");
    int2 rw[2][3];
    int rsc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = tid + j * kBlock < nd ? tid + j * kBlock : 0;
      const int2* src = reinterpret_cast<const int2*>(lrows) + 3 * (int64_t)(base + t);
      rw[j][0] = src[0]; rw[j][1] = src[1]; rw[j][2] = src[2];
      rsc[j] = tab.row_scan[base + t];
    }
    for (int t = tid; t < 3 * CAP; t += kBlock) s_g[t] = 0ull;
    __syncthreads();                                        // the poses are in LDS
    // ---- staging: the world point of every distinct row from its local basis row, the current pose and weights ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = tid + j * kBlock;
      if (t >= nd) continue;
      const float c0 = __int_as_float(rw[j][2].x), c1f = __int_as_float(rw[j][2].y);
      const int sc = rsc[j];
      double dp = (double)__int_as_float(rw[j][0].x) + wq[0] * (double)c0;
      if constexpr (P > 1) dp += wq[1] * (double)c1f;
      const double dl[3] = {(double)__int_as_float(rw[j][0].y), (double)__int_as_float(rw[j][1].x), (double)__int_as_float(rw[j][1].y)};
      const double* Tp = s_pose + sc * 12;
      double u[3], x[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        u[a] = Tp[4 * a] * dl[0] + Tp[4 * a + 1] * dl[1] + Tp[4 * a + 2] * dl[2];
        x[a] = Tp[4 * a + 3] + dp * u[a];
      }
      tile[t] = make_int4(quantize(x[0], qp.origin[0], qp.inv_scale, qp.flag), quantize(x[1], qp.origin[1], qp.inv_scale, qp.flag),
                          quantize(x[2], qp.origin[2], qp.inv_scale, qp.flag), __float_as_int((float)u[0]));
      tile[CAP + t] = make_int4(__float_as_int((float)u[1]), __float_as_int((float)u[2]), rw[j][2].x, rw[j][2].y);
    }
    __syncthreads();
    // ---- the centre: moments, smallest eigenpair, loss (as consistency_step_q32_kernel) and a bound of its edges' gradients ----
    const bool work = live && (!mask || __any((int)in_mask));
    Pt<q32>::Raw ci;
    double cm[3] = {0.0, 0.0, 0.0}, v0[3] = {0.0, 0.0, 0.0}, c1 = 0.0, c2 = 0.0;
    float bound = 0.0f;
    if (work) {
      const char* tb = reinterpret_cast<const char*>(tile);
      ci = Pt<q32>::from_row(reinterpret_cast<const int4*>(tb + own));
      CovAcc acc;
      cov_init(acc);
      uint32_t mo = pre[0];
#pragma unroll
      for (int q = 1; q < NS; ++q) mo |= pre[q];
      const bool any_miss = __any((int)(mo & 1u)) != 0;
      int n_have;
      if (any_miss) n_have = gather_fixed<q32, NS, true>(tile, CAP, ci, pre, acc);
      else n_have = gather_fixed<q32, NS, false>(tile, CAP, ci, pre, acc);
      acc.W = (double)n_have;
      double se2;
      step_point2<q32, NS>(acc, n_have, !any_miss, in_mask, lp, qp, acc2, cm, v0, &c1, &c2, &se2);
      // |g_ij| <= (|c1| + |c2|) |e_j| and |e_j|^2 <= sum_j |e_j|^2 (at least one grid unit, so that c S stays finite)
      const float r = sqrtf((float)se2);
      bound = (float)(fabs(c1) + fabs(c2)) * (r > 1.0f ? r : 1.0f);
    }
    bound = bound == bound ? bound : INFINITY;              // a NaN coefficient poisons the block's sums like an infinite one
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) bound = fmaxf(bound, __shfl_xor(bound, o, kWave));
    if ((tid & (kWave - 1)) == 0) s_bound[tid / kWave] = bound;
    __syncthreads();
    bound = s_bound[0];
#pragma unroll
    for (int q = 1; q < kWavesPerBlock; ++q) bound = fmaxf(bound, s_bound[q]);
    const bool poisoned = !(bound < INFINITY);              // block-uniform
    int sh = 0;
    if (bound > 0.0f && !poisoned) {
      int ex;
      (void)frexpf(bound, &ex);                             // bound < 2^ex
      sh = 50 - ex;
      sh = sh > 100 ? 100 : sh;
    }
    const double S = ldexp(1.0, sh), invS = ldexp(1.0, -sh);
    // ---- second sweep: dL/dw of the centre, and its edges' gradients into the rows' planes ----
    if (work && !poisoned) {
      float cmf[3], vs[3], vu[3], gwf[P];
      const double c1s = c1 * S;
#pragma unroll
      for (int a = 0; a < 3; ++a) { cmf[a] = (float)cm[a]; vs[a] = (float)(c1s * v0[a]); vu[a] = (float)v0[a]; }
      const float c2f = (float)(c2 * S);
#pragma unroll
      for (int k = 0; k < P; ++k) gwf[k] = 0.0f;
      if (__any((int)(c1 != 0.0 || c2 != 0.0))) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          if (q % 4 == 0 && q > 0) __builtin_amdgcn_sched_barrier(0);
          chain_term_pose<P, CAP>(tile, s_g, pre[q], pre[q] != kNoLoc && (c1 != 0.0 || c2 != 0.0), ci, cmf, vs, vu, c2f, gwf);
        }
      }
#pragma unroll
      for (int k = 0; k < P; ++k) gw[k] = (double)gwf[k] * (qp.scale * invS);
    }
    __syncthreads();                                        // every edge has been added
    // ---- dL/d[R|t]_s = sum_j g_j (x) [x_local_j, 1] over the rows of scan s, contiguous in the block's list.  With
    //      x_local = R^T (x - t) the sum is (sum_j g_j (x_j - t)^T) R: eight lanes per scan (n_scans <= 32) take every eighth row
    //      each and sum g (x) q and g over them -- q the row's grid coordinates, g its three integer sums, all in LDS -- the eight
    //      lanes' twelve sums are added by DPP quad / row operations that also halve what a lane carries (as wave_sum4_dpp), and
    //      lanes 0..2 of the eight finish row a of [dL/dR | dL/dt] with the scan's pose ----
    {
      const uint16_t* seg = tab.row_seg + blk * (n_scans + 1);
      const int sc = tid >> 3, part = tid & 7;
      const bool mine = sc < n_scans;
      const int end = mine ? (int)seg[sc + 1] : 0;
      // slot (p & 1) 6 + (p >> 1) 3 + c ends on lane p of the eight: lanes 0..2 get {sum g_a q_c}, a = p; lane 3 {sum g_a}
      double sacc[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) sacc[q] = 0.0;
      for (int p = mine ? (int)seg[sc] + part : 0; p < end; p += 8) {
        const int4 xr = tile[p];
        const double qd[3] = {(double)xr.x, (double)xr.y, (double)xr.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const unsigned long long bits = s_g[a * CAP + p];
          const double g = fma((double)(int)(uint32_t)(bits >> 32), 4294967296.0, (double)(uint32_t)bits);
#pragma unroll
          for (int c = 0; c < 3; ++c) sacc[(a & 1) * 6 + (a >> 1) * 3 + c] = fma(g, qd[c], sacc[(a & 1) * 6 + (a >> 1) * 3 + c]);
          sacc[9 + a] += g;                                 // lane 3: (3 & 1) 6 + (3 >> 1) 3 = 9
        }
      }
      const bool up1 = (part & 1) != 0, up2 = (part & 2) != 0;
      double h[6], r3[3], gs[3];
#pragma unroll
      for (int q = 0; q < 6; ++q) h[q] = (up1 ? sacc[6 + q] : sacc[q]) + dpp_f64<kDppXor1>(up1 ? sacc[q] : sacc[6 + q]);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        r3[q] = (up2 ? h[3 + q] : h[q]) + dpp_f64<kDppXor2>(up2 ? h[q] : h[3 + q]);
        r3[q] += dpp_f64<kDppShl4>(r3[q]);                  // lanes 0..3 of the eight: + lanes 4..7
        gs[q] = dpp_f64<kDppQuad3>(r3[q]);                  // {sum g_a} from lane 3 of the quad
      }
      if (mine && part < 3) {
        const double unscale = qp.scale * invS;             // the block's gradient unit
        const double* Tp = s_pose + sc * 12;
        const double ga = (part == 0 ? gs[0] : (part == 1 ? gs[1] : gs[2])) * unscale;
        double m[3];                                        // row a of sum_j g_j (x_j - t)^T
#pragma unroll
        for (int c = 0; c < 3; ++c) m[c] = fma(r3[c] * unscale, qp.scale, ga * (qp.origin[c] - Tp[4 * c + 3]));
        double* dst = pcol + (sc * 12 + part * 4) * rs;
#pragma unroll
        for (int b2 = 0; b2 < 3; ++b2) {
          const double v = fma(m[2], Tp[8 + b2], fma(m[1], Tp[4 + b2], m[0] * Tp[b2]));
          dst[b2 * rs] = poisoned ? (double)NAN : v;
        }
        dst[3 * rs] = poisoned ? (double)NAN : ga;
      }
    }
  }
  // {sum loss, count, dL/dw} of the wavefront; the exponent-gradient columns [P, 2P) of this evaluation are zero
  if (tid < P) p_bwd[(P + tid) * rs + blockIdx.x] = 0.0;
  step_partials<P, true>(acc2, gw, p_fwd, p_bwd, true, 0);
}

// Backward in basis form over a run table: the point itself and the chain to the weights come from the basis rows.
// partial rows: [0, P) dL/dw (the exponent slots [P, 2P) are written as zeros).
template <typename PT, int P>
__global__ __launch_bounds__(kBlock) void consistency_bwd_basis_kernel(
    PointBasis pb, const PT* __restrict__ rec, RunTab tab, int cap, int64_t n, QParams qp, double* __restrict__ partials) {
  constexpr int RR = RecRaw<PT>::kRow16;
  constexpr int NP = P > 0 ? P : DC_MAX_MODEL_TERMS;
  extern __shared__ int4 tile[];
  __shared__ double s_w[DC_MAX_MODEL_TERMS];
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  const int64_t blk = xcd_block(nblocks);
  double gw[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) gw[k] = 0.0;
  const int64_t j = blk * kBlock + threadIdx.x;
  const bool active = blk >= 0 && j < n;
  uint2 pre[kPreRuns];
  int32_t nruns = 0;
  uint32_t nd = 0;
  const uint2* runs = reinterpret_cast<const uint2*>(tab.loc);
  if (blk >= 0) {
    if (active) {
      const int32_t r0 = tab.run_ptr[j];
      nruns = tab.run_ptr[j + 1] - r0;
      runs += r0;
    }
#pragma unroll
    for (int t = 0; t < kPreRuns; ++t) pre[t] = t < nruns ? runs[t] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
    stage_weights(pb, s_w);
    nd = (uint32_t)stage_rows<RR>(tab.blk_ptr, tab.blk_ids, blk, reinterpret_cast<const int4*>(rec), tile, cap);
    if (threadIdx.x < RR) tile[threadIdx.x * cap + nd] = make_int4(0, 0, 0, 0);
  }
  const uint32_t nd16 = nd * 16u;
  __syncthreads();
  if (active) {
    double wq[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) wq[k] = (P > 0 || k < pb.n_terms) ? s_w[k] : 0.0;
    const typename Pt<PT>::Raw cj = Basis<PT>::template point<P>(pb, wq, j);
    double g[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < kPreRuns; ++t)
      if (__any((int)(t < nruns))) run_edges<PT>(tile, cap, pre[t], nd16, cj, g);
    if (__any((int)(nruns > kPreRuns))) {
      uint2 nxt = kPreRuns < nruns ? runs[kPreRuns] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
      for (int t = kPreRuns; __any((int)(t < nruns)); ++t) {
        const uint2 r = nxt;
        nxt = t + 1 < nruns ? runs[t + 1] : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        run_edges<PT>(tile, cap, r, nd16, cj, g);
      }
    }
    const double u = Pt<PT>::unit(qp);
    g[0] *= u; g[1] *= u; g[2] *= u;
    // u_j and c_j again (the row is still in the cache): holding them across the edge loop costs a wavefront of occupancy
    Basis<PT>::template chain<NP>(pb, P > 0 ? P : pb.n_terms, j, g, gw);
  }
  // per-wavefront partial rows, as reduce_param_grads writes them
  const int64_t rs = (int64_t)gridDim.x * kWavesPerBlock;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  double* prow = partials + (int64_t)blockIdx.x * kWavesPerBlock + wave;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    if (k < pb.n_terms) {
      const double sw = wave_sum(gw[k]);
      if (lane == 0) { prow[k * rs] = sw; prow[(pb.n_terms + k) * rs] = 0.0; }
    }
  }
}

// Quantile-inlier gating of the pointwise loss (loss.py:256-277) for the fused path: `raw` is the forward's raw loss
// (DC_LOSS_RAW_POINTWISE), *threshold the bound the caller derived from it (quantile x multiplier, or the given maximum).
// A masked centre with raw loss above the bound (or NaN: `<=` fails as in torch) is dropped: its record's coefficients
// are zeroed, so the backward passes nothing through it; the sums become {sum of relu / sqrt loss, count} of the inliers.
template <typename PT> struct RecWord { using type = PT; };
template <> struct RecWord<q32> { using type = int32_t; };          // c1 / c2 are float bits: all-zero bits = 0.0f
template <typename T, typename PT>
__global__ __launch_bounds__(kBlock) void consistency_gate_kernel(const T* __restrict__ raw, const uint8_t* __restrict__ mask, int64_t n,
                                                                  const double* __restrict__ threshold, int sqrt_,
                                                                  PT* __restrict__ rec, double* __restrict__ partials) {
  const int64_t nblocks = (n + kBlock - 1) / kBlock;
  const int64_t blk = xcd_block(nblocks);
  double acc2[2] = {0.0, 0.0};
  const int64_t i = blk * kBlock + threadIdx.x;
  if (blk >= 0 && i < n) {
    const bool m = mask ? mask[i] != 0 : true;
    const T r = raw[i];
    const bool in = m && (r <= (T)*threshold);        // compared in the cloud dtype, like the reference's tensors
    if (in) {
      double l = (double)r;
      l = l > 0.0 ? l : 0.0;
      if (sqrt_) l = sqrt(l);
      acc2[0] = l; acc2[1] = 1.0;
    } else if (m) {
      using W = typename RecWord<PT>::type;
      W* row = reinterpret_cast<W*>(rec) + i * 8;
      row[3] = W(0); row[7] = W(0);
    }
  }
  wave_partials<2>(acc2, partials);
}

// Stand-alone point epilogue for the un-fused API path (grad of points given).
template <typename T, int STRIDE>
__global__ __launch_bounds__(kBlock) void points_bwd_kernel(const T* __restrict__ grad_x, const int32_t* __restrict__ perm,
                                                            int64_t n, PointInputs in, int want_e, int want_pose,
                                                            double* __restrict__ partials, int n_acc) {
  __shared__ double lds[(kBlock / kWave) * 2 * DC_MAX_MODEL_TERMS];
  __shared__ double s_pose[kLdsScans * 12];
  const PoseTile poses = stage_poses(in, s_pose);
  __syncthreads();
  ModelParams mp;
  load_model(in, mp);
  double gw[DC_MAX_MODEL_TERMS], ge[DC_MAX_MODEL_TERMS], gT[6];
#pragma unroll
  for (int k = 0; k < DC_MAX_MODEL_TERMS; ++k) gw[k] = ge[k] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) gT[k] = 0.0;
  int scan = -1;
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool active = j < n;
  if (active) {
    double g[3];
    Row3<T, STRIDE>::load(grad_x, perm ? (int64_t)perm[j] : j, g, QParams{});      // perm: grad rows live in another point order
    points_bwd_point<T>(in, poses, mp, j, g, gw, ge, gT, want_e != 0, want_pose != 0, &scan);
  }
  reduce_param_grads<T>(in, active, want_e, want_pose, gw, ge, gT, scan, lds, partials);
}

// Sum the block partials [n_acc][n_rows] in a fixed order into out[n_acc]: one 1024-lane block per accumulator,
// contiguous (coalesced) reads.
constexpr int kRedBlock = 1024;
// Sum of p[threadIdx.x], p[threadIdx.x + 1024], ... in a fixed order with 32 loads in flight per lane: the rows were
// written by blocks on every XCD, so each read is a trip to the fabric, and with four in flight the ~31 rows per lane
// (N = 2 M) took eight dependent round trips.
constexpr int kPoseRedCols = 8;          // columns of the row-major pose partials a block of reduce_eval_kernel sums: one 64-byte line
__device__ __forceinline__ double strided_sum(const double* __restrict__ p, int64_t n_rows, int64_t stride = 1, int first = -1,
                                              int step = kRedBlock) {
  constexpr int U = 32;                  // 32 k rows (N = 2 M: one row per wavefront) in ONE round trip per lane
  double acc[U];
#pragma unroll
  for (int u_ = 0; u_ < U; ++u_) acc[u_] = 0.0;
  int64_t r = first < 0 ? (int)threadIdx.x : first;        // this lane's rows: r, r + step, ...
  for (; r + (int64_t)(U - 1) * step < n_rows; r += (int64_t)U * step) {
#pragma unroll
    for (int u_ = 0; u_ < U; ++u_) acc[u_] += p[(r + (int64_t)u_ * step) * stride];
  }
  {
    double last[U];                      // the tail's loads are issued together, too
#pragma unroll
    for (int u_ = 0; u_ < U; ++u_) last[u_] = (r + (int64_t)u_ * step < n_rows) ? p[(r + (int64_t)u_ * step) * stride] : 0.0;
#pragma unroll
    for (int u_ = 0; u_ < U; ++u_) acc[u_] += last[u_];
  }
#pragma unroll
  for (int w = U / 2; w > 0; w >>= 1) {
#pragma unroll
    for (int u_ = 0; u_ < w; ++u_) acc[u_] += acc[u_ + w];
  }
  return acc[0];
}
__global__ __launch_bounds__(kRedBlock) void reduce_partials_kernel(const double* __restrict__ partials, int64_t n_rows,
                                                                    double* __restrict__ out) {
  __shared__ double lds[kRedBlock / kWave];
  const double* p = partials + (int64_t)blockIdx.x * n_rows;
  const double s = wave_sum(strided_sum(p, n_rows));
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == 0) lds[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int wv = 0; wv < kRedBlock / kWave; ++wv) t += lds[wv];
    out[blockIdx.x] = t;
  }
}

// One launch for a whole evaluation: out[0..2) <- forward partials, out[2..2+n_red) <- backward partials,
// out[2+n_red..2+n_acc) <- 0 (gradient slots that were not requested).  With `adam.p` the block that finishes
// dL/dw_i also takes the Adam step of w_i (dc_sequence_step: the kernels of the next evaluation come after it on
// the stream).
__global__ __launch_bounds__(kRedBlock) void reduce_eval_kernel(const double* __restrict__ p_fwd, const double* __restrict__ p_bwd,
                                                                int64_t rows_fwd, int64_t rows_bwd, int n_red, int n_out,
                                                                double* __restrict__ out, AdamArgs adam,
                                                                const int32_t* __restrict__ status, int pose_first = -1,
                                                                int pose_cols = 0) {
  // pose_first >= 0: the accumulators from 2 + pose_first on (the pose slots) were written ROW-major [rows_bwd / 4][pose_cols]
  // behind the column-major ones (one row per block of the backward / pose kernel).  A block of this kernel then sums EIGHT
  // neighbouring columns -- one 64-byte line of every row: lane & 7 is the column, the other lane bits the row -- so every line is
  // fetched once, by one XCD (a block per column fetched each line eight times: 16 us at N = 2 M, 20 scans, against 4).
  __shared__ double lds[kRedBlock / kWave * kPoseRedCols];
  const int a = blockIdx.x;                // grid = red_grid(): only the sums that were asked for get a block
  if (a == 0)                              // block 0 also clears the slots of gradients that were not requested
    for (int z = 2 + n_red + threadIdx.x; z < n_out; z += kRedBlock) out[z] = 0.0;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (pose_first >= 0 && a >= 2 + pose_first) {
    const int col = (a - 2 - pose_first) * kPoseRedCols + (threadIdx.x & (kPoseRedCols - 1));
    const bool have = col < pose_cols;
    const double* p = p_bwd + (int64_t)pose_first * rows_bwd + (have ? col : 0);
    double s = strided_sum(p, have ? rows_bwd / kWavesPerBlock : 0, pose_cols, threadIdx.x / kPoseRedCols, kRedBlock / kPoseRedCols);
#pragma unroll
    for (int o = kPoseRedCols; o < kWave; o <<= 1) s += __shfl_xor(s, o, kWave);
    if (lane < kPoseRedCols) lds[wave * kPoseRedCols + lane] = s;
    __syncthreads();
    if (threadIdx.x < kPoseRedCols && have) {
      double t = 0.0;
      for (int wv = 0; wv < kRedBlock / kWave; ++wv) t += lds[wv * kPoseRedCols + threadIdx.x];
      out[2 + pose_first + col] = t;
    }
    return;
  }
  const bool flagged = a == 0 && threadIdx.x == 0 && status && *status != 0;      // requested before the rows, not after
  const int64_t n_rows = a < 2 ? rows_fwd : rows_bwd;
  const double* p = a < 2 ? p_fwd + (int64_t)a * rows_fwd : p_bwd + (int64_t)(a - 2) * rows_bwd;
  const bool step = adam.p && a >= 2 && a - 2 < adam.n && threadIdx.x == 0;
  double p0 = 0.0, m0 = 0.0, v0 = 0.0;
  if (step) { p0 = adam.p[a - 2]; m0 = adam.m[a - 2]; v0 = adam.v[a - 2]; }      // in flight together with the partial rows
  const double s = wave_sum(strided_sum(p, n_rows));
  if (lane == 0) lds[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int wv = 0; wv < kRedBlock / kWave; ++wv) t += lds[wv];
    // points that did not fit the fixed-point extent (or were NaN) make the evaluation meaningless: say so in the loss
    if (flagged) t = __longlong_as_double(0x7ff8000000000000ll);
    out[a] = t;
    if (step) adam_apply(adam, a - 2, t, p0, m0, v0);
  }
}

// blocks of reduce_eval_kernel: one per column-major sum, one per kPoseRedCols columns of row-major pose partials
static inline unsigned red_grid(int n_red, int pose_first, int pose_cols) {
  if (pose_first < 0 || n_red <= pose_first) return 2u + (unsigned)n_red;
  return 2u + (unsigned)pose_first + (unsigned)((pose_cols + kPoseRedCols - 1) / kPoseRedCols);
}

__global__ void adam_kernel(const double* __restrict__ grad, AdamArgs adam) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < adam.n) adam_update(adam, i, grad[i]);
}

// The same with the step counter on the device (one block): t = *step + 1 enters the bias corrections and is written back,
// so the launch can be captured into a hipGraph and replayed -- a host-side counter would be frozen into the graph.
__global__ __launch_bounds__(256) void adam_device_step_kernel(const double* __restrict__ grad, AdamArgs adam, int64_t* __restrict__ step) {
  const double t = (double)(*step + 1);
  adam.bias1 = 1.0 - pow(adam.b1, t);
  adam.bias2_sqrt = sqrt(1.0 - pow(adam.b2, t));
  for (int i = threadIdx.x; i < adam.n; i += blockDim.x) adam_update(adam, i, grad[i]);
  __syncthreads();                                   // every lane has read the counter
  if (threadIdx.x == 0) *step = (int64_t)t;
}

}  // namespace dc

// ================================================================================================
// C ABI
// ================================================================================================
using namespace dc;

#define DC_CHECK_LAUNCH()                                  \
  do {                                                     \
    hipError_t err__ = hipGetLastError();                  \
    if (err__ != hipSuccess) return (int)err__;            \
  } while (0)

static inline int64_t n_blocks(int64_t n) { return (n + kBlock - 1) / kBlock; }
// dc_set_option(0, 1): ignore block tables, gather from global memory (A-B measurements); process-wide, read per launch
static std::atomic<bool> g_no_reverse{false};        // dc_set_option(8, 1): every launch of a chain walks the blocks forwards (A-B, chain_block_of)
static std::atomic<bool> g_pose_three_pass{false};   // dc_set_option(7, 1): pose gradients through the three-kernel general path (A-B, tests)
static std::atomic<bool> g_no_tab{false};
static std::atomic<int> g_fwd_generic{0};
static std::atomic<bool> g_two_pass{false};     // dc_set_option(4, 1): basis form with separate forward and backward kernels
static std::atomic<int> g_step_var{1};          // dc_set_option(6, v): 1 = consistency_step_q32_kernel for float32 clouds with a [rows, K] table (default),
                                                // 7 = consistency_step_basis_kernel<.., kStepVar> for them too, 0 = its round-2 form (K = 10, P = 2 only: A-B baseline)
static std::atomic<int> g_chain_spin{1 << 22};  // dc_set_option(5, n): polls of a chained launch's wait for its weights (tests force 0)
static std::atomic<bool> g_no_basis{false};    // dc_set_option(3, 1): ignore a sequence's basis rows (general path)    // dc_set_option(1, 1): run-time slot loop instead of the fixed-K forward kernels

// consistency_step_ragged_q32_kernel with a tile of CAP rows: more than 64 KB of LDS per block needs the attribute (once per
// instantiation, process and device).  Capacities up to the table's own limit (4095 rows), so every ball-neighbourhood table that can be
// built runs fused -- at one block per CU for the densest (voxel grid 0.1 m, r = 0.25 m: 2 000 distinct rows per block).
template <int P, int CAP>
static int ragged_launch(dim3 grid, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, const PointBasis& pb, const BlockTab& tab,
                         const dcBlockTable* t, int64_t n_rows, const uint8_t* mask, const LossParams& lp, const QParams& qp, double* p_fwd,
                         double* p_bwd, const StepChain& ch) {
  constexpr size_t bytes = (size_t)StepRow<q32, P>::kPieces * CAP * 16;
  // function attributes are per device: one bit per device of the process (a second GPU of the same process sets its own)
  static std::atomic<uint64_t> attr_set{0};
  if (bytes > 60 * 1024) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_set.fetch_or(bit) & bit)) {
      hipError_t err = hipFuncSetAttribute((const void*)consistency_step_ragged_q32_kernel<P, CAP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
      if (err != hipSuccess) { attr_set.fetch_and(~bit); return (int)err; }
    }
  }
  hipExtLaunchKernelGGL((consistency_step_ragged_q32_kernel<P, CAP>), grid, dim3(kBlock), bytes, stream, ev0, ev1, 0, pb, tab, t->own_base, t->row_ptr, n_rows,
                        mask, lp, qp, p_fwd, p_bwd, ch);
  return DC_OK;
}

// a usable table of the wanted layout -> LDS bytes / rows of the staged tile (+ `extra_rows`), which must fit `lds_limit`
static bool use_table(const dcBlockTable* t, int layout, int stride, uint32_t row_bytes, int extra_rows, size_t lds_limit,
                      size_t* lds_bytes, int* lds_rows) {
  if (!t || g_no_tab.load() || stride != 4 || t->layout != layout || !t->blk_ptr || !t->loc || t->max_rows < 0) return false;
  if (layout == DC_TABLE_SLOTS ? !t->slot_ptr : !t->run_ptr) return false;
  if (t->max_rows > 0 && !t->blk_ids) return false;
  const size_t rows = (size_t)t->max_rows + extra_rows + (t->max_rows + extra_rows == 0 ? 1 : 0);
  const size_t need = rows * row_bytes;
  if (need > lds_limit || t->max_rows >= 0xFFF) return false;       // positions are 16 x row in 16 bits
  *lds_bytes = need;
  *lds_rows = (int)rows;
  return true;
}

extern "C" {

int dc_version(void) { return 100; }

int64_t dc_partial_rows(int64_t n) { return xcd_grid(n_blocks(n)) * kWavesPerBlock; }

int dc_param_grad_count(int n_terms, int n_scans) { return 2 * n_terms + 12 * n_scans; }

// ordinary evaluations: rows x (2 + 2 P + 12 S) columns; chained steps: two buffers of (2 + P) columns behind them (chain_buffer)
int64_t dc_sequence_partials_count(int64_t n, int n_terms, int n_scans) {
  if (n < 0 || n_terms < 0 || n_scans < 0) return 0;
  const int64_t rows = xcd_grid(n_blocks(n)) * kWavesPerBlock;
  return rows * (2 + 2 * (int64_t)n_terms + 12 * (int64_t)n_scans) + 2 * (2 + (int64_t)n_terms) * rows;
}

static PointInputs make_inputs(const void* vps, const void* dirs, const void* depth, const void* inc,
                               const uint8_t* lmask, const int32_t* scan_id, const double* poses, int n_scans,
                               int model_kind, int n_terms, const double* w, const double* e) {
  PointInputs in;
  in.vps = vps; in.dirs = dirs; in.depth = depth; in.inc = inc; in.lmask = lmask; in.scan_id = scan_id;
  in.poses = poses; in.w = w; in.e = e; in.model_kind = model_kind; in.n_terms = n_terms; in.n_scans = n_scans;
  return in;
}

static int check_model(int model_kind, int n_terms, const void* inc, const double* w, const double* e) {
  if (model_kind < DC_MODEL_NONE || model_kind > DC_MODEL_LAST) return DC_ERR_ARG;
  if (model_kind != DC_MODEL_NONE) {
    if (n_terms < 1 || n_terms > DC_MAX_MODEL_TERMS || !inc || !w || !e) return DC_ERR_ARG;
    if (model_kind == DC_MODEL_LINEAR && n_terms != 3) return DC_ERR_ARG;
    if ((model_kind == DC_MODEL_INVCOS || model_kind == DC_MODEL_SCALED_INVCOS) && n_terms != 1) return DC_ERR_ARG;
  }
  return DC_OK;
}

static int make_qparams(int point_fmt, int dtype, int stride, const double* qparams, QParams* qp, int32_t* flag = nullptr) {
  *qp = QParams{};
  qp->flag = flag;
  if (point_fmt == DC_Q32) {
    if (!qparams || stride != 4 || dtype != DC_F32 || !(qparams[3] > 0.0)) return DC_ERR_ARG;
    qp->origin[0] = qparams[0]; qp->origin[1] = qparams[1]; qp->origin[2] = qparams[2];
    qp->scale = qparams[3];
    qp->inv_scale = 1.0 / qparams[3];
    return DC_OK;
  }
  return point_fmt == dtype ? DC_OK : DC_ERR_DTYPE;      // float / double points share the inputs' dtype
}

// Dispatch over (input dtype, point format, row stride).
#define DC_DISPATCH_FMT(dtype, fmt, stride, LAUNCH)                       \
  do {                                                                    \
    if ((fmt) == DC_Q32) { LAUNCH(float, q32, 4); }                       \
    else if ((dtype) == DC_F32) { if ((stride) == 3) { LAUNCH(float, float, 3); } else { LAUNCH(float, float, 4); } } \
    else if ((dtype) == DC_F64) { if ((stride) == 3) { LAUNCH(double, double, 3); } else { LAUNCH(double, double, 4); } } \
    else return DC_ERR_DTYPE;                                             \
  } while (0)

int dc_points_fwd(const void* vps, const void* dirs, const void* depth, const void* inc, const uint8_t* lmask,
                  const int32_t* scan_id, const double* poses, int n_scans, int model_kind, int n_terms,
                  const double* w, const double* e, int64_t n, int dtype, int point_fmt, const double* qparams,
                  int out_stride, void* points_out, void* vps_out, void* dirs_out, void* depth_out, int32_t* status,
                  hipStream_t stream) {
  if (n == 0) return DC_OK;
  if (n < 0 || !dirs || !depth || !points_out || (out_stride != 3 && out_stride != 4)) return DC_ERR_ARG;
  if (scan_id && (!poses || n_scans < 1)) return DC_ERR_ARG;
  int rc = check_model(model_kind, n_terms, inc, w, e);
  if (rc) return rc;
  QParams qp;
  rc = make_qparams(point_fmt, dtype, out_stride, qparams, &qp, status);
  if (rc) return rc;
  if (n == 0) return DC_OK;
  if (model_kind == DC_MODEL_NONE) n_terms = 0;
  PointInputs in = make_inputs(vps, dirs, depth, inc, lmask, scan_id, poses, n_scans, model_kind, n_terms, w, e);
  dim3 grid((unsigned)n_blocks(n)), block(kBlock);
#define LAUNCH(T, PT, S) \
  DC_TIMED_LAUNCH((points_fwd_kernel<T, PT, S>), grid, block, 0, stream, in, n, qp, (PT*)points_out, (T*)vps_out, (T*)dirs_out, (T*)depth_out)
  { ProfScope prof(0); DC_DISPATCH_FMT(dtype, point_fmt, out_stride, LAUNCH); }
#undef LAUNCH
  DC_CHECK_LAUNCH();
  return DC_OK;
}

int dc_points_basis(const void* vps, const void* dirs, const void* depth, const void* inc, const uint8_t* lmask,
                    const int32_t* scan_id, const double* poses, int n_scans, int model_kind, int n_terms, const double* e,
                    int64_t n, int dtype, const double* qparams, void* rows_out, int32_t* status, hipStream_t stream) {
  if (n == 0) return DC_OK;
  if (n < 0 || !dirs || !depth || !rows_out) return DC_ERR_ARG;
  if (scan_id && (!poses || n_scans < 1)) return DC_ERR_ARG;
  if (dtype != DC_F32 && dtype != DC_F64) return DC_ERR_DTYPE;
  if (dtype == DC_F32 && !qparams) return DC_ERR_ARG;         // float32 clouds: rows on the q32 grid
  // the weights do not enter the basis rows: a dummy non-null pointer satisfies the model check, load_model reads e only... and w
  int rc = check_model(model_kind, n_terms, inc, e, e);
  if (rc || model_kind == DC_MODEL_NONE) return rc ? rc : DC_ERR_ARG;
  QParams qp;
  rc = make_qparams(dtype == DC_F32 ? DC_Q32 : DC_F64, dtype, 4, qparams, &qp, status);
  if (rc) return rc;
  PointInputs in = make_inputs(vps, dirs, depth, inc, lmask, scan_id, poses, n_scans, model_kind, n_terms, e, e);
  if (dtype == DC_F32)
    hipLaunchKernelGGL((points_basis_kernel<q32>), dim3((unsigned)n_blocks(n)), dim3(kBlock), 0, stream, in, n, qp, rows_out);
  else
    hipLaunchKernelGGL((points_basis_kernel<double>), dim3((unsigned)n_blocks(n)), dim3(kBlock), 0, stream, in, n, qp, rows_out);
  DC_CHECK_LAUNCH();
  return DC_OK;
}

int dc_points_local_basis(const void* dirs, const void* depth, const void* inc, const uint8_t* lmask, const int32_t* scan_id,
                          int model_kind, int n_terms, const double* e, int64_t n, int dtype, const dcPoseTable* table, void* rows_out,
                          hipStream_t stream) {
  if (n == 0) return DC_OK;
  if (n < 0 || !dirs || !depth || !rows_out || n_terms < 1 || n_terms > 2 || !table || !table->blk_ptr || !table->ids) return DC_ERR_ARG;
  if (dtype != DC_F32) return DC_ERR_DTYPE;                  // float32 clouds (q32 points): what the pose kernel takes
  int rc = check_model(model_kind, n_terms, inc, e, e);
  if (rc || model_kind == DC_MODEL_NONE) return rc ? rc : DC_ERR_ARG;
  PointInputs in = make_inputs(nullptr, dirs, depth, inc, lmask, scan_id, nullptr, 1, model_kind, n_terms, e, e);
  PoseTab tab{table->blk_ptr, table->ids, table->loc, table->own_pos, table->row_seg, table->row_scan};
  hipLaunchKernelGGL((points_local_basis_kernel<float>), dim3((unsigned)n_blocks(n)), dim3(kBlock), 0, stream, in, tab, (int32_t*)rows_out);
  DC_CHECK_LAUNCH();
  return DC_OK;
}

int dc_pose_table_build(const dcBlockTable* fwd, const int32_t* scan_id, int64_t n, int n_scans, int k, int32_t* ids_out,
                        uint16_t* loc_out, uint16_t* own_pos, uint16_t* row_seg, uint8_t* row_scan, int32_t* info, hipStream_t stream) {
  if (!fwd || fwd->layout != DC_TABLE_SLOTS || !fwd->blk_ptr || !fwd->blk_ids || !fwd->slot_ptr || !fwd->loc || !fwd->own_base) return DC_ERR_ARG;
  if (n < 1 || n_scans < 1 || n_scans > kMaxBlockScans || !ids_out || !loc_out || !own_pos || !row_seg || !row_scan || !info) return DC_ERR_ARG;
  if (k != 4 && k != 8 && k != 10 && k != 16) return DC_ERR_UNSUPPORTED;
  hipError_t err = hipMemsetAsync(info, 0, sizeof(int32_t), stream);
  if (err != hipSuccess) return (int)err;
  BlockTab tab{fwd->blk_ptr, fwd->blk_ids, fwd->slot_ptr, fwd->loc};
  const dim3 grid((unsigned)n_blocks(n)), block(kBlock);
#define PT_K(KK) hipLaunchKernelGGL((pose_table_kernel<KK>), grid, block, 0, stream, tab, fwd->own_base, scan_id, n, n_scans, ids_out, loc_out, own_pos, \
                                    row_seg, row_scan, info)
  if (k == 10) PT_K(10); else if (k == 4) PT_K(4); else if (k == 8) PT_K(8); else PT_K(16);
#undef PT_K
  DC_CHECK_LAUNCH();
  return DC_OK;
}

// `reduce` = false leaves the block partials in partials_ws for a later combined reduction (dc_sequence_eval).
static int consistency_fwd_impl(const void* points, int stride, int dtype, int point_fmt, const double* qparams,
                                const int32_t* nbr, const int32_t* centre_idx, const dcBlockTable* table, int64_t n, int k,
                                const uint8_t* mask,
                                const void* offset, int loss_kind,
                                int normalization, int sqrt_, void* rec, void* pointwise, void* eigvals, double* partials_ws,
                                double* sums_out, hipStream_t stream, bool reduce) {
  if (n == 0 && sums_out) return (int)hipMemsetAsync(sums_out, 0, 2 * sizeof(double), stream);
  if (n < 0 || k < 1 || !points || !partials_ws || !sums_out || (stride != 3 && stride != 4)) return DC_ERR_ARG;
  if ((loss_kind & 0xFF) != DC_LOSS_MIN_EIGVAL && (loss_kind & 0xFF) != DC_LOSS_TRACE) return DC_ERR_ARG;
  if (loss_kind & ~(0xFF | DC_LOSS_RAW_POINTWISE | DC_LOSS_SKIP_NANS | DC_LOSS_ONLY_FINITE)) return DC_ERR_ARG;
  BlockTab tab{};
  size_t lds_bytes = 0;
  int lds_rows = 0;
  const bool staged = use_table(table, DC_TABLE_SLOTS, stride, point_fmt == DC_F64 ? 32u : 16u, 0, 60 * 1024, &lds_bytes, &lds_rows);
  if (staged) tab = BlockTab{table->blk_ptr, table->blk_ids, table->slot_ptr, table->loc};
  const int fixed_k = g_fwd_generic.load() ? 0 : k;        // a table of [rows, k] has k slots in every block
  if (!staged && !nbr) return table ? DC_ERR_UNSUPPORTED : DC_ERR_ARG;
  QParams qp;
  int rc = make_qparams(point_fmt, dtype, stride, qparams, &qp);
  if (rc) return rc;
  if (n == 0) return (int)hipMemsetAsync(sums_out, 0, 2 * sizeof(double), stream);
  const LossParams lp = make_loss_params(loss_kind, normalization, sqrt_);
  const int64_t rows = xcd_grid(n_blocks(n));
  dim3 grid((unsigned)rows), block(kBlock);
#define FWD_ARGS(T, PT) (const PT*)points, nbr, centre_idx, n, k, mask, (const T*)offset, lp, qp, (PT*)rec, (T*)pointwise, (T*)eigvals, partials_ws
#define FWD_STAGED_ARGS(T, PT) (const PT*)points, tab, lds_rows, centre_idx, n, mask, (const T*)offset, lp, qp, (PT*)rec, (T*)pointwise, (T*)eigvals, partials_ws
#define FWD_FIXED(T, PT, NS) \
  do { \
    if (eigvals) DC_TIMED_LAUNCH((consistency_fwd_fixed_kernel<T, PT, true, NS>), grid, block, lds_bytes, stream, FWD_STAGED_ARGS(T, PT)); \
    else DC_TIMED_LAUNCH((consistency_fwd_fixed_kernel<T, PT, false, NS>), grid, block, lds_bytes, stream, FWD_STAGED_ARGS(T, PT)); \
  } while (0)
#define LAUNCH(T, PT, S) \
  do { \
    if (S == 4 && staged) { /* padded rows + block table: gathers served from LDS */ \
      if (fixed_k == 10) FWD_FIXED(T, PT, 10); \
      else if (fixed_k == 4) FWD_FIXED(T, PT, 4); \
      else if (fixed_k == 8) FWD_FIXED(T, PT, 8); \
      else if (fixed_k == 16) FWD_FIXED(T, PT, 16); \
      else if (eigvals) DC_TIMED_LAUNCH((consistency_fwd_staged_kernel<T, PT, true>), grid, block, lds_bytes, stream, FWD_STAGED_ARGS(T, PT)); \
      else DC_TIMED_LAUNCH((consistency_fwd_staged_kernel<T, PT, false>), grid, block, lds_bytes, stream, FWD_STAGED_ARGS(T, PT)); \
    } else { \
      if (eigvals) DC_TIMED_LAUNCH((consistency_fwd_kernel<T, PT, S, true>), grid, block, 0, stream, FWD_ARGS(T, PT)); \
      else DC_TIMED_LAUNCH((consistency_fwd_kernel<T, PT, S, false>), grid, block, 0, stream, FWD_ARGS(T, PT)); \
    } \
  } while (0)
  { ProfScope prof(1); DC_DISPATCH_FMT(dtype, point_fmt, stride, LAUNCH); }
#undef LAUNCH
#undef FWD_FIXED
  DC_CHECK_LAUNCH();
  if (!reduce) return DC_OK;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(2), dim3(kRedBlock), 0, stream, partials_ws, rows * kWavesPerBlock, sums_out);
  DC_CHECK_LAUNCH();
  return DC_OK;
}

int dc_consistency_fwd(const void* points, int stride, int dtype, int point_fmt, const double* qparams,
                       const int32_t* nbr, const int32_t* centre_idx, const dcBlockTable* table, int64_t n, int k,
                       const uint8_t* mask, const void* offset, int loss_kind, int normalization, int sqrt_, void* rec,
                       void* pointwise, void* eigvals, double* partials_ws, double* sums_out, hipStream_t stream) {
  return consistency_fwd_impl(points, stride, dtype, point_fmt, qparams, nbr, centre_idx, table, n, k, mask, offset, loss_kind, normalization,
                              sqrt_, rec, pointwise, eigvals, partials_ws, sums_out, stream, true);
}

static int consistency_bwd_impl(const void* points, int stride, int dtype, int point_fmt, const double* qparams, const void* rec,
                                const int32_t* csr_ptr, const int32_t* csr_src, const uint8_t* lane_perm,
                                const dcBlockTable* table, int64_t n,
                                const void* vps, const void* dirs, const void* depth, const void* inc, const uint8_t* lmask,
                                const int32_t* scan_id, const double* poses, int n_scans, int model_kind, int n_terms,
                                const double* w, const double* e, int want_exponent_grad, int want_pose_grad,
                                void* grad_points, double* partials_ws, double* grads_out, hipStream_t stream, bool reduce,
                                int64_t rec_rows, const uint16_t* scan_seg = nullptr) {
  if (n < 0 || !points || !rec || (stride != 3 && stride != 4)) return DC_ERR_ARG;
  BlockTab tab{};
  RunTab rtab{};
  size_t lds_bytes = 0;
  int lds_rows = 0;
  // the pose variants hold up to 16 KB of static LDS for the per-scan sums: staged records within 44 KB fit the default 64 KB of a
  // workgroup; longer lists (ball neighbourhoods: 1 500 centres reference a block at r = 0.4 m) take the run kernel with a larger
  // dynamic allocation (hipFuncSetAttribute below) -- one or two blocks per CU, but gathers from LDS: un-staged, the backward of the
  // r = 0.4 m table took 535 us
  const uint32_t rec_row = point_fmt == DC_F64 ? 64u : 32u;
  const bool by_runs = !lane_perm && use_table(table, DC_TABLE_RUNS, stride, rec_row, 1, 128 * 1024, &lds_bytes, &lds_rows);
  const bool staged = by_runs || (!lane_perm && use_table(table, DC_TABLE_SLOTS, stride, rec_row, 1, 44 * 1024, &lds_bytes, &lds_rows));
  if (by_runs) rtab = RunTab{table->blk_ptr, table->blk_ids, table->run_ptr, table->loc};
  else if (staged) tab = BlockTab{table->blk_ptr, table->blk_ids, table->slot_ptr, table->loc};
  if (!staged && (!csr_ptr || !csr_src)) return table ? DC_ERR_UNSUPPORTED : DC_ERR_ARG;
  const bool params = dirs != nullptr;
  if (!params && !grad_points) return DC_ERR_ARG;
  if (params) {
    if (!depth || !partials_ws || !grads_out) return DC_ERR_ARG;
    if (scan_id && (!poses || n_scans < 1)) return DC_ERR_ARG;
    if (want_pose_grad && n_scans < 1) return DC_ERR_ARG;
    int rc = check_model(model_kind, n_terms, inc, w, e);
    if (rc) return rc;
  }
  QParams qp;
  int rc = make_qparams(point_fmt, dtype, stride, qparams, &qp);
  if (rc) return rc;
  if (model_kind == DC_MODEL_NONE) n_terms = 0;
  const int n_acc = 2 * n_terms + 12 * n_scans;
  if (n == 0) return params ? (int)hipMemsetAsync(grads_out, 0, n_acc * sizeof(double), stream) : DC_OK;
  PointInputs in = make_inputs(vps, dirs, depth, inc, lmask, scan_id, poses, n_scans, model_kind, n_terms, w, e);
  // points grouped by scan inside every block (the plan's layout): per-scan pose sums over static lane ranges -- for the
  // kernels whose lanes are the plan's points in order (not with a lane map), and while a block's scans fit the item loop
  const bool grouped = scan_seg && want_pose_grad && !lane_perm && n_scans <= kMaxBlockScans && !reduce && params;   // (the caller's reduction knows the row-major pose rows)
  in.seg_start = grouped ? scan_seg : nullptr;
  const int64_t rows = xcd_grid(n_blocks(n));
  const int64_t prows = rows * kWavesPerBlock;               // partial rows: one per wavefront
  dim3 grid((unsigned)rows), block(kBlock);
  // byte size of the record array for the buffer resource (rows <= n; a compact centre list has fewer)
  const uint64_t rec_total = (uint64_t)(rec_rows > 0 ? rec_rows : n) * (point_fmt == DC_F64 ? 64u : 32u);
  if (rec_total >= (1ull << 32)) return DC_ERR_UNSUPPORTED;
  const uint32_t rec_bytes = (uint32_t)rec_total;
  const int n_red = want_pose_grad ? n_acc : 2 * n_terms;       // slots the kernel produces
  if (params && want_pose_grad && !grouped) {               // (the grouped reduction writes every pose slot of every row itself)
    hipError_t err = hipMemsetAsync(partials_ws + (size_t)prows * 2 * n_terms, 0, (size_t)prows * 12 * n_scans * sizeof(double), stream);
    if (err != hipSuccess) return (int)err;
  }
  if (params && n_red < n_acc && reduce) {
    hipError_t err = hipMemsetAsync(grads_out + n_red, 0, (size_t)(n_acc - n_red) * sizeof(double), stream);
    if (err != hipSuccess) return (int)err;
  }
#define BWD_ARGS(T, PT) (const PT*)points, (const PT*)rec, csr_ptr, csr_src, lane_perm, n, in, qp, (T*)grad_points, partials_ws, n_acc, rec_bytes
#define BWD_STAGED_ARGS(T, PT) (const PT*)points, (const PT*)rec, tab, lds_rows, n, in, qp, (T*)grad_points, partials_ws, n_acc
#define BWD_RUNS_ARGS(T, PT) (const PT*)points, (const PT*)rec, rtab, lds_rows, n, in, qp, (T*)grad_points, partials_ws, n_acc
#define LAUNCH(T, PT, S) \
  do { \
    if (S == 4 && by_runs) { \
      if (want_pose_grad && want_exponent_grad) DC_TIMED_LAUNCH((consistency_bwd_runs_kernel<T, PT, true, true>), grid, block, lds_bytes, stream, BWD_RUNS_ARGS(T, PT)); \
      else if (want_pose_grad) DC_TIMED_LAUNCH((consistency_bwd_runs_kernel<T, PT, false, true>), grid, block, lds_bytes, stream, BWD_RUNS_ARGS(T, PT)); \
      else if (want_exponent_grad) DC_TIMED_LAUNCH((consistency_bwd_runs_kernel<T, PT, true, false>), grid, block, lds_bytes, stream, BWD_RUNS_ARGS(T, PT)); \
      else DC_TIMED_LAUNCH((consistency_bwd_runs_kernel<T, PT, false, false>), grid, block, lds_bytes, stream, BWD_RUNS_ARGS(T, PT)); \
    } else if (S == 4 && staged) { \
      if (want_pose_grad && want_exponent_grad) DC_TIMED_LAUNCH((consistency_bwd_staged_kernel<T, PT, true, true>), grid, block, lds_bytes, stream, BWD_STAGED_ARGS(T, PT)); \
      else if (want_pose_grad) DC_TIMED_LAUNCH((consistency_bwd_staged_kernel<T, PT, false, true>), grid, block, lds_bytes, stream, BWD_STAGED_ARGS(T, PT)); \
      else if (want_exponent_grad) DC_TIMED_LAUNCH((consistency_bwd_staged_kernel<T, PT, true, false>), grid, block, lds_bytes, stream, BWD_STAGED_ARGS(T, PT)); \
      else DC_TIMED_LAUNCH((consistency_bwd_staged_kernel<T, PT, false, false>), grid, block, lds_bytes, stream, BWD_STAGED_ARGS(T, PT)); \
    } else \
    { \
      if (want_pose_grad && want_exponent_grad) DC_TIMED_LAUNCH((consistency_bwd_kernel<T, PT, S, true, true>), grid, block, 0, stream, BWD_ARGS(T, PT)); \
      else if (want_pose_grad) DC_TIMED_LAUNCH((consistency_bwd_kernel<T, PT, S, false, true>), grid, block, 0, stream, BWD_ARGS(T, PT)); \
      else if (want_exponent_grad) DC_TIMED_LAUNCH((consistency_bwd_kernel<T, PT, S, true, false>), grid, block, 0, stream, BWD_ARGS(T, PT)); \
      else DC_TIMED_LAUNCH((consistency_bwd_kernel<T, PT, S, false, false>), grid, block, 0, stream, BWD_ARGS(T, PT)); \
    } \
  } while (0)
  if (by_runs && lds_bytes > 44 * 1024) {
    int attr_rc = DC_OK;
#define BIG_LDS(T, PT, S) \
  do { \
    if (S == 4) { \
      hipError_t e_; \
      if (want_pose_grad && want_exponent_grad) e_ = hipFuncSetAttribute((const void*)consistency_bwd_runs_kernel<T, PT, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
      else if (want_pose_grad) e_ = hipFuncSetAttribute((const void*)consistency_bwd_runs_kernel<T, PT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
      else if (want_exponent_grad) e_ = hipFuncSetAttribute((const void*)consistency_bwd_runs_kernel<T, PT, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
      else e_ = hipFuncSetAttribute((const void*)consistency_bwd_runs_kernel<T, PT, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
      if (e_ != hipSuccess) attr_rc = (int)e_; \
    } \
  } while (0)
    DC_DISPATCH_FMT(dtype, point_fmt, stride, BIG_LDS);
#undef BIG_LDS
    if (attr_rc) return attr_rc;
  }
  { ProfScope prof(2); DC_DISPATCH_FMT(dtype, point_fmt, stride, LAUNCH); }
#undef LAUNCH
  DC_CHECK_LAUNCH();
  if (params && n_red > 0 && reduce) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(n_red), dim3(kRedBlock), 0, stream, partials_ws, prows, grads_out);
    DC_CHECK_LAUNCH();
  }
  return DC_OK;
}

int dc_consistency_bwd(const void* points, int stride, int dtype, int point_fmt, const double* qparams, const void* rec,
                       const int32_t* csr_ptr, const int32_t* csr_src, const uint8_t* lane_perm, const dcBlockTable* table,
                       int64_t n, const void* vps, const void* dirs, const void* depth, const void* inc, const uint8_t* lmask,
                       const int32_t* scan_id, const double* poses, int n_scans, int model_kind, int n_terms, const double* w,
                       const double* e, int want_exponent_grad, int want_pose_grad, void* grad_points, double* partials_ws,
                       double* grads_out, hipStream_t stream) {
  return consistency_bwd_impl(points, stride, dtype, point_fmt, qparams, rec, csr_ptr, csr_src, lane_perm, table, n, vps, dirs, depth,
                              inc, lmask, scan_id, poses, n_scans, model_kind, n_terms, w, e, want_exponent_grad,
                              want_pose_grad, grad_points, partials_ws, grads_out, stream, true, 0);
}

int dc_points_bwd(const void* grad_points, const int32_t* perm, int stride, int dtype, int64_t n, const void* vps,
                  const void* dirs, const void* depth, const void* inc, const uint8_t* lmask, const int32_t* scan_id,
                  const double* poses, int n_scans, int model_kind, int n_terms, const double* w, const double* e,
                  int want_exponent_grad, int want_pose_grad, double* partials_ws, double* grads_out,
                  hipStream_t stream) {
  if (n < 0 || !grad_points || !dirs || !depth || !partials_ws || !grads_out || (stride != 3 && stride != 4))
    return DC_ERR_ARG;
  if (scan_id && (!poses || n_scans < 1)) return DC_ERR_ARG;
  int rc = check_model(model_kind, n_terms, inc, w, e);
  if (rc) return rc;
  if (model_kind == DC_MODEL_NONE) n_terms = 0;
  const int n_acc = 2 * n_terms + 12 * n_scans;
  if (n_acc == 0) return DC_OK;
  if (n == 0) return (int)hipMemsetAsync(grads_out, 0, n_acc * sizeof(double), stream);
  PointInputs in = make_inputs(vps, dirs, depth, inc, lmask, scan_id, poses, n_scans, model_kind, n_terms, w, e);
  const int64_t rows = n_blocks(n);
  const int64_t prows = rows * kWavesPerBlock;
  dim3 grid((unsigned)rows), block(kBlock);
  const int n_red = want_pose_grad ? n_acc : 2 * n_terms;
  if (n_red < n_acc) {
    hipError_t err = hipMemsetAsync(grads_out + n_red, 0, (size_t)(n_acc - n_red) * sizeof(double), stream);
    if (err != hipSuccess) return (int)err;
  }
  if (n_red == 0) return DC_OK;
  if (want_pose_grad) {      // blocks only write the pose slots of the scans they contain
    hipError_t err = hipMemsetAsync(partials_ws + (size_t)prows * 2 * n_terms, 0, (size_t)prows * 12 * n_scans * sizeof(double), stream);
    if (err != hipSuccess) return (int)err;
  }
#define LAUNCH(T, S) \
  hipLaunchKernelGGL((points_bwd_kernel<T, S>), grid, block, 0, stream, (const T*)grad_points, perm, n, in, \
                     want_exponent_grad, want_pose_grad, partials_ws, n_acc)
  if (dtype == DC_F32) { if (stride == 3) LAUNCH(float, 3); else LAUNCH(float, 4); }
  else if (dtype == DC_F64) { if (stride == 3) LAUNCH(double, 3); else LAUNCH(double, 4); }
  else return DC_ERR_DTYPE;
#undef LAUNCH
  DC_CHECK_LAUNCH();
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(n_red), dim3(kRedBlock), 0, stream, partials_ws, prows, grads_out);
  DC_CHECK_LAUNCH();
  return DC_OK;
}

// option 0: 1 = ignore block tables and gather from global memory (ablation / A-B measurements), 0 = default.
#ifdef DC_BLOCK_TRACE
int dc_debug_block_trace(void* buf) {        // diagnostic build only: [4 x grid] uint64 (or null to stop recording)
  unsigned long long* p = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(dc::g_block_trace), &p, sizeof(p));
}
#endif
// The switches exist for A-B measurements and for the tests that hold one path against another; a process that has not asked for them
// (DC_ENABLE_ABLATIONS=1 in its environment when the library is first used) cannot change them: the product has no mutable
// process-wide state.
int dc_set_option(int option, int value) {
  static const bool enabled = [] { const char* e = getenv("DC_ENABLE_ABLATIONS"); return e && atoi(e) != 0; }();
  if (!enabled) return DC_ERR_UNSUPPORTED;
  if (option == 0) { g_no_tab.store(value != 0); return DC_OK; }
  if (option == 1) { g_fwd_generic.store(value); return DC_OK; }
  if (option == 3) { g_no_basis.store(value != 0); return DC_OK; }
  if (option == 4) { g_two_pass.store(value != 0); return DC_OK; }
  if (option == 5) { g_chain_spin.store(value < 0 ? (1 << 22) : value); return DC_OK; }
  if (option == 6) { g_step_var.store(value); return DC_OK; }
  if (option == 7) { g_pose_three_pass.store(value != 0); return DC_OK; }
  if (option == 8) { g_no_reverse.store(value != 0); return DC_OK; }
  return DC_ERR_ARG;
}

// ---- profiler control ---------------------------------------------------------------------------------------
int dc_profiler_enable(int every) {
  std::lock_guard<std::mutex> lock(g_prof.mu);
  g_prof.every = every < 0 ? 0 : every;
  return DC_OK;
}
int dc_profiler_reset(void) {
  std::lock_guard<std::mutex> lock(g_prof.mu);
  for (int k = 0; k < kProfKinds; ++k) { g_prof.count[k] = 0; g_prof.seen[k] = 0; }
  return DC_OK;
}
// Source-level name of the kernel instantiation the last launch of `kind` used, e.g.
// "(consistency_fwd_fixed_kernel<float, q32, false, 10>)"; empty before the first launch.
int dc_profiler_kernel(int kind, char* buf, int len) {
  if (kind < 0 || kind >= kProfKinds || !buf || len < 1) return DC_ERR_ARG;
  const char* nm = g_prof.last_kernel[kind];
  snprintf(buf, (size_t)len, "%s", nm ? nm : "");
  return DC_OK;
}
// kind: 0 points_fwd, 1 consistency_fwd, 2 consistency_bwd, 3 features_fwd.  Waits for the recorded launches to finish.
int dc_profiler_read(int kind, double* total_ms, int64_t* launches) {
  if (kind < 0 || kind >= kProfKinds || !total_ms || !launches) return DC_ERR_ARG;
  double tot = 0.0;
  int n;
  { std::lock_guard<std::mutex> lock(g_prof.mu); n = g_prof.count[kind]; }
  for (int i = 0; i < n; ++i) {
    hipError_t err = hipEventSynchronize(g_prof.stop[kind][i]);
    if (err != hipSuccess) return (int)err;
    float ms = 0.f;
    err = hipEventElapsedTime(&ms, g_prof.start[kind][i], g_prof.stop[kind][i]);
    if (err != hipSuccess) return (int)err;
    tot += ms;
  }
  *total_ms = tot;
  *launches = n;
  return DC_OK;
}

static int make_adam(double* param, double* exp_avg, double* exp_avg_sq, int64_t n, int64_t step, double grad_scale, double lr,
                     double beta1, double beta2, double eps, double weight_decay, AdamArgs* a) {
  if (!param || !exp_avg || !exp_avg_sq || n < 0 || n > 0x7fffffff || step < 1) return DC_ERR_ARG;
  *a = AdamArgs{param, exp_avg, exp_avg_sq, (int)n, grad_scale, lr, beta1, beta2, eps, weight_decay,
                1.0 - pow(beta1, (double)step), sqrt(1.0 - pow(beta2, (double)step))};
  return DC_OK;
}

int dc_adam_step(double* param, const double* grad, double* exp_avg, double* exp_avg_sq, int64_t n, int64_t step,
                 double grad_scale, double lr, double beta1, double beta2, double eps, double weight_decay,
                 hipStream_t stream) {
  AdamArgs a;
  int rc = make_adam(param, exp_avg, exp_avg_sq, n, step, grad_scale, lr, beta1, beta2, eps, weight_decay, &a);
  if (rc || !grad) return rc ? rc : DC_ERR_ARG;
  if (n == 0) return DC_OK;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, grad, a);
  DC_CHECK_LAUNCH();
  return DC_OK;
}

int dc_adam_step_device(double* param, const double* grad, double* exp_avg, double* exp_avg_sq, int64_t n, int64_t* step,
                        double grad_scale, double lr, double beta1, double beta2, double eps, double weight_decay,
                        hipStream_t stream) {
  AdamArgs a;
  int rc = make_adam(param, exp_avg, exp_avg_sq, n, 1, grad_scale, lr, beta1, beta2, eps, weight_decay, &a);
  if (rc || !grad || !step) return rc ? rc : DC_ERR_ARG;
  hipLaunchKernelGGL(adam_device_step_kernel, dim3(1), dim3(256), 0, stream, grad, a, step);
  DC_CHECK_LAUNCH();
  return DC_OK;
}

// One evaluation of a whole sequence (eval.py:85-112 + backward) from a caller-filled descriptor: three kernels
// (+ two fixed-order reductions).  out fp64 [2 + 2 P + 12 S] = {sum loss over mask, mask count, grads of the sum}.
// a chained step (dc_sequence_step_chained): this launch also finishes the previous evaluation (StepChain)
struct ChainCall {
  int32_t* ready;
  int parity, has_prev;
  double* out_prev;
  AdamArgs adam_prev;
  const double* grad_sum;    // nullptr: the previous evaluation's rows are summed by this launch
  bool reduce_now;           // also launch the ordinary reduction of THIS evaluation's rows into `out` (no Adam)
  double* w_prev_out = nullptr;
  uint32_t stamp = 0;        // the launch's number (its step): marks the weights it publishes
  const double* prev_rows = nullptr;   // linked chains: the PREVIOUS launch's partial rows when that was another sequence's (else this
  int64_t prev_count = 0;              // sequence's other buffer), how many there are, and the running sums they are added to
  const double* acc_in = nullptr;
  const int32_t* prev_status = nullptr;
};
// the two partial-row buffers of a chain: behind the columns ordinary evaluations use, so that an evaluation of the same
// sequence between two chained steps (a validation pass, a lazily produced loss cloud) cannot overwrite a pending step
static inline double* chain_buffer(const dcSequenceDesc* d, int n_terms, int parity) {
  const int64_t rows = xcd_grid(n_blocks(d->n)) * kWavesPerBlock;
  const int n_acc = 2 * n_terms + 12 * d->n_scans;
  return d->partials + (int64_t)(2 + n_acc) * rows + (int64_t)parity * (2 + n_terms) * rows;
}

static int sequence_eval_impl(const dcSequenceDesc* d, const double* w, const double* e, const double* poses, int want_grad,
                              int want_exponent_grad, int want_pose_grad, double* out, hipStream_t stream,
                              const AdamArgs& adam, const ChainCall* chain = nullptr, bool one_pass_only = false) {
  if (!d || !out || !poses || !d->partials) return DC_ERR_ARG;
  // the one-pass evaluation first tries the table that lists only what the centres INSIDE the mask gather (dcSequenceDesc.fwd_table_loss)
  if (!one_pass_only && d->fwd_table_loss && d->mask && !d->centre_idx && want_grad && !want_exponent_grad && !want_pose_grad && d->basis &&
      d->model_kind != DC_MODEL_NONE && d->n_terms >= 1 && d->n_terms <= 3 && d->n > 0) {
    dcSequenceDesc dd = *d;
    dd.fwd_table = d->fwd_table_loss;
    dd.fwd_table_loss = nullptr;
    dd.fwd_rows_active = d->fwd_rows_active_loss;
    const int rc = sequence_eval_impl(&dd, w, e, poses, want_grad, want_exponent_grad, want_pose_grad, out, stream, adam, chain, true);
    if (rc != DC_ERR_UNSUPPORTED) return rc;
  }
  const int stride = 4;
  const int n_terms = d->model_kind == DC_MODEL_NONE ? 0 : d->n_terms;
  const int n_acc = 2 * n_terms + 12 * d->n_scans;
  if (d->n == 0) return (int)hipMemsetAsync(out, 0, (size_t)(2 + n_acc) * sizeof(double), stream);
  if (d->partials_count < dc_sequence_partials_count(d->n, n_terms, d->n_scans)) return DC_ERR_WORKSPACE;
  const int64_t n_rows = d->centre_idx ? d->n_centres : d->n;      // forward rows (centres); the backward runs over all points
  if (d->centre_idx && (d->n_centres < 0 || d->n_centres > d->n)) return DC_ERR_ARG;
  const int64_t rows = xcd_grid(n_blocks(d->n)) * kWavesPerBlock;      // partial rows: one per wavefront
  double* p_fwd = d->partials;
  double* p_bwd = d->partials + 2 * rows;
  const int n_red = !want_grad ? 0 : (want_pose_grad ? n_acc : 2 * n_terms);

  // ---- basis form: the basis rows of the current poses and exponents are valid (the caller says so by passing them), only the
  // weights change between evaluations -> no pass over the points, no model / pose arithmetic in the loop
  size_t lds_f = 0, lds_b = 0;
  int rows_f = 0, rows_b = 0;
  const bool q32_pts = d->point_fmt == DC_Q32 && d->dtype == DC_F32, f64_pts = d->point_fmt == DC_F64 && d->dtype == DC_F64;
  // ball neighbourhoods on float32 clouds (a packed table from CSR lists that knows its rows' lengths): the ragged one-pass
  // kernel, whose tile may take up to 128 KB of LDS
  const int rag_pieces = n_terms <= 2 ? 2 : 3;
  const int rag_rows = !d->fwd_table ? 0 : ((d->blk_skip && d->mask && !d->centre_idx && d->fwd_rows_active > 0 && d->fwd_rows_active < d->fwd_table->max_rows)
                                              ? d->fwd_rows_active : d->fwd_table->max_rows);
  const bool ragged_ok = q32_pts && d->fwd_table && d->fwd_table->layout == DC_TABLE_SLOTS && d->fwd_table->packed == 1 && d->fwd_table->row_ptr &&
                         d->fwd_table->own_base && !d->centre_idx && d->fwd_table->max_rows > 0 && !g_no_tab.load() && g_step_var.load() != 8 &&
                         (size_t)rag_rows * rag_pieces * 16 <= 128 * 1024 && rag_rows <= (n_terms <= 2 ? 4096 : 2560);
  const bool basis_fwd = d->basis && (q32_pts || f64_pts) && n_terms > 0 && w &&
                         !want_exponent_grad && !want_pose_grad && !g_no_basis.load() &&
                         (use_table(d->fwd_table, DC_TABLE_SLOTS, stride, q32_pts ? 16u : 32u, 0, 60 * 1024, &lds_f, &rows_f) || (ragged_ok && want_grad && n_terms <= 3 && !g_two_pass.load()));
  // loss and dL/dw in ONE pass (forward-mode) for up to three weights: no record, no backward launch, no transposed table
  size_t lds_s = 0;
  int rows_s = 0;
  const unsigned step_row_bytes = 16u * (unsigned)(q32_pts ? (6 + n_terms + 3) / 4 : 3);        // StepRow<PT, P>::kPieces
  // (the one-pass kernels never stage the blocks they skip: their tile is sized by the longest list among the others)
  dcBlockTable ft_active{};
  if (d->fwd_table) {
    ft_active = *d->fwd_table;
    if (d->blk_skip && d->mask && !d->centre_idx && d->fwd_rows_active > 0 && d->fwd_rows_active < ft_active.max_rows) ft_active.max_rows = d->fwd_rows_active;
  }
  const bool one_pass = basis_fwd && want_grad && n_terms <= 3 && !g_two_pass.load() &&
                        (use_table(d->fwd_table ? &ft_active : nullptr, DC_TABLE_SLOTS, stride, step_row_bytes, 0, 60 * 1024, &lds_s, &rows_s) || ragged_ok);
  // pose gradients (no exponent gradients) of a float32 sequence with a [rows, K] table whose pose tables and local basis rows the
  // plan has built: one launch, no transposed lists (consistency_step_pose_kernel)
  const bool pose_one_pass = want_grad && want_pose_grad && !want_exponent_grad && d->pose_table && d->local_basis && !d->vps && !d->centre_idx &&
                             d->point_fmt == DC_Q32 && d->dtype == DC_F32 && (n_terms == 1 || n_terms == 2) && d->n_scans <= kLdsScans && w &&
                             (d->k == 4 || d->k == 8 || d->k == 10 || d->k == 16) && !g_pose_three_pass.load();
  // every other way to a gradient walks the transposed neighbour lists: the caller provides them on demand
  if (want_grad && !one_pass && !pose_one_pass && (!d->csr_ptr || !d->csr_src)) return DC_ERR_BACKWARD_TABLES;
  const bool basis = basis_fwd &&
                     (!want_grad || one_pass || use_table(d->bwd_table, DC_TABLE_RUNS, stride, q32_pts ? 32u : 64u, 1, 44 * 1024, &lds_b, &rows_b));
  const int fixed_k = g_fwd_generic.load() ? 0 : d->k;
  // a chained step exists for the one-pass kernels only: the caller steps without a chain otherwise
  if ((chain || one_pass_only) && !(basis && one_pass)) return DC_ERR_UNSUPPORTED;
  if (basis) {
    QParams qp;
    int rc = make_qparams(d->point_fmt, d->dtype, stride, d->qparams, &qp);
    if (rc) return rc;
    PointBasis pb{d->basis, w, n_terms, q32_pts ? qp.inv_scale : 1.0};
    const LossParams lp = make_loss_params(d->loss_kind & ~DC_LOSS_RAW_POINTWISE, d->normalization, d->sqrt_);
    BlockTab tab{d->fwd_table->blk_ptr, d->fwd_table->blk_ids, d->fwd_table->slot_ptr, d->fwd_table->loc};
    const dim3 block(kBlock);
    if (one_pass) {
      const int64_t g_blocks = xcd_grid(n_blocks(n_rows));
      StepChain ch{};
      ch.blk_skip = (d->mask && !d->centre_idx) ? d->blk_skip : nullptr;
      if (chain) {
        p_fwd = chain_buffer(d, n_terms, chain->parity);
        p_bwd = p_fwd + 2 * g_blocks;
        ch.ready = chain->ready; ch.parity = chain->parity; ch.has_prev = chain->has_prev; ch.n_front = kChainFront;
        ch.n_out = 2 + n_acc; ch.prev = chain_buffer(d, n_terms, chain->parity ^ 1); ch.prev_rows = g_blocks;
        if (chain->prev_rows) { ch.prev = chain->prev_rows; ch.prev_rows = chain->prev_count; }
        ch.acc_in = chain->acc_in; ch.prev_status = chain->prev_status;
        ch.out_prev = chain->out_prev; ch.w_prev_out = chain->w_prev_out; ch.status = d->status; ch.spin_limit = g_chain_spin.load(); ch.adam = chain->adam_prev;
        ch.grad_sum = chain->grad_sum;
        ch.stamp = chain->stamp; ch.w_now = w;
        ch.reverse = (chain->parity && !g_no_reverse.load()) ? 1 : 0;
      }
      const dim3 grid((unsigned)(g_blocks + (chain ? kChainFront : 0)));
      const int var = g_step_var.load();
      // float32 clouds with a [rows, K] table: the kernel with fp64 row differences in LDS (48-B rows + its scratch, all dynamic)
      const bool q32_step = q32_pts && var == 1 && (fixed_k == 10 || fixed_k == 4 || fixed_k == 8 || fixed_k == 16) && rows_s <= 768;
      const bool ragged_step = ragged_ok;
      if (ragged_step) {
        ProfScope prof(1);
        const int mr = rag_rows;
        int rc = DC_OK;
#define RAGGED(P, CAP) (prof.name("(consistency_step_ragged_q32_kernel<" #P ", " #CAP ">)"), \
                        ragged_launch<P, CAP>(grid, stream, prof.start(), prof.stop(), pb, tab, d->fwd_table, n_rows, d->mask, lp, qp, p_fwd, p_bwd, ch))
#define RAGGED_CAPS(P) (mr <= 1024 ? RAGGED(P, 1024) : mr <= 1280 ? RAGGED(P, 1280) : mr <= 1600 ? RAGGED(P, 1600) : mr <= 2048 ? RAGGED(P, 2048) : \
                        mr <= 2560 ? RAGGED(P, 2560) : RAGGED(P, 4096))
        if (n_terms == 2) rc = RAGGED_CAPS(2);
        else if (n_terms == 1) rc = RAGGED_CAPS(1);
        else rc = mr <= 1024 ? RAGGED(3, 1024) : mr <= 1600 ? RAGGED(3, 1600) : RAGGED(3, 2560);
#undef RAGGED_CAPS
#undef RAGGED
        if (rc) return rc;
      } else if (q32_step) {
        ProfScope prof(1);
        static_assert(kStepQ32Cap == 512, "the profiler names the instantiation by its literal arguments");
#define STEPQ_LAUNCH(NS, P, CAP) DC_TIMED_LAUNCH((consistency_step_q32_kernel<NS, P, CAP>), grid, block, 0, stream, pb, tab, d->fwd_table->own_base, \
                                                 d->centre_idx, n_rows, d->mask, lp, qp, p_fwd, p_bwd, ch)
#define STEPQ_P(NS, CAP) do { if (n_terms == 2) STEPQ_LAUNCH(NS, 2, CAP); else if (n_terms == 1) STEPQ_LAUNCH(NS, 1, CAP); else STEPQ_LAUNCH(NS, 3, CAP); } while (0)
#define STEPQ_K(CAP) do { if (fixed_k == 10) STEPQ_P(10, CAP); else if (fixed_k == 4) STEPQ_P(4, CAP); else if (fixed_k == 8) STEPQ_P(8, CAP); else STEPQ_P(16, CAP); } while (0)
        if (rows_s <= 512) STEPQ_K(512); else STEPQ_K(768);
#undef STEPQ_K
#undef STEPQ_P
#undef STEPQ_LAUNCH
      } else {
        ProfScope prof(1);
#define STEP_LAUNCH(K) DC_TIMED_LAUNCH(K, grid, block, lds_s, stream, pb, tab, d->fwd_table->own_base, rows_s, d->centre_idx, n_rows, \
                                       d->mask, lp, qp, p_fwd, p_bwd, ch)
#define STEP_NS(PT, P) do { if (fixed_k == 10 && P == 2 && var == 0) STEP_LAUNCH((consistency_step_basis_kernel<PT, 10, 2, 0>)); \
                            else if (fixed_k == 10) STEP_LAUNCH((consistency_step_basis_kernel<PT, 10, P, kStepVar>)); \
                            else if (fixed_k == 4) STEP_LAUNCH((consistency_step_basis_kernel<PT, 4, P, kStepVar>)); \
                            else if (fixed_k == 8) STEP_LAUNCH((consistency_step_basis_kernel<PT, 8, P, kStepVar>)); \
                            else if (fixed_k == 16) STEP_LAUNCH((consistency_step_basis_kernel<PT, 16, P, kStepVar>)); \
                            else DC_TIMED_LAUNCH((consistency_step_basis_slots_kernel<PT, P>), grid, block, lds_s, stream, pb, tab, d->fwd_table->own_base, rows_s, \
                                                 d->centre_idx, n_rows, d->mask, lp, qp, p_fwd, p_bwd, ch, d->fwd_table->packed); } while (0)
#define STEP(PT) do { if (n_terms == 2) STEP_NS(PT, 2); else if (n_terms == 1) STEP_NS(PT, 1); else STEP_NS(PT, 3); } while (0)
        if (q32_pts) STEP(q32); else STEP(double);
#undef STEP
#undef STEP_NS
#undef STEP_LAUNCH
      }
      DC_CHECK_LAUNCH();
      if (chain && chain->reduce_now) {        // several ranks: this evaluation's sums now (they go through an all-reduce next)
        hipLaunchKernelGGL(reduce_eval_kernel, dim3(2 + n_terms), dim3(kRedBlock), 0, stream, p_fwd, p_bwd, g_blocks, g_blocks, n_terms,
                           2 + n_acc, out, AdamArgs{}, (const int32_t*)d->status);
        DC_CHECK_LAUNCH();
      }
      if (chain) return DC_OK;                 // its sums are taken by the next launch of the chain, or by the flush
      const int64_t rows_g = (int64_t)grid.x * kWavesPerBlock;
      hipLaunchKernelGGL(reduce_eval_kernel, dim3(2 + n_terms), dim3(kRedBlock), 0, stream, p_fwd, p_bwd, rows_g, rows_g, n_terms, 2 + n_acc, out,
                         adam, (const int32_t*)d->status);
      DC_CHECK_LAUNCH();
      return DC_OK;
    }
    {
      ProfScope prof(1);
      const dim3 grid((unsigned)xcd_grid(n_blocks(n_rows)));
#define FWD_BASIS_P(PT, T, NS, P) DC_TIMED_LAUNCH((consistency_fwd_basis_kernel<PT, false, NS, P>), grid, block, lds_f, stream, pb, tab, \
                                                 d->fwd_table->own_base, rows_f, d->centre_idx, n_rows, d->mask, (const T*)nullptr, lp, qp, \
                                                 (PT*)(want_grad ? d->rec : nullptr), (T*)nullptr, (T*)nullptr, p_fwd)
#define FWD_BASIS_SLOTS(PT, T, P) DC_TIMED_LAUNCH((consistency_fwd_basis_slots_kernel<PT, false, P>), grid, block, lds_f, stream, pb, tab, \
                                                 d->fwd_table->own_base, rows_f, d->centre_idx, n_rows, d->mask, (const T*)nullptr, lp, qp, \
                                                 (PT*)(want_grad ? d->rec : nullptr), (T*)nullptr, (T*)nullptr, p_fwd)
#define FWD_BASIS_NS(PT, T, P) do { if (fixed_k == 10) FWD_BASIS_P(PT, T, 10, P); else if (fixed_k == 4) FWD_BASIS_P(PT, T, 4, P); \
                                    else if (fixed_k == 8) FWD_BASIS_P(PT, T, 8, P); else if (fixed_k == 16) FWD_BASIS_P(PT, T, 16, P); \
                                    else FWD_BASIS_SLOTS(PT, T, P); } while (0)
#define FWD_BASIS(PT, T) do { if (n_terms == 2) FWD_BASIS_NS(PT, T, 2); else if (n_terms == 1) FWD_BASIS_NS(PT, T, 1); \
                              else if (n_terms == 3) FWD_BASIS_NS(PT, T, 3); else FWD_BASIS_NS(PT, T, 0); } while (0)
      if (q32_pts) FWD_BASIS(q32, float); else FWD_BASIS(double, double);
#undef FWD_BASIS
#undef FWD_BASIS_NS
#undef FWD_BASIS_SLOTS
#undef FWD_BASIS_P
    }
    DC_CHECK_LAUNCH();
    if (want_grad) {
      RunTab rtab{d->bwd_table->blk_ptr, d->bwd_table->blk_ids, d->bwd_table->run_ptr, d->bwd_table->loc};
      ProfScope prof(2);
#define BWD_BASIS_P(PT, P) DC_TIMED_LAUNCH((consistency_bwd_basis_kernel<PT, P>), dim3((unsigned)xcd_grid(n_blocks(d->n))), block, lds_b, stream, pb, \
                                          (const PT*)d->rec, rtab, rows_b, d->n, qp, p_bwd)
#define BWD_BASIS(PT) do { if (n_terms == 2) BWD_BASIS_P(PT, 2); else if (n_terms == 1) BWD_BASIS_P(PT, 1); \
                           else if (n_terms == 3) BWD_BASIS_P(PT, 3); else BWD_BASIS_P(PT, 0); } while (0)
      if (q32_pts) BWD_BASIS(q32); else BWD_BASIS(double);
#undef BWD_BASIS
#undef BWD_BASIS_P
      DC_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(reduce_eval_kernel, dim3(2 + n_red), dim3(kRedBlock), 0, stream, p_fwd, p_bwd,
                       xcd_grid(n_blocks(n_rows)) * kWavesPerBlock, rows, n_red, 2 + n_acc, out, adam, (const int32_t*)d->status);
    DC_CHECK_LAUNCH();
    return DC_OK;
  }

  // pose gradients (no exponent gradients) of a float32 sequence with a [rows, K] table: everything in ONE launch
  // (consistency_step_pose_kernel) when the plan has built the pose tables and the local basis rows for these exponents
  if (pose_one_pass) {
    QParams qp;
    int rc = make_qparams(d->point_fmt, d->dtype, stride, d->qparams, &qp, d->status);
    if (rc) return rc;
    const dcPoseTable* pt = d->pose_table;
    PoseTab tab{pt->blk_ptr, pt->ids, pt->loc, pt->own_pos, pt->row_seg, pt->row_scan};
    const LossParams lp = make_loss_params(d->loss_kind & ~DC_LOSS_RAW_POINTWISE, d->normalization, d->sqrt_);
    const dim3 grid((unsigned)xcd_grid(n_blocks(d->n))), block(kBlock);
    {
      ProfScope prof(1);
#define POSE_NS(NS, P) DC_TIMED_LAUNCH((consistency_step_pose_kernel<NS, P>), grid, block, 0, stream, (const int32_t*)d->local_basis, tab, poses, d->n_scans, \
                                       w, d->n, d->mask, lp, qp, p_fwd, p_bwd)
#define POSE_P(NS) do { if (n_terms == 2) POSE_NS(NS, 2); else POSE_NS(NS, 1); } while (0)
      if (d->k == 10) POSE_P(10); else if (d->k == 4) POSE_P(4); else if (d->k == 8) POSE_P(8); else POSE_P(16);
#undef POSE_P
#undef POSE_NS
    }
    DC_CHECK_LAUNCH();
    const int64_t rows_b = xcd_grid(n_blocks(d->n));         // one row per block, every column contiguous
    hipLaunchKernelGGL(reduce_eval_kernel, dim3(2 + n_acc), dim3(kRedBlock), 0, stream, p_fwd, p_bwd, rows_b, rows_b, n_acc, 2 + n_acc, out, adam,
                       (const int32_t*)d->status, -1, 0);
    DC_CHECK_LAUNCH();
    return DC_OK;
  }
  // (measured in round 4 and dropped: a forward that FORMS the rows it stages from the raw inputs -- model, pose, ray end point per
  // staged row, the owning block writing x for the backward -- instead of dc_points_fwd + a forward over x.  It removes a 22 us launch
  // and 48 B per point of traffic, but a block stages 1.47 rows per point and each costs five scattered loads and ~80 fp64
  // instructions in front of the staging barrier: 106 us against 22 + 46.)
  int rc = dc_points_fwd(d->vps, d->dirs, d->depth, d->inc, d->lmask, d->scan_id, poses, d->n_scans, d->model_kind,
                         d->n_terms, w, e, d->n, d->dtype, d->point_fmt, d->qparams, stride, d->x, nullptr, nullptr,
                         nullptr, d->status, stream);
  if (rc) return rc;
  rc = consistency_fwd_impl(d->x, stride, d->dtype, d->point_fmt, d->qparams, d->nbr, d->centre_idx, d->fwd_table, n_rows, d->k, d->mask,
                            nullptr, d->loss_kind, d->normalization, d->sqrt_, want_grad ? d->rec : nullptr, nullptr, nullptr, p_fwd, out, stream,
                            false);
  if (!rc && want_grad)
    rc = consistency_bwd_impl(d->x, stride, d->dtype, d->point_fmt, d->qparams, d->rec, d->csr_ptr, d->csr_src, d->lane_perm,
                              d->bwd_table, d->n, d->vps, d->dirs, d->depth, d->inc, d->lmask, d->scan_id, poses, d->n_scans, d->model_kind,
                              d->n_terms, w, e, want_exponent_grad, want_pose_grad, nullptr, p_bwd, out + 2, stream, false,
                              n_rows, d->scan_seg);
  if (rc) return rc;
  // (as consistency_bwd_impl decides: the grouped backward leaves the pose slots row-major, one row per block)
  const bool grouped = want_grad && want_pose_grad && d->scan_seg && !d->lane_perm && d->n_scans <= kMaxBlockScans;
  hipLaunchKernelGGL(reduce_eval_kernel, dim3(red_grid(n_red, grouped ? 2 * n_terms : -1, 12 * d->n_scans)), dim3(kRedBlock), 0, stream, p_fwd, p_bwd,
                     xcd_grid(n_blocks(n_rows)) * kWavesPerBlock, rows, n_red, 2 + n_acc, out, adam, (const int32_t*)d->status,
                     grouped ? 2 * n_terms : -1, 12 * d->n_scans);
  DC_CHECK_LAUNCH();
  return DC_OK;
}

int dc_consistency_gate(const void* raw_pointwise, int dtype, int point_fmt, const uint8_t* mask, int64_t n,
                        const double* threshold, int sqrt_, void* rec, double* partials_ws, double* sums_out,
                        hipStream_t stream) {
  if (n < 0 || !partials_ws || !sums_out || !threshold) return DC_ERR_ARG;
  if (n == 0) return (int)hipMemsetAsync(sums_out, 0, 2 * sizeof(double), stream);
  if (!raw_pointwise || !rec) return DC_ERR_ARG;
  const int64_t rows = xcd_grid(n_blocks(n));
  const dim3 grid((unsigned)rows), block(kBlock);
  if (point_fmt == DC_Q32 && dtype == DC_F32)
    hipLaunchKernelGGL((consistency_gate_kernel<float, q32>), grid, block, 0, stream, (const float*)raw_pointwise, mask, n, threshold, sqrt_, (q32*)rec, partials_ws);
  else if (point_fmt == DC_F32 && dtype == DC_F32)
    hipLaunchKernelGGL((consistency_gate_kernel<float, float>), grid, block, 0, stream, (const float*)raw_pointwise, mask, n, threshold, sqrt_, (float*)rec, partials_ws);
  else if (point_fmt == DC_F64 && dtype == DC_F64)
    hipLaunchKernelGGL((consistency_gate_kernel<double, double>), grid, block, 0, stream, (const double*)raw_pointwise, mask, n, threshold, sqrt_, (double*)rec, partials_ws);
  else
    return DC_ERR_DTYPE;
  DC_CHECK_LAUNCH();
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(2), dim3(kRedBlock), 0, stream, partials_ws, rows * kWavesPerBlock, sums_out);
  DC_CHECK_LAUNCH();
  return DC_OK;
}

int dc_sequence_eval(const dcSequenceDesc* d, const double* w, const double* e, const double* poses, int want_grad,
                     int want_exponent_grad, int want_pose_grad, double* out, hipStream_t stream) {
  return sequence_eval_impl(d, w, e, poses, want_grad, want_exponent_grad, want_pose_grad, out, stream, AdamArgs{});
}

// Evaluation + optimiser step of the model weights in one host call (train.py:220-312 for a single sequence on this
// rank): the block of the final reduction that finishes dL/dw_i also applies torch.optim.Adam's update to w_i.
int dc_sequence_step(const dcSequenceDesc* d, double* w, const double* e, const double* poses, double* exp_avg,
                     double* exp_avg_sq, int64_t step, double grad_scale, double lr, double beta1, double beta2, double eps,
                     double weight_decay, double* out, hipStream_t stream) {
  if (!d || d->model_kind == DC_MODEL_NONE || d->n_terms < 1 || d->n == 0) return DC_ERR_ARG;
  AdamArgs a;
  int rc = make_adam(w, exp_avg, exp_avg_sq, d->n_terms, step, grad_scale, lr, beta1, beta2, eps, weight_decay, &a);
  if (rc) return rc;
  return sequence_eval_impl(d, w, e, poses, 1, 0, 0, out, stream, a);
}

int dc_sequence_step_chained(const dcSequenceDesc* d, double* w, const double* e, const double* poses, double* exp_avg,
                             double* exp_avg_sq, int64_t step, int has_prev, double grad_scale, double lr, double beta1, double beta2,
                             double eps, double weight_decay, int32_t* ready, double* out_prev, hipStream_t stream) {
  if (!d || d->model_kind == DC_MODEL_NONE || d->n_terms < 1 || d->n == 0 || !ready || !out_prev || step < 1) return DC_ERR_ARG;
  if (has_prev && step < 2) return DC_ERR_ARG;
  ChainCall c{ready, (int)(step & 1), has_prev ? 1 : 0, out_prev, AdamArgs{}, nullptr, false};
  c.stamp = (uint32_t)step;
  if (has_prev) {
    int rc = make_adam(w, exp_avg, exp_avg_sq, d->n_terms, step - 1, grad_scale, lr, beta1, beta2, eps, weight_decay, &c.adam_prev);
    if (rc) return rc;
  }
  return sequence_eval_impl(d, w, e, poses, 1, 0, 0, out_prev, stream, AdamArgs{}, &c);
}

int dc_sequence_step_chained_rec(const dcSequenceDesc* d, double* w, const double* e, const double* poses, double* exp_avg,
                                 double* exp_avg_sq, int64_t step, int has_prev, double grad_scale, double lr, double beta1, double beta2,
                                 double eps, double weight_decay, int32_t* ready, double* out_prev, double* w_used_prev,
                                 hipStream_t stream) {
  if (!d || d->model_kind == DC_MODEL_NONE || d->n_terms < 1 || d->n == 0 || !ready || !out_prev || step < 1) return DC_ERR_ARG;
  if (has_prev && step < 2) return DC_ERR_ARG;
  ChainCall c{ready, (int)(step & 1), has_prev ? 1 : 0, out_prev, AdamArgs{}, nullptr, false, has_prev ? w_used_prev : nullptr};
  c.stamp = (uint32_t)step;
  if (has_prev) {
    int rc = make_adam(w, exp_avg, exp_avg_sq, d->n_terms, step - 1, grad_scale, lr, beta1, beta2, eps, weight_decay, &c.adam_prev);
    if (rc) return rc;
  }
  return sequence_eval_impl(d, w, e, poses, 1, 0, 0, out_prev, stream, AdamArgs{}, &c);
}

// ---- a chain over the SEVERAL sequences of one loss (train.py:172-175; eval.py:85-112 pools their sums) ---------------------------
// Launch (step, i) evaluates sequence i and first finishes the launch before it -- sequence i - 1 of this step, or the last sequence
// of the previous step -- by summing ITS rows (d_prev, parity prev_parity) onto the running sums of the step (acc_in, NULL for the
// step's second launch).  finish: 0 nothing is pending (the very first launch), 1 sum into out_prev (the running sums; may be
// acc_in itself), 2 the sums complete a step: Adam update `step - 1` on w, out_prev <- the step's totals, w_used_prev.
static int linked_call(const dcSequenceDesc* d, const dcSequenceDesc* d_prev, int prev_parity, int finish, const double* acc_in, int n_terms,
                       ChainCall* c) {
  if (finish == 0) return DC_OK;
  if (!d_prev || !d_prev->partials || d_prev->n_terms != n_terms || d_prev->n < 1) return DC_ERR_ARG;
  if (d_prev->partials_count < dc_sequence_partials_count(d_prev->n, n_terms, d_prev->n_scans)) return DC_ERR_WORKSPACE;
  const int64_t n_rows = d_prev->centre_idx ? d_prev->n_centres : d_prev->n;
  c->prev_rows = chain_buffer(d_prev, n_terms, prev_parity & 1);
  c->prev_count = xcd_grid(n_blocks(n_rows));
  c->acc_in = acc_in;
  c->prev_status = d_prev->status;
  (void)d;
  return DC_OK;
}

int dc_sequence_step_linked(const dcSequenceDesc* d, const dcSequenceDesc* d_prev, int prev_parity, int finish, const double* acc_in,
                            double* w, const double* e, const double* poses, double* exp_avg, double* exp_avg_sq, int64_t step,
                            int64_t stamp, double grad_scale, double lr, double beta1, double beta2, double eps, double weight_decay,
                            int32_t* ready, double* out_prev, double* w_used_prev, hipStream_t stream) {
  if (!d || d->model_kind == DC_MODEL_NONE || d->n_terms < 1 || d->n == 0 || !ready || !out_prev || step < 1 || stamp < 1) return DC_ERR_ARG;
  if (finish < 0 || finish > 2 || (finish == 2 && step < 2)) return DC_ERR_ARG;
  ChainCall c{ready, (int)(step & 1), finish ? 1 : 0, out_prev, AdamArgs{}, nullptr, false, finish == 2 ? w_used_prev : nullptr};
  c.stamp = (uint32_t)stamp;
  int rc = linked_call(d, d_prev, prev_parity, finish, acc_in, d->n_terms, &c);
  if (rc) return rc;
  if (finish == 2) {
    rc = make_adam(w, exp_avg, exp_avg_sq, d->n_terms, step - 1, grad_scale, lr, beta1, beta2, eps, weight_decay, &c.adam_prev);
    if (rc) return rc;
  }
  return sequence_eval_impl(d, w, e, poses, 1, 0, 0, out_prev, stream, AdamArgs{}, &c);
}

// the last launch of a linked chain finished on its own: d_prev's rows (+ acc_in) -> out, Adam update `step` (one small launch)
int dc_sequence_chain_flush_linked(const dcSequenceDesc* d_prev, int prev_parity, const double* acc_in, double* w, double* exp_avg,
                                   double* exp_avg_sq, int64_t step, int64_t stamp, double grad_scale, double lr, double beta1, double beta2,
                                   double eps, double weight_decay, int32_t* ready, double* out, hipStream_t stream) {
  if (!d_prev || d_prev->model_kind == DC_MODEL_NONE || d_prev->n_terms < 1 || d_prev->n_terms > 3 || !ready || !out || step < 1) return DC_ERR_ARG;
  ChainCall c{ready, 0, 1, out, AdamArgs{}, nullptr, false, nullptr};
  int rc = linked_call(d_prev, d_prev, prev_parity, 2, acc_in, d_prev->n_terms, &c);
  if (rc) return rc;
  rc = make_adam(w, exp_avg, exp_avg_sq, d_prev->n_terms, step, grad_scale, lr, beta1, beta2, eps, weight_decay, &c.adam_prev);
  if (rc) return rc;
  StepChain ch{};
  ch.ready = ready; ch.stamp = (uint32_t)stamp; ch.parity = 0; ch.has_prev = 1; ch.n_front = kChainFront;
  ch.n_out = 2 + 2 * d_prev->n_terms + 12 * d_prev->n_scans;
  ch.prev = c.prev_rows; ch.prev_rows = c.prev_count; ch.acc_in = acc_in; ch.out_prev = out; ch.w_prev_out = nullptr; ch.status = d_prev->status;
  ch.spin_limit = 0; ch.adam = c.adam_prev; ch.grad_sum = nullptr; ch.w_now = w; ch.prev_status = d_prev->status;
  const int P = d_prev->n_terms;
  if (P == 1) hipLaunchKernelGGL(chain_front_only_kernel<1>, dim3(kChainFront), dim3(kBlock), 0, stream, ch);
  else if (P == 2) hipLaunchKernelGGL(chain_front_only_kernel<2>, dim3(kChainFront), dim3(kBlock), 0, stream, ch);
  else hipLaunchKernelGGL(chain_front_only_kernel<3>, dim3(kChainFront), dim3(kBlock), 0, stream, ch);
  DC_CHECK_LAUNCH();
  return DC_OK;
}

int dc_sequence_chain_flush(const dcSequenceDesc* d, double* w, double* exp_avg, double* exp_avg_sq, int64_t step, double grad_scale,
                            double lr, double beta1, double beta2, double eps, double weight_decay, double* out, hipStream_t stream) {
  if (!d || d->model_kind == DC_MODEL_NONE || d->n_terms < 1 || d->n == 0 || !out || step < 1) return DC_ERR_ARG;
  AdamArgs a;
  int rc = make_adam(w, exp_avg, exp_avg_sq, d->n_terms, step, grad_scale, lr, beta1, beta2, eps, weight_decay, &a);
  if (rc) return rc;
  const int n_terms = d->n_terms, n_acc = 2 * n_terms + 12 * d->n_scans;
  if (!d->partials || d->partials_count < dc_sequence_partials_count(d->n, n_terms, d->n_scans)) return DC_ERR_WORKSPACE;
  const int64_t n_rows = d->centre_idx ? d->n_centres : d->n;
  const int64_t g_blocks = xcd_grid(n_blocks(n_rows));
  const double* buf = chain_buffer(d, n_terms, (int)(step & 1));
  hipLaunchKernelGGL(reduce_eval_kernel, dim3(2 + n_terms), dim3(kRedBlock), 0, stream, buf, buf + 2 * g_blocks, g_blocks, g_blocks,
                     n_terms, 2 + n_acc, out, a, (const int32_t*)d->status);
  DC_CHECK_LAUNCH();
  return DC_OK;
}

int dc_sequence_eval_after_update(const dcSequenceDesc* d, double* w, const double* e, const double* poses, double* exp_avg,
                                  double* exp_avg_sq, int64_t step, const double* grad_sum, double grad_scale, double lr, double beta1,
                                  double beta2, double eps, double weight_decay, int32_t* ready, double* out, double* w_used,
                                  hipStream_t stream) {
  if (!d || d->model_kind == DC_MODEL_NONE || d->n_terms < 1 || d->n == 0 || !ready || !out || step < 1) return DC_ERR_ARG;
  if (grad_sum && step < 2) return DC_ERR_ARG;
  ChainCall c{ready, (int)(step & 1), grad_sum ? 1 : 0, out, AdamArgs{}, grad_sum ? grad_sum : w, true, w_used};
  c.stamp = (uint32_t)step;
  if (grad_sum) {
    int rc = make_adam(w, exp_avg, exp_avg_sq, d->n_terms, step - 1, grad_scale, lr, beta1, beta2, eps, weight_decay, &c.adam_prev);
    if (rc) return rc;
  }
  return sequence_eval_impl(d, w, e, poses, 1, 0, 0, out, stream, AdamArgs{}, &c);
}

}  // extern "C"
