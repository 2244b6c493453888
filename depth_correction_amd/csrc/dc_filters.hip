// Mask / filter kernels of the map-consistency set-up phase (gfx950).  Reference: filters.py:85-113
// (within_bounds), :184-193 (valid neighbours), :196-254 (eigenvalue / ratio bounds),
// depth_cloud.py:314-326 (dir / vp dispersion), preproc.py:122-164 (global_cloud_mask).
// HBM-bound elementwise passes; one lane per point, coalesced.
#include <cstring>
#include <cstdlib>
#include "dc_common.h"
#include "../../include/dc_hip.h"
#include "dc_device.h"
#include <cstring>
#include "dc_hostutil.h"
#include "dc_sort.h"

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


// Scan-shadow filter (filters.py:257-309, the laser_filters ScanShadowsFilter idea): a point is kept when the angles
// between the ray back to its viewpoint (o - x) and the vectors to its direction-neighbours (x_j - x) all lie in
// [lo, hi].  One lane per point walks its neighbour row; nothing of the reference's [N, K, 3] tensors is materialised.
// Arithmetic in the cloud's dtype and in torch's operation order (cosine_similarity normalises each vector by
// max(|v|, 1e-8) first, then sums the three products; acos of a value an ulp outside [-1, 1] is NaN and, like the
// reference's amin / amax over a row holding a NaN, removes the point).  Missing neighbours (-1) count as `fill`
// (the reference overwrites them with the mean of the bounds).
template <typename T>
__global__ __launch_bounds__(kBlock) void shadow_mask_kernel(const T* __restrict__ x, const T* __restrict__ vps, int vps_rows,
                                                             const int32_t* __restrict__ dnbr, int64_t n, int k, T lo, T hi,
                                                             T fill, uint8_t* __restrict__ mask) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const T eps = (T)1e-8;
  const T xi0 = x[i * 3], xi1 = x[i * 3 + 1], xi2 = x[i * 3 + 2];
  const T* o = vps + (vps_rows == 1 ? 0 : i * 3);
  T a0 = o[0] - xi0, a1 = o[1] - xi1, a2 = o[2] - xi2;
  const T na = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
  const T da = na > eps ? na : eps;
  a0 /= da; a1 /= da; a2 /= da;
  T amin = (T)INFINITY, amax = -(T)INFINITY;
  bool bad = false;
  for (int q = 0; q < k; ++q) {
    const int32_t j = dnbr[i * k + q];
    T ang = fill;
    if (j >= 0) {
      T b0 = x[(int64_t)j * 3] - xi0, b1 = x[(int64_t)j * 3 + 1] - xi1, b2 = x[(int64_t)j * 3 + 2] - xi2;
      const T nb = sqrt(b0 * b0 + b1 * b1 + b2 * b2);
      const T db = nb > eps ? nb : eps;
      b0 /= db; b1 /= db; b2 /= db;
      const T c = a0 * b0 + a1 * b1 + a2 * b2;
      ang = acos(c);
    }
    bad = bad || (ang != ang);
    amin = ang < amin ? ang : amin;
    amax = ang > amax ? ang : amax;
  }
  mask[i] = (!bad && amin >= lo && amax <= hi) ? 1 : 0;
}

// ---- the model applied to a cloud outside the training loop (model.py:181-215, 250-274; the node: scripts/depth_correction:52)
// depth'[i] = mask[i] ? f(depth[i], bias(gamma[i])) : depth[i],  bias = sum_k w_k gamma^e_k, in torch's operation order:
// pow in fp64 (the exponents are fp64 [1,P], so torch.pow promotes), the [n,P] x [P,1] product accumulated term by term,
// the correction in fp64, rounded to the cloud's precision at the end.  The reference does this with a boolean gather, pow,
// a GEMM, two elementwise passes and an index_put (0.4 ms for a 200 k-point scan here, mostly launches); no autograd here.
// op: 0 d - b | 1 d + b | 2 d (1 - b) | 3 d / (1 - b)
template <typename T>
__global__ __launch_bounds__(kBlock) void correct_depth_kernel(const T* __restrict__ depth, const T* __restrict__ gamma,
                                                               const uint8_t* __restrict__ mask, const double* __restrict__ w,
                                                               const double* __restrict__ ex, int n_terms, int op, int64_t n,
                                                               T* __restrict__ out) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const T d = depth[i];
  if (mask && !mask[i]) { out[i] = d; return; }
  const double g = (double)gamma[i];
  double b = 0.0;
  for (int k = 0; k < n_terms; ++k) b = fma(pow(g, ex[k]), w[k], b);
  const double dd = (double)d;
  double r;
  if (op == 0) r = dd - b;
  else if (op == 1) r = dd + b;
  else if (op == 2) r = dd * (1.0 - b);
  else r = dd / (1.0 - b);
  out[i] = (T)r;
}

// ---- voxel-grid filter: one survivor per voxel, the reference's dict semantics (filters.py:24-82) --------------------
// The reference feeds points to a dict {voxel -> index} in a processing sequence (identity, reversed, or a seeded
// shuffle): the LAST point of the sequence falling into a voxel survives, and voxels are listed in order of FIRST
// appearance.  On the GPU: voxel keys in sequence order -> stable radix sort (key, t) -> runs; the head of a run gives
// the first appearance, its tail the survivor; runs are finally ordered by first appearance (or survivors by index).
template <typename T>
__global__ __launch_bounds__(kBlock) void voxel_coords_kernel(const T* __restrict__ xyz, int stride, int64_t n, double res,
                                                              int32_t* __restrict__ vox, int32_t* __restrict__ vmin,
                                                              int32_t* __restrict__ vmax, int32_t* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const T r = (T)res;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const T f = floor(xyz[i * stride + a] / r);              // in the cloud's precision, like numpy does
    int32_t v = 0;
    if (f >= (T)-1073741824.0 && f <= (T)1073741824.0) v = (int32_t)f; else atomicOr(bad, 1);
    vox[i * 3 + a] = v;
    atomicMin(vmin + a, v);
    atomicMax(vmax + a, v);
  }
}

__global__ void voxel_box_init_kernel(int32_t* __restrict__ box) {
  const int t = threadIdx.x;                 // box[0..3) running minima, box[4..7) running maxima
  if (t < 8) box[t] = t < 3 ? 0x7fffffff : ((t >= 4 && t < 7) ? (int32_t)0x80000000 : 0);
}

__global__ __launch_bounds__(kBlock) void voxel_keys_kernel(const int32_t* __restrict__ vox, const int32_t* __restrict__ seq,
                                                            int64_t n, const int32_t* __restrict__ vmin,
                                                            const int32_t* __restrict__ vmax, int32_t* __restrict__ bad,
                                                            uint64_t* __restrict__ keys, int32_t* __restrict__ vals) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= n) return;
  const int64_t i = seq ? seq[t] : t;
  uint64_t key = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int64_t range = (int64_t)vmax[a] - vmin[a];
    if (range >= (1 << 21)) atomicOr(bad, 2);
    key |= (uint64_t)((int64_t)vox[i * 3 + a] - vmin[a]) << (21 * a);
  }
  keys[t] = key;
  vals[t] = (int32_t)t;
}

__global__ __launch_bounds__(kBlock) void run_heads_kernel(const uint64_t* __restrict__ skeys, int64_t n, int32_t* __restrict__ head) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  head[p] = (p == 0 || skeys[p] != skeys[p - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void run_emit_kernel(const uint64_t* __restrict__ skeys, const int32_t* __restrict__ svals,
                                                          const int32_t* __restrict__ run_id, const int32_t* __restrict__ seq,
                                                          int64_t n, int preserve_order, int32_t* __restrict__ sort_key,
                                                          int32_t* __restrict__ surv, int32_t* __restrict__ count) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  const int32_t r = run_id[p] - 1;
  const bool head = p == 0 || skeys[p] != skeys[p - 1];
  const bool tail = p == n - 1 || skeys[p] != skeys[p + 1];
  if (tail) {
    const int32_t t = svals[p];
    const int32_t s = seq ? seq[t] : t;
    surv[r] = s;
    if (preserve_order) sort_key[r] = s;
    if (p == n - 1) *count = r + 1;
  }
  if (head && !preserve_order) sort_key[r] = svals[p];        // first appearance in the processing sequence
}


// ---- K17: inlier correspondences of two scans (train.py:186-193, 202-209; loss.py:440-452) -------------------------------------
// dists, ids = tree2.query(points1); th = np.quantile(dists[~isnan], ratio); mask1 = dists <= th; (mask1, ids[mask1]).  The 1-NN is
// dc_knn_build's; here the quantile WITHOUT a sort: a radix select over the bit patterns of the non-negative fp64 distances (their
// order is the numbers' order) finds the order statistic below the quantile position in eight passes of one byte -- a 256-bin
// histogram per pass, kept in LDS per block and merged -- one more pass finds the next distinct value and how many elements lie
// at or below, numpy's linear interpolation (lerp with its t >= 0.5 branch) gives the threshold, and the survivors' indices are
// compacted in their order.  Nothing returns to the host in between.
struct Nn1State {
  unsigned long long prefix;      // the bits found so far
  long long rank;                 // rank still to find among the elements that share the prefix
  long long n_valid, lo;          // non-NaN elements; index of the order statistic below the quantile position
  double gamma;                   // fractional part of the position
  long long n_le;                 // elements <= a[lo]
  unsigned long long next_bits;   // smallest value > a[lo] (bits), ~0 when none
  unsigned int hist[256];
};

__global__ __launch_bounds__(kBlock) void nn1_hist_kernel(const double* __restrict__ dist, int64_t n, Nn1State* st, int shift, int first) {
  __shared__ unsigned int s_hist[256];
  s_hist[threadIdx.x] = 0;
  __syncthreads();
  const unsigned long long prefix = first ? 0ull : st->prefix;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const double d = dist[i];
    if (d != d) continue;
    const unsigned long long b = (unsigned long long)__double_as_longlong(d);
    if (!first && (b >> (shift + 8)) != (prefix >> (shift + 8))) continue;
    atomicAdd(&s_hist[(b >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (s_hist[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], s_hist[threadIdx.x]);
}

__global__ void nn1_pick_kernel(Nn1State* st, int shift, int first, double ratio) {
  if (threadIdx.x != 0) return;
  if (first) {
    long long total = 0;
    for (int q = 0; q < 256; ++q) total += st->hist[q];
    st->n_valid = total;
    const double pos = (double)(total - 1) * ratio;                   // numpy: virtual_indexes = (n - 1) * q
    const double fl = floor(pos);
    st->lo = total > 0 ? (long long)fl : 0;
    st->gamma = pos - fl;
    st->rank = st->lo;
    st->prefix = 0ull;
    st->n_le = 0;
    st->next_bits = ~0ull;
  }
  long long r = st->rank;
  int bin = 255;
  for (int q = 0; q < 256; ++q) {
    const long long c = st->hist[q];
    if (r < c) { bin = q; break; }
    r -= c;
  }
  st->rank = r;
  st->prefix |= (unsigned long long)bin << shift;
  for (int q = 0; q < 256; ++q) st->hist[q] = 0;
}

__global__ __launch_bounds__(kBlock) void nn1_next_kernel(const double* __restrict__ dist, int64_t n, Nn1State* st) {
  __shared__ long long s_cnt[kBlock / kWave];
  __shared__ unsigned long long s_min[kBlock / kWave];
  const unsigned long long v = st->prefix;
  long long cnt = 0;
  unsigned long long mn = ~0ull;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const double d = dist[i];
    if (d != d) continue;
    const unsigned long long b = (unsigned long long)__double_as_longlong(d);
    if (b <= v) ++cnt;
    else mn = b < mn ? b : mn;
  }
  for (int o = kWave / 2; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, kWave);
    const unsigned long long other = __shfl_xor(mn, o, kWave);
    mn = other < mn ? other : mn;
  }
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == 0) { s_cnt[wave] = cnt; s_min[wave] = mn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int wv = 1; wv < kBlock / kWave; ++wv) { cnt += s_cnt[wv]; mn = s_min[wv] < mn ? s_min[wv] : mn; }
    atomicAdd((unsigned long long*)&st->n_le, (unsigned long long)cnt);
    atomicMin(&st->next_bits, mn);
  }
}

__global__ void nn1_threshold_kernel(Nn1State* st, double* __restrict__ threshold) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (st->n_valid <= 0) { *threshold = __longlong_as_double(0x7ff8000000000000ll); return; }      // np.quantile of nothing: NaN
  const double a = __longlong_as_double((long long)st->prefix);
  // a[lo + 1]: another copy of a[lo] when more than lo + 1 elements are <= it, else the next distinct value (a[lo] itself at the end)
  double b = a;
  if (st->lo + 1 < st->n_valid) b = (st->n_le >= st->lo + 2) ? a : __longlong_as_double((long long)st->next_bits);
  const double t = st->gamma, diff = b - a;
  *threshold = t >= 0.5 ? b - diff * (1.0 - t) : a + diff * t;       // numpy's _lerp
}

__global__ __launch_bounds__(kBlock) void nn1_flag_kernel(const double* __restrict__ dist, int64_t n, const double* __restrict__ threshold,
                                                          uint8_t* __restrict__ mask, int32_t* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const bool in = dist[i] <= *threshold;                               // (NaN compares false on either side, as in numpy)
  mask[i] = in ? 1 : 0;
  flags[i] = in ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void nn1_scatter_kernel(const int32_t* __restrict__ idx, const uint8_t* __restrict__ mask,
                                                             const int32_t* __restrict__ pos, int64_t n, int32_t* __restrict__ idx_out,
                                                             int64_t* __restrict__ count_out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  if (mask[i]) idx_out[pos[i]] = idx[i];
  if (i == n - 1) *count_out = (int64_t)pos[i] + (mask[i] ? 1 : 0);
}


// ---- stable compaction of the rows of several arrays by one mask (depth_cloud.py:126-134: cloud[mask] slices every per-point field) ----
// torch does this per field -- nonzero (count, partition, prefix sums) and one index_select each: ~25 launches for the four fields of
// an incoming scan.  Here: one counting kernel (kept rows of every 512-row block) and one kernel that places every field of the
// kept rows (a block's first output row = the sum of the counts before it, read by the block itself; beyond kCompactSelfScan blocks a
// prefix sum over the counts runs in between).
constexpr int kCompactPer = 2;                    // rows per lane
constexpr int kCompactRows = kCompactPer * kBlock; // rows of one block
constexpr int kCompactFields = 8;
constexpr int kCompactSelfScan = 4096;             // blocks up to which a block sums the counts before it by itself (2 M rows)
struct CompactFields {
  const void* src[kCompactFields];
  void* dst[kCompactFields];
  int32_t row_bytes[kCompactFields];
  int32_t n_fields;
};

__device__ __forceinline__ int block_sum_i32(int v, int* lds /* [kBlock / kWave] */) {
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x / kWave] = v;
  __syncthreads();
  int s = 0;
#pragma unroll
  for (int w = 0; w < kBlock / kWave; ++w) s += lds[w];
  __syncthreads();
  return s;
}

// (row = block base + q * kBlock + lane, q < kCompactPer: consecutive lanes read consecutive rows and, where they are kept, write consecutive rows)
__global__ __launch_bounds__(kBlock) void compact_count_kernel(const uint8_t* __restrict__ mask, int64_t n, int32_t* __restrict__ counts) {
  __shared__ int lds[kBlock / kWave];
  const int64_t base = (int64_t)blockIdx.x * kCompactRows + threadIdx.x;
  int c = 0;
#pragma unroll
  for (int q = 0; q < kCompactPer; ++q) { const int64_t i = base + q * kBlock; c += (i < n && mask[i]) ? 1 : 0; }
  const int tot = block_sum_i32(c, lds);
  if (threadIdx.x == 0) counts[blockIdx.x] = tot;
}

template <bool SCANNED>
__global__ __launch_bounds__(kBlock) void compact_place_kernel(const uint8_t* __restrict__ mask, int64_t n, const int32_t* __restrict__ counts,
                                                               CompactFields f, int32_t* __restrict__ index_out, int64_t* __restrict__ count_out) {
  constexpr int kWaves = kBlock / kWave;
  __shared__ int lds[kWaves];
  __shared__ int s_cnt[kCompactPer * kWaves];
  int before = 0;
  if (SCANNED) before = counts[blockIdx.x];                      // (exclusive prefix sums of the counts)
  else {
    int part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += kBlock) part += counts[b];
    before = block_sum_i32(part, lds);
  }
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int64_t base = (int64_t)blockIdx.x * kCompactRows + threadIdx.x;
  bool keep[kCompactPer];
  int below[kCompactPer];
#pragma unroll
  for (int q = 0; q < kCompactPer; ++q) {
    const int64_t i = base + q * kBlock;
    keep[q] = i < n && mask[i] != 0;
    const unsigned long long bal = __ballot(keep[q]);
    below[q] = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_cnt[q * kWaves + wave] = __popcll(bal);
  }
  __syncthreads();
  int run = before;
#pragma unroll
  for (int q = 0; q < kCompactPer; ++q) {
    int mine = run;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) { const int c = s_cnt[q * kWaves + w]; if (w < wave) mine += c; run += c; }
    if (!keep[q]) continue;
    const int64_t i = base + q * kBlock, o = mine + below[q];
    if (index_out) index_out[o] = (int32_t)i;
    for (int a = 0; a < f.n_fields; ++a) {
      const int rb = f.row_bytes[a];
      if ((rb & 3) == 0) {
        const int words = rb >> 2;
        const uint32_t* sp = static_cast<const uint32_t*>(f.src[a]) + i * words;
        uint32_t* dp = static_cast<uint32_t*>(f.dst[a]) + o * words;
        // (the usual rows -- [N,1], [N,3] of float32 / float64 -- with their loads issued together)
        if (words == 1) dp[0] = sp[0];
        else if (words == 2) { const uint32_t v0 = sp[0], v1 = sp[1]; dp[0] = v0; dp[1] = v1; }
        else if (words == 3) { const uint32_t v0 = sp[0], v1 = sp[1], v2 = sp[2]; dp[0] = v0; dp[1] = v1; dp[2] = v2; }
        else if (words == 6) {
          uint32_t v[6];
#pragma unroll
          for (int w = 0; w < 6; ++w) v[w] = sp[w];
#pragma unroll
          for (int w = 0; w < 6; ++w) dp[w] = v[w];
        } else for (int w = 0; w < words; ++w) dp[w] = sp[w];
      } else {
        const uint8_t* sp = static_cast<const uint8_t*>(f.src[a]) + i * rb;
        uint8_t* dp = static_cast<uint8_t*>(f.dst[a]) + o * rb;
        for (int w = 0; w < rb; ++w) dp[w] = sp[w];
      }
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && count_out) *count_out = (int64_t)run;
}

// points = vps + depth * dirs (depth_cloud.py:251-252), product and sum rounded separately like the two tensor operations
template <typename T>
__global__ __launch_bounds__(kBlock) void to_points_kernel(const T* __restrict__ vps, int vps_rows, const T* __restrict__ dirs,
                                                           const T* __restrict__ depth, int64_t n, T* __restrict__ out) {
#pragma clang fp contract(off)
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= 3 * n) return;
  const int64_t i = e / 3;
  const T prod = depth[i] * dirs[e];
  out[e] = vps[vps_rows == 1 ? e - 3 * i : e] + prod;
}

// weights = valid_neighbor_mask().float() (depth_cloud.py:341-343)
__global__ __launch_bounds__(kBlock) void valid_weights_kernel(const int32_t* __restrict__ nbr, int64_t count, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e < count) out[e] = nbr[e] >= 0 ? 1.0f : 0.0f;
}

// every bound on the columns (and column ratios) of one array in one pass: mask = [mask &] AND_b lo_b <= v[i, num_b] (/ v[i, den_b]) <= hi_b
constexpr int kMaxBounds = 8;
struct BoundSpecs {
  int32_t num[kMaxBounds], den[kMaxBounds];        // den < 0: the plain value
  double lo[kMaxBounds], hi[kMaxBounds];
  int32_t n_bounds;
};
template <typename T>
__global__ __launch_bounds__(kBlock) void mask_bounds_multi_kernel(const T* __restrict__ v, int stride, int64_t n, BoundSpecs b, int init,
                                                                   uint8_t* __restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  bool keep = init ? true : mask[i] != 0;
  for (int q = 0; q < b.n_bounds; ++q) {
    T x = v[i * stride + b.num[q]];
    if (b.den[q] >= 0) x = x / v[i * stride + b.den[q]];          // in storage precision, like the reference's tensor division
    keep = keep && in_bounds((double)x, b.lo[q], b.hi[q]);
  }
  mask[i] = keep ? 1 : 0;
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

static size_t voxel_ws(void* base, int64_t n, int32_t** vox, int32_t** box, uint64_t** keys, uint64_t** skeys, int32_t** vals,
                       int32_t** svals, int32_t** head, int32_t** run_id, int32_t** skey, int32_t** surv, int32_t** skey2,
                       void** tmp, size_t* tmp_bytes) {
  size_t off = 0;
  auto take = [&](size_t bytes) { off = (off + 255) & ~(size_t)255; void* p = base ? (char*)base + off : nullptr; off += bytes; return p; };
  const size_t m = (size_t)(n > 0 ? n : 1);
  *vox = (int32_t*)take(3 * m * 4); *box = (int32_t*)take(8 * 4);
  *keys = (uint64_t*)take(m * 8); *skeys = (uint64_t*)take(m * 8);
  *vals = (int32_t*)take(m * 4); *svals = (int32_t*)take(m * 4);
  *head = (int32_t*)take(m * 4); *run_id = (int32_t*)take(m * 4);
  *skey = (int32_t*)take(m * 4); *surv = (int32_t*)take(m * 4); *skey2 = (int32_t*)take(m * 4);
  const size_t a = sort_pairs_bytes(m, 64), b = sort_pairs_bytes(m, 32), c = scan_bytes(m);
  *tmp_bytes = a > b ? (a > c ? a : c) : (b > c ? b : c);
  *tmp = take(*tmp_bytes);
  return off + 256;
}

// mask_out[i] = 1 when every angle of point i lies in [lo, hi]; vps [n,3] or a single row (vps_rows = 1).
int dc_shadow_mask(const void* points, const void* vps, int vps_rows, int dtype, const int32_t* dir_nbr, int64_t n, int k,
                   double lo, double hi, double fill, uint8_t* mask_out, hipStream_t stream) {
  if (n < 0 || k < 1 || (n > 0 && (!points || !vps || !dir_nbr || !mask_out)) || (vps_rows != 1 && vps_rows != n)) return DC_ERR_ARG;
  if (n == 0) return DC_OK;
  const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
  if (dtype == DC_F32)
    hipLaunchKernelGGL((shadow_mask_kernel<float>), grid, block, 0, stream, (const float*)points, (const float*)vps, vps_rows, dir_nbr,
                       n, k, (float)lo, (float)hi, (float)fill, mask_out);
  else if (dtype == DC_F64)
    hipLaunchKernelGGL((shadow_mask_kernel<double>), grid, block, 0, stream, (const double*)points, (const double*)vps, vps_rows,
                       dir_nbr, n, k, lo, hi, fill, mask_out);
  else return DC_ERR_DTYPE;
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? DC_OK : (int)err;
}

// depth / gamma / out [n] in `dtype`; mask uint8 [n] or NULL; w / exponent fp64 [n_terms] on the device.
int dc_correct_depth(const void* depth, const void* gamma, const uint8_t* mask, const double* w, const double* exponent, int n_terms,
                     int op, int dtype, int64_t n, void* depth_out, hipStream_t stream) {
  if (n == 0) return DC_OK;
  if (!depth || !gamma || !w || !exponent || !depth_out || n < 0 || n_terms < 0 || op < 0 || op > 3) return DC_ERR_ARG;
  const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
  if (dtype == DC_F32)
    hipLaunchKernelGGL((correct_depth_kernel<float>), grid, block, 0, stream, (const float*)depth, (const float*)gamma, mask, w, exponent,
                       n_terms, op, n, (float*)depth_out);
  else if (dtype == DC_F64)
    hipLaunchKernelGGL((correct_depth_kernel<double>), grid, block, 0, stream, (const double*)depth, (const double*)gamma, mask, w,
                       exponent, n_terms, op, n, (double*)depth_out);
  else return DC_ERR_DTYPE;
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? DC_OK : (int)err;
}

size_t dc_voxel_filter_workspace_bytes(int64_t n) {
  int32_t *a, *b, *e, *f, *g, *h, *i, *j, *k; uint64_t *c, *d; void* t; size_t tb;
  return n < 0 ? 0 : voxel_ws(nullptr, n, &a, &b, &c, &d, &e, &f, &g, &h, &i, &j, &k, &t, &tb);
}

// seq int32 [n] or NULL: processing sequence (seq[t] = index of the t-th point offered to the dict).  out_idx int32 [n]
// (first *count_out entries valid), count_out device int32; status_out device int32: 0 ok, != 0 voxel range too large
// for the 3 x 21-bit key (use the host filter then).
int dc_voxel_filter(const void* points, int stride, int dtype, int64_t n, double grid_res, const int32_t* seq,
                    int preserve_order, int32_t* out_idx, int32_t* count_out, int32_t* status_out, void* ws, size_t ws_bytes,
                    hipStream_t stream) {
  if (n == 0 && count_out && status_out) {
    hipError_t e0 = hipMemsetAsync(count_out, 0, 4, stream);
    if (e0 == hipSuccess) e0 = hipMemsetAsync(status_out, 0, 4, stream);
    return (int)e0;
  }
  if (!points || n < 0 || stride < 3 || !(grid_res > 0.0) || !out_idx || !count_out || !status_out || !ws) return DC_ERR_ARG;
  if (n >= (int64_t)0x7fffffff) return DC_ERR_UNSUPPORTED;
  int32_t *vox, *box, *vals, *svals, *head, *run_id, *skey, *surv, *skey2; uint64_t *keys, *skeys; void* tmp; size_t tmp_bytes;
  if (ws_bytes < voxel_ws(ws, n, &vox, &box, &keys, &skeys, &vals, &svals, &head, &run_id, &skey, &surv, &skey2, &tmp, &tmp_bytes))
    return DC_ERR_WORKSPACE;
  hipLaunchKernelGGL(voxel_box_init_kernel, dim3(1), dim3(8), 0, stream, box);       // no host buffer behind an async copy
  hipError_t err = hipMemsetAsync(status_out, 0, 4, stream);
  if (err != hipSuccess) return (int)err;
  err = hipMemsetAsync(skey, 0x7f, (size_t)n * 4, stream);               // 0x7f7f7f7f: unused slots sort last
  if (err != hipSuccess) return (int)err;
  const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
  if (dtype == DC_F32)
    hipLaunchKernelGGL((voxel_coords_kernel<float>), grid, block, 0, stream, (const float*)points, stride, n, grid_res, vox, box, box + 4, status_out);
  else if (dtype == DC_F64)
    hipLaunchKernelGGL((voxel_coords_kernel<double>), grid, block, 0, stream, (const double*)points, stride, n, grid_res, vox, box, box + 4, status_out);
  else return DC_ERR_DTYPE;
  hipLaunchKernelGGL(voxel_keys_kernel, grid, block, 0, stream, vox, seq, n, box, box + 4, status_out, keys, vals);
  err = sort_pairs_u64(tmp, tmp_bytes, keys, skeys, vals, svals, (size_t)n, 0, 63, stream);
  if (err != hipSuccess) return (int)err;
  hipLaunchKernelGGL(run_heads_kernel, grid, block, 0, stream, skeys, n, head);
  err = inclusive_scan_32(tmp, tmp_bytes, head, run_id, (size_t)n, stream);
  if (err != hipSuccess) return (int)err;
  hipLaunchKernelGGL(run_emit_kernel, grid, block, 0, stream, skeys, svals, run_id, seq, n, preserve_order, skey, surv, count_out);
  err = sort_pairs_u32(tmp, tmp_bytes, skey, skey2, surv, out_idx, (size_t)n, 0, 32, stream);      // (keys never negative)
  if (err != hipSuccess) return (int)err;
  err = hipGetLastError();
  return err == hipSuccess ? DC_OK : (int)err;
}


size_t dc_nn1_corr_workspace_bytes(int64_t n) {
  if (n < 0) return 0;
  const size_t m = (size_t)(n > 0 ? n : 1);
  return 256 + ((sizeof(dc::Nn1State) + 255) / 256) * 256 + 2 * ((m * sizeof(int32_t) + 255) / 256) * 256 + dc::scan_bytes(m);
}

int dc_nn1_corr(const double* dist, const int32_t* idx, int64_t n, double ratio, uint8_t* mask_out, int32_t* idx_out, int64_t* count_out,
                double* threshold_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n < 0 || !(ratio >= 0.0 && ratio <= 1.0) || !count_out || !threshold_out) return DC_ERR_ARG;
  if (n == 0) {
    hipError_t e0 = hipMemsetAsync(count_out, 0, sizeof(int64_t), stream);
    if (e0 == hipSuccess) e0 = hipMemsetAsync(threshold_out, 0xff, sizeof(double), stream);      // (a NaN)
    return e0 == hipSuccess ? DC_OK : (int)e0;
  }
  if (!dist || !idx || !mask_out || !idx_out || !ws || n > 0x7fffffff) return DC_ERR_ARG;
  if (ws_bytes < dc_nn1_corr_workspace_bytes(n)) return DC_ERR_WORKSPACE;
  char* base = static_cast<char*>(ws);
  dc::Nn1State* st = reinterpret_cast<dc::Nn1State*>(base);
  const size_t st_bytes = ((sizeof(dc::Nn1State) + 255) / 256) * 256, arr = (((size_t)n * sizeof(int32_t) + 255) / 256) * 256;
  int32_t* flags = reinterpret_cast<int32_t*>(base + st_bytes);
  int32_t* pos = reinterpret_cast<int32_t*>(base + st_bytes + arr);
  void* scan_ws = base + st_bytes + 2 * arr;
  hipError_t err = hipMemsetAsync(st, 0, sizeof(dc::Nn1State), stream);
  if (err != hipSuccess) return (int)err;
  const unsigned blocks_all = (unsigned)((n + dc::kBlock - 1) / dc::kBlock);
  const unsigned blocks = blocks_all < 512u ? blocks_all : 512u;
  for (int pass = 0; pass < 8; ++pass) {
    const int shift = 56 - 8 * pass;
    hipLaunchKernelGGL(dc::nn1_hist_kernel, dim3(blocks), dim3(dc::kBlock), 0, stream, dist, n, st, shift, pass == 0 ? 1 : 0);
    hipLaunchKernelGGL(dc::nn1_pick_kernel, dim3(1), dim3(64), 0, stream, st, shift, pass == 0 ? 1 : 0, ratio);
  }
  hipLaunchKernelGGL(dc::nn1_next_kernel, dim3(blocks), dim3(dc::kBlock), 0, stream, dist, n, st);
  hipLaunchKernelGGL(dc::nn1_threshold_kernel, dim3(1), dim3(64), 0, stream, st, threshold_out);
  hipLaunchKernelGGL(dc::nn1_flag_kernel, dim3(blocks_all), dim3(dc::kBlock), 0, stream, dist, n, (const double*)threshold_out, mask_out, flags);
  err = dc::exclusive_scan_32(scan_ws, dc::scan_bytes((size_t)n), flags, pos, (size_t)n, stream);
  if (err != hipSuccess) return (int)err;
  hipLaunchKernelGGL(dc::nn1_scatter_kernel, dim3(blocks_all), dim3(dc::kBlock), 0, stream, idx, (const uint8_t*)mask_out, (const int32_t*)pos, n,
                     idx_out, count_out);
  err = hipGetLastError();
  return err == hipSuccess ? DC_OK : (int)err;
}

size_t dc_compact_rows_workspace_bytes(int64_t n) {
  if (n < 0) return 0;
  const size_t nb = (size_t)((n + dc::kCompactRows - 1) / dc::kCompactRows) + 1;
  return 512 + 2 * ((nb * sizeof(int32_t) + 255) / 256) * 256 + dc::scan_bytes(nb);
}

int dc_compact_rows(const uint8_t* mask, int64_t n, int n_fields, const void* const* src, void* const* dst, const int32_t* row_bytes,
                    int32_t* index_out, int64_t* count_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n < 0 || n_fields < 0 || n_fields > dc::kCompactFields || !count_out) return DC_ERR_ARG;
  if (n == 0) { hipError_t e0 = hipMemsetAsync(count_out, 0, sizeof(int64_t), stream); return e0 == hipSuccess ? DC_OK : (int)e0; }
  if (!mask || !ws || n > 0x7fffffff || (n_fields > 0 && (!src || !dst || !row_bytes))) return DC_ERR_ARG;
  if (ws_bytes < dc_compact_rows_workspace_bytes(n)) return DC_ERR_WORKSPACE;
  dc::CompactFields f;
  memset(&f, 0, sizeof(f));
  f.n_fields = n_fields;
  for (int a = 0; a < n_fields; ++a) {
    if (!src[a] || !dst[a] || row_bytes[a] < 1) return DC_ERR_ARG;
    // (word copies need 4-byte aligned rows)
    f.src[a] = src[a]; f.dst[a] = dst[a]; f.row_bytes[a] = row_bytes[a];
    if ((row_bytes[a] & 3) == 0 && ((((uintptr_t)src[a]) | ((uintptr_t)dst[a])) & 3) != 0) return DC_ERR_ARG;
  }
  const unsigned nb = (unsigned)((n + dc::kCompactRows - 1) / dc::kCompactRows);
  dc::Carver c(ws);
  int32_t* counts = c.take<int32_t>(nb + 1);
  int32_t* offsets = c.take<int32_t>(nb + 1);
  void* scan_ws = c.take<char>(dc::scan_bytes(nb + 1));
  hipLaunchKernelGGL(dc::compact_count_kernel, dim3(nb), dim3(dc::kBlock), 0, stream, mask, n, counts);
  if ((int)nb <= dc::kCompactSelfScan) {
    hipLaunchKernelGGL((dc::compact_place_kernel<false>), dim3(nb), dim3(dc::kBlock), 0, stream, mask, n, (const int32_t*)counts, f, index_out,
                       count_out);
  } else {
    hipError_t err = dc::exclusive_scan_32(scan_ws, dc::scan_bytes(nb + 1), counts, offsets, (size_t)nb, stream);
    if (err != hipSuccess) return (int)err;
    hipLaunchKernelGGL((dc::compact_place_kernel<true>), dim3(nb), dim3(dc::kBlock), 0, stream, mask, n, (const int32_t*)offsets, f, index_out,
                       count_out);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DC_OK : (int)e;
}

int dc_to_points(const void* vps, int vps_rows, const void* dirs, const void* depth, int dtype, int64_t n, void* points_out, hipStream_t stream) {
  if (n == 0) return DC_OK;
  if (n < 0 || !vps || !dirs || !depth || !points_out || (vps_rows != 1 && vps_rows != n)) return DC_ERR_ARG;
  const dim3 grid((unsigned)((3 * n + dc::kBlock - 1) / dc::kBlock)), block(dc::kBlock);
  if (3 * n > (int64_t)0x7fffffff * dc::kBlock) return DC_ERR_UNSUPPORTED;
  if (dtype == DC_F32)
    hipLaunchKernelGGL((dc::to_points_kernel<float>), grid, block, 0, stream, (const float*)vps, vps_rows, (const float*)dirs, (const float*)depth, n,
                       (float*)points_out);
  else if (dtype == DC_F64)
    hipLaunchKernelGGL((dc::to_points_kernel<double>), grid, block, 0, stream, (const double*)vps, vps_rows, (const double*)dirs,
                       (const double*)depth, n, (double*)points_out);
  else return DC_ERR_DTYPE;
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DC_OK : (int)e;
}

int dc_valid_weights(const int32_t* nbr, int64_t count, float* weights_out, hipStream_t stream) {
  if (count == 0) return DC_OK;
  if (count < 0 || !nbr || !weights_out) return DC_ERR_ARG;
  if (count > (int64_t)0x7fffffff * dc::kBlock) return DC_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(dc::valid_weights_kernel, dim3((unsigned)((count + dc::kBlock - 1) / dc::kBlock)), dim3(dc::kBlock), 0, stream, nbr, count,
                     weights_out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DC_OK : (int)e;
}

int dc_mask_bounds_multi(const void* values, int stride, int dtype, int64_t n, int n_bounds, const int32_t* num_index, const int32_t* den_index,
                         const double* lo, const double* hi, int init, uint8_t* mask, hipStream_t stream) {
  if (n == 0) return DC_OK;
  if (n < 0 || !values || !mask || stride < 1 || n_bounds < 0 || n_bounds > dc::kMaxBounds) return DC_ERR_ARG;
  if (n_bounds > 0 && (!num_index || !den_index || !lo || !hi)) return DC_ERR_ARG;
  dc::BoundSpecs b;
  memset(&b, 0, sizeof(b));
  b.n_bounds = n_bounds;
  for (int q = 0; q < n_bounds; ++q) {
    if (num_index[q] < 0 || num_index[q] >= stride || den_index[q] >= stride) return DC_ERR_ARG;
    b.num[q] = num_index[q]; b.den[q] = den_index[q]; b.lo[q] = lo[q]; b.hi[q] = hi[q];
  }
  const dim3 grid((unsigned)((n + dc::kBlock - 1) / dc::kBlock)), block(dc::kBlock);
  if (dtype == DC_F32)
    hipLaunchKernelGGL((dc::mask_bounds_multi_kernel<float>), grid, block, 0, stream, (const float*)values, stride, n, b, init, mask);
  else if (dtype == DC_F64)
    hipLaunchKernelGGL((dc::mask_bounds_multi_kernel<double>), grid, block, 0, stream, (const double*)values, stride, n, b, init, mask);
  else return DC_ERR_DTYPE;
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? DC_OK : (int)e;
}

// ---- first stage of the online node in one call (scripts/depth_correction:31-58 -> preproc.py:44-47): from_points, points, the
// scan-shadow mask off the direction grid, cloud[mask].  A composition of the entry points above and in dc_knn.hip / dc_scanio.hip
// -- what it saves is the host's time between their launches (five Python calls, ~150 us during which the device waits).
static size_t prefilter_field_bytes(int64_t n, int out_dtype) { return (((size_t)n * 3 * (out_dtype == DC_F32 ? 4 : 8)) + 255) / 256 * 256; }

size_t dc_scan_prefilter_workspace_bytes(int64_t n) {
  if (n < 0) return 0;
  const size_t m = (size_t)(n > 0 ? n : 1);
  size_t sub = dc_knn_workspace_bytes((int64_t)m, 0);
  const size_t b = dc_cloud_from_points_workspace_bytes((int64_t)m), c = dc_compact_rows_workspace_bytes((int64_t)m);
  sub = sub > b ? sub : b;
  sub = sub > c ? sub : c;
  return 4 * prefilter_field_bytes((int64_t)m, DC_F64) + ((m + 255) / 256) * 256 + 512 + sub;
}

int dc_scan_prefilter(const void* points, int stride, int in_dtype, const void* vps, int64_t n, int out_dtype, double shadow_r,
                      double shadow_lo, double shadow_hi, void* vps_out, void* dirs_out, void* depth_out, void* points_out,
                      int64_t* count_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n < 0 || stride < 3 || !count_out || !(shadow_r > 0.0)) return DC_ERR_ARG;
  if ((in_dtype != DC_F32 && in_dtype != DC_F64) || (out_dtype != DC_F32 && out_dtype != DC_F64)) return DC_ERR_DTYPE;
  if (n == 0) { hipError_t e0 = hipMemsetAsync(count_out, 0, sizeof(int64_t), stream); return e0 == hipSuccess ? DC_OK : (int)e0; }
  if (!points || !vps_out || !dirs_out || !depth_out || !points_out || !ws) return DC_ERR_ARG;
  if (ws_bytes < dc_scan_prefilter_workspace_bytes(n)) return DC_ERR_WORKSPACE;
  char* base = static_cast<char*>(ws);
  const size_t fb = prefilter_field_bytes(n, DC_F64);
  void* t_vps = base; void* t_dirs = base + fb; void* t_depth = base + 2 * fb; void* t_pts = base + 3 * fb;
  uint8_t* mask = reinterpret_cast<uint8_t*>(base + 4 * fb);
  int64_t* count_all = reinterpret_cast<int64_t*>(base + 4 * fb + (((size_t)n + 255) / 256) * 256);
  void* sub = base + 4 * fb + (((size_t)n + 255) / 256) * 256 + 512;
  const size_t sub_bytes = ws_bytes - (size_t)((char*)sub - base);
  const double nan = __builtin_nan("");
  int rc = dc_cloud_from_points(points, stride, in_dtype, vps, n, 0.0, nan, nan, out_dtype, t_dirs, t_depth, t_vps, nullptr, count_all, sub,
                                sub_bytes, stream);
  if (rc) return rc;
  if ((rc = dc_to_points(t_vps, (int)n, t_dirs, t_depth, out_dtype, n, t_pts, stream))) return rc;
  if ((rc = dc_shadow_filter(t_pts, t_vps, (int)n, t_dirs, out_dtype, n, shadow_r, shadow_lo, shadow_hi, mask, sub, sub_bytes, stream))) return rc;
  const int es = out_dtype == DC_F32 ? 4 : 8;
  const void* src[4] = {t_vps, t_dirs, t_depth, t_pts};
  void* dst[4] = {vps_out, dirs_out, depth_out, points_out};
  const int32_t rb[4] = {3 * es, 3 * es, es, 3 * es};
  return dc_compact_rows(mask, n, 4, src, dst, rb, nullptr, count_out, sub, sub_bytes, stream);
}

}  // extern "C"
