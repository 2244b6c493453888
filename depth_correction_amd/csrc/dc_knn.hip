// Neighbourhood builder on the GPU (gfx950): k-NN / k-within-radius / radius / cross-cloud 1-NN over lidar
// points, plus the transposed neighbour list used by the backward and the spatial (Morton) order used
// to lay a sequence out in HBM.  Replaces scipy.spatial.cKDTree at reference nearest_neighbors.py:46-51
// (and loss.py:442-443, train.py:188-189).  C ABI at the bottom; see include/dc_hip.h.
//
// Contract reproduced from cKDTree (fp64 tree even for fp32 input): neighbours ordered by ascending
// fp64 squared distance s = ((dx*dx + dy*dy) + dz*dz) evaluated WITHOUT fused multiply-add, query()
// keeps d2 < r*r, query_ball_point() keeps d2 <= r*r and returns ascending indices, missing -> -1 / inf.
// Exact distance ties are outside the contract (cKDTree orders them by tree traversal); here they are
// ordered by index, so results are deterministic.
//
// Structure: points are binned into a uniform grid whose cell keys are 63-bit Morton codes; a radix
// sort (rocPRIM) groups the cells, an open-addressing hash table maps key -> [begin, end) in the sorted
// array.  One lane per query walks cube shells of cells outwards until the k-th best distance is
// provably final.  Queries are processed in Morton order, so the lanes of a wavefront read the same
// few cells (L1/L2 hits) and the sorted coordinates they stream are contiguous.
#include <cstring>
#include <cstdlib>
#include <atomic>
#include "dc_common.h"
#include "../../include/dc_hip.h"
#include "dc_device.h"
#include "dc_hostutil.h"
#include "dc_sort.h"

namespace dc {

struct Grid {
  double origin[3];
  double h, inv_h;
  int32_t dim[3];
};

constexpr uint64_t kEmptyKey = ~0ull;

__device__ __forceinline__ uint64_t spread21(uint32_t v) {
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
__device__ __forceinline__ uint64_t morton3(int32_t x, int32_t y, int32_t z) {
  return spread21((uint32_t)x) | (spread21((uint32_t)y) << 1) | (spread21((uint32_t)z) << 2);
}
// TWO LEVELS.  The cell edge h is chosen from the cloud's AVERAGE areal density, but lidar density is anything but uniform: weighted
// by query, the 27 cells around a query hold 110 points on the 2 M-point test cloud (median 77, 90th percentile 245) for k = 10.  So
// the grid has a second, fine level of edge h / 2: the points are ordered by the 30-bit Morton code of their FINE cell (10 bits per
// axis; grid_from_box keeps the coarse level below 512 cells per axis), a coarse cell is then eight consecutive fine cells --
// one sorted array serves both levels -- and the hash table's entry of a coarse cell carries the starts of its eight fine cells.
// knn_group_kernel searches at the fine level where the query's own coarse cell holds at least g_knn_fine_min points (mean
// candidates 110 -> 53 at 16, 11.6 % instead of 11.2 % of the queries need a second stage); every other kernel uses the coarse level.
typedef uint32_t GridKey;
constexpr int kGridKeyBits = 30;
constexpr int kGridAxisCells = 511;              // most cells per axis of a search grid (its fine level has twice as many: 10 bits)
__device__ __forceinline__ uint32_t spread10(uint32_t v) {
  uint32_t x = v & 0x3ffu;
  x = (x | x << 16) & 0x030000ffu;
  x = (x | x << 8) & 0x0300f00fu;
  x = (x | x << 4) & 0x030c30c3u;
  x = (x | x << 2) & 0x09249249u;
  return x;
}
__device__ __forceinline__ uint32_t compact10(uint32_t x) {             // inverse of spread10
  x &= 0x09249249u;
  x = (x | x >> 2) & 0x030c30c3u;
  x = (x | x >> 4) & 0x0300f00fu;
  x = (x | x >> 8) & 0x030000ffu;
  x = (x | x >> 16) & 0x3ffu;
  return x;
}
__device__ __forceinline__ GridKey grid_key(int32_t x, int32_t y, int32_t z) {
  return spread10((uint32_t)x) | (spread10((uint32_t)y) << 1) | (spread10((uint32_t)z) << 2);
}
// cell of p at `level` (0: edge h, 1: edge h / 2); the coarse cell of a point is its fine cell >> 1 exactly (2 x is exact)
__device__ __forceinline__ void cell_of_level(const Grid& g, const double* p, int level, int32_t* c) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    double t = (p[a] - g.origin[a]) * g.inv_h;
    if (level) t *= 2.0;
    const double f = floor(t);
    const int32_t top = (g.dim[a] << level) - 1;
    c[a] = (f > 0.0) ? ((f < (double)top) ? (int32_t)f : top) : 0;   // NaN -> 0
  }
}
__device__ __forceinline__ void cell_of(const Grid& g, const double* p, int32_t* c) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    double f = floor((p[a] - g.origin[a]) * g.inv_h);
    int32_t ci = (f > 0.0) ? ((f < (double)(g.dim[a] - 1)) ? (int32_t)f : g.dim[a] - 1) : 0;   // NaN -> 0
    c[a] = ci;
  }
}
// The hash table of the cells is keyed by the PACKED cell coordinates (21 bits each) and hashed by a three-multiply mix of
// them; the Morton code only orders the sorted array.  A query probes ~27-125 cells, and interleaving three coordinates
// (~45 instructions) plus a 64-bit finaliser (~20) per probe was a quarter of knn_query_kernel's instructions.
__device__ __forceinline__ uint64_t cell_key(int32_t x, int32_t y, int32_t z) {
  return (uint64_t)(uint32_t)x | ((uint64_t)(uint32_t)y << 21) | ((uint64_t)(uint32_t)z << 42);
}
__device__ __forceinline__ uint32_t cell_hash(int32_t x, int32_t y, int32_t z) {
  uint32_t h = (uint32_t)x * 0x9E3779B1u ^ (uint32_t)y * 0x85EBCA77u ^ (uint32_t)z * 0xC2B2AE3Du;
  return h ^ (h >> 15);
}

template <typename T>
__device__ __forceinline__ void load_xyz(const T* p, int64_t i, int stride, double* o) {
  const T* q = p + i * stride;
  o[0] = (double)q[0]; o[1] = (double)q[1]; o[2] = (double)q[2];
}

// ---- bounding box and first two moments per axis (finite points only) -----------------------------
constexpr int kBoxVals = 13;             // min[3], max[3], sum[3], sum of squares[3], count
template <typename T>
__global__ __launch_bounds__(kBlock) void bbox_partial_kernel(const T* __restrict__ xyz, int stride, int64_t n,
                                                              double* __restrict__ part) {
  __shared__ double lds[(kBlock / kWave) * kBoxVals];
  double v[kBoxVals] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    double p[3];
    load_xyz(xyz, i, stride, p);
    if (isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2])) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        v[a] = fmin(v[a], p[a]); v[3 + a] = fmax(v[3 + a], p[a]);
        v[6 + a] += p[a]; v[9 + a] += p[a] * p[a];
      }
      v[12] += 1.0;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < kBoxVals; ++q) {
    double s = v[q];
    for (int off = 32; off > 0; off >>= 1) {
      double o = __shfl_down(s, off, 64);
      s = q < 3 ? fmin(s, o) : (q < 6 ? fmax(s, o) : s + o);
    }
    if (lane == 0) lds[wave * kBoxVals + q] = s;
  }
  __syncthreads();
  if (threadIdx.x < kBoxVals) {
    const int q = threadIdx.x;
    double s = lds[q];
    for (int wv = 1; wv < kBlock / kWave; ++wv) {
      const double o = lds[wv * kBoxVals + q];
      s = q < 3 ? fmin(s, o) : (q < 6 ? fmax(s, o) : s + o);
    }
    part[(int64_t)blockIdx.x * kBoxVals + q] = s;
  }
}

// Combine the partial rows of bbox_partial_kernel inside one block: thread q < kBoxVals ends up with value q
// (min for 0..2, max for 3..5, sum for the rest).  A single thread walking all rows took ~70 us per build.
__device__ __forceinline__ void combine_box_partials(const double* __restrict__ part, int n_part, double* out /* [kBoxVals], shared */) {
  __shared__ double s_row[kBlock / kWave][kBoxVals];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  double v[kBoxVals];
#pragma unroll
  for (int q = 0; q < kBoxVals; ++q) v[q] = q < 3 ? INFINITY : (q < 6 ? -INFINITY : 0.0);
  for (int b = threadIdx.x; b < n_part; b += blockDim.x) {
#pragma unroll
    for (int q = 0; q < kBoxVals; ++q) {
      const double o = part[(int64_t)b * kBoxVals + q];
      v[q] = q < 3 ? fmin(v[q], o) : (q < 6 ? fmax(v[q], o) : v[q] + o);
    }
  }
#pragma unroll
  for (int q = 0; q < kBoxVals; ++q) {
    double s = v[q];
    for (int off = kWave / 2; off > 0; off >>= 1) {
      const double o = __shfl_down(s, off, kWave);
      s = q < 3 ? fmin(s, o) : (q < 6 ? fmax(s, o) : s + o);
    }
    if (lane == 0) s_row[wave][q] = s;
  }
  __syncthreads();
  if (threadIdx.x < kBoxVals) {
    const int q = threadIdx.x;
    double s = s_row[0][q];
    for (int wv = 1; wv < (int)(blockDim.x / kWave); ++wv) {
      const double o = s_row[wv][q];
      s = q < 3 ? fmin(s, o) : (q < 6 ? fmax(s, o) : s + o);
    }
    out[q] = s;
  }
  __syncthreads();
}

// One block: finish the bounding box and choose the cell size.
//   cell_hint > 0 : use it (radius searches use r).
//   otherwise     : lidar points lie on surfaces; with areal density sigma ~ N / A(bbox) a cell of edge
//                   h holds ~ sigma h^2 points and a ball of radius h ~ pi sigma h^2, so h = sqrt(A k / (2N))
//                   lets most queries finish after the first shell.
__device__ void grid_from_box(const double* tot, int64_t n, int k, double cell_hint, Grid* g) {
  double lo[3] = {tot[0], tot[1], tot[2]}, hi[3] = {tot[3], tot[4], tot[5]};
  const double sum[3] = {tot[6], tot[7], tot[8]}, sq[3] = {tot[9], tot[10], tot[11]}, cnt = tot[12];
  // The grid covers the bulk of the cloud: the bounding box cut to mean +- 6 sigma per axis.  A few far outliers
  // would otherwise inflate the box, and with it the cell size chosen from the areal density below, until whole
  // surfaces fall into single cells; points outside the grid are clamped into its boundary cells (cell_of), which
  // keeps every distance bound valid (their true position is farther away than the cell they sit in).
  double L[3], Lmax = 0.0;
  for (int a = 0; a < 3; ++a) {
    if (!(hi[a] >= lo[a])) { lo[a] = 0.0; hi[a] = 0.0; }
    if (cnt > 1.0) {
      const double mean = sum[a] / cnt;
      double var = sq[a] / cnt - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const double reach = 6.0 * sqrt(var);
      if (isfinite(mean) && isfinite(reach)) { lo[a] = fmax(lo[a], mean - reach); hi[a] = fmin(hi[a], mean + reach); }
      if (!(hi[a] >= lo[a])) { lo[a] = hi[a] = mean; }
    }
    L[a] = hi[a] - lo[a];
    Lmax = fmax(Lmax, L[a]);
  }
  double h = cell_hint;
  if (!(h > 0.0)) {
    const double area = 2.0 * (L[0] * L[1] + L[1] * L[2] + L[0] * L[2]);
    const double kk = k > 1 ? (double)k : 2.0;
    h = area > 0.0 ? sqrt(area * kk / (2.0 * (double)(n > 0 ? n : 1))) : Lmax / 16.0;
    if (!(h > 0.0)) h = 1.0;
  }
  const double hmin = Lmax / (double)(kGridAxisCells - 1);          // at most kGridAxisCells cells per axis (30-bit cell order key)
  if (h < hmin) h = hmin;
  g->h = h;
  g->inv_h = 1.0 / h;
  for (int a = 0; a < 3; ++a) {
    g->origin[a] = lo[a];
    double d = floor(L[a] / h) + 1.0;
    g->dim[a] = d < (double)kGridAxisCells ? (int32_t)d : kGridAxisCells;
  }
}

__global__ void grid_setup_kernel(const double* __restrict__ part, int n_part, int64_t n, int k, double cell_hint, Grid* __restrict__ g) {
  __shared__ double tot[kBoxVals];
  combine_box_partials(part, n_part, tot);
  if (threadIdx.x == 0 && blockIdx.x == 0) grid_from_box(tot, n, k, cell_hint, g);
}

// Every block finishes the box itself (a few hundred partial rows) and derives the grid from it -- the launch of a one-block set-up
// kernel between two passes over the points costs more than the 20 MB of cached re-reads --; block 0 stores the grid for the kernels
// that follow.  The blocks also clear the hash table of the cells (filled after the sort) and the pending-query counter.
template <typename T>
__global__ __launch_bounds__(kBlock) void cell_keys_kernel(const T* __restrict__ xyz, int stride, int64_t n, const double* __restrict__ part,
                                                           int n_part, int k, double cell_hint, Grid* __restrict__ gp,
                                                           GridKey* __restrict__ keys, int32_t* __restrict__ ids,
                                                           uint64_t* __restrict__ tab_key, uint32_t tab_n, int32_t* __restrict__ n_pending) {
  __shared__ double tot[kBoxVals];
  __shared__ Grid s_g;
  if (n_part > 0) {
    combine_box_partials(part, n_part, tot);
    if (threadIdx.x == 0) {
      grid_from_box(tot, n, k, cell_hint, &s_g);
      if (blockIdx.x == 0) *gp = s_g;
    }
  } else if (threadIdx.x == 0) s_g = *gp;          // (large clouds: grid_setup_kernel ran before -- thousands of blocks re-reading the partials cost more than its launch)
  if (threadIdx.x == 0 && blockIdx.x == 0) *n_pending = 0;
  for (uint64_t e = (uint64_t)blockIdx.x * kBlock + threadIdx.x; e < tab_n; e += (uint64_t)gridDim.x * kBlock) tab_key[e] = kEmptyKey;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const Grid g = s_g;
  double p[3];
  int32_t c[3];
  load_xyz(xyz, i, stride, p);
  cell_of_level(g, p, 1, c);                      // the order key is the FINE cell's; coarse cell = fine >> 1 per axis
  keys[i] = grid_key(c[0], c[1], c[2]);
  ids[i] = (int32_t)i;
}

// Fine Morton key over the bounding box (21 bits per axis) for the spatial layout order.
__global__ void box_finish_kernel(const double* __restrict__ part, int n_part, double* __restrict__ box) {
  __shared__ double tot[kBoxVals];
  combine_box_partials(part, n_part, tot);
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double lo[3] = {tot[0], tot[1], tot[2]}, hi[3] = {tot[3], tot[4], tot[5]};
  double Lmax = 0.0;
  for (int a = 0; a < 3; ++a) { if (!(hi[a] >= lo[a])) { lo[a] = hi[a] = 0.0; } Lmax = fmax(Lmax, hi[a] - lo[a]); }
  box[0] = lo[0]; box[1] = lo[1]; box[2] = lo[2];
  box[3] = Lmax > 0.0 ? 2097151.0 / Lmax : 0.0;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void fine_keys_kernel(const T* __restrict__ xyz, int stride, int64_t n,
                                                           const double* __restrict__ box,
                                                           uint64_t* __restrict__ keys, int32_t* __restrict__ ids) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const double s = box[3];
  double p[3];
  int32_t c[3];
  load_xyz(xyz, i, stride, p);
  for (int a = 0; a < 3; ++a) {
    double f = (p[a] - box[a]) * s;
    c[a] = (f > 0.0) ? (f < 2097151.0 ? (int32_t)f : 2097151) : 0;
  }
  keys[i] = morton3(c[0], c[1], c[2]);
  ids[i] = (int32_t)i;
}

// One hash table, keyed by the COARSE cell (open addressing, linear probing; empty key = all ones); its payload is the nine
// sorted positions s[0..8] at which the cell's eight fine cells start (s[8] = the cell's end; Morton sub-index
// i = fx&1 | (fy&1) << 1 | (fz&1) << 2, an empty fine cell has s[i] == s[i+1]).  A coarse lookup reads s[0], s[8]; a fine lookup
// hashes the coarse cell it lies in and reads s[i], s[i+1]: no second table, and the 27 fine cells around a query lie in only
// 8 coarse cells, so their probes share cache lines.  The payload is requested TOGETHER with the key (its address only needs the
// slot): a probe is one memory round trip unless the slot holds another cell's key.  Keys (8 B) and payloads (48 B) are separate arrays: most probes of a
// neighbourhood hit empty cells and only ever touch the key array (16-byte {key, begin, end} entries doubled the bytes those
// probes pull in and made the 2 M-point build memory-bound).
constexpr int kCellStride = 12;                  // ints per payload: s[0..8], pad, then (begin, end) again as one aligned pair
struct CellTable {
  const uint64_t* key;
  const int32_t* s;
  uint32_t mask;
};

// Gather the sorted fp64 coordinates and fill the hash table: the first point of every FINE cell and the last point of every coarse
// cell find -- or claim -- the table slot of their coarse cell (one compare-and-swap serves both) and write the positions they know:
// the first point of a fine cell its position into s[i] of its coarse cell and into the entries of the empty fine cells before it
// (back to the previous point's fine cell, or to 0 at the coarse cell's first point); the last point of the coarse cell the end into
// the rest.  (Two kernels until round 5: all keys first, then the positions.)
template <typename T>
__global__ __launch_bounds__(kBlock) void sorted_cells_kernel(const T* __restrict__ xyz, int stride, int64_t n,
                                                              const GridKey* __restrict__ skeys, const int32_t* __restrict__ sids,
                                                              double* __restrict__ sp, uint64_t* __restrict__ tab_key,
                                                              int32_t* __restrict__ tab_s, uint32_t tab_mask) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  double x[3];
  load_xyz(xyz, sids[p], stride, x);
  sp[p * 3] = x[0]; sp[p * 3 + 1] = x[1]; sp[p * 3 + 2] = x[2];
  const GridKey mk = skeys[p];
  const GridKey prev_k = p > 0 ? skeys[p - 1] : 0u, next_k = p < n - 1 ? skeys[p + 1] : 0u;
  const bool c_head = p == 0 || (prev_k >> 3) != (mk >> 3);
  const bool f_head = p == 0 || prev_k != mk;
  const bool c_tail = p == n - 1 || (next_k >> 3) != (mk >> 3);
  if (!f_head && !c_tail) return;
  const int32_t cx = (int32_t)compact10(mk) >> 1, cy = (int32_t)compact10(mk >> 1) >> 1, cz = (int32_t)compact10(mk >> 2) >> 1;
  const uint64_t key = cell_key(cx, cy, cz);
  uint32_t slot = cell_hash(cx, cy, cz) & tab_mask;
  while (true) {
    const unsigned long long prev = atomicCAS((unsigned long long*)&tab_key[slot], (unsigned long long)kEmptyKey, (unsigned long long)key);
    if (prev == kEmptyKey || prev == key) break;
    slot = (slot + 1) & tab_mask;
  }
  int32_t* s = tab_s + (int64_t)slot * kCellStride;
  const int i = (int)(mk & 7u);                    // Morton sub-index of the fine cell: x | y << 1 | z << 2
  if (f_head) {
    const int from = c_head ? 0 : (int)(prev_k & 7u) + 1;
    for (int j = from; j <= i; ++j) s[j] = (int32_t)p;
  }
  if (c_tail) {
    for (int j = i + 1; j <= 8; ++j) s[j] = (int32_t)(p + 1);
    s[11] = (int32_t)(p + 1);
  }
  if (c_head) s[10] = (int32_t)p;
}

// fine[p] = 1 when the coarse cell of sorted point p holds at least fine_min points: the level knn_group_kernel searches for it
__global__ __launch_bounds__(kBlock) void cell_level_kernel(int64_t n, const GridKey* __restrict__ skeys, const uint64_t* __restrict__ tab_key,
                                                            const int32_t* __restrict__ tab_s, uint32_t tab_mask, int fine_min,
                                                            uint8_t* __restrict__ fine) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  const GridKey mk = skeys[p];
  const int32_t cx = (int32_t)compact10(mk) >> 1, cy = (int32_t)compact10(mk >> 1) >> 1, cz = (int32_t)compact10(mk >> 2) >> 1;
  const uint64_t key = cell_key(cx, cy, cz);
  uint32_t slot = cell_hash(cx, cy, cz) & tab_mask;
  while (tab_key[slot] != key) slot = (slot + 1) & tab_mask;
  const int2 be = *reinterpret_cast<const int2*>(tab_s + (int64_t)slot * kCellStride + 10);
  fine[p] = be.y - be.x >= fine_min ? 1 : 0;
}

static uint32_t table_size(int64_t n) {
  uint64_t s = 1024;
  while (s < (uint64_t)(2 * n + 2)) s <<= 1;             // at most n coarse cells: load factor <= 1/2
  return (uint32_t)s;
}

// [b, e) of sorted positions of cell (x, y, z) of `level` (0: coarse, 1: fine); false when its coarse cell is empty
__device__ __forceinline__ bool find_cell(const CellTable& t, int32_t x, int32_t y, int32_t z, int32_t* b, int32_t* e, int level = 0) {
  const int32_t cx = x >> level, cy = y >> level, cz = z >> level;
  const uint64_t key = cell_key(cx, cy, cz);
  uint32_t slot = cell_hash(cx, cy, cz) & t.mask;
  const int i = level ? ((x & 1) | ((y & 1) << 1) | ((z & 1) << 2)) : 10;       // the payload pair wanted: s[i], s[i+1]
  // key and payload of the first slot are requested together
  uint64_t k = t.key[slot];
  const int32_t* s = t.s + (int64_t)slot * kCellStride + i;
  int32_t vb = s[0], ve = s[1];
  while (k != key) {
    if (k == kEmptyKey) return false;
    slot = (slot + 1) & t.mask;                    // another cell's key: linear probing (rare at load factor <= 1/2)
    k = t.key[slot];
    s = t.s + (int64_t)slot * kCellStride + i;
    vb = s[0]; ve = s[1];
  }
  *b = vb; *e = ve;
  return true;
}

// cKDTree's squared distance: products and sums individually rounded, in axis order.
__device__ __forceinline__ double sqdist(const double* a, const double* b) {
  const double d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
  double s = __dmul_rn(d0, d0);
  s = __dadd_rn(s, __dmul_rn(d1, d1));
  s = __dadd_rn(s, __dmul_rn(d2, d2));
  return s;
}

// Visit the cells of the cube shell at Chebyshev distance r around cell c.
template <typename F>
__device__ __forceinline__ void for_shell(const Grid& g, const int32_t* c, int r, F&& f) {
  const int z0 = max(c[2] - r, 0), z1 = min(c[2] + r, g.dim[2] - 1);
  const int y0 = max(c[1] - r, 0), y1 = min(c[1] + r, g.dim[1] - 1);
  const int x0 = max(c[0] - r, 0), x1 = min(c[0] + r, g.dim[0] - 1);
  for (int z = z0; z <= z1; ++z) {
    const bool zf = (z == c[2] - r) || (z == c[2] + r);
    for (int y = y0; y <= y1; ++y) {
      const bool yf = zf || (y == c[1] - r) || (y == c[1] + r);
      if (yf) {
        for (int x = x0; x <= x1; ++x) f(x, y, z);
      } else {
        if (c[0] - r >= 0) f(c[0] - r, y, z);
        if (r > 0 && c[0] + r <= g.dim[0] - 1) f(c[0] + r, y, z);
      }
    }
  }
}

__device__ __forceinline__ bool shell_in_grid(const Grid& g, const int32_t* c, int r) {
  return c[0] - r >= 0 || c[1] - r >= 0 || c[2] - r >= 0 || c[0] + r < g.dim[0] || c[1] + r < g.dim[1] ||
         c[2] + r < g.dim[2];
}

// Lower bound on the distance from q (in cell c) to any point outside the shells 0..r.
__device__ __forceinline__ double shell_bound(const Grid& g, const double* q, const int32_t* c, int r) {
  double fmin_ = INFINITY;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double lo = g.origin[a] + (double)c[a] * g.h, hi = lo + g.h;
    fmin_ = fmin(fmin_, fmin(q[a] - lo, hi - q[a]));
  }
  if (!(fmin_ > 0.0)) fmin_ = 0.0;
  const double b = (double)r * g.h + fmin_ - 1e-7 * g.h;      // guard against cell-assignment rounding
  return b > 0.0 ? b : 0.0;
}

// ---- k nearest neighbours ---------------------------------------------------------------------------
constexpr int kKnnQueue = 3;             // accepted candidates a lane holds before the wavefront merges its queues
template <int KMAX>
__global__ __launch_bounds__(kBlock, KMAX <= 16 ? 4 : 1) void knn_query_kernel(const double* __restrict__ sp, const int32_t* __restrict__ sids,
                                                           const double* __restrict__ queries,
                                                           const int32_t* __restrict__ qids, int64_t n_query,
                                                           const Grid* __restrict__ gp, CellTable tab, int k, double r_max,
                                                           int64_t n_points, int r_exhaust,
                                                           int32_t* __restrict__ idx_out, double* __restrict__ dist_out,
                                                           int r_budget, int32_t* __restrict__ pending, int32_t* __restrict__ n_pending,
                                                           int64_t* __restrict__ idx64_out) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= n_query) return;
  const Grid g = *gp;
  const double q[3] = {queries[t * 3], queries[t * 3 + 1], queries[t * 3 + 2]};
  const int64_t row = qids ? qids[t] : t;
  double bd[KMAX];
  int32_t bi[KMAX];
#pragma unroll
  for (int s = 0; s < KMAX; ++s) { bd[s] = INFINITY; bi[s] = 0x7fffffff; }
  const double ub2 = r_max > 0.0 ? r_max * r_max : INFINITY;
  double worst_d = INFINITY;
  int32_t worst_i = 0x7fffffff;
  int32_t c[3];
  cell_of(g, q, c);
  const bool finite_q = isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2]);
  // The sorted insertion by (distance, index) -- cKDTree's order -- costs ~10 instructions per slot and runs for the whole
  // wavefront whenever ONE lane accepts a candidate, i.e. at nearly every step.  So an accepted candidate first goes into a small
  // per-lane queue (a shift register of kKnnQueue entries), and the queues are merged into the sorted lists only when some lane's
  // queue is full -- every lane then merges whatever it holds -- or a shell ends: the number of insertion passes a wavefront
  // executes falls from (steps at which any lane accepts) to about (accepts of its busiest lane).  The acceptance test uses
  // the k-th best as of the last merge, which is never smaller than the current one: nothing is rejected wrongly.
  double qd[kKnnQueue];
  int32_t qi[kKnnQueue];
  int qn = 0;
#pragma unroll
  for (int s = 0; s < kKnnQueue; ++s) { qd[s] = INFINITY; qi[s] = 0x7fffffff; }
  auto insert = [&](double d, int32_t id) {
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
      const bool lt = (d < bd[s]) || (d == bd[s] && id < bi[s]);
      const double td = bd[s];
      const int32_t ti = bi[s];
      bd[s] = lt ? d : td; bi[s] = lt ? id : ti;
      d = lt ? td : d; id = lt ? ti : id;
    }
  };
  auto merge = [&]() {                      // (every lane that reaches this: queue slots beyond qn hold +inf and fall through)
#pragma unroll
    for (int s = 0; s < kKnnQueue; ++s) {
      if (__any((int)(s < qn))) insert(qd[s], qi[s]);
      qd[s] = INFINITY; qi[s] = 0x7fffffff;
    }
    qn = 0;
#pragma unroll
    for (int s = 0; s < KMAX; ++s) if (s == k - 1) { worst_d = bd[s]; worst_i = bi[s]; }
  };
  auto offer = [&](double d, int32_t id) {
    const bool acc = d < ub2 && (d < worst_d || (d == worst_d && id < worst_i));
    if (acc) {
#pragma unroll
      for (int s = kKnnQueue - 1; s > 0; --s) { qd[s] = qd[s - 1]; qi[s] = qi[s - 1]; }
      qd[0] = d; qi[0] = id;
      ++qn;
    }
    if (__any((int)(qn == kKnnQueue))) merge();
  };
  // (the scan of all points, a rare path, inserts directly: another inlined copy of the queue logic would cost registers)
  auto offer_direct = [&](double d, int32_t id) {
    if (!(d < ub2)) return;
    if (!(d < worst_d || (d == worst_d && id < worst_i))) return;
    insert(d, id);
#pragma unroll
    for (int s = 0; s < KMAX; ++s) if (s == k - 1) { worst_d = bd[s]; worst_i = bi[s]; }
  };
  auto consider = [&](int32_t p) {
    const double pp[3] = {sp[(int64_t)p * 3], sp[(int64_t)p * 3 + 1], sp[(int64_t)p * 3 + 2]};
    offer_direct(sqdist(pp, q), sids[p]);
  };
  bool exhaustive = false;
  for (int r = 0; finite_q && shell_in_grid(g, c, r); ++r) {
    // A query that r_budget shells do not settle sits in a sparse region (lidar density falls with the square of the range) and
    // would walk hundreds of mostly empty cells alone while the 63 other lanes of its wavefront idle: it goes on the pending
    // list, and knn_tail_kernel finishes it with a whole wavefront (64 cells probed per trip).
    if (pending && r > r_budget) {
      // one atomic per wavefront; its unsettled queries stay next to each other on the list (they are neighbours in space)
      const uint64_t m = __ballot(1);
      const int lane = threadIdx.x & (kWave - 1);
      const int leader = __ffsll((long long)m) - 1;
      int32_t base = 0;
      if (lane == leader) base = atomicAdd(n_pending, (int32_t)__popcll(m));
      base = __shfl(base, leader, kWave);
      pending[base + (int32_t)__popcll(m & ((1ull << lane) - 1ull))] = (int32_t)t;
      return;
    }
    if (r > r_exhaust) { exhaustive = true; break; }
    // the cube of cells at Chebyshev distance <= r minus its interior (visited by the earlier shells), as one loop
    // nest with ONE inlined copy of the candidate code: for_shell's three call sites tripled it, and with it the
    // register count (143 -> 104 VGPRs at 10 slots, 3 -> 4 wavefronts per SIMD for a latency-bound kernel)
    const int z0 = max(c[2] - r, 0), z1 = min(c[2] + r, g.dim[2] - 1);
    const int y0 = max(c[1] - r, 0), y1 = min(c[1] + r, g.dim[1] - 1);
    const int x0 = max(c[0] - r, 0), x1 = min(c[0] + r, g.dim[0] - 1);
    for (int z = z0; z <= z1; ++z)
      for (int y = y0; y <= y1; ++y) {
        // cells of this row that belong to the shell: the whole row on a face, else only its two ends
        const bool face = abs(z - c[2]) == r || abs(y - c[1]) == r;
        int xs, cnt, step;
        if (face) { xs = x0; cnt = x1 - x0 + 1; step = 1; }
        else {
          const bool lo_in = c[0] - r >= 0, hi_in = c[0] + r <= g.dim[0] - 1;
          xs = lo_in ? c[0] - r : c[0] + r;
          cnt = (lo_in ? 1 : 0) + ((hi_in && r > 0) ? 1 : 0);
          step = 2 * r;
        }
        // Four cells per trip: their hash probes, then their (begin, end) loads, are in flight together -- a lane that
        // walks hundreds of mostly empty cells (sparse regions, the tail of small clouds) otherwise pays one dependent
        // load latency per probe.  The candidate code below stays a single inlined copy (runtime loop over the hits).
        for (int j0 = 0; j0 < cnt; j0 += 4) {
          uint64_t key[4], got[4];
          uint32_t slot[4];
#pragma unroll
          for (int u_ = 0; u_ < 4; ++u_) {
            key[u_] = cell_key(xs + (j0 + u_) * step, y, z);
            slot[u_] = cell_hash(xs + (j0 + u_) * step, y, z) & tab.mask;
            got[u_] = (j0 + u_ < cnt) ? tab.key[slot[u_]] : kEmptyKey;
          }
          int32_t bb[4], ee[4];
#pragma unroll
          for (int u_ = 0; u_ < 4; ++u_) {
            while (got[u_] != key[u_] && got[u_] != kEmptyKey) {      // collision: linear probing (rare)
              slot[u_] = (slot[u_] + 1) & tab.mask;
              got[u_] = tab.key[slot[u_]];
            }
            const bool hit = got[u_] == key[u_];
            bb[u_] = hit ? tab.s[(int64_t)slot[u_] * kCellStride + 10] : 0;
            ee[u_] = hit ? tab.s[(int64_t)slot[u_] * kCellStride + 11] : 0;
          }
#pragma unroll 1
          for (int u_ = 0; u_ < 4; ++u_) {
            const int32_t b = u_ == 0 ? bb[0] : (u_ == 1 ? bb[1] : (u_ == 2 ? bb[2] : bb[3]));
            const int32_t e = u_ == 0 ? ee[0] : (u_ == 1 ? ee[1] : (u_ == 2 ? ee[2] : ee[3]));
            if (b >= e) continue;
            // the next point's coordinates are requested before the current one is offered
            double cx = sp[(int64_t)b * 3], cy = sp[(int64_t)b * 3 + 1], cz = sp[(int64_t)b * 3 + 2];
            int32_t cid = sids[b];
            for (int32_t p = b; p < e; ++p) {
              const int32_t pn = p + 1 < e ? p + 1 : p;
              const double nx = sp[(int64_t)pn * 3], ny = sp[(int64_t)pn * 3 + 1], nz = sp[(int64_t)pn * 3 + 2];
              const int32_t nid = sids[pn];
              const double pp[3] = {cx, cy, cz};
              offer(sqdist(pp, q), cid);
              cx = nx; cy = ny; cz = nz; cid = nid;
            }
          }
        }
      }
    if (qn > 0) merge();
    const double bound = shell_bound(g, q, c, r);
    const double b2 = bound * bound;
    if (worst_d < b2) break;              // k-th best is closer than anything unvisited
    if (b2 >= ub2) break;                 // everything unvisited is beyond the radius
  }
  if (exhaustive) {
    // A query far from the bulk of the cloud (an outlier) would need thousands of empty shells, whose cell count grows
    // with the square of the radius; past r_exhaust shells a plain scan of all points is cheaper.  Start over: the
    // scan sees every point once.
#pragma unroll
    for (int s = 0; s < KMAX; ++s) { bd[s] = INFINITY; bi[s] = 0x7fffffff; }
#pragma unroll
    for (int s = 0; s < kKnnQueue; ++s) { qd[s] = INFINITY; qi[s] = 0x7fffffff; }
    qn = 0;
    worst_d = INFINITY; worst_i = 0x7fffffff;
    int64_t p = 0;
    for (; p + 4 <= n_points; p += 4) {                     // four points' loads in flight per trip
      double pp[4][3];
      int32_t id[4];
#pragma unroll
      for (int u_ = 0; u_ < 4; ++u_) {
        pp[u_][0] = sp[(p + u_) * 3]; pp[u_][1] = sp[(p + u_) * 3 + 1]; pp[u_][2] = sp[(p + u_) * 3 + 2];
        id[u_] = sids[p + u_];
      }
#pragma unroll 1
      for (int u_ = 0; u_ < 4; ++u_) offer_direct(sqdist(pp[u_], q), id[u_]);
    }
    for (; p < n_points; ++p) consider((int32_t)p);
  }
#pragma unroll
  for (int s = 0; s < KMAX; ++s) {
    if (s < k) {
      const bool ok = bi[s] != 0x7fffffff;
      idx_out[row * k + s] = ok ? bi[s] : -1;
      if (idx64_out) idx64_out[row * k + s] = ok ? (int64_t)bi[s] : -1;
      if (dist_out) dist_out[row * k + s] = ok ? sqrt(bd[s]) : INFINITY;
    }
  }
}

// ---- the pending queries of knn_query_kernel: ONE WAVEFRONT PER QUERY --------------------------------------------------------
// Lane l probes cell l, l + 64, ... of the current shell (hash probe -> its points -> squared distances, cKDTree's arithmetic);
// candidates that beat the current k-th best are appended to a per-wavefront LDS buffer (ballot + prefix count), and after every
// shell (or when the buffer fills) the k smallest of {current best, buffer} by (distance, index) are extracted by k rounds of a
// wave-wide lexicographic minimum -- the order knn_query_kernel's sorted insertion produces, so indices and distances are
// bit-identical.  Same termination rule, same fall-back to a scan of all points beyond r_exhaust shells.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ bool lex_less(double ad, int32_t ai, double bd, int32_t bi) { return ad < bd || (ad == bd && ai < bi); }
// one step of a row-wide (16 lanes) lexicographic minimum of (distance, index) over DPP
template <int CTRL>
__device__ __forceinline__ void lex_min_dpp(double& md, int32_t& mi) {
  const double od = __hiloint2double(dpp_i32<CTRL>(__double2hiint(md)), dpp_i32<CTRL>(__double2loint(md)));
  const int32_t oi = dpp_i32<CTRL>(mi);
  if (lex_less(od, oi, md, mi)) { md = od; mi = oi; }
}
constexpr int kTailCap = 320;            // candidate entries per wavefront behind the KMAX best ones
__device__ __forceinline__ int udiv_small(int a, int d, float inv_d) {      // a / d for 0 <= a < 2^22, d > 0
  int q = (int)((float)a * inv_d);
  q -= (q * d > a) ? 1 : 0;
  q += ((q + 1) * d <= a) ? 1 : 0;
  return q;
}
// cell j of the cube shell at Chebyshev distance r (side = 2 r + 1): the two z faces first, then the perimeters of the slices between
__device__ __forceinline__ void shell_cell(int r, int side, float inv_side, float inv_ring, int j, int* dx, int* dy, int* dz) {
  if (r == 0) { *dx = *dy = *dz = 0; return; }
  const int face = side * side, ring = 4 * side - 4;
  if (j < 2 * face) {
    const int f = j >= face ? 1 : 0, jj = j - f * face;
    const int y = udiv_small(jj, side, inv_side);
    *dx = jj - y * side - r; *dy = y - r; *dz = f ? r : -r;
    return;
  }
  const int jj = j - 2 * face;
  const int sl = udiv_small(jj, ring, inv_ring), pos = jj - sl * ring;
  *dz = sl + 1 - r;
  if (pos < side) { *dx = pos - r; *dy = -r; }
  else if (pos < 2 * side) { *dx = pos - side - r; *dy = r; }
  else if (pos < 3 * side - 2) { *dx = -r; *dy = pos - 2 * side + 1 - r; }
  else { *dx = r; *dy = pos - (3 * side - 2) + 1 - r; }
}

// LDS hand-over between the lanes of one wavefront: the wavefront's LDS operations execute in program order, so no hardware
// barrier is needed, but the compiler must not move memory operations across the hand-over (wave_barrier alone is IntrNoMem)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

template <int KMAX>
__global__ __launch_bounds__(kBlock) void knn_tail_kernel(const double* __restrict__ sp, const int32_t* __restrict__ sids,
                                                          const double* __restrict__ queries, const int32_t* __restrict__ qids,
                                                          const int32_t* __restrict__ pending, const int32_t* __restrict__ n_pending,
                                                          const Grid* __restrict__ gp, CellTable tab, int k, double r_max, int64_t n_points,
                                                          int r_exhaust, int32_t* __restrict__ idx_out, double* __restrict__ dist_out,
                                                          int64_t* __restrict__ idx64_out, int seeded_stages) {
  __shared__ double s_d[kWavesPerBlock][KMAX + kTailCap];
  __shared__ int32_t s_i[kWavesPerBlock][KMAX + kTailCap];
  __shared__ double s_nd[kWavesPerBlock][KMAX];
  __shared__ int32_t s_ni[kWavesPerBlock][KMAX];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  double* pd = s_d[wave];
  int32_t* pi = s_i[wave];
  const Grid g = *gp;
  const int total_waves = gridDim.x * kWavesPerBlock;
  const int n_pend = *n_pending;
  const double ub2 = r_max > 0.0 ? r_max * r_max : INFINITY;
  for (int wq = blockIdx.x * kWavesPerBlock + wave; wq < n_pend; wq += total_waves) {
    // (knn_group_kernel marks the queries it searched at the fine level by the complement of their number)
    const int32_t tv = pending[wq];
    const bool fine = tv < 0;
    const int64_t t = fine ? (int64_t)~tv : (int64_t)tv;
    const double q[3] = {queries[t * 3], queries[t * 3 + 1], queries[t * 3 + 2]};
    const int64_t row = qids ? qids[t] : t;
    int32_t c[3];
    cell_of(g, q, c);
    for (int s = lane; s < k; s += kWave) { pd[s] = INFINITY; pi[s] = 0x7fffffff; }
    int nc = 0;                                   // candidates behind the k best (wave-uniform)
    double worst_d = INFINITY;
    int32_t worst_i = 0x7fffffff;
    // the k smallest DISTINCT entries of pool [0, k + nc) by (distance, index) -> pool [0, k).  Every lane takes its (up to kPer) entries
    // into registers and orders them; every round then pops the wave-wide minimum of the lanes' heads -- four DPP steps inside the rows
    // of 16, the four row results read lane by lane.  An entry equal to the one before it (a member of the list handed over by
    // knn_group_kernel met again in its cell) is popped without being written.  (Until round 5 every round re-read the pool from
    // LDS and reduced over six ds_bpermute steps: 13 us per selection, 21 of the 37 us of a pending query -- timed on the device.)
    auto select = [&]() {
      const int np = k + nc;
      constexpr int kPer = (KMAX + kTailCap + kWave - 1) / kWave;
      double hd[kPer];
      int32_t hi[kPer];
#pragma unroll
      for (int u = 0; u < kPer; ++u) {
        const int e = lane + u * kWave;
        const bool in = e < np;
        hd[u] = in ? pd[e] : INFINITY;
        hi[u] = in ? pi[e] : 0x7fffffff;
      }
#pragma unroll
      for (int a = 1; a < kPer; ++a) {
#pragma unroll
        for (int b = a; b > 0; --b) {
          if (lex_less(hd[b], hi[b], hd[b - 1], hi[b - 1])) {
            const double td = hd[b]; hd[b] = hd[b - 1]; hd[b - 1] = td;
            const int32_t ti = hi[b]; hi[b] = hi[b - 1]; hi[b - 1] = ti;
          }
        }
      }
      double last_d = -1.0;
      int32_t last_i = -1;
      int s = 0;
      for (int round = 0; round < 2 * k + 2 && s < k; ++round) {
        double md = hd[0];
        int32_t mi = hi[0];
        lex_min_dpp<0xB1>(md, mi);
        lex_min_dpp<0x4E>(md, mi);
        lex_min_dpp<0x124>(md, mi);
        lex_min_dpp<0x128>(md, mi);
        double gd = __shfl(md, 0, kWave);
        int32_t gi = __shfl(mi, 0, kWave);
#pragma unroll
        for (int r = 1; r < kWave / 16; ++r) {
          const double od = __shfl(md, 16 * r, kWave);
          const int32_t oi = __shfl(mi, 16 * r, kWave);
          if (lex_less(od, oi, gd, gi)) { gd = od; gi = oi; }
        }
        if (!(gd < INFINITY)) break;               // nothing finite is left: the remaining slots are empty
        if (!(gd == last_d && gi == last_i)) {
          if (lane == 0) { s_nd[wave][s] = gd; s_ni[wave][s] = gi; }
          ++s;
          last_d = gd; last_i = gi;
        }
        if (hd[0] == gd && hi[0] == gi) {          // the lane(s) that held it move on to their next entry
#pragma unroll
          for (int u = 0; u + 1 < kPer; ++u) { hd[u] = hd[u + 1]; hi[u] = hi[u + 1]; }
          hd[kPer - 1] = INFINITY; hi[kPer - 1] = 0x7fffffff;
        }
      }
      for (int s2 = s + lane; s2 < k; s2 += kWave) { s_nd[wave][s2] = INFINITY; s_ni[wave][s2] = 0x7fffffff; }
      wave_sync();
      for (int s = lane; s < k; s += kWave) { pd[s] = s_nd[wave][s]; pi[s] = s_ni[wave][s]; }
      wave_sync();
      worst_d = pd[k - 1]; worst_i = pi[k - 1];
      nc = 0;
    };
    // one candidate per lane (has = false: none): filter, wave-aggregated append, selection when the buffer runs full
    auto offer = [&](bool has, double d, int32_t id) {
      const bool acc = has && d < ub2 && (d < worst_d || (d == worst_d && id < worst_i));
      const unsigned long long m = __ballot(acc);
      if (m == 0ull) return;
      const int pos = k + nc + __popcll(m & ((1ull << lane) - 1ull));
      if (acc) { pd[pos] = d; pi[pos] = id; }
      nc += __popcll(m);
      wave_sync();
      if (nc > kTailCap - kWave) select();
    };
    const bool finite_q = isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2]);
    bool exhaustive = false;
    // seeded_stages > 0: the row of the output tables holds the k best (index, SQUARED distance) knn_group_kernel found in its stages
    // without settling the query.  A query searched at the coarse level goes on with the first shell those stages did not cover;
    // one searched at the fine level (half-size cells: less ground covered) walks the coarse shells from its own cell again, the
    // list only as the threshold that keeps nearly every candidate out of the pool -- its members met again are dropped by select().
    int r_first = 0;
    if (seeded_stages > 0) {
      for (int s = lane; s < k; s += kWave) {
        const int32_t id = idx_out[row * k + s];
        pd[s] = id >= 0 ? dist_out[row * k + s] : INFINITY;
        pi[s] = id >= 0 ? id : 0x7fffffff;
      }
      wave_sync();
      worst_d = pd[k - 1]; worst_i = pi[k - 1];
      r_first = fine ? 0 : seeded_stages + 1;
    }
    for (int r = r_first; finite_q && shell_in_grid(g, c, r); ++r) {
      if (r > r_exhaust) { exhaustive = true; break; }
      const int side = 2 * r + 1, ring = 4 * side - 4;
      const int n_cells = r == 0 ? 1 : 2 * side * side + (side - 2) * ring;
      const float inv_side = 1.0f / (float)side, inv_ring = r == 0 ? 1.0f : 1.0f / (float)ring;
      for (int j0 = 0; j0 < n_cells; j0 += kWave) {
        const int j = j0 + lane;
        int32_t b = 0, e = 0;
        if (j < n_cells) {
          int dx, dy, dz;
          shell_cell(r, side, inv_side, inv_ring, j, &dx, &dy, &dz);
          const int x = c[0] + dx, y = c[1] + dy, z = c[2] + dz;
          if (x >= 0 && y >= 0 && z >= 0 && x < g.dim[0] && y < g.dim[1] && z < g.dim[2]) {
            if (!find_cell(tab, x, y, z, &b, &e)) { b = 0; e = 0; }
          }
        }
        for (int32_t p = b; __any((int)(p < e)); ++p) {
          const bool has = p < e;
          const int32_t pc = has ? p : 0;
          const double pp[3] = {sp[(int64_t)pc * 3], sp[(int64_t)pc * 3 + 1], sp[(int64_t)pc * 3 + 2]};
          offer(has, sqdist(pp, q), sids[pc]);
        }
      }
      if (nc > 0) select();
      const double bound = shell_bound(g, q, c, r);
      const double b2 = bound * bound;
      if (worst_d < b2) break;
      if (b2 >= ub2) break;
    }
    if (exhaustive) {
      for (int s = lane; s < k; s += kWave) { pd[s] = INFINITY; pi[s] = 0x7fffffff; }
      nc = 0; worst_d = INFINITY; worst_i = 0x7fffffff;
      wave_sync();
      for (int64_t p0 = 0; p0 < n_points; p0 += kWave) {
        const int64_t p = p0 + lane;
        const bool has = p < n_points;
        const int64_t pc = has ? p : 0;
        const double pp[3] = {sp[pc * 3], sp[pc * 3 + 1], sp[pc * 3 + 2]};
        offer(has, sqdist(pp, q), sids[pc]);
      }
      if (nc > 0) select();
    }
    wave_sync();
    for (int s = lane; s < k; s += kWave) {
      const bool ok = pi[s] != 0x7fffffff;
      idx_out[row * k + s] = ok ? pi[s] : -1;
      if (idx64_out) idx64_out[row * k + s] = ok ? (int64_t)pi[s] : -1;
      if (dist_out) dist_out[row * k + s] = ok ? sqrt(pd[s]) : INFINITY;
    }
    wave_sync();
  }
}

// ---- k nearest neighbours, SIXTEEN LANES PER QUERY (k <= 16) ----------------------------------------------------------------
// knn_query_kernel gives every query one lane: a wavefront then walks 27 (+98) cells one after the other and each cell's points
// one after the other -- some 250 dependent memory round trips per wavefront, with the trip count of every loop set by the
// busiest of 64 lanes -- and one unsettled query in 64 sends all of them through the next shell.  A 200 k-point scan is less
// than one wavefront per SIMD: its build time WAS that chain of round trips.  Here a query owns one DPP row:
//   * its 16 lanes probe 16 cells of the current stage at once (stage 1: the 27 cells around the query's own; stage r >= 2: the
//     shell at Chebyshev distance r);
//   * the points of those 16 cells are numbered through and taken 16 per step: candidate jj lies in the cell of lane
//     L = #{lanes whose inclusive point count <= jj}, found without a search -- every lane drops a one into bin
//     (inclusive count - step base) of a 16-bin LDS histogram and the running sum of the bins up to a lane's own is its L --
//     so the trip count is the row's candidate count / 16, not the sum over cells of the fullest cell;
//   * candidates that beat the current k-th best go to the row's LDS pool (ballot + prefix count); the k best of {best list,
//     pool} by (distance, index) -- cKDTree's order -- are then found by a float32 threshold (every lane sorts its five rounded
//     distances once, k rounds of a row-wide minimum over the heads pop at least k entries: at least k keys are <= the last
//     minimum, and rounding is monotone, so nothing at or below the k-th distance is lost), after which the ~k finalists are
//     ranked, one per lane, by 15 row rotations of their rounded distances -- of (fp64 distance, index) when two share one.
//     More than 16 finalists (points repeated many times) take k rounds of an exact row-wide lexicographic minimum instead.
// The best list lives one slot per lane (lane s of the row = the s-th neighbour).  Same distances (sqdist), same order, same
// termination rule as knn_query_kernel: bit-identical tables.  Queries not settled after r_budget stages go to knn_tail_kernel.
// Measured (k = 10): 200 k-point scan 0.47 -> 0.38 ms, 2 M-point cloud 1.95 -> 1.89 ms per build.
constexpr int kGrp = 16;                           // lanes per query: one DPP row
constexpr int kGrpPerBlock = kBlock / kGrp;
constexpr int kGrpPool = 64;                       // pool entries per query: four per lane in the selection

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) { return __int_as_float(dpp_i32<CTRL>(__float_as_int(v))); }
__device__ __forceinline__ float row_all_min(float m) {
  m = fminf(m, dpp_f32<0xB1>(m));                  // quad_perm [1,0,3,2]
  m = fminf(m, dpp_f32<0x4E>(m));                  // quad_perm [2,3,0,1]
  m = fminf(m, dpp_f32<0x124>(m));                 // row_ror:4
  m = fminf(m, dpp_f32<0x128>(m));                 // row_ror:8
  return m;
}
__device__ __forceinline__ int row_incl_scan(int v) {          // inclusive prefix sum along the row (row_shr shifts zeros in)
  v += dpp_i32<0x111>(v);
  v += dpp_i32<0x112>(v);
  v += dpp_i32<0x114>(v);
  v += dpp_i32<0x118>(v);
  return v;
}
template <int CTRL>
__device__ __forceinline__ int rot_less(double md, int32_t mi) {          // 1 when the entry CTRL lanes away sorts before mine
  const double od = __hiloint2double(dpp_i32<CTRL>(__double2hiint(md)), dpp_i32<CTRL>(__double2loint(md)));
  const int32_t oi = dpp_i32<CTRL>(mi);
  return lex_less(od, oi, md, mi) ? 1 : 0;
}
template <int CTRL>
__device__ __forceinline__ void row_lex_min_step(double& md, int32_t& mi) {
  const double od = __hiloint2double(dpp_i32<CTRL>(__double2hiint(md)), dpp_i32<CTRL>(__double2loint(md)));
  const int32_t oi = dpp_i32<CTRL>(mi);
  if (lex_less(od, oi, md, mi)) { md = od; mi = oi; }
}

// (five wavefronts per SIMD: 96 VGPRs, 8 B of scratch.  Six -- 80 VGPRs, 96 B of scratch -- cost the two-level kernel 10 % at 2 M points,
//  four 14 %)
__global__ __launch_bounds__(kBlock, 5) void knn_group_kernel(const double* __restrict__ sp, const int32_t* __restrict__ sids,
                                                           const double* __restrict__ queries, const int32_t* __restrict__ qids,
                                                           int64_t n_query, const Grid* __restrict__ gp, CellTable tab, int k,
                                                           double r_max, int32_t* __restrict__ idx_out, double* __restrict__ dist_out,
                                                           int r_budget, int32_t* __restrict__ pending, int32_t* __restrict__ n_pending,
                                                           int fine_min, const uint8_t* __restrict__ fine_flag, int64_t* __restrict__ idx64_out) {
  __shared__ int32_t s_hist[kGrpPerBlock][kGrp];
  __shared__ double s_pd[kGrpPerBlock][kGrpPool];
  __shared__ int32_t s_pi[kGrpPerBlock][kGrpPool];
  __shared__ double s_fd[kGrpPerBlock][kGrp];
  __shared__ int32_t s_fi[kGrpPerBlock][kGrp];
  constexpr int32_t kNone = 0x7fffffff;
  const int lane = threadIdx.x & (kWave - 1), sub = lane & (kGrp - 1), grp = threadIdx.x / kGrp;
  // (blocks stay in launch order: giving every XCD one contiguous stretch of the Morton-ordered queries -- xcd_block -- made the
  //  2 M-point build 12 % slower: the stretches differ in density, and the XCD with the crowded one finishes last)
  const int64_t t = (int64_t)blockIdx.x * kGrpPerBlock + grp;
  const bool valid = t < n_query;
  const int64_t tc = valid ? t : 0;
  const Grid g = *gp;
  const double q[3] = {queries[tc * 3], queries[tc * 3 + 1], queries[tc * 3 + 2]};
  const int64_t row = qids ? qids[tc] : tc;
  // the level of the search: fine (cell edge h / 2) where the query's own coarse cell is crowded
  int32_t c[3];
  cell_of_level(g, q, 0, c);
  int lv = 0;
  if (fine_flag) lv = valid ? (int)fine_flag[tc] : 0;          // self search: decided per sorted point by cell_level_kernel
  else if (fine_min != 0x7fffffff && valid) {                   // another cloud's queries: one more probe
    int32_t b0 = 0, e0 = 0;
    if (find_cell(tab, c[0], c[1], c[2], &b0, &e0) && e0 - b0 >= fine_min) lv = 1;
  }
  if (lv) cell_of_level(g, q, 1, c);
  const double h_lv = lv ? 0.5 * g.h : g.h;
  const int32_t dim_lv[3] = {g.dim[0] << lv, g.dim[1] << lv, g.dim[2] << lv};
  // (shell_in_grid / shell_bound at the row's level)
  auto stage_in_grid = [&](int r) {
    return c[0] - r >= 0 || c[1] - r >= 0 || c[2] - r >= 0 || c[0] + r < dim_lv[0] || c[1] + r < dim_lv[1] || c[2] + r < dim_lv[2];
  };
  auto stage_bound = [&](int r) {
    double fmin_ = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double lo = g.origin[a] + (double)c[a] * h_lv, hi = lo + h_lv;
      fmin_ = fmin(fmin_, fmin(q[a] - lo, hi - q[a]));
    }
    if (!(fmin_ > 0.0)) fmin_ = 0.0;
    const double bnd = (double)r * h_lv + fmin_ - 1e-7 * h_lv;      // guard against cell-assignment rounding
    return bnd > 0.0 ? bnd : 0.0;
  };
  const double ub2 = r_max > 0.0 ? r_max * r_max : INFINITY;
  const bool finite_q = isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2]);
  bool done = !(valid && finite_q);                // (all of these are uniform over the row)
  bool unsettled = false;                          // done without an answer: the tail kernel's
  double bd = INFINITY;                            // lane s: the s-th best so far
  int32_t bi = kNone;
  double worst_d = INFINITY;
  int32_t worst_i = kNone;
  int nc = 0;                                      // pool entries (uniform over the row)
  const int row_lane0 = lane & ~(kGrp - 1);

  // ---- the k best of {best list, pool} -> best list ----
  auto select = [&]() {
    double ed[5];
    int32_t ei[5];
    float k0[5], kf[5];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = sub + kGrp * i;
      const bool ok = e < nc;
      ed[i] = ok ? s_pd[grp][e] : INFINITY;
      ei[i] = ok ? s_pi[grp][e] : kNone;
    }
    ed[4] = bd; ei[4] = bi;
#pragma unroll
    for (int i = 0; i < 5; ++i) { k0[i] = ei[i] != kNone ? (float)ed[i] : INFINITY; kf[i] = k0[i]; }
    // every lane sorts its five keys once (nine compare-exchanges); a round then looks at the heads only
#define DC_CE(a, b) { const float lo_ = fminf(kf[a], kf[b]), hi_ = fmaxf(kf[a], kf[b]); kf[a] = lo_; kf[b] = hi_; }
    DC_CE(0, 1) DC_CE(3, 4) DC_CE(2, 4) DC_CE(2, 3) DC_CE(1, 4) DC_CE(0, 3) DC_CE(0, 2) DC_CE(1, 3) DC_CE(1, 2)
#undef DC_CE
    float tau = INFINITY;                    // bound on the k-th smallest rounded distance
    for (int s = 0; s < k; ++s) {
      const float mm = row_all_min(kf[0]);
      tau = mm;
      // lanes whose head is the minimum pop it: every round removes at least one entry, all of them <= the last minimum,
      // so after k rounds at least k keys are <= tau (and tau is at most the k-th distinct key)
      const bool pop = kf[0] == mm;
      kf[0] = pop ? kf[1] : kf[0]; kf[1] = pop ? kf[2] : kf[1]; kf[2] = pop ? kf[3] : kf[2]; kf[3] = pop ? kf[4] : kf[3];
      kf[4] = pop ? INFINITY : kf[4];
    }
    bool fin[5];
    int cf = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) { fin[i] = ei[i] != kNone && k0[i] <= tau; cf += fin[i] ? 1 : 0; }
    const int fincl = row_incl_scan(cf);
    const int nf = __shfl(fincl, row_lane0 + kGrp - 1, kWave);
    if (!__any((int)(nf > kGrp))) {
      int off = fincl - cf;
#pragma unroll
      for (int i = 0; i < 5; ++i) if (fin[i]) { s_fd[grp][off] = ed[i]; s_fi[grp][off] = ei[i]; ++off; }
      wave_sync();
      const bool mine = sub < nf;
      const double md = mine ? s_fd[grp][sub] : INFINITY;
      const int32_t mi = mine ? s_fi[grp][sub] : kNone;
      // rank among the finalists: by the rounded distance (one register per rotation); the exact (fp64, index) comparison
      // only when two finalists of some row share a rounded distance
      const float mk = mine ? (float)md : INFINITY;
      int rank = 0;
      bool tie = false;
#define DC_ROT(C) { const float ok_ = dpp_f32<C>(mk); rank += ok_ < mk ? 1 : 0; tie = tie || (ok_ == mk && mine); }
      DC_ROT(0x121) DC_ROT(0x122) DC_ROT(0x123) DC_ROT(0x124) DC_ROT(0x125) DC_ROT(0x126) DC_ROT(0x127) DC_ROT(0x128)
      DC_ROT(0x129) DC_ROT(0x12A) DC_ROT(0x12B) DC_ROT(0x12C) DC_ROT(0x12D) DC_ROT(0x12E) DC_ROT(0x12F)
#undef DC_ROT
      if (__any((int)tie)) {
        rank = 0;
        rank += rot_less<0x121>(md, mi); rank += rot_less<0x122>(md, mi); rank += rot_less<0x123>(md, mi);
        rank += rot_less<0x124>(md, mi); rank += rot_less<0x125>(md, mi); rank += rot_less<0x126>(md, mi);
        rank += rot_less<0x127>(md, mi); rank += rot_less<0x128>(md, mi); rank += rot_less<0x129>(md, mi);
        rank += rot_less<0x12A>(md, mi); rank += rot_less<0x12B>(md, mi); rank += rot_less<0x12C>(md, mi);
        rank += rot_less<0x12D>(md, mi); rank += rot_less<0x12E>(md, mi); rank += rot_less<0x12F>(md, mi);
      }
      wave_sync();
      if (mine) { s_fd[grp][rank] = md; s_fi[grp][rank] = mi; }
      wave_sync();
      const bool take = mine && sub < k;
      bd = take ? s_fd[grp][sub] : INFINITY;
      bi = take ? s_fi[grp][sub] : kNone;
      wave_sync();
    } else {
      // more finalists than lanes (points repeated many times): k rounds of an exact row-wide minimum
      double last_d = -1.0, nbd = INFINITY;
      int32_t last_i = -1, nbi = kNone;
      for (int s = 0; s < k; ++s) {
        double md = INFINITY;
        int32_t mi = kNone;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const bool gt = ed[i] > last_d || (ed[i] == last_d && ei[i] > last_i);
          if (ei[i] != kNone && gt && lex_less(ed[i], ei[i], md, mi)) { md = ed[i]; mi = ei[i]; }
        }
        row_lex_min_step<0xB1>(md, mi);
        row_lex_min_step<0x4E>(md, mi);
        row_lex_min_step<0x124>(md, mi);
        row_lex_min_step<0x128>(md, mi);
        if (sub == s) { nbd = md; nbi = mi; }
        last_d = md; last_i = mi;
        if (mi == kNone) last_d = INFINITY;                   // nothing is left: the remaining slots stay empty
      }
      bd = nbd; bi = nbi;
    }
    worst_d = __shfl(bd, row_lane0 + k - 1, kWave);
    worst_i = __shfl(bi, row_lane0 + k - 1, kWave);
    nc = 0;
  };

  for (int r = 1; __any((int)!done); ++r) {
    const int side = 2 * r + 1, ring = 4 * side - 4;
    const int n_cells = r == 1 ? 27 : 2 * side * side + (side - 2) * ring;
    const float inv_side = 1.0f / (float)side, inv_ring = 1.0f / (float)ring;
    for (int j0 = 0; j0 < n_cells; j0 += kGrp) {
      // sixteen cells of the stage: one hash probe per lane (a single memory round trip, find_cell)
      const int j = j0 + sub;
      int32_t b = 0, e = 0;
      if (!done && j < n_cells) {
        int dx, dy, dz;
        if (r == 1) { dz = j / 9; const int jj = j - dz * 9; dy = jj / 3; dx = jj - dy * 3; dx -= 1; dy -= 1; dz -= 1; }
        else shell_cell(r, side, inv_side, inv_ring, j, &dx, &dy, &dz);
        const int x = c[0] + dx, y = c[1] + dy, z = c[2] + dz;
        if (x >= 0 && y >= 0 && z >= 0 && x < dim_lv[0] && y < dim_lv[1] && z < dim_lv[2]) {
          if (!find_cell(tab, x, y, z, &b, &e, lv)) { b = 0; e = 0; }
        }
      }
      const int cnt = e - b;
      const int incl = row_incl_scan(cnt);
      const int excl = incl - cnt;
      const int T = __shfl(incl, row_lane0 + kGrp - 1, kWave);       // points in the row's sixteen cells
      // their points numbered through, sixteen per step: candidate jj lies in the cell of lane L = #{lanes: incl <= jj}.  Every
      // lane drops a one into the bin min(incl - s0, .) of a 16-bin LDS histogram; the running sum of the bins up to `sub` is
      // that count for candidate s0 + sub (a load-balanced search without a search).
      for (int s0 = 0; __any((int)(s0 < T)); s0 += kGrp) {
        s_hist[grp][sub] = 0;
        wave_sync();
        const int v = incl - s0;
        if (v < kGrp) atomicAdd(&s_hist[grp][v > 0 ? v : 0], 1);
        wave_sync();
        const int L = row_incl_scan(s_hist[grp][sub]);
        wave_sync();
        const int jj = s0 + sub;
        const bool has = jj < T;
        const int src = row_lane0 + (L & (kGrp - 1));
        // (the shuffles outside the conditional: ds_bpermute returns 0 from lanes that are masked off, and the cell of an active
        //  lane's candidate is often held by a lane with no candidate of its own in this step)
        const int32_t b_src = __shfl(b, src, kWave), excl_src = __shfl(excl, src, kWave);
        const int32_t p = has ? b_src + (jj - excl_src) : 0;
        const double pp[3] = {sp[(int64_t)p * 3], sp[(int64_t)p * 3 + 1], sp[(int64_t)p * 3 + 2]};
        const int32_t id = sids[p];
        const double d = sqdist(pp, q);
        const bool acc = has && d < ub2 && lex_less(d, id, worst_d, worst_i);
        const unsigned long long m = __ballot(acc);
        const unsigned m16 = (unsigned)(m >> row_lane0) & 0xffffu;
        const int pos = nc + __popc(m16 & ((1u << sub) - 1u));
        if (acc) { s_pd[grp][pos] = d; s_pi[grp][pos] = id; }
        nc += __popc(m16);
        wave_sync();
        if (__any((int)(nc > kGrpPool - kGrp))) select();
      }
    }
    if (__any((int)(nc > 0))) select();
    if (!done) {
      const double bound = stage_bound(r);
      const double b2 = bound * bound;
      if (worst_d < b2 || b2 >= ub2 || !stage_in_grid(r + 1)) done = true;       // settled
      else if (r >= r_budget) { done = true; unsettled = true; }
    }
  }
  // unsettled queries: one entry per row, one atomic per wavefront
  const bool pend = unsettled && sub == 0;
  const unsigned long long pm = __ballot(pend);
  if (pm != 0ull) {
    const int leader = __ffsll((long long)pm) - 1;
    int32_t base = 0;
    if (lane == leader) base = atomicAdd(n_pending, (int32_t)__popcll(pm));
    base = __shfl(base, leader, kWave);
    if (pend) pending[base + (int32_t)__popcll(pm & ((1ull << lane) - 1ull))] = lv ? (int32_t)~t : (int32_t)t;     // (< 0: fine level)
  }
  // an unsettled row hands its best list to knn_tail_kernel through its rows of the output tables: indices as they will stand,
  // distances SQUARED (the tail kernel writes the roots); without a distance table the tail kernel starts from nothing
  if (valid && unsettled && dist_out && sub < k) {
    const bool ok = bi != kNone;
    idx_out[row * k + sub] = ok ? bi : -1;
    dist_out[row * k + sub] = ok ? bd : INFINITY;
  }
  if (valid && !unsettled && sub < k) {
    const bool ok = bi != kNone;
    idx_out[row * k + sub] = ok ? bi : -1;
    if (idx64_out) idx64_out[row * k + sub] = ok ? (int64_t)bi : -1;
    if (dist_out) dist_out[row * k + sub] = ok ? sqrt(bd) : INFINITY;
  }
}

// ---- scan-shadow filter straight off the direction grid (filters.py:257-309 after depth_cloud.py:352-360) ---------------
// The reference builds the padded table of every point's direction-neighbours (all rays within `rad` chord length of its own)
// and then takes min / max over the table's rows of the angle between the ray back to the viewpoint and the vector to the
// neighbour.  The mask only depends on the SET of neighbours (the fill value of short rows, the mean of the bounds, never
// violates them), so the walk over the grid cells evaluates the angles as it meets the neighbours and no table is written:
// one pass instead of count + fill + row sort + mask.  The inclusion test is radius_kernel's (fp64, on the fp64 copies of the
// directions), the angle arithmetic is shadow_mask_kernel's (dc_filters.hip), in the cloud's precision.
// SIXTEEN LANES PER RAY, knn_group_kernel's division of labour (a lane-per-ray walk of a 200 k-point scan is one wavefront per
// SIMD following ~100 dependent loads: 0.23 ms): 16 cells probed at once, their points dealt 16 per step.
template <int CTRL> __device__ __forceinline__ float dpp_val_f(float v) { return dpp_f32<CTRL>(v); }
template <int CTRL> __device__ __forceinline__ double dpp_val_d(double v) {
  return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}
template <int CTRL> __device__ __forceinline__ float dpp_any(float v) { return dpp_val_f<CTRL>(v); }
template <int CTRL> __device__ __forceinline__ double dpp_any(double v) { return dpp_val_d<CTRL>(v); }
template <typename T>
__device__ __forceinline__ void row_min_max(T& lo, T& hi) {          // all lanes of the row end up with the row's min / max
#define DC_STEP(C) { const T ol = dpp_any<C>(lo), oh = dpp_any<C>(hi); lo = ol < lo ? ol : lo; hi = oh > hi ? oh : hi; }
  DC_STEP(0xB1) DC_STEP(0x4E) DC_STEP(0x124) DC_STEP(0x128)
#undef DC_STEP
}

template <typename T>
__global__ __launch_bounds__(kBlock) void shadow_group_kernel(const double* __restrict__ sp, const int32_t* __restrict__ sids,
                                                              int64_t n, const Grid* __restrict__ gp, CellTable tab, double rad,
                                                              const T* __restrict__ x, const T* __restrict__ vps, int vps_rows,
                                                              T lo, T hi, uint8_t* __restrict__ mask) {
#pragma clang fp contract(off)
  __shared__ int32_t s_hist[kGrpPerBlock][kGrp];
  const int lane = threadIdx.x & (kWave - 1), sub = lane & (kGrp - 1), grp = threadIdx.x / kGrp;
  const int row_lane0 = lane & ~(kGrp - 1);
  const int64_t t = (int64_t)blockIdx.x * kGrpPerBlock + grp;
  const bool valid = t < n;
  const int64_t tc = valid ? t : 0;
  const Grid g = *gp;
  const double q[3] = {sp[tc * 3], sp[tc * 3 + 1], sp[tc * 3 + 2]};
  const int64_t i = sids[tc];
  const double r2 = rad * rad;
  const T eps = (T)1e-8;
  const T xi0 = x[i * 3], xi1 = x[i * 3 + 1], xi2 = x[i * 3 + 2];
  const T* o = vps + (vps_rows == 1 ? 0 : i * 3);
  T a0 = o[0] - xi0, a1 = o[1] - xi1, a2 = o[2] - xi2;
  const T na = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
  const T da = na > eps ? na : eps;
  a0 /= da; a1 /= da; a2 /= da;
  // The extreme ANGLES are the arc cosines of the extreme COSINES (acos does not increase anywhere): the walk keeps the largest and the
  // smallest cosine and takes two arc cosines per ray at the end instead of one per neighbour (a fifth of the kernel's instructions,
  // which it issues at 0.86 of the VALU rate).  A cosine an ulp outside [-1, 1] or a NaN -- whose arc cosine is NaN in the reference's
  // row -- removes the ray as before.
  T cmax = -(T)INFINITY, cmin = (T)INFINITY;
  bool bad = false;
  int32_t c[3];
  cell_of(g, q, c);
  const bool finite_q = isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2]);
  bool done = !(valid && finite_q && shell_in_grid(g, c, 0));
  for (int r = 1; __any((int)!done); ++r) {
    const int side = 2 * r + 1, ring = 4 * side - 4;
    const int n_cells = r == 1 ? 27 : 2 * side * side + (side - 2) * ring;
    const float inv_side = 1.0f / (float)side, inv_ring = 1.0f / (float)ring;
    for (int j0 = 0; j0 < n_cells; j0 += kGrp) {
      const int j = j0 + sub;
      int32_t b = 0, e = 0;
      if (!done && j < n_cells) {
        int dx, dy, dz;
        if (r == 1) { dz = j / 9; const int jj = j - dz * 9; dy = jj / 3; dx = jj - dy * 3; dx -= 1; dy -= 1; dz -= 1; }
        else shell_cell(r, side, inv_side, inv_ring, j, &dx, &dy, &dz);
        const int cx = c[0] + dx, cy = c[1] + dy, cz = c[2] + dz;
        if (cx >= 0 && cy >= 0 && cz >= 0 && cx < g.dim[0] && cy < g.dim[1] && cz < g.dim[2]) {
          if (!find_cell(tab, cx, cy, cz, &b, &e)) { b = 0; e = 0; }
        }
      }
      const int cnt = e - b;
      const int incl = row_incl_scan(cnt);
      const int excl = incl - cnt;
      const int Tn = __shfl(incl, row_lane0 + kGrp - 1, kWave);
      for (int s0 = 0; __any((int)(s0 < Tn)); s0 += kGrp) {        // (see knn_group_kernel)
        s_hist[grp][sub] = 0;
        wave_sync();
        const int v = incl - s0;
        if (v < kGrp) atomicAdd(&s_hist[grp][v > 0 ? v : 0], 1);
        wave_sync();
        const int L = row_incl_scan(s_hist[grp][sub]);
        wave_sync();
        const int jj = s0 + sub;
        const bool has = jj < Tn;
        const int src = row_lane0 + (L & (kGrp - 1));
        const int32_t b_src = __shfl(b, src, kWave), excl_src = __shfl(excl, src, kWave);
        const int32_t p = has ? b_src + (jj - excl_src) : 0;
        const double pp[3] = {sp[(int64_t)p * 3], sp[(int64_t)p * 3 + 1], sp[(int64_t)p * 3 + 2]};
        const int64_t jn = sids[p];
        if (has && sqdist(pp, q) <= r2) {
          T b0 = x[jn * 3] - xi0, b1 = x[jn * 3 + 1] - xi1, b2 = x[jn * 3 + 2] - xi2;
          const T nb = sqrt(b0 * b0 + b1 * b1 + b2 * b2);
          const T db = nb > eps ? nb : eps;
          b0 /= db; b1 /= db; b2 /= db;
          const T cs = a0 * b0 + a1 * b1 + a2 * b2;
          bad = bad || !(cs >= (T)-1 && cs <= (T)1);
          cmax = cs > cmax ? cs : cmax;
          cmin = cs < cmin ? cs : cmin;
        }
      }
    }
    if (!done && (shell_bound(g, q, c, r) > rad || !shell_in_grid(g, c, r + 1))) done = true;
  }
  row_min_max(cmin, cmax);
  const bool none = cmin > cmax;                     // no neighbour met
  const T amin = none ? (T)INFINITY : (T)acos(cmax), amax = none ? -(T)INFINITY : (T)acos(cmin);
  const unsigned long long bm = __ballot(bad);
  const bool row_bad = ((unsigned)(bm >> row_lane0) & 0xffffu) != 0u;
  // a ray with no neighbour at all (not even itself: non-finite direction) has an all-fill row in the reference: kept
  if (valid && sub == 0) mask[i] = (!row_bad && (amin > amax ? lo <= hi : (amin >= lo && amax <= hi))) ? 1 : 0;
}

// ---- radius search: count, then fill (ascending index, -1 padded) ------------------------------------
// Rows are put into ascending order (cKDTree's query_ball_point returns sorted lists) by radius_sort_rows_kernel: ONE WAVEFRONT
// per row, a bitonic network over the row padded to a power of two in the wavefront's part of LDS (-1 padding sorts last as
// INT_MAX).  Round 3 sorted one row per LANE by insertion -- a chain of dependent LDS round trips per element, and rows of more
// than 192 entries by insertion straight into the global row: ball neighbourhoods of r = 0.4 m on 0.2 m voxels have 280, and
// their fill pass took 27 ms of a 28 ms search.
constexpr int kRadiusSortMax = 4096;     // longest row the network sorts (16 KB of LDS per wavefront); longer ones: insertion in place
template <int N>
__global__ __launch_bounds__(kBlock) void radius_sort_rows_kernel(int32_t* __restrict__ idx, int64_t n, int kmax) {
  __shared__ int32_t s_all[kWavesPerBlock * N];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int64_t rowi = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (rowi >= n) return;
  int32_t* s = s_all + wave * N;
  int32_t* row = idx + rowi * kmax;
  int mine = 0;
  for (int t = lane; t < N; t += kWave) {
    const int32_t id = t < kmax ? row[t] : -1;
    s[t] = id < 0 ? 0x7fffffff : id;
    mine += id >= 0 ? 1 : 0;
  }
  // the network of THIS row: the power of two that holds its entries (the template argument is the longest row of the table:
  // at r = 0.4 m on 0.2 m voxels 290, i.e. 512, while most rows fit 256 -- 45 stages over 256 pairs against 36 over 128)
  for (int off = kWave / 2; off > 0; off >>= 1) mine += __shfl_xor(mine, off, kWave);
  const int len = __builtin_amdgcn_readfirstlane(mine);
  int nr = kWave;
  while (nr < len) nr <<= 1;
  wave_sync();
#pragma unroll 1
  for (int k = 2; k <= nr; k <<= 1) {
#pragma unroll 1
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < nr / 2; t += kWave) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));       // pair t: i and i | j
        const int32_t a = s[i], c = s[i | j];
        const bool up = (i & k) == 0;
        if ((a > c) == up) { s[i] = c; s[i | j] = a; }
      }
      wave_sync();
    }
  }
  for (int t = lane; t < kmax; t += kWave) {
    const int32_t id = s[t];
    row[t] = id == 0x7fffffff ? -1 : id;
  }
}
static int launch_radius_sort(int32_t* idx, int64_t n, int kmax, hipStream_t stream) {
  const dim3 grid((unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock)), block(kBlock);
#define SORT_N(N) hipLaunchKernelGGL((radius_sort_rows_kernel<N>), grid, block, 0, stream, idx, n, kmax)
  if (kmax <= 64) SORT_N(64); else if (kmax <= 128) SORT_N(128); else if (kmax <= 256) SORT_N(256); else if (kmax <= 512) SORT_N(512);
  else if (kmax <= 1024) SORT_N(1024); else if (kmax <= 2048) SORT_N(2048); else SORT_N(4096);
#undef SORT_N
  return (int)hipGetLastError();
}
// queries / qids: the query points and their output rows (self search: the sorted points and their original indices; another
// cloud: its points in fp64, qids == nullptr -> row t)
template <bool FILL>
__global__ __launch_bounds__(kBlock) void radius_kernel(const double* __restrict__ sp, const int32_t* __restrict__ sids,
                                                        const double* __restrict__ queries, const int32_t* __restrict__ qids,
                                                        int64_t n, const Grid* __restrict__ gp, CellTable tab, double rad,
                                                        int32_t* __restrict__ count, int32_t* __restrict__ idx_out,
                                                        int kmax) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= n) return;
  const Grid g = *gp;
  const double q[3] = {queries[t * 3], queries[t * 3 + 1], queries[t * 3 + 2]};
  const int64_t row = qids ? qids[t] : t;
  const double r2 = rad * rad;
  int32_t c[3];
  cell_of(g, q, c);
  int32_t cnt = 0;
  // FILL: hits are appended to the row in the order the cells are visited; radius_sort_rows_kernel then puts every row into
  // ascending order inside LDS (the shifting insertion straight into the global row cost 0.42 ms per 200 k-point scan, three
  // times the counting pass).  Rows longer than kRadiusSortMax are still built by insertion in place.
  const bool in_lds = FILL && kmax <= kRadiusSortMax;      // append now, rows sorted by radius_sort_rows_kernel afterwards
  int32_t* out = FILL ? idx_out + row * kmax : nullptr;
  const bool finite_q = isfinite(q[0]) && isfinite(q[1]) && isfinite(q[2]);
  for (int r = 0; finite_q && shell_in_grid(g, c, r); ++r) {
    for_shell(g, c, r, [&](int x, int y, int z) {
      int32_t b, e;
      if (!find_cell(tab, x, y, z, &b, &e)) return;
      for (int32_t p = b; p < e; ++p) {
        const double pp[3] = {sp[(int64_t)p * 3], sp[(int64_t)p * 3 + 1], sp[(int64_t)p * 3 + 2]};
        if (sqdist(pp, q) <= r2) {
          if (FILL) {
            const int32_t id = sids[p];
            int32_t s = cnt;
            if (in_lds) {
              out[s] = id;
            } else {
              while (s > 0 && out[s - 1] > id) { out[s] = out[s - 1]; --s; }
              out[s] = id;
            }
          }
          ++cnt;
        }
      }
    });
    const double bound = shell_bound(g, q, c, r);
    if (bound > rad) break;
  }
  if (FILL) {
    for (int32_t s = cnt; s < kmax; ++s) out[s] = -1;
  } else {
    count[row] = cnt;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void to_f64_kernel(const T* __restrict__ xyz, int stride, int64_t n,
                                                        double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  double p[3];
  load_xyz(xyz, i, stride, p);
  out[i * 3] = p[0]; out[i * 3 + 1] = p[1]; out[i * 3 + 2] = p[2];
}

// ---- transposed neighbour list ------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void edge_keys_kernel(const int32_t* __restrict__ nbr, int64_t n_edges, int k,
                                                           int32_t n, uint32_t* __restrict__ keys,
                                                           int32_t* __restrict__ src) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n_edges) return;
  const int32_t j = nbr[e];
  keys[e] = (j >= 0 && j < n) ? (uint32_t)j : (uint32_t)n;      // missing neighbours sort to the end
  src[e] = (int32_t)(e / k);
}

__global__ __launch_bounds__(kBlock) void csr_ptr_kernel(const uint32_t* __restrict__ skeys, int64_t n_edges, int32_t n,
                                                         int32_t* __restrict__ ptr) {
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j > n) return;
  int64_t lo = 0, hi = n_edges;             // first position with key >= j
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (skeys[mid] < (uint32_t)j) lo = mid + 1; else hi = mid;
  }
  ptr[j] = (int32_t)lo;
}

__global__ __launch_bounds__(kBlock) void max_i32_kernel(const int32_t* __restrict__ v, int64_t n, int32_t* __restrict__ out) {
  int32_t m = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) m = max(m, v[i]);
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

constexpr int kBoxBlocks = 256;

struct GridWs {
  double* part; Grid* grid; GridKey* keys; GridKey* skeys; int32_t* ids; int32_t* sids; double* sp;
  uint64_t* tab_key; int32_t* tab_s; uint8_t* fine; void* sort_tmp; size_t sort_bytes; uint32_t tab_n;
  double* qf64; int32_t* pending; int32_t* n_pending; size_t total;
};

static GridWs carve_grid(void* ws, int64_t n, int64_t n_query_extra) {
  GridWs g;
  Carver c(ws);
  g.part = c.take<double>(kBoxBlocks * kBoxVals);
  g.grid = c.take<Grid>(1);
  g.keys = c.take<GridKey>(n);
  g.skeys = c.take<GridKey>(n);
  g.ids = c.take<int32_t>(n);
  g.sids = c.take<int32_t>(n);
  g.sp = c.take<double>(3 * n);
  g.tab_n = table_size(n);
  g.tab_key = c.take<uint64_t>(g.tab_n);
  g.tab_s = c.take<int32_t>((size_t)g.tab_n * kCellStride);
  g.fine = c.take<uint8_t>(n);
  g.qf64 = c.take<double>(3 * n_query_extra);
  g.pending = c.take<int32_t>(n > n_query_extra ? n : n_query_extra);      // queries the query kernels hand to knn_tail_kernel
  g.n_pending = c.take<int32_t>(16);
  static_assert(sizeof(GridKey) == 4, "the grid keys sort as 32-bit words");
  g.sort_bytes = sort_pairs_bytes((size_t)(n > 0 ? n : 1), 32);
  g.sort_tmp = c.take<char>(g.sort_bytes);
  g.total = c.off + 256;
  return g;
}


template <typename T>
static int build_grid(const T* xyz, int stride, int64_t n, int k, double cell_hint, GridWs& w, hipStream_t st) {
  const unsigned nb = (unsigned)((n + kBlock - 1) / kBlock);
  // (the box in at most kBoxBlocks partial rows, four points per lane: every block of cell_keys_kernel re-reads them)
  const int64_t want_box = (n + 4 * kBlock - 1) / (4 * kBlock);
  const int n_box = (int)(want_box < 1 ? 1 : (want_box > kBoxBlocks ? kBoxBlocks : want_box));
  hipLaunchKernelGGL((bbox_partial_kernel<T>), dim3(n_box), dim3(kBlock), 0, st, xyz, stride, n, w.part);
  // (one scan: every key block derives the grid itself, a launch saved; from half a million points on the set-up kernel is the cheaper way)
  const bool own_setup = n <= 500000;
  if (!own_setup) hipLaunchKernelGGL(grid_setup_kernel, dim3(1), dim3(kBlock), 0, st, (const double*)w.part, n_box, n, k, cell_hint, w.grid);
  hipLaunchKernelGGL((cell_keys_kernel<T>), dim3(nb), dim3(kBlock), 0, st, xyz, stride, n, (const double*)w.part, own_setup ? n_box : 0, k, cell_hint,
                     w.grid, w.keys, w.ids, w.tab_key, w.tab_n, w.n_pending);
  DC_HIP(sort_pairs_u32(w.sort_tmp, w.sort_bytes, w.keys, w.skeys, w.ids, w.sids, (size_t)n, 0, kGridKeyBits, st));
  hipLaunchKernelGGL((sorted_cells_kernel<T>), dim3(nb), dim3(kBlock), 0, st, xyz, stride, n, w.skeys, w.sids, w.sp, w.tab_key, w.tab_s,
                     w.tab_n - 1);
  DC_HIP(hipGetLastError());
  return DC_OK;
}

static std::atomic<int> g_knn_fine_min{14};      // dc_knn_set_fine_cell_count: points in a query's coarse cell from which it searches the fine level
constexpr int kKnnBudgetAuto = 1000;   // dc_knn_set_shell_budget value for "by size": 2 stages below a million queries, 4 above
static std::atomic<int> g_knn_budget{kKnnBudgetAuto};        // dc_knn_set_shell_budget

// sixteen lanes per query (knn_group_kernel, both grid levels) or one (knn_query_kernel, coarse level)?
static bool knn_rows_per_query(int k) {
  const int raw = g_knn_budget.load();
  return k <= 16 && (raw == kKnnBudgetAuto || (raw >= 1 && raw < 100));
}

static int launch_knn(int k, const double* sp, const int32_t* sids, int64_t n, const double* q, const int32_t* qids, int64_t nq,
                      const Grid* g, CellTable tab, double r, int32_t* idx, double* dist, int32_t* pending, int32_t* n_pending,
                      const uint8_t* fine_flag, int64_t* idx64, hipStream_t st) {
  const dim3 grid((unsigned)((nq + kBlock - 1) / kBlock)), block(kBlock);
  // shells 0..R hold (2R+1)^3 cells at ~4 candidates' worth of work each; a scan of all n points costs n candidates
  int r_exhaust = (int)(cbrt((double)n * 0.25) * 0.5);
  r_exhaust = r_exhaust < 4 ? 4 : (r_exhaust > 64 ? 64 : r_exhaust);
  // dc_knn_set_shell_budget(b): b < 0 one lane per query to the end (round 2); 0 <= b < 100: stages before the tail kernel,
  // sixteen lanes per query when k <= 16 (knn_group_kernel); b >= 100: the same with b - 100 and ONE lane per query (A-B).
  int raw = g_knn_budget.load();
  // measured (k = 10): one 200 k-point scan 0.39 / 0.40 / 0.41 ms with 2 / 3 / 4 stages before the tail kernel, the 2 M-point cloud
  // 1.90 / 1.87 / 1.84 ms (its tail queries are few and each costs the tail kernel a wavefront's restart)
  if (raw == kKnnBudgetAuto) raw = nq >= 1000000 ? 4 : 2;
  const int budget = raw >= 100 ? raw - 100 : raw;
  if (budget < 0) pending = nullptr;
  // (n_pending was zeroed by cell_keys_kernel of the grid build that precedes every call)
  int seeded = 0;            // stages whose best list knn_group_kernel hands to knn_tail_kernel (through the distance table)
  if (pending && knn_rows_per_query(k)) {
    seeded = (dist && budget > 0) ? budget : 0;
    const dim3 ggrid((unsigned)((nq + kGrpPerBlock - 1) / kGrpPerBlock));
    hipLaunchKernelGGL(knn_group_kernel, ggrid, block, 0, st, sp, sids, q, qids, nq, g, tab, k, r, idx, dist, budget, pending, n_pending,
                       g_knn_fine_min.load(), fine_flag, idx64);
  } else {
#define LK(KM) hipLaunchKernelGGL((knn_query_kernel<KM>), grid, block, 0, st, sp, sids, q, qids, nq, g, tab, k, r, n, r_exhaust, idx, dist, \
                                  budget, pending, n_pending, idx64)
    // the sorted insertion costs ~12 instructions per slot and runs for a whole wavefront whenever one lane accepts a
    // candidate, so the slot count follows k closely (10 = the reference's default nn_k)
    if (k <= 4) LK(4); else if (k <= 8) LK(8); else if (k <= 10) LK(10); else if (k <= 16) LK(16); else if (k <= 32) LK(32); else LK(64);
#undef LK
  }
  if (pending) {
    // one wavefront per pending query; their number stays on the device (the grid is fixed, wavefronts stride over the list)
    const int64_t want = (nq + kWavesPerBlock - 1) / kWavesPerBlock;
    const dim3 tgrid((unsigned)(want < 2048 ? (want < 1 ? 1 : want) : 2048));
#define LT(KM) hipLaunchKernelGGL((knn_tail_kernel<KM>), tgrid, block, 0, st, sp, sids, q, qids, pending, n_pending, g, tab, k, r, n, r_exhaust, idx, dist, idx64, seeded)
    if (k <= 16) LT(16); else LT(64);
#undef LT
  }
  DC_HIP(hipGetLastError());
  return DC_OK;
}


// ---- several scans in one build (the set-up's local feature clouds, preproc.py:35-64 per scan) ---------------------------------
// The scans are set side by side on a lattice -- scan s shifted by an integer offset per axis, two box widths apart -- so that one
// k-NN build serves all of them.  The result equals the per-scan builds bit for bit when every shifted coordinate is exact in fp64
// (every difference, distance and tie is then what it was) and no neighbourhood crosses scans; both are checked, not assumed.
constexpr int kLatticeScans = 64;

// One block: the box of all scans -> the pitch per axis (an integer: 2 x extent + 1, rounded up) -> the lattice (nx, ny, nz) whose
// longest side is shortest -> the offset of every scan.  info |= 1 when the box is not finite.
__global__ void lattice_setup_kernel(const double* __restrict__ part, int n_part, int n_scans, double* __restrict__ offsets,
                                     int32_t* __restrict__ info) {
  __shared__ double tot[kBoxVals];
  combine_box_partials(part, n_part, tot);
  if (threadIdx.x != 0) return;
  double pitch[3];
  bool ok = true;
  for (int a = 0; a < 3; ++a) {
    pitch[a] = ceil(2.0 * (tot[3 + a] - tot[a]) + 1.0);
    ok = ok && isfinite(pitch[a]) && pitch[a] >= 1.0 && pitch[a] < 1e9;
  }
  if (!ok) { atomicOr(info, 1); pitch[0] = pitch[1] = pitch[2] = 1.0; }
  int best[3] = {n_scans, 1, 1};
  double best_side = INFINITY;
  for (int nx = 1; nx <= n_scans; ++nx) {
    for (int ny = 1; nx * (ny - 1) < n_scans; ++ny) {
      const int nz = (n_scans + nx * ny - 1) / (nx * ny);
      const double side = fmax(fmax(nx * pitch[0], ny * pitch[1]), nz * pitch[2]);
      if (side < best_side) { best_side = side; best[0] = nx; best[1] = ny; best[2] = nz; }
    }
  }
  for (int s = 0; s < n_scans; ++s) {
    offsets[3 * s + 0] = (double)(s % best[0]) * pitch[0];
    offsets[3 * s + 1] = (double)((s / best[0]) % best[1]) * pitch[1];
    offsets[3 * s + 2] = (double)(s / (best[0] * best[1])) * pitch[2];
  }
}

__device__ __forceinline__ int scan_of_row(const int64_t* __restrict__ scan_ptr, int n_scans, int64_t i) {
  int lo = 0, hi = n_scans;                       // scan_ptr[lo] <= i < scan_ptr[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (scan_ptr[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void lattice_shift_kernel(const T* __restrict__ xyz, int64_t n, const int64_t* __restrict__ scan_ptr,
                                                               int n_scans, const double* __restrict__ offsets,
                                                               double* __restrict__ shifted, int32_t* __restrict__ info) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int s = scan_of_row(scan_ptr, n_scans, i);
  bool exact = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double x = (double)xyz[i * 3 + a], o = offsets[3 * s + a];
    const double y = x + o;
    exact = exact && (y - o == x);                // also false for NaN
    shifted[i * 3 + a] = y;
  }
  if (!exact) atomicOr(info, 1);
}

// Row i of the table built over the shifted scans -> indices inside its own scan; info |= 2 when a neighbour lies in another scan.
__global__ __launch_bounds__(kBlock) void lattice_localize_kernel(int32_t* __restrict__ nbr, int64_t n, int k, const int64_t* __restrict__ scan_ptr,
                                                                  int n_scans, int32_t* __restrict__ info) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n * k) return;
  const int64_t i = e / k;
  const int s = scan_of_row(scan_ptr, n_scans, i);
  const int32_t v = nbr[e];
  if (v < 0) return;
  const int64_t b = scan_ptr[s], en = scan_ptr[s + 1];
  if (v < b || v >= en) { atomicOr(info, 2); return; }
  nbr[e] = (int32_t)(v - b);
}

// Bounding box of a cloud: out[0..2] = min, out[3..5] = max per axis (non-finite coordinates are skipped by the partial kernel's
// min / max; an empty cloud gives +inf / -inf).
__global__ void extent_finish_kernel(const double* __restrict__ part, int n_part, double* __restrict__ out) {
  __shared__ double tot[kBoxVals];
  combine_box_partials(part, n_part, tot);
  if (threadIdx.x < 6) out[threadIdx.x] = tot[threadIdx.x];
}

// out[i] = the scan of row i (scan_ptr: [n_scans + 1] ascending row offsets)
__global__ __launch_bounds__(kBlock) void scan_ids_kernel(const int64_t* __restrict__ scan_ptr, int n_scans, int64_t n, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = scan_of_row(scan_ptr, n_scans, i);
}

}  // namespace dc

using namespace dc;

extern "C" {

// (A-B switches: refused unless the process asked for them with DC_ENABLE_ABLATIONS=1, like dc_set_option)
static bool knn_ablations_enabled() {
  static const bool enabled = [] { const char* e = getenv("DC_ENABLE_ABLATIONS"); return e && atoi(e) != 0; }();
  return enabled;
}
int dc_knn_set_shell_budget(int shells) { if (!knn_ablations_enabled()) return DC_ERR_UNSUPPORTED; g_knn_budget.store(shells); return DC_OK; }
int dc_knn_set_fine_cell_count(int points) {       // default 14
  if (!knn_ablations_enabled()) return DC_ERR_UNSUPPORTED;
  g_knn_fine_min.store(points < 1 ? 0x7fffffff : points);
  return DC_OK;
}

size_t dc_knn_workspace_bytes(int64_t n, int64_t n_query) {
  if (n < 0 || n_query < 0) return 0;
  return carve_grid(nullptr, n, n_query).total;
}

// Self k-NN (query == NULL) or cross-cloud k-NN of `query` in `points`.
int dc_knn_build(const void* points, int stride, int dtype, int64_t n, const void* query, int q_stride, int64_t n_query,
                 int k, double r, double cell_hint, int32_t* idx_out, double* dist_out, void* ws, size_t ws_bytes,
                 hipStream_t stream) {
  return dc_knn_build_i64(points, stride, dtype, n, query, q_stride, n_query, k, r, cell_hint, idx_out, nullptr, dist_out, ws, ws_bytes, stream);
}

// The same, with the table written a second time as int64 (the index type of the reference's tensors, nearest_neighbors.py:78): the
// conversion pass over the table -- a launch of its own per build -- folds into the kernels' stores.
int dc_knn_build_i64(const void* points, int stride, int dtype, int64_t n, const void* query, int q_stride, int64_t n_query,
                     int k, double r, double cell_hint, int32_t* idx_out, int64_t* idx64_out, double* dist_out, void* ws, size_t ws_bytes,
                     hipStream_t stream) {
  if (n == 0 && !query) return DC_OK;
  if (!points || n < 0 || k < 1 || k > 64 || !idx_out || !ws || stride < 3) return DC_ERR_ARG;
  if (n >= (int64_t)0x7fffffff) return DC_ERR_UNSUPPORTED;
  if (query && (n_query < 0 || q_stride < 3)) return DC_ERR_ARG;
  const int64_t nq = query ? n_query : 0;
  GridWs w = carve_grid(ws, n, nq);
  if (ws_bytes < w.total) return DC_ERR_WORKSPACE;
  if (n == 0) {
    if (query && n_query > 0) {
      DC_HIP(hipMemsetAsync(idx_out, 0xff, (size_t)n_query * k * sizeof(int32_t), stream));
      if (idx64_out) DC_HIP(hipMemsetAsync(idx64_out, 0xff, (size_t)n_query * k * sizeof(int64_t), stream));
      // dist = inf is not a byte pattern; callers treat idx = -1 as authoritative
    }
    return DC_OK;
  }
  int rc;
  if (dtype == DC_F32) rc = build_grid((const float*)points, stride, n, k, cell_hint, w, stream);
  else if (dtype == DC_F64) rc = build_grid((const double*)points, stride, n, k, cell_hint, w, stream);
  else return DC_ERR_DTYPE;
  if (rc) return rc;
  CellTable tab{w.tab_key, w.tab_s, w.tab_n - 1};
  if (!query) {
    const uint8_t* flags = nullptr;
    const int fine_min = g_knn_fine_min.load();
    if (knn_rows_per_query(k) && fine_min != 0x7fffffff) {          // the level every sorted point searches at (it is its own query)
      hipLaunchKernelGGL(cell_level_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, n, w.skeys, w.tab_key, w.tab_s,
                         w.tab_n - 1, fine_min, w.fine);
      flags = w.fine;
    }
    return launch_knn(k, w.sp, w.sids, n, w.sp, w.sids, n, w.grid, tab, r, idx_out, dist_out, w.pending, w.n_pending, flags, idx64_out, stream);
  }
  if (n_query == 0) return DC_OK;
  const dim3 grid((unsigned)((n_query + kBlock - 1) / kBlock)), block(kBlock);
  if (dtype == DC_F32) hipLaunchKernelGGL((to_f64_kernel<float>), grid, block, 0, stream, (const float*)query, q_stride, n_query, w.qf64);
  else hipLaunchKernelGGL((to_f64_kernel<double>), grid, block, 0, stream, (const double*)query, q_stride, n_query, w.qf64);
  return launch_knn(k, w.sp, w.sids, n, w.qf64, nullptr, n_query, w.grid, tab, r, idx_out, dist_out, w.pending, w.n_pending, nullptr, idx64_out, stream);
}

// Radius search, pass 1: per-point neighbour counts and their maximum (device scalars).
int dc_radius_count(const void* points, int stride, int dtype, int64_t n, double r, int32_t* count_out,
                    int32_t* kmax_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!points || n < 0 || !(r > 0.0) || !count_out || !kmax_out || !ws || stride < 3) return DC_ERR_ARG;
  if (n >= (int64_t)0x7fffffff) return DC_ERR_UNSUPPORTED;
  GridWs w = carve_grid(ws, n, 0);
  if (ws_bytes < w.total) return DC_ERR_WORKSPACE;
  DC_HIP(hipMemsetAsync(kmax_out, 0, sizeof(int32_t), stream));
  if (n == 0) return DC_OK;
  int rc;
  if (dtype == DC_F32) rc = build_grid((const float*)points, stride, n, 0, r, w, stream);
  else if (dtype == DC_F64) rc = build_grid((const double*)points, stride, n, 0, r, w, stream);
  else return DC_ERR_DTYPE;
  if (rc) return rc;
  CellTable tab{w.tab_key, w.tab_s, w.tab_n - 1};
  const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
  hipLaunchKernelGGL((radius_kernel<false>), grid, block, 0, stream, w.sp, w.sids, w.sp, w.sids, n, w.grid, tab, r, count_out, nullptr, 0);
  hipLaunchKernelGGL(max_i32_kernel, dim3(256), block, 0, stream, count_out, n, kmax_out);
  DC_HIP(hipGetLastError());
  return DC_OK;
}

// Scan-shadow mask without the direction-neighbour table: grid over `dirs` (cell edge r), one walk.  mask_out uint8 [n].
int dc_shadow_filter(const void* points, const void* vps, int vps_rows, const void* dirs, int dtype, int64_t n, double r, double lo,
                     double hi, uint8_t* mask_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n == 0) return DC_OK;
  if (!points || !vps || !dirs || n < 0 || !(r > 0.0) || !mask_out || !ws || (vps_rows != 1 && vps_rows != n)) return DC_ERR_ARG;
  if (n >= (int64_t)0x7fffffff) return DC_ERR_UNSUPPORTED;
  GridWs w = carve_grid(ws, n, 0);
  if (ws_bytes < w.total) return DC_ERR_WORKSPACE;
  int rc;
  if (dtype == DC_F32) rc = build_grid((const float*)dirs, 3, n, 0, r, w, stream);
  else if (dtype == DC_F64) rc = build_grid((const double*)dirs, 3, n, 0, r, w, stream);
  else return DC_ERR_DTYPE;
  if (rc) return rc;
  CellTable tab{w.tab_key, w.tab_s, w.tab_n - 1};
  const dim3 grid((unsigned)((n + kGrpPerBlock - 1) / kGrpPerBlock)), block(kBlock);
  if (dtype == DC_F32)
    hipLaunchKernelGGL((shadow_group_kernel<float>), grid, block, 0, stream, w.sp, w.sids, n, w.grid, tab, r, (const float*)points,
                       (const float*)vps, vps_rows, (float)lo, (float)hi, mask_out);
  else
    hipLaunchKernelGGL((shadow_group_kernel<double>), grid, block, 0, stream, w.sp, w.sids, n, w.grid, tab, r, (const double*)points,
                       (const double*)vps, vps_rows, lo, hi, mask_out);
  DC_HIP(hipGetLastError());
  return DC_OK;
}

// The same for the points of ANOTHER cloud as queries (nearest_neighbors.py:50-51 accepts any query): counts per query row.
int dc_radius_count_query(const void* points, int stride, int dtype, int64_t n, const void* query, int q_stride, int64_t n_query,
                          double r, int32_t* count_out, int32_t* kmax_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!points || n < 0 || n_query < 0 || !(r > 0.0) || !kmax_out || !ws || stride < 3 || q_stride < 3) return DC_ERR_ARG;
  if (n_query > 0 && (!query || !count_out)) return DC_ERR_ARG;
  if (n >= (int64_t)0x7fffffff || n_query >= (int64_t)0x7fffffff) return DC_ERR_UNSUPPORTED;
  GridWs w = carve_grid(ws, n, n_query);
  if (ws_bytes < w.total) return DC_ERR_WORKSPACE;
  DC_HIP(hipMemsetAsync(kmax_out, 0, sizeof(int32_t), stream));
  if (n_query == 0) return DC_OK;
  if (n == 0) return (int)hipMemsetAsync(count_out, 0, (size_t)n_query * sizeof(int32_t), stream);
  int rc;
  if (dtype == DC_F32) rc = build_grid((const float*)points, stride, n, 0, r, w, stream);
  else if (dtype == DC_F64) rc = build_grid((const double*)points, stride, n, 0, r, w, stream);
  else return DC_ERR_DTYPE;
  if (rc) return rc;
  CellTable tab{w.tab_key, w.tab_s, w.tab_n - 1};
  const dim3 grid((unsigned)((n_query + kBlock - 1) / kBlock)), block(kBlock);
  if (dtype == DC_F32) hipLaunchKernelGGL((to_f64_kernel<float>), grid, block, 0, stream, (const float*)query, q_stride, n_query, w.qf64);
  else hipLaunchKernelGGL((to_f64_kernel<double>), grid, block, 0, stream, (const double*)query, q_stride, n_query, w.qf64);
  hipLaunchKernelGGL((radius_kernel<false>), grid, block, 0, stream, w.sp, w.sids, w.qf64, (const int32_t*)nullptr, n_query, w.grid, tab, r,
                     count_out, nullptr, 0);
  hipLaunchKernelGGL(max_i32_kernel, dim3(256), block, 0, stream, count_out, n_query, kmax_out);
  DC_HIP(hipGetLastError());
  return DC_OK;
}

// pass 2 of dc_radius_count_query (grid and fp64 queries still in `ws`)
int dc_radius_fill_query(int64_t n, int64_t n_query, double r, int kmax, int32_t* idx_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n < 0 || n_query < 0 || !(r > 0.0) || kmax < 1 || !idx_out || !ws) return DC_ERR_ARG;
  GridWs w = carve_grid(ws, n, n_query);
  if (ws_bytes < w.total) return DC_ERR_WORKSPACE;
  if (n_query == 0) return DC_OK;
  if (n == 0) return (int)hipMemsetAsync(idx_out, 0xff, (size_t)n_query * kmax * sizeof(int32_t), stream);
  CellTable tab{w.tab_key, w.tab_s, w.tab_n - 1};
  const dim3 grid((unsigned)((n_query + kBlock - 1) / kBlock)), block(kBlock);
  hipLaunchKernelGGL((radius_kernel<true>), grid, block, 0, stream, w.sp, w.sids, w.qf64, (const int32_t*)nullptr, n_query, w.grid, tab, r,
                     nullptr, idx_out, kmax);
  DC_HIP(hipGetLastError());
  if (kmax <= kRadiusSortMax && kmax > 1) { const int rc = launch_radius_sort(idx_out, n_query, kmax, stream); if (rc) return rc; }
  return DC_OK;
}

// Radius search, pass 2: the grid of pass 1 must still be in `ws` (same points, same r, same stream).
int dc_radius_fill(int64_t n, double r, int kmax, int32_t* idx_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n < 0 || !(r > 0.0) || kmax < 1 || !idx_out || !ws) return DC_ERR_ARG;
  GridWs w = carve_grid(ws, n, 0);
  if (ws_bytes < w.total) return DC_ERR_WORKSPACE;
  if (n == 0) return DC_OK;
  CellTable tab{w.tab_key, w.tab_s, w.tab_n - 1};
  const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
  hipLaunchKernelGGL((radius_kernel<true>), grid, block, 0, stream, w.sp, w.sids, w.sp, w.sids, n, w.grid, tab, r, nullptr, idx_out, kmax);
  DC_HIP(hipGetLastError());
  if (kmax <= kRadiusSortMax && kmax > 1) { const int rc = launch_radius_sort(idx_out, n, kmax, stream); if (rc) return rc; }
  return DC_OK;
}

size_t dc_knn_transpose_workspace_bytes(int64_t n, int k) {
  if (n < 0 || k < 1) return 0;
  const int64_t ne = n * k;
  Carver c(nullptr);
  c.take<uint32_t>(ne); c.take<uint32_t>(ne); c.take<int32_t>(ne);
  const size_t sb = sort_pairs_bytes((size_t)(ne > 0 ? ne : 1), 32);
  c.take<char>(sb);
  return c.off + 256;
}

// csr_ptr[N+1], csr_src[N*K]: for every point j the ascending list of centres i whose neighbourhood contains j.
int dc_knn_transpose(const int32_t* nbr, int64_t n, int k, int64_t n_dst, int32_t* csr_ptr, int32_t* csr_src, void* ws,
                     size_t ws_bytes, hipStream_t stream) {
  // n rows of k neighbour indices into [0, n_dst); n_dst <= 0 means n_dst = n (the square self-neighbourhood case)
  if (n_dst <= 0) n_dst = n;
  if (n == 0 && csr_ptr) return (int)hipMemsetAsync(csr_ptr, 0, (size_t)(n_dst + 1) * sizeof(int32_t), stream);
  if (!nbr || n < 0 || k < 1 || !csr_ptr || !csr_src || !ws) return DC_ERR_ARG;
  const int64_t ne = n * k;
  if (ne >= (int64_t)0x7fffffff) return DC_ERR_UNSUPPORTED;
  if (ws_bytes < dc_knn_transpose_workspace_bytes(n, k)) return DC_ERR_WORKSPACE;
  if (n_dst >= (int64_t)0x7fffffff) return DC_ERR_UNSUPPORTED;
  Carver c(ws);
  uint32_t* keys = c.take<uint32_t>(ne);
  uint32_t* skeys = c.take<uint32_t>(ne);
  int32_t* src = c.take<int32_t>(ne);
  const size_t sb = sort_pairs_bytes((size_t)ne, 32);
  void* tmp = c.take<char>(sb);
  int bits = 1;
  while (((int64_t)1 << bits) <= n_dst) ++bits;
  const dim3 block(kBlock);
  hipLaunchKernelGGL(edge_keys_kernel, dim3((unsigned)((ne + kBlock - 1) / kBlock)), block, 0, stream, nbr, ne, k, (int32_t)n_dst, keys, src);
  DC_HIP(sort_pairs_u32(tmp, sb, keys, skeys, src, csr_src, (size_t)ne, 0, (unsigned)bits, stream));
  hipLaunchKernelGGL(csr_ptr_kernel, dim3((unsigned)((n_dst + 1 + kBlock - 1) / kBlock)), block, 0, stream, skeys, ne, (int32_t)n_dst, csr_ptr);
  DC_HIP(hipGetLastError());
  return DC_OK;
}

size_t dc_spatial_order_workspace_bytes(int64_t n) {
  if (n < 0) return 0;
  Carver c(nullptr);
  c.take<double>(kBoxBlocks * kBoxVals); c.take<double>(4); c.take<uint64_t>(n); c.take<uint64_t>(n); c.take<int32_t>(n);
  const size_t sb = sort_pairs_bytes((size_t)(n > 0 ? n : 1), 64);
  c.take<char>(sb);
  return c.off + 256;
}

// order_out[p] = index of the point at position p of the Morton (Z-curve) order over the bounding box.
int dc_spatial_order(const void* points, int stride, int dtype, int64_t n, int32_t* order_out, void* ws, size_t ws_bytes,
                     hipStream_t stream) {
  if (n == 0) return DC_OK;
  if (!points || n < 0 || !order_out || !ws || stride < 3) return DC_ERR_ARG;
  if (n >= (int64_t)0x7fffffff) return DC_ERR_UNSUPPORTED;
  if (ws_bytes < dc_spatial_order_workspace_bytes(n)) return DC_ERR_WORKSPACE;
  if (n == 0) return DC_OK;
  Carver c(ws);
  double* part = c.take<double>(kBoxBlocks * kBoxVals);
  double* box = c.take<double>(4);
  uint64_t* keys = c.take<uint64_t>(n);
  uint64_t* skeys = c.take<uint64_t>(n);
  int32_t* ids = c.take<int32_t>(n);
  const size_t sb = sort_pairs_bytes((size_t)n, 64);
  void* tmp = c.take<char>(sb);
  const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
  if (dtype == DC_F32) {
    hipLaunchKernelGGL((bbox_partial_kernel<float>), dim3(kBoxBlocks), block, 0, stream, (const float*)points, stride, n, part);
    hipLaunchKernelGGL(box_finish_kernel, dim3(1), dim3(kBlock), 0, stream, part, kBoxBlocks, box);
    hipLaunchKernelGGL((fine_keys_kernel<float>), grid, block, 0, stream, (const float*)points, stride, n, box, keys, ids);
  } else if (dtype == DC_F64) {
    hipLaunchKernelGGL((bbox_partial_kernel<double>), dim3(kBoxBlocks), block, 0, stream, (const double*)points, stride, n, part);
    hipLaunchKernelGGL(box_finish_kernel, dim3(1), dim3(kBlock), 0, stream, part, kBoxBlocks, box);
    hipLaunchKernelGGL((fine_keys_kernel<double>), grid, block, 0, stream, (const double*)points, stride, n, box, keys, ids);
  } else return DC_ERR_DTYPE;
  DC_HIP(sort_pairs_u64(tmp, sb, keys, skeys, ids, order_out, (size_t)n, 0, 63, stream));
  DC_HIP(hipGetLastError());
  return DC_OK;
}

// ---- several scans in one k-NN build ---------------------------------------------------------------------------------------------
size_t dc_scan_lattice_workspace_bytes(int n_scans) { return (size_t)(kBoxBlocks * kBoxVals + 3 * (n_scans > 0 ? n_scans : 1)) * sizeof(double); }

int dc_scan_lattice_shift(const void* points, int dtype, int64_t n, const int64_t* scan_ptr, int n_scans, double* shifted, int32_t* info,
                          void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!points || n < 1 || !scan_ptr || n_scans < 1 || n_scans > kLatticeScans || !shifted || !info || !ws) return DC_ERR_ARG;
  if (ws_bytes < dc_scan_lattice_workspace_bytes(n_scans)) return DC_ERR_WORKSPACE;
  double* part = (double*)ws;
  double* offsets = part + kBoxBlocks * kBoxVals;
  const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
  if (dtype == DC_F32) hipLaunchKernelGGL((bbox_partial_kernel<float>), dim3(kBoxBlocks), block, 0, stream, (const float*)points, 3, n, part);
  else if (dtype == DC_F64) hipLaunchKernelGGL((bbox_partial_kernel<double>), dim3(kBoxBlocks), block, 0, stream, (const double*)points, 3, n, part);
  else return DC_ERR_DTYPE;
  hipLaunchKernelGGL(lattice_setup_kernel, dim3(1), block, 0, stream, part, kBoxBlocks, n_scans, offsets, info);
  if (dtype == DC_F32)
    hipLaunchKernelGGL((lattice_shift_kernel<float>), grid, block, 0, stream, (const float*)points, n, scan_ptr, n_scans, offsets, shifted, info);
  else
    hipLaunchKernelGGL((lattice_shift_kernel<double>), grid, block, 0, stream, (const double*)points, n, scan_ptr, n_scans, offsets, shifted, info);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? DC_OK : (int)err;
}

int dc_scan_lattice_localize(int32_t* nbr, int64_t n, int k, const int64_t* scan_ptr, int n_scans, int32_t* info, hipStream_t stream) {
  if (!nbr || n < 1 || k < 1 || !scan_ptr || n_scans < 1 || !info) return DC_ERR_ARG;
  hipLaunchKernelGGL(lattice_localize_kernel, dim3((unsigned)((n * k + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, nbr, n, k, scan_ptr, n_scans, info);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? DC_OK : (int)err;
}

size_t dc_points_extent_workspace_bytes(void) { return (size_t)kBoxBlocks * kBoxVals * sizeof(double); }

int dc_points_extent(const void* points, int stride, int dtype, int64_t n, double* out6, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!points || n < 1 || (stride != 3 && stride != 4) || !out6 || !ws) return DC_ERR_ARG;
  if (ws_bytes < dc_points_extent_workspace_bytes()) return DC_ERR_WORKSPACE;
  double* part = (double*)ws;
  if (dtype == DC_F32) hipLaunchKernelGGL((bbox_partial_kernel<float>), dim3(kBoxBlocks), dim3(kBlock), 0, stream, (const float*)points, stride, n, part);
  else if (dtype == DC_F64) hipLaunchKernelGGL((bbox_partial_kernel<double>), dim3(kBoxBlocks), dim3(kBlock), 0, stream, (const double*)points, stride, n, part);
  else return DC_ERR_DTYPE;
  hipLaunchKernelGGL(extent_finish_kernel, dim3(1), dim3(kBlock), 0, stream, part, kBoxBlocks, out6);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? DC_OK : (int)err;
}

int dc_scan_ids(const int64_t* scan_ptr, int n_scans, int64_t n, int32_t* out, hipStream_t stream) {
  if (!scan_ptr || n_scans < 1 || n < 0 || (n > 0 && !out)) return DC_ERR_ARG;
  if (n == 0) return DC_OK;
  hipLaunchKernelGGL(scan_ids_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, scan_ptr, n_scans, n, out);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? DC_OK : (int)err;
}

}  // extern "C"
