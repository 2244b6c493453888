// Block-local gather tables for the hot kernels (gfx950).
//
// The forward kernel gathers the K neighbour rows of every point (depth_cloud.py:303-304 `points[neighbors]`), the
// backward kernel the records of every centre whose neighbourhood contains the point (autograd's scatter of the same
// index).  In Morton order the 256 points of a block reference only ~1.5 x 256 DISTINCT rows (375 of 2560 references
// at K = 10), yet a per-lane gather costs one L1 tag lookup per reference, and the L1 serves one lookup per cycle --
// that, not HBM and not the ALUs, bounded both kernels (rocprofv3: TCP_TOTAL_CACHE_ACCESSES ~ 1.06 per CU cycle).
//
// A block table, built once per neighbourhood set, lists for every block of 256 rows the distinct rows it
// references (`blk_ids[blk_ptr[b] .. blk_ptr[b+1])`, ascending) and replaces every reference by a 16-bit position in
// that list (stored as the byte offset 16 x position of the row in the LDS tile).  Layouts: slot-major
// (`loc[(slot_ptr[b] + s) * 256 + lane]`, 0xFFFF = empty slot: a wavefront reads its slot s with one coalesced 128-B
// load; exact for the forward's [rows, K] table) or per-row runs padded to four positions (`loc[4 * run_ptr[r] + s]`:
// the backward's incoming-edge lists, whose length varies per point).  The kernels stage the distinct rows into LDS once
// per block and gather from LDS.
//
// Build: keys (block << 32 | id) of all references -> radix sort -> heads of runs -> prefix sum -> scatter.
#include <atomic>
#include <cstring>
#include <cstdlib>
#include "dc_common.h"
#include "../../include/dc_hip.h"
#include "dc_device.h"
#include "dc_hostutil.h"
#include "dc_sort.h"

namespace dc {

constexpr uint64_t kNoKey = ~0ull;
constexpr int32_t kMaxBlockRows = 0xFFF;     // 16 x position must stay below the 0xFFFF padding marker

// slots per block: the longest reference list among the block's rows (fixed k: k)
__global__ __launch_bounds__(kBlock) void bt_slot_count_kernel(const int32_t* __restrict__ row_ptr, int64_t n_rows, int k,
                                                               int32_t* __restrict__ cnt) {
  __shared__ int s_max;
  if (threadIdx.x == 0) s_max = 0;
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  int deg = 0;
  if (r < n_rows) deg = row_ptr ? row_ptr[r + 1] - row_ptr[r] : k;
  for (int off = kWave / 2; off > 0; off >>= 1) deg = max(deg, __shfl_down(deg, off, kWave));
  if ((threadIdx.x & (kWave - 1)) == 0) atomicMax(&s_max, deg);
  __syncthreads();
  // lists of varying length (CSR): whole trips of eight slots (the ragged kernels read eight positions per trip, unguarded)
  if (threadIdx.x == 0) cnt[blockIdx.x] = row_ptr ? (s_max + 7) & ~7 : s_max;
}

// exclusive prefix sum of cnt[0..m) into out[0..m]; one block (m is the number of 256-row blocks)
__global__ __launch_bounds__(1024) void bt_scan_kernel(const int32_t* __restrict__ cnt, int64_t m, int32_t* __restrict__ out) {
  __shared__ int s_part[1024];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < m; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const int v = i < m ? cnt[i] : 0;
    s_part[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int t = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0;
      __syncthreads();
      s_part[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < m) out[i] = s_carry + s_part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry += s_part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[m] = s_carry;
}

// CSR lists: the row every reference belongs to (references beyond row_ptr[n_rows] keep -1)
__global__ __launch_bounds__(kBlock) void bt_row_of_kernel(const int32_t* __restrict__ row_ptr, int64_t n_rows,
                                                           int32_t* __restrict__ row_of) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r >= n_rows) return;
  for (int32_t e = row_ptr[r]; e < row_ptr[r + 1]; ++e) row_of[e] = (int32_t)r;
}

__global__ __launch_bounds__(kBlock) void bt_keys_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ row_of,
                                                         int64_t n_refs, int k, uint64_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n_refs) return;
  const int64_t row = row_of ? (int64_t)row_of[e] : e / k;
  const int32_t id = ids[e];
  keys[e] = (row >= 0 && id >= 0) ? (((uint64_t)(row / kBlock)) << 32) | (uint32_t)id : kNoKey;
  vals[e] = (uint32_t)e;
}

__global__ __launch_bounds__(kBlock) void bt_heads_kernel(const uint64_t* __restrict__ skeys, int64_t n_refs,
                                                          uint32_t* __restrict__ heads) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n_refs) return;
  const uint64_t key = skeys[p];
  heads[p] = (key != kNoKey && (p == 0 || skeys[p - 1] != key)) ? 1u : 0u;
}

// blk_ptr[b] = number of distinct (block, id) pairs of the blocks before b
__global__ __launch_bounds__(kBlock) void bt_blk_ptr_kernel(const uint64_t* __restrict__ skeys, const uint32_t* __restrict__ rank1,
                                                            int64_t n_refs, int64_t n_blocks, int32_t* __restrict__ blk_ptr) {
  const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (b > n_blocks) return;
  const uint64_t want = (uint64_t)b << 32;          // first key of block b (invalid keys are larger than any block's)
  int64_t lo = 0, hi = n_refs;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (skeys[mid] < want) lo = mid + 1; else hi = mid;
  }
  blk_ptr[b] = lo == 0 ? 0 : (int32_t)rank1[lo - 1];
}

__global__ __launch_bounds__(kBlock) void bt_scatter_kernel(const uint64_t* __restrict__ skeys, const uint32_t* __restrict__ svals,
                                                            const uint32_t* __restrict__ heads, const uint32_t* __restrict__ rank1,
                                                            const int32_t* __restrict__ row_of, const int32_t* __restrict__ row_ptr,
                                                            int64_t n_refs, int k, const int32_t* __restrict__ blk_ptr,
                                                            const int32_t* __restrict__ slot_ptr, const int32_t* __restrict__ run_ptr,
                                                            int32_t* __restrict__ blk_ids, uint16_t* __restrict__ loc,
                                                            int32_t* __restrict__ info) {
  const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (p >= n_refs) return;
  const uint64_t key = skeys[p];
  if (key == kNoKey) return;
  const int64_t b = (int64_t)(key >> 32);
  const int32_t r = (int32_t)rank1[p] - 1;
  if (heads[p]) blk_ids[r] = (int32_t)(uint32_t)key;
  const int32_t local = r - blk_ptr[b];
  if (local >= kMaxBlockRows) { atomicMax(&info[2], 1); return; }
  const int64_t e = svals[p];
  const int64_t row = row_of ? (int64_t)row_of[e] : e / k;
  const int64_t slot = row_ptr ? e - row_ptr[row] : e - row * k;
  const int64_t at = run_ptr ? (int64_t)run_ptr[row] * 4 + slot : ((int64_t)slot_ptr[b] + slot) * kBlock + (row & (kBlock - 1));
  loc[at] = (uint16_t)(local << 4);                       // byte offset of the row in the LDS tile
}

// runs per row of a CSR list: ceil(length / 4); entry n_rows = 0, so that the exclusive scan ends with the total
__global__ __launch_bounds__(kBlock) void bt_run_len_kernel(const int32_t* __restrict__ row_ptr, int64_t n_rows,
                                                            int32_t* __restrict__ len) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r > n_rows) return;
  len[r] = r < n_rows ? (row_ptr[r + 1] - row_ptr[r] + 3) >> 2 : 0;
}

// own_base[b] = position of row 256 b in block b's distinct list when ALL rows of the block appear in it (then they are
// contiguous there, the list being sorted), else -1.  A k-NN table lists every point among its own neighbours, so the
// forward kernels can take a lane's centre point from the staged LDS rows instead of fetching / forming it again.
__global__ __launch_bounds__(kBlock) void bt_own_base_kernel(const int32_t* __restrict__ blk_ptr, const int32_t* __restrict__ blk_ids,
                                                             int64_t n_rows, int64_t n_blocks, int32_t* __restrict__ own_base) {
  const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (b >= n_blocks) return;
  const int32_t lo0 = blk_ptr[b], hi0 = blk_ptr[b + 1];
  const int64_t first = b * kBlock, cnt = (first + kBlock <= n_rows ? kBlock : n_rows - first);
  int32_t lo = lo0, hi = hi0;
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    if ((int64_t)blk_ids[mid] < first) lo = mid + 1; else hi = mid;
  }
  const bool all = lo + cnt <= hi0 && (int64_t)blk_ids[lo] == first && (int64_t)blk_ids[lo + cnt - 1] == first + cnt - 1;
  own_base[b] = all ? lo - lo0 : -1;
}

__global__ __launch_bounds__(kBlock) void bt_info_kernel(const int32_t* __restrict__ blk_ptr, int64_t n_blocks,
                                                         int32_t* __restrict__ info) {
  const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (b < n_blocks) atomicMax(&info[1], blk_ptr[b + 1] - blk_ptr[b]);
  if (b == 0) info[0] = blk_ptr[n_blocks];
}

// Inside every block of 256 consecutive positions of `order` (the plan's Morton order): the points inside the loss mask first, by
// scan id, then those outside, by scan id -- a stable counting sort of the block's lanes by the key (outside ? S : 0) + scan, and
// the segment starts seg[b][0 .. 2 S] (seg[b][2 S] = the block's point count).  What the set of points of a block is does not
// change.  (dcSequenceDesc.scan_seg; round 4 -- as torch ops this was a stable argsort of [blocks, 256] keys, whose first use in a
// process loads torch's sort kernels: 0.4 s of the first set-up.)
constexpr int kMaxGroupKeys = 2 * 64 + 1;
__global__ __launch_bounds__(kBlock) void bt_block_group_kernel(const int32_t* __restrict__ order_in, const int32_t* __restrict__ scan_id,
                                                                const uint8_t* __restrict__ mask, int64_t n, int n_scans,
                                                                int32_t* __restrict__ order_out, uint16_t* __restrict__ seg,
                                                                uint8_t* __restrict__ blk_skip, int32_t* __restrict__ skipped) {
  constexpr int NW = kBlock / kWave;
  __shared__ int s_cnt[NW][kMaxGroupKeys];
  __shared__ int s_start[kMaxGroupKeys + 1];
  const int V = 2 * n_scans;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int64_t p = (int64_t)blockIdx.x * kBlock + tid;
  const bool valid = p < n;
  int32_t o = 0;
  int key = V;                                               // positions past the end sort last
  if (valid) {
    o = order_in[p];
    key = scan_id[o] + ((mask && !mask[o]) ? n_scans : 0);
  }
  int rank_in_wave = 0;
  for (int q = 0; q <= V; ++q) {
    const unsigned long long m = __ballot(key == q);
    if (lane == 0) s_cnt[wave][q] = __popcll(m);
    if (key == q) rank_in_wave = __popcll(m & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  if (tid == 0) {
    int at = 0;
    for (int q = 0; q <= V; ++q) {
      s_start[q] = at;
      for (int w = 0; w < NW; ++w) at += s_cnt[w][q];
    }
  }
  __syncthreads();
  if (tid <= V) seg[(int64_t)blockIdx.x * (V + 1) + tid] = (uint16_t)s_start[tid];
  if (tid == 0 && (blk_skip || skipped)) {
    // the block's points inside the mask come first (keys below n_scans): a block without any is skipped by the one-pass kernels,
    // and of a mixed block the wavefronts behind its last inside point are
    const int inside = s_start[n_scans], cnt = s_start[V];
    if (blk_skip) blk_skip[blockIdx.x] = inside == 0 ? 1 : 0;
    if (skipped) atomicAdd(skipped, (cnt + kWave - 1) / kWave - (inside + kWave - 1) / kWave);
  }
  if (valid) {
    int pos = s_start[key] + rank_in_wave;
    for (int w = 0; w < wave; ++w) pos += s_cnt[w][key];
    order_out[(int64_t)blockIdx.x * kBlock + pos] = o;
  }
}


// ---- [rows, K] tables, K <= 16: every block on its own, in LDS ---------------------------------------------------------------------
// The radix-sort build above moves every (block, id) reference through five sort passes, a scan and a scatter (1.8 ms for the
// 20 M references of the C2 table: the largest item of the set-up after the two k-NN builds).  A block has at most 256 K <= 4096
// references and a few hundred distinct rows, so one workgroup can do it alone: the references go into an open-addressing hash
// table in LDS (atomicCAS; the slot of every reference is kept in a register), the occupied slots are compacted into a list, the
// list -- 444 entries at C2 -- is sorted by a bitonic network, every distinct row's rank goes back into the table, and every
// reference reads its rank.  The distinct rows leave through a fixed-stride scratch row; a scan of the counts gives blk_ptr and a
// copy packs blk_ids.  Same tables as the radix build (ascending ids per block), 0.25 ms.
constexpr int kBtTab = 4096;
__device__ __forceinline__ uint32_t bt_hash(int32_t id) { return ((uint32_t)id * 2654435761u) >> 20; }       // 12 bits

template <int K>
__global__ __launch_bounds__(kBlock) void bt_block_unique_kernel(const int32_t* __restrict__ ids, int64_t n_rows,
                                                                 const int32_t* __restrict__ slot_ptr, int32_t* __restrict__ uniq,
                                                                 int32_t* __restrict__ cnt, uint16_t* __restrict__ loc,
                                                                 int32_t* __restrict__ info) {
  static_assert(kBlock * K <= kBtTab, "a block's references fit the table");
  __shared__ uint32_t s_key[kBtTab];            // id + 1; 0 = empty
  __shared__ uint16_t s_rank[kBtTab];
  __shared__ int32_t s_list[kBtTab];
  __shared__ int s_wave[kBlock / kWave];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int64_t b = blockIdx.x;
  const int64_t r = b * kBlock + tid;
  for (int t = tid; t < kBtTab; t += kBlock) s_key[t] = 0u;
  int32_t id[K];
#pragma unroll
  for (int q = 0; q < K; ++q) id[q] = r < n_rows ? ids[r * K + q] : -1;
  __syncthreads();
  uint32_t at[K];
#pragma unroll
  for (int q = 0; q < K; ++q) {
    at[q] = 0;
    if (id[q] >= 0) {
      uint32_t h = bt_hash(id[q]);
      const uint32_t key = (uint32_t)id[q] + 1u;
      for (;;) {
        const uint32_t old = atomicCAS(&s_key[h], 0u, key);
        if (old == 0u || old == key) break;
        h = (h + 1u) & (kBtTab - 1);
      }
      at[q] = h;
    }
  }
  __syncthreads();
  // the occupied slots, compacted in slot order (any order: the list is sorted next)
  constexpr int PER = kBtTab / kBlock;
  int mine = 0;
#pragma unroll
  for (int u_ = 0; u_ < PER; ++u_) mine += s_key[tid * PER + u_] != 0u ? 1 : 0;
  int incl = mine;
  for (int off = 1; off < kWave; off <<= 1) {
    const int o = __shfl_up(incl, off, kWave);
    if (lane >= off) incl += o;
  }
  if (lane == kWave - 1) s_wave[wave] = incl;
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kBlock / kWave; ++w) { if (w < wave) base += s_wave[w]; total += s_wave[w]; }
  int pos = base + incl - mine;
#pragma unroll
  for (int u_ = 0; u_ < PER; ++u_) {
    const uint32_t key = s_key[tid * PER + u_];
    if (key != 0u) s_list[pos++] = (int32_t)(key - 1u);
  }
  int np2 = kWave;
  while (np2 < total) np2 <<= 1;
  for (int t = total + tid; t < np2; t += kBlock) s_list[t] = 0x7fffffff;
  __syncthreads();
  for (int k2 = 2; k2 <= np2; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < np2 / 2; t += kBlock) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int32_t a = s_list[i], c = s_list[i | j];
        const bool up = (i & k2) == 0;
        if ((a > c) == up) { s_list[i] = c; s_list[i | j] = a; }
      }
      __syncthreads();
    }
  }
  // every distinct row's rank back into the table; the sorted list out through the block's scratch row
  int32_t* urow = uniq + b * (int64_t)(kBlock * K);
  for (int p = tid; p < total; p += kBlock) {
    const int32_t v = s_list[p];
    uint32_t h = bt_hash(v);
    while (s_key[h] != (uint32_t)v + 1u) h = (h + 1u) & (kBtTab - 1);
    s_rank[h] = (uint16_t)p;
    urow[p] = v;
  }
  if (tid == 0) {
    cnt[b] = total;
    if (total >= kMaxBlockRows) atomicMax(&info[2], 1);
  }
  __syncthreads();
  const int32_t s0 = slot_ptr[b];
#pragma unroll
  for (int q = 0; q < K; ++q)
    loc[((int64_t)s0 + q) * kBlock + tid] = id[q] >= 0 ? (uint16_t)(s_rank[at[q]] << 4) : (uint16_t)0xFFFF;
}

__global__ __launch_bounds__(kBlock) void bt_pack_unique_kernel(const int32_t* __restrict__ uniq, const int32_t* __restrict__ blk_ptr,
                                                                int stride, int32_t* __restrict__ blk_ids) {
  const int64_t b = blockIdx.x;
  const int32_t lo = blk_ptr[b], n = blk_ptr[b + 1] - lo;
  for (int p = threadIdx.x; p < n; p += kBlock) blk_ids[lo + p] = uniq[b * (int64_t)stride + p];
}

// dst row i = src row order[i], rows of `words` 32-bit words (or `bytes` single bytes when words == 0): the plan's permutation of its
// per-point arrays in one kernel family (as tensor indexing every dtype / shape loads its own torch kernel in a fresh process).
__global__ __launch_bounds__(kBlock) void bt_gather_rows_kernel(const uint32_t* __restrict__ src, const int64_t* __restrict__ order,
                                                                int64_t n, int words, uint32_t* __restrict__ dst) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n * words) return;
  const int64_t i = e / words;
  dst[e] = src[order[i] * words + (e - i * words)];
}
__global__ __launch_bounds__(kBlock) void bt_gather_bytes_kernel(const uint8_t* __restrict__ src, const int64_t* __restrict__ order,
                                                                 int64_t n, uint8_t* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) dst[i] = src[order[i]];
}

static inline int64_t blocks_of(int64_t n) { return (n + kBlock - 1) / kBlock; }
static inline unsigned grid_of(int64_t n) { return (unsigned)(n > 0 ? (n + kBlock - 1) / kBlock : 1); }

// The neighbour table in a new point order: out[i][q] = rank[nbr[order[i]][q]] (-1 stays -1); rank = the inverse of order.  One pass
// instead of a chain of index / clamp / where passes over [n, k] 64-bit intermediates (the set-up's permute stage: 1.2 -> 0.4 ms).
__global__ __launch_bounds__(kBlock) void bt_inverse_order_kernel(const int64_t* __restrict__ order, int64_t n, int32_t* __restrict__ rank) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) rank[order[i]] = (int32_t)i;
}
__global__ __launch_bounds__(kBlock) void bt_table_permute_kernel(const int32_t* __restrict__ nbr, const int64_t* __restrict__ order,
                                                                  const int32_t* __restrict__ rank, int64_t n, int k,
                                                                  int32_t* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n * k) return;
  const int64_t i = e / k;
  const int q = (int)(e - i * k);
  const int32_t v = nbr[order[i] * k + q];
  out[e] = v >= 0 ? rank[v] : v;
}

}  // namespace dc

using namespace dc;

// A-B switch for measurements and tests (dc_block_table_set_lds_build): 0 sends [rows, K] tables through the radix-sort build too
static std::atomic<int> g_bt_lds{1};

extern "C" {

int dc_block_table_set_lds_build(int on) { return g_bt_lds.exchange(on ? 1 : 0); }

int dc_block_table_slots(const int32_t* row_ptr, int64_t n_rows, int k, int32_t* slot_cnt_ws, int32_t* slot_ptr,
                         hipStream_t stream) {
  if (n_rows < 0 || !slot_ptr || (!row_ptr && k < 1) || (n_rows > 0 && !slot_cnt_ws)) return DC_ERR_ARG;
  const int64_t nb = blocks_of(n_rows);
  if (nb == 0) return (int)hipMemsetAsync(slot_ptr, 0, sizeof(int32_t), stream);
  hipLaunchKernelGGL(bt_slot_count_kernel, dim3((unsigned)nb), dim3(kBlock), 0, stream, row_ptr, n_rows, k, slot_cnt_ws);
  hipLaunchKernelGGL(bt_scan_kernel, dim3(1), dim3(1024), 0, stream, slot_cnt_ws, nb, slot_ptr);
  DC_HIP(hipGetLastError());
  return DC_OK;
}

size_t dc_block_table_workspace_bytes(int64_t n_refs) {
  if (n_refs < 0) return 0;
  const size_t ne = (size_t)(n_refs > 0 ? n_refs : 1);
  Carver c(nullptr);
  c.take<uint64_t>(ne); c.take<uint64_t>(ne); c.take<uint32_t>(ne); c.take<uint32_t>(ne);
  c.take<uint32_t>(ne); c.take<uint32_t>(ne); c.take<int32_t>(ne);
  const size_t sb = sort_pairs_bytes(ne, 64), cb = scan_bytes(ne);
  c.take<char>(sb > cb ? sb : cb);
  return c.off + 256;
}

// A [rows, K] table with K = 4 / 8 / 10 / 16 is built block by block in LDS and needs only the blocks' scratch rows (one int32 per
// reference slot) -- a tenth of the radix build's keys, values and sort storage (800 MB for the C2 table: its allocation alone took
// 0.2 s of the first set-up of a process).
size_t dc_block_table_slots_workspace_bytes(int64_t n_rows, int k) {
  if (n_rows < 0 || k < 1) return 0;
  if ((k == 4 || k == 8 || k == 10 || k == 16) && g_bt_lds.load()) {
    const size_t nb = (size_t)blocks_of(n_rows);
    Carver c(nullptr);
    c.take<int32_t>((nb > 0 ? nb : 1) * kBlock * (size_t)k); c.take<int32_t>(nb + 1);
    return c.off + 256;
  }
  return dc_block_table_workspace_bytes(n_rows * (int64_t)k);
}

// loc_entries: number of uint16 entries of loc (all set to 0xFFFF first); run_ptr != NULL selects the run layout
static int block_table_build_impl(const int32_t* row_ptr, const int32_t* ids, int64_t n_rows, int k, int64_t n_refs,
                                  const int32_t* slot_ptr, const int32_t* run_ptr, int64_t loc_entries, int32_t* blk_ptr,
                                  int32_t* blk_ids, uint16_t* loc, int32_t* info, void* ws, size_t ws_bytes, hipStream_t stream,
                                  int32_t* own_base = nullptr) {
  const int64_t n_slot_rows = loc_entries / kBlock;
  if (n_rows < 0 || n_refs < 0 || loc_entries < 0 || !blk_ptr || !info || (!row_ptr && k < 1)) return DC_ERR_ARG;
  DC_HIP(hipMemsetAsync(info, 0, 4 * sizeof(int32_t), stream));
  const int64_t nb = blocks_of(n_rows);
  if (n_rows == 0 || n_refs == 0) return (int)hipMemsetAsync(blk_ptr, 0, (size_t)(nb + 1) * sizeof(int32_t), stream);
  if (!ids || (!slot_ptr && !run_ptr) || !blk_ids || !loc || !ws) return DC_ERR_ARG;
  if (!row_ptr && n_refs != n_rows * (int64_t)k) return DC_ERR_ARG;
  if (n_refs >= (int64_t)0x7fffffff || n_slot_rows * kBlock >= ((int64_t)1 << 40)) return DC_ERR_UNSUPPORTED;
  const bool lds_build = !row_ptr && !run_ptr && (k == 4 || k == 8 || k == 10 || k == 16) && g_bt_lds.load();
  if (ws_bytes < (lds_build ? dc_block_table_slots_workspace_bytes(n_rows, k) : dc_block_table_workspace_bytes(n_refs))) return DC_ERR_WORKSPACE;
  if (lds_build) {
    // a [rows, K] table: every block on its own in LDS (bt_block_unique_kernel); the workspace holds the blocks' scratch rows
    Carver c2(ws);
    int32_t* uniq = c2.take<int32_t>((size_t)nb * kBlock * k);       // (<= n_refs + 255 k entries: inside the radix build's arrays)
    int32_t* cnt = c2.take<int32_t>((size_t)nb + 1);
    const dim3 grid_b((unsigned)nb), block_b(kBlock);
#define BT_K(KK) hipLaunchKernelGGL((bt_block_unique_kernel<KK>), grid_b, block_b, 0, stream, ids, n_rows, slot_ptr, uniq, cnt, loc, info)
    if (k == 10) BT_K(10); else if (k == 4) BT_K(4); else if (k == 8) BT_K(8); else BT_K(16);
#undef BT_K
    hipLaunchKernelGGL(bt_scan_kernel, dim3(1), dim3(1024), 0, stream, cnt, nb, blk_ptr);
    hipLaunchKernelGGL(bt_pack_unique_kernel, grid_b, block_b, 0, stream, uniq, blk_ptr, (int)(kBlock * k), blk_ids);
    hipLaunchKernelGGL(bt_info_kernel, dim3(grid_of(nb)), block_b, 0, stream, blk_ptr, nb, info);
    if (own_base) hipLaunchKernelGGL(bt_own_base_kernel, dim3(grid_of(nb)), block_b, 0, stream, blk_ptr, blk_ids, n_rows, nb, own_base);
    DC_HIP(hipGetLastError());
    return DC_OK;
  }
  Carver c(ws);
  uint64_t* keys = c.take<uint64_t>(n_refs);
  uint64_t* skeys = c.take<uint64_t>(n_refs);
  uint32_t* vals = c.take<uint32_t>(n_refs);
  uint32_t* svals = c.take<uint32_t>(n_refs);
  uint32_t* heads = c.take<uint32_t>(n_refs);
  uint32_t* rank1 = c.take<uint32_t>(n_refs);
  int32_t* row_of = c.take<int32_t>(n_refs);
  const size_t sb = sort_pairs_bytes((size_t)n_refs, 64), cb = scan_bytes((size_t)n_refs);
  void* tmp = c.take<char>(sb > cb ? sb : cb);
  const dim3 block(kBlock);
  if (row_ptr) {
    DC_HIP(hipMemsetAsync(row_of, 0xff, (size_t)n_refs * sizeof(int32_t), stream));
    hipLaunchKernelGGL(bt_row_of_kernel, dim3(grid_of(n_rows)), block, 0, stream, row_ptr, n_rows, row_of);
  }
  const int32_t* rows = row_ptr ? row_of : nullptr;
  DC_HIP(hipMemsetAsync(loc, 0xff, (size_t)loc_entries * sizeof(uint16_t), stream));
  hipLaunchKernelGGL(bt_keys_kernel, dim3(grid_of(n_refs)), block, 0, stream, ids, rows, n_refs, k, keys, vals);
  // 32 id bits + enough block bits that the all-ones block of an invalid reference exceeds every real block: invalid
  // references sort behind all blocks, and the sort runs 6 radix passes instead of 8
  unsigned key_bits = 33;
  while (key_bits < 64 && (((uint64_t)1 << (key_bits - 32)) - 1) < (uint64_t)nb) ++key_bits;
  DC_HIP(sort_pairs_u64(tmp, sb, keys, skeys, vals, svals, (size_t)n_refs, 0, key_bits, stream));
  hipLaunchKernelGGL(bt_heads_kernel, dim3(grid_of(n_refs)), block, 0, stream, skeys, n_refs, heads);
  DC_HIP(inclusive_scan_32(tmp, cb, heads, rank1, (size_t)n_refs, stream));
  hipLaunchKernelGGL(bt_blk_ptr_kernel, dim3(grid_of(nb + 1)), block, 0, stream, skeys, rank1, n_refs, nb, blk_ptr);
  hipLaunchKernelGGL(bt_scatter_kernel, dim3(grid_of(n_refs)), block, 0, stream, skeys, svals, heads, rank1, rows, row_ptr,
                     n_refs, k, blk_ptr, slot_ptr, run_ptr, blk_ids, loc, info);
  hipLaunchKernelGGL(bt_info_kernel, dim3(grid_of(nb)), block, 0, stream, blk_ptr, nb, info);
  if (own_base) hipLaunchKernelGGL(bt_own_base_kernel, dim3(grid_of(nb)), block, 0, stream, blk_ptr, blk_ids, n_rows, nb, own_base);
  DC_HIP(hipGetLastError());
  return DC_OK;
}

int dc_block_table_own_base(const int32_t* blk_ptr, const int32_t* blk_ids, int64_t n_rows, int32_t* own_base, hipStream_t stream) {
  if (n_rows < 0 || !blk_ptr || !own_base) return DC_ERR_ARG;
  const int64_t nb = blocks_of(n_rows);
  if (nb == 0) return DC_OK;
  hipLaunchKernelGGL(bt_own_base_kernel, dim3(grid_of(nb)), dim3(kBlock), 0, stream, blk_ptr, blk_ids, n_rows, nb, own_base);
  DC_HIP(hipGetLastError());
  return DC_OK;
}

int dc_block_group(const int32_t* order_in, const int32_t* scan_id, const uint8_t* mask, int64_t n, int n_scans, int32_t* order_out,
                   uint16_t* seg_out, uint8_t* blk_skip_out, int32_t* skipped_out, hipStream_t stream) {
  if (n < 0 || n_scans < 1 || n_scans > 64 || !order_in || !scan_id || !order_out || !seg_out || order_in == order_out) return DC_ERR_ARG;
  if (skipped_out) DC_HIP(hipMemsetAsync(skipped_out, 0, sizeof(int32_t), stream));
  if (n == 0) return DC_OK;
  hipLaunchKernelGGL(bt_block_group_kernel, dim3((unsigned)blocks_of(n)), dim3(kBlock), 0, stream, order_in, scan_id, mask, n, n_scans,
                     order_out, seg_out, blk_skip_out, skipped_out);
  DC_HIP(hipGetLastError());
  return DC_OK;
}

int dc_gather_rows(const void* src, int row_bytes, const int64_t* order, int64_t n, void* dst, hipStream_t stream) {
  if (n < 0 || row_bytes < 1 || (n > 0 && (!src || !order || !dst || src == dst))) return DC_ERR_ARG;
  if (row_bytes != 1 && (row_bytes & 3)) return DC_ERR_UNSUPPORTED;
  if (n == 0) return DC_OK;
  if (row_bytes == 1)
    hipLaunchKernelGGL(bt_gather_bytes_kernel, dim3((unsigned)blocks_of(n)), dim3(kBlock), 0, stream, (const uint8_t*)src, order, n, (uint8_t*)dst);
  else
    hipLaunchKernelGGL(bt_gather_rows_kernel, dim3((unsigned)blocks_of(n * (row_bytes / 4))), dim3(kBlock), 0, stream, (const uint32_t*)src, order, n,
                       row_bytes / 4, (uint32_t*)dst);
  DC_HIP(hipGetLastError());
  return DC_OK;
}

int dc_table_permute(const int32_t* nbr, int64_t n, int k, const int64_t* order, int32_t* rank_out, int32_t* nbr_out, hipStream_t stream) {
  if (n < 0 || k < 1 || !nbr || !order || !rank_out || !nbr_out || nbr == nbr_out) return DC_ERR_ARG;
  if (n == 0) return DC_OK;
  hipLaunchKernelGGL(bt_inverse_order_kernel, dim3((unsigned)blocks_of(n)), dim3(kBlock), 0, stream, order, n, rank_out);
  hipLaunchKernelGGL(bt_table_permute_kernel, dim3((unsigned)blocks_of(n * k)), dim3(kBlock), 0, stream, nbr, order, rank_out, n, k, nbr_out);
  DC_HIP(hipGetLastError());
  return DC_OK;
}

int dc_block_table_build(const int32_t* row_ptr, const int32_t* ids, int64_t n_rows, int k, int64_t n_refs,
                         const int32_t* slot_ptr, int64_t n_slot_rows, int32_t* blk_ptr, int32_t* blk_ids, uint16_t* loc,
                         int32_t* info, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n_slot_rows < 0) return DC_ERR_ARG;
  return block_table_build_impl(row_ptr, ids, n_rows, k, n_refs, slot_ptr, nullptr, n_slot_rows * kBlock, blk_ptr, blk_ids, loc,
                                info, ws, ws_bytes, stream);
}

// upper bound of the number of runs: every row wastes at most three positions
int64_t dc_block_table_run_capacity(int64_t n_rows, int64_t n_refs) {
  if (n_rows < 0 || n_refs < 0) return 0;
  return (n_refs + 3 * n_rows) / 4 + 1;
}

int dc_block_table_build_runs(const int32_t* row_ptr, const int32_t* ids, int64_t n_rows, int64_t n_refs, int32_t* run_ptr,
                              int32_t* blk_ptr, int32_t* blk_ids, uint16_t* loc, int32_t* info, void* ws, size_t ws_bytes,
                              hipStream_t stream) {
  if (!row_ptr || !run_ptr || n_rows < 0 || n_refs < 0) return DC_ERR_ARG;
  if (ws_bytes < dc_block_table_workspace_bytes(n_refs > n_rows + 1 ? n_refs : n_rows + 1) || !ws) return DC_ERR_WORKSPACE;
  if (n_rows + 1 >= (int64_t)0x7fffffff) return DC_ERR_UNSUPPORTED;
  // run_ptr = exclusive scan of ceil(length / 4); the scan's scratch and input live in the (not yet used) workspace
  Carver c(ws);
  int32_t* len = c.take<int32_t>((size_t)n_rows + 1);
  const size_t cb = scan_bytes((size_t)n_rows + 1);
  void* tmp = c.take<char>(cb);
  if (c.off > ws_bytes) return DC_ERR_WORKSPACE;
  hipLaunchKernelGGL(bt_run_len_kernel, dim3(grid_of(n_rows + 1)), dim3(kBlock), 0, stream, row_ptr, n_rows, len);
  DC_HIP(exclusive_scan_32(tmp, cb, len, run_ptr, (size_t)n_rows + 1, stream));
  return block_table_build_impl(row_ptr, ids, n_rows, 0, n_refs, nullptr, run_ptr, dc_block_table_run_capacity(n_rows, n_refs) * 4,
                                blk_ptr, blk_ids, loc, info, ws, ws_bytes, stream);
}

}  // extern "C"
