/* libdc_hip.so -- C ABI of the MI355X (gfx950) map-consistency hot path of ctu-vras/depth_correction.
 *
 * The reference is pure Python; this boundary is what a Python binding of its hot path (ctypes, see
 * INTEGRATION.md) calls instead of torch / scipy / LAPACK.  Each entry point names the reference code it
 * replaces (paths relative to the reference's src/depth_correction/).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch's allocator), unless marked "host";
 *   - float arrays have element type `dtype` (DC_F32 | DC_F64); on-chip arithmetic is always fp64;
 *   - indices are int32, -1 = missing neighbour (the reference's int64 is converted at the Python boundary);
 *   - every call is asynchronous on `stream` and never allocates: scratch comes from a caller workspace whose
 *     size the *_workspace_bytes / dc_partial_rows twins report;
 *   - return value: 0 = ok, < 0 = invalid argument (DC_ERR_*), > 0 = hipError_t;
 *   - results are bitwise reproducible (no floating-point atomics anywhere).
 */
#ifndef DC_HIP_H
#define DC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef DC_F32
#define DC_F32 0
#define DC_F64 1
#define DC_Q32 2 /* internal point format: int32 fixed point rows [n,4], x = origin + q * scale (16 B / point) */
#define DC_LOSS_MIN_EIGVAL 0
#define DC_LOSS_TRACE 1
#define DC_LOSS_RAW_POINTWISE 0x100 /* OR-ed into loss_kind of dc_consistency_fwd: `pointwise` receives the loss before relu / sqrt */
/* OR-ed into loss_kind (dc_consistency_fwd, dcSequenceDesc.loss_kind): the reduction drops NaN (skip_nans) / non-finite
 * (only_finite) pointwise losses -- neither the sum nor the count nor any gradient sees them (loss.py:125-137) */
#define DC_LOSS_SKIP_NANS 0x200
#define DC_LOSS_ONLY_FINITE 0x400
#define DC_MODEL_NONE 0
#define DC_MODEL_POLYNOMIAL 1
#define DC_MODEL_SCALED_POLYNOMIAL 2
#define DC_MODEL_LINEAR 3            /* w = [w0, w1, b]: d' = w0 d + w1 gamma + b   (model.py:113-146) */
#define DC_MODEL_INVCOS 4            /* w = [p0]: d' = d - p0 / cos(gamma)          (model.py:289-313) */
#define DC_MODEL_SCALED_INVCOS 5     /* w = [p0]: d' = d (1 - p0 / |cos(gamma)|)    (model.py:316-349) */
#define DC_MODEL_LAST DC_MODEL_SCALED_INVCOS
#define DC_MAX_MODEL_TERMS 8
#define DC_OK 0
#define DC_ERR_ARG (-1)
#define DC_ERR_DTYPE (-2)
#define DC_ERR_WORKSPACE (-3)
#define DC_ERR_UNSUPPORTED (-4)
#define DC_ERR_BACKWARD_TABLES (-5) /* dc_sequence_eval / _step: this evaluation needs dcSequenceDesc.csr_ptr / csr_src (and bwd_table) */
#endif

typedef struct ihipStream_t* dcStream_t; /* == hipStream_t */

int dc_version(void);

/* ---- neighbourhood builder: nearest_neighbors.py:22-80 (scipy cKDTree), depth_cloud.py:210-215 --------- */

/* Self k-NN (query == NULL) or k-NN of `query` [n_query, q_stride] in `points` [n, stride]; k <= 64.
 * r > 0: keep only d < r (cKDTree distance_upper_bound), missing -> idx -1 / dist inf.  cell_hint <= 0: auto.
 * idx_out int32 [rows, k], dist_out fp64 [rows, k] or NULL.  Ordering: ascending fp64 distance, self first. */
size_t dc_knn_workspace_bytes(int64_t n, int64_t n_query);
/* Stages of cells a query walks (stage 1: the 27 cells around its own, stage r: the shell at Chebyshev distance r) before it is
 * handed to the wavefront-per-query tail kernel: default (the value 1000) 2 below a million queries, 4 above.  0 <= shells < 100: sixteen lanes per query for
 * k <= 16 (knn_group_kernel), one lane per query above; shells + 100: one lane per query for every k (the round-2/3 kernel,
 * kept for A-B runs); negative: one lane per query to the end, no tail kernel.  Results do not depend on it. */
int dc_knn_set_shell_budget(int shells);
/* The search grid has two levels (cell edge h from the cloud's average density, and h / 2): a query of the sixteen-lanes-per-query
 * kernel searches the fine level when its own coarse cell holds at least `points` points (default 14; < 1: never).  Lidar density
 * is far from uniform: weighted by query the 27 coarse cells around a query hold ~110 points for k = 10.  Results do not depend on it. */
int dc_knn_set_fine_cell_count(int points);
int dc_knn_build(const void* points, int stride, int dtype, int64_t n, const void* query, int q_stride,
                 int64_t n_query, int k, double r, double cell_hint, int32_t* idx_out, double* dist_out, void* ws,
                 size_t ws_bytes, dcStream_t stream);
/* dc_knn_build that also writes the table as int64 [., k] (idx64_out, optional): the reference's index dtype (nearest_neighbors.py:78)
 * without a conversion pass. */
int dc_knn_build_i64(const void* points, int stride, int dtype, int64_t n, const void* query, int q_stride,
                 int64_t n_query, int k, double r, double cell_hint, int32_t* idx_out, int64_t* idx64_out, double* dist_out, void* ws,
                 size_t ws_bytes, dcStream_t stream);

/* Radius search (query_ball_point, nearest_neighbors.py:50-51,69-73), two passes over one workspace
 * (size dc_knn_workspace_bytes(n, 0)): counts + their maximum, then rows of ascending indices padded with -1. */
int dc_radius_count(const void* points, int stride, int dtype, int64_t n, double r, int32_t* count_out,
                    int32_t* kmax_out, void* ws, size_t ws_bytes, dcStream_t stream);
int dc_radius_fill(int64_t n, double r, int kmax, int32_t* idx_out, void* ws, size_t ws_bytes, dcStream_t stream);
/* Radius search of the points of ANOTHER cloud (nearest_neighbors.py:50-51: query_ball_point takes any query): the same two
 * passes, rows = queries; pass 2 needs the workspace of pass 1 untouched (grid of `points` + the queries in fp64).
 * ws: dc_knn_workspace_bytes(n, n_query). */
int dc_radius_count_query(const void* points, int stride, int dtype, int64_t n, const void* query, int q_stride, int64_t n_query,
                          double r, int32_t* count_out, int32_t* kmax_out, void* ws, size_t ws_bytes, dcStream_t stream);
int dc_radius_fill_query(int64_t n, int64_t n_query, double r, int kmax, int32_t* idx_out, void* ws, size_t ws_bytes,
                         dcStream_t stream);

/* Transposed neighbour list for the backward (replaces autograd's index_put scatter of depth_cloud.py:303-304):
 * nbr int32 [n,k] with values in [0, n_dst) (n_dst <= 0: n_dst = n); csr_ptr int32 [n_dst+1], csr_src int32 [n*k]
 * (first csr_ptr[n_dst] entries valid, ascending row index). */
size_t dc_knn_transpose_workspace_bytes(int64_t n, int k);
/* Several scans in ONE k-NN build (the set-up's local feature clouds: preproc.py:35-64 once per scan, cKDTree per scan).
 * dc_scan_lattice_shift sets the scans (rows scan_ptr[s] .. scan_ptr[s + 1] of `points` [n,3], scan_ptr a DEVICE int64 [n_scans + 1],
 * n_scans <= 64) side by side on a lattice: scan s is shifted by an integer offset per axis, two box widths from its neighbours;
 * shifted fp64 [n,3] is what dc_knn_build then takes.  dc_scan_lattice_localize turns the rows of that table into indices inside
 * each row's own scan (in place).  info (device int32, zeroed by the caller): bit 0 <- some shifted coordinate is not exact in fp64
 * (or not finite), bit 1 <- some neighbour lies in another scan.  With info == 0 the table equals the per-scan tables bit for bit
 * (equal distances are ordered by index); otherwise the caller builds the scans one by one. */
size_t dc_scan_lattice_workspace_bytes(int n_scans);
int dc_scan_lattice_shift(const void* points, int dtype, int64_t n, const int64_t* scan_ptr, int n_scans, double* shifted, int32_t* info,
                          void* ws, size_t ws_bytes, dcStream_t stream);
int dc_scan_lattice_localize(int32_t* nbr, int64_t n, int k, const int64_t* scan_ptr, int n_scans, int32_t* info, dcStream_t stream);
/* Bounding box of a cloud [n, stride] (stride 3 or 4): out6 device fp64 <- {min x, y, z, max x, y, z} (the extent a fixed-point point
 * format is sized for: torch.aminmax over dim 0 of an [n,3] tensor took 1.1 ms at n = 2 M; this takes 0.03).  dc_scan_ids: out[i] =
 * the scan of row i for rows partitioned by scan_ptr (device int64 [n_scans + 1]): torch.repeat_interleave(arange(S), sizes). */
size_t dc_points_extent_workspace_bytes(void);
int dc_points_extent(const void* points, int stride, int dtype, int64_t n, double* out6, void* ws, size_t ws_bytes, dcStream_t stream);
int dc_scan_ids(const int64_t* scan_ptr, int n_scans, int64_t n, int32_t* out, dcStream_t stream);
int dc_knn_transpose(const int32_t* nbr, int64_t n, int k, int64_t n_dst, int32_t* csr_ptr, int32_t* csr_src, void* ws,
                     size_t ws_bytes, dcStream_t stream);

/* Morton (Z-curve) order of the points over their bounding box: order_out[p] = index of the p-th point.
 * Used once per sequence to lay the global cloud out so that neighbour gathers stay cache-local. */
size_t dc_spatial_order_workspace_bytes(int64_t n);
int dc_spatial_order(const void* points, int stride, int dtype, int64_t n, int32_t* order_out, void* ws,
                     size_t ws_bytes, dcStream_t stream);

/* ---- corrected points: model.py:243-261 (ScaledPolynomial), :181-199 (Polynomial), BaseModel.forward :76-78,
 *      DepthCloud.transform depth_cloud.py:135-152, to_points :122-124, preproc.global_cloud :80-119 ---------
 * d' = model(depth, inc) on points with lmask != 0 (NULL = all), scan s = scan_id[i] (NULL = 0) is moved by
 * poses[s] (fp64 [n_scans, 12] row-major [R|t], NULL = identity), x = vps' + d' dirs'.  vps == NULL means all
 * viewpoints sit at the sensor origin (the usual case for scans in their own frame): 12 B/point less to read.
 * w, e: fp64 device arrays [n_terms] (model weights / exponents).  out_stride 3 or 4 (4 = padded rows).
 * point_fmt: format of points_out -- `dtype` itself, or DC_Q32 (dtype DC_F32, stride 4) with qparams = HOST
 * fp64 [4] {origin.xyz, scale}: fixed-point rows with uniform resolution `scale` (fp32 traffic, ~2^8 finer
 * than fp32 at the range limit; used for the internal buffers of the fused loss).
 * vps_out / dirs_out [n,3], depth_out [n] are optional (NULL).
 * status: device int32 or NULL; bit 0 is raised (never cleared) when a DC_Q32 coordinate does not fit the 32-bit range
 * around `origin` or is NaN -- the row is then stored saturated, so no kernel faults, and the condition stays visible. */
int dc_points_fwd(const void* vps, const void* dirs, const void* depth, const void* inc, const uint8_t* lmask,
                  const int32_t* scan_id, const double* poses, int n_scans, int model_kind, int n_terms,
                  const double* w, const double* e, int64_t n, int dtype, int point_fmt, const double* qparams,
                  int out_stride, void* points_out, void* vps_out, void* dirs_out, void* depth_out, int32_t* status,
                  dcStream_t stream);

/* Basis form of dc_points_fwd for fixed poses and exponents.  Every model is affine in its weights, so
 *   x_j(w) = X0_j + (sum_k w_k c_kj) u_j,  X0 = R (vp + d0 dir) + t (d0 = d' at w = 0),  u = R dir,  c_kj = dd'/dw_k  (0 outside lmask).
 * dtype DC_F32 (float32 clouds, points in DC_Q32): rows_out int32 [n, 6 + n_terms]: per point X0 on the DC_Q32 grid of qparams
 * (status as in dc_points_fwd), then the float32 bits of u and of c_0 .. c_{P-1} (metres per unit weight); 32 bytes for the
 * two-term models = one sector per gathered point.  dtype DC_F64 (float64 clouds, points in DC_F64): rows_out fp64
 * [n, 6 + n_terms] = X0, u, c; qparams unused.  dc_sequence_eval / _step use the rows (dcSequenceDesc.basis) instead of
 * launching dc_points_fwd when neither pose nor exponent gradients are requested. */
int dc_points_basis(const void* vps, const void* dirs, const void* depth, const void* inc, const uint8_t* lmask,
                    const int32_t* scan_id, const double* poses, int n_scans, int model_kind, int n_terms, const double* e,
                    int64_t n, int dtype, const double* qparams, void* rows_out, int32_t* status, dcStream_t stream);

/* Backward of dc_points_fwd for a given dL/dpoints: grads_out fp64 [2*n_terms + 12*n_scans] =
 * {dL/dw, dL/dexponent, dL/d[R|t] per scan}.  partials_ws: fp64 [dc_partial_rows(n) * that count].
 * perm int32 [n] or NULL: point i takes its gradient from row perm[i] of grad_points (the gradient then lives in another
 * point order, e.g. the Morton layout of the fused kernels while the inputs here are scan-major, where a block of 256
 * points contains one or two scans and the per-scan pose reduction is cheap). */
int64_t dc_partial_rows(int64_t n);
int dc_param_grad_count(int n_terms, int n_scans);
int dc_points_bwd(const void* grad_points, const int32_t* perm, int stride, int dtype, int64_t n, const void* vps,
                  const void* dirs, const void* depth, const void* inc, const uint8_t* lmask, const int32_t* scan_id,
                  const double* poses, int n_scans, int model_kind, int n_terms, const double* w, const double* e,
                  int want_exponent_grad, int want_pose_grad, double* partials_ws, double* grads_out,
                  dcStream_t stream);

/* ---- neighbourhood features: DepthCloud.update_features depth_cloud.py:426-433 = update_mean :291-295,
 *      update_weights :356-364, update_cov :366-369 -> utils.covs utils.py:109-149, compute_eig :376-399
 *      (torch.linalg.eigh, CPU-forced in the reference), update_normals :401-415, update_incidence_angles :417-424.
 * All outputs optional (NULL): mean [n,3], cov [n,3,3], eigvals [n,3] ascending, eigvecs [n,3,3] (columns),
 * normals [n,3], inc_angles [n], nvalid int32 [n], weights_out [n,k]; cmean_out [n,3] / invd_out [n] are the
 * saved tensors of dc_features_bwd.  mean_weights [n,k] or NULL (validity); scale <= 0: no Gaussian weights. */
int dc_features_fwd(const void* points, int stride, int dtype, const int32_t* nbr, int64_t n, int k,
                    const void* mean_weights, double scale, const void* dirs, void* mean, void* cov, void* eigvals,
                    void* eigvecs, void* normals, void* inc_angles, int32_t* nvalid, void* weights_out,
                    void* cmean_out, void* invd_out, dcStream_t stream);

/* A-B switch for measurements (process-wide atomic, read once per launch, NOT synchronised with concurrent dc_features_fwd calls
 * of other threads): 0 sends every dc_features_fwd call to the general run-time-k kernel, 1 (default) lets calls with
 * k = 4 / 8 / 10 / 16, validity weights and 16-B aligned arrays take the tiled kernel (features_fwd_tile_kernel: coalesced index
 * tile, k gathers in flight, outputs through LDS).  Same reference lines as dc_features_fwd; results agree to round-off. */
int dc_features_set_tiled(int on);   /* returns the previous setting; changes nothing unless DC_ENABLE_ABLATIONS=1 (see dc_set_option) */

/* dL/dpoints from dL/d(mean, cov, eigvals) (any may be NULL); grec_ws: dtype [n,12] scratch. */
int dc_features_bwd(const void* points, int stride, int dtype, const int32_t* csr_ptr, const int32_t* csr_src,
                    int64_t n, const void* cmean, const void* invd, const int32_t* nvalid, const void* eigvecs,
                    const void* grad_mean, const void* grad_cov, const void* grad_eigvals, void* grec_ws,
                    void* grad_points, dcStream_t stream);

/* ---- block tables: LDS-staged gathers for the two hot kernels -------------------------------------------------
 * The forward gathers `points[neighbors]` (depth_cloud.py:303-304), the backward the records of every centre whose
 * neighbourhood contains the point (autograd's scatter of that index).  A table, built once per neighbourhood set,
 * lists for every block of 256 rows the DISTINCT rows it references (blk_ids[blk_ptr[b] .. blk_ptr[b+1]), ascending)
 * and gives every reference its 16-bit position in that list, slot-major: loc[(slot_ptr[b] + s) * 256 + lane],
 * 0xFFFF = empty slot; block b has slot_ptr[b+1] - slot_ptr[b] slots (the longest list among its rows).
 * The kernels copy the distinct rows of a block into LDS once and gather from LDS: same results bit for bit, ~7x
 * fewer L1 lookups (in Morton order a block's 2560 references at K = 10 hit ~375 rows).
 * A stored position is the BYTE offset of the row's first 16-B piece in the LDS tile, i.e. 16 x (index in the block's
 * distinct list); a block may therefore reference at most 4094 distinct rows (the LDS tile is smaller anyway).
 * References come either as a table ids[n_rows, k] (row_ptr == NULL; forward: the neighbour table) or as CSR lists
 * row_ptr[n_rows + 1], ids[...] (backward: dc_knn_transpose's output); negative ids are empty.
 * Two layouts of the positions:
 *   DC_TABLE_SLOTS  slot-major, as above (dc_block_table_slots + dc_block_table_build).  Exact for a table [rows, k]
 *                   (every block has k slots: the forward's layout); for lists of varying length every block is padded
 *                   to its longest list.
 *   DC_TABLE_RUNS   per row a contiguous run padded to a multiple of FOUR positions (8 B = one trip of the backward's
 *                   edge loop): loc[4 * run_ptr[r] + s], run_ptr int32 [n_rows + 1] in units of runs, 0xFFFF = padding
 *                   (dc_block_table_build_runs).  The backward's layout: 12 instead of 16.2 stored positions per point
 *                   at K = 10, and every lane stops at its own in-degree.
 *   1. dc_block_table_slots  -> slot_ptr int32 [n_blocks + 1] (device); slot_ptr[n_blocks] = number of slot rows, the
 *      host reads it to size loc (uint16 [n_slot_rows * 256]); slot_cnt_ws: int32 [n_blocks] scratch.
 *   2. dc_block_table_build  -> blk_ptr int32 [n_blocks + 1], blk_ids int32 [n_refs] (first blk_ptr[n_blocks] used),
 *      loc, info int32 [4] = {total distinct, max distinct rows of a block, overflow flag (a block with >= 4095
 *      distinct rows: table unusable), 0}.  n_refs = length of ids (table: n_rows * k).
 *   or dc_block_table_build_runs (CSR lists only) -> run_ptr int32 [n_rows + 1], loc uint16 [dc_block_table_run_capacity
 *      (n_rows, n_refs) * 4] (an upper bound known on the host: no synchronisation), blk_ptr, blk_ids, info as above.
 * n_blocks = ceil(n_rows / 256).  ws: dc_block_table_workspace_bytes(n_refs). */
#define DC_TABLE_SLOTS 0
#define DC_TABLE_RUNS 1
typedef struct dcBlockTable {
  const int32_t* blk_ptr;
  const int32_t* blk_ids;
  const int32_t* slot_ptr;    /* DC_TABLE_SLOTS */
  const uint16_t* loc;
  int32_t max_rows;           /* info[1]: sizes the LDS tile */
  int32_t layout;             /* DC_TABLE_SLOTS | DC_TABLE_RUNS */
  const int32_t* run_ptr;     /* DC_TABLE_RUNS */
  const int32_t* own_base;    /* int32 [n_blocks] or NULL (dc_block_table_own_base): position of row 256 b in block b's list when
                                 all of the block's own rows are in it (a k-NN table references every point from its own row),
                                 else -1; lets the forward take each lane's centre point from the staged rows */
  int32_t packed;             /* DC_TABLE_SLOTS: 1 = every row's references fill its slots from 0 upwards without holes (tables built
                                 from CSR lists are; a [rows, K] table with -1 entries between valid ones is not): a wavefront of
                                 the forward kernels may then stop at the first trip of slots that is empty for all its lanes */
  int32_t reserved;
  const int32_t* row_ptr;     /* packed tables whose own_base is >= 0 for EVERY block: int32 [rows + 1], the CSR offsets the table was
                                 built from (row lengths), else NULL.  With it the one-pass evaluation of float32 clouds takes
                                 consistency_step_ragged_q32_kernel (ball neighbourhoods: config.py:187-189) */
} dcBlockTable;
int dc_block_table_slots(const int32_t* row_ptr, int64_t n_rows, int k, int32_t* slot_cnt_ws, int32_t* slot_ptr,
                         dcStream_t stream);
size_t dc_block_table_workspace_bytes(int64_t n_refs);
/* A-B switch (measurements, tests; process-wide, not thread-safe like dc_set_option): 0 = [rows, K] tables through the radix-sort
 * build as well; returns the previous setting.  Default 1: K = 4 / 8 / 10 / 16 tables are built block by block in LDS. */
int dc_block_table_set_lds_build(int on);
/* Workspace of dc_block_table_build for a table [n_rows, k] (row_ptr NULL): the LDS build's scratch rows where it applies, else
 * dc_block_table_workspace_bytes(n_rows * k). */
size_t dc_block_table_slots_workspace_bytes(int64_t n_rows, int k);
int dc_block_table_build(const int32_t* row_ptr, const int32_t* ids, int64_t n_rows, int k, int64_t n_refs,
                         const int32_t* slot_ptr, int64_t n_slot_rows, int32_t* blk_ptr, int32_t* blk_ids, uint16_t* loc,
                         int32_t* info, void* ws, size_t ws_bytes, dcStream_t stream);
int dc_block_table_own_base(const int32_t* blk_ptr, const int32_t* blk_ids, int64_t n_rows, int32_t* own_base, dcStream_t stream);
/* The layout step behind dcSequenceDesc.scan_seg: inside every block of 256 consecutive entries of `order_in` (point indices in the
 * plan's order) the points inside `mask` (uint8 [n] in the ORIGINAL order, or NULL: all) first, by scan id (`scan_id` int32 [n],
 * original order), then those outside, by scan id; stable.  order_out int32 [n] (not order_in), seg_out uint16
 * [ceil(n / 256), 2 n_scans + 1].  n_scans <= 64.  Optional: blk_skip_out uint8 [ceil(n / 256)] <- 1 for blocks without a point inside
 * the mask (dcSequenceDesc.blk_skip), skipped_out int32 [1] <- the number of 64-lane wavefronts with no point inside the mask.
 * (No reference counterpart: the reference keeps scans one after the other.) */
int dc_block_group(const int32_t* order_in, const int32_t* scan_id, const uint8_t* mask, int64_t n, int n_scans, int32_t* order_out,
                   uint16_t* seg_out, uint8_t* blk_skip_out, int32_t* skipped_out, dcStream_t stream);
/* The neighbour table in a new point order (`order` int64 [n], a permutation: new row i = old row order[i]): rank_out[order[i]] = i,
 * nbr_out[i][q] = rank_out[nbr[order[i]][q]], -1 stays -1.  (The layout step of a sequence: Morton order of the global cloud.) */
int dc_table_permute(const int32_t* nbr, int64_t n, int k, const int64_t* order, int32_t* rank_out, int32_t* nbr_out, dcStream_t stream);
/* dst row i = src row order[i] for rows of row_bytes bytes (1, or a multiple of 4): a per-point array into the plan's order. */
int dc_gather_rows(const void* src, int row_bytes, const int64_t* order, int64_t n, void* dst, dcStream_t stream);
int64_t dc_block_table_run_capacity(int64_t n_rows, int64_t n_refs);
int dc_block_table_build_runs(const int32_t* row_ptr, const int32_t* ids, int64_t n_rows, int64_t n_refs, int32_t* run_ptr,
                              int32_t* blk_ptr, int32_t* blk_ids, uint16_t* loc, int32_t* info, void* ws, size_t ws_bytes,
                              dcStream_t stream);

/* ---- fused map-consistency loss: compute_neighborhood_features preproc.py:195-217 + min_eigval_loss
 *      loss.py:216-294 / trace_loss :297-370 + their autograd backward (train.py:300-307) ------------------------
 * Forward: per point the covariance of its neighbourhood, smallest eigenpair, pointwise loss
 *   l = relu(lam0 [/ clamp(sum lam, 1e-6)] - offset) [sqrt], and the 8-element backward record rec [n,8]
 *   (same format as the points: point_fmt / qparams as in dc_points_fwd).
 * sums_out fp64 [2] = {sum of l over mask, number of masked points}; mask u8 [n] or NULL; offset [n] or NULL;
 * pointwise [n], eigvals [n,3] optional.  partials_ws: fp64 [dc_partial_rows(n) * 2].
 * centre_idx int32 [n] or NULL: when given, row r of nbr / rec / pointwise / mask belongs to point centre_idx[r]
 * (a compact list of centres, e.g. only the masked points -- the others contribute nothing to loss or gradient).
 * table: block table of nbr (host struct of device pointers) or NULL; used for stride-4 rows when its LDS tile fits,
 * nbr may then be NULL. */
int dc_consistency_fwd(const void* points, int stride, int dtype, int point_fmt, const double* qparams,
                       const int32_t* nbr, const int32_t* centre_idx, const dcBlockTable* table, int64_t n, int k,
                       const uint8_t* mask, const void* offset, int loss_kind, int normalization, int sqrt_, void* rec,
                       void* pointwise, void* eigvals, double* partials_ws, double* sums_out, dcStream_t stream);

/* Backward of sum-over-mask of l: dL/dx_j gathered over incoming edges, chained in the same kernel to
 * dL/dw, dL/dexponent, dL/d[R|t] (grads_out as in dc_points_bwd; pass dirs == NULL to get only grad_points).
 * grad_points [n, stride] optional.  partials_ws: fp64 [dc_partial_rows(n) * dc_param_grad_count()].
 * lane_perm u8 [256 * ceil(n / 256)] or NULL: for every block of 256 consecutive points a permutation of 0..255
 * (ascending in-degree) telling which point each lane handles -- lanes of a wavefront then walk edge lists of
 * similar length; results do not depend on it.
 * table: block table of (csr_ptr, csr_src) or NULL (as in dc_consistency_fwd; csr_ptr / csr_src may then be NULL;
 * ignored together with lane_perm). */
int dc_consistency_bwd(const void* points, int stride, int dtype, int point_fmt, const double* qparams, const void* rec,
                       const int32_t* csr_ptr, const int32_t* csr_src, const uint8_t* lane_perm, const dcBlockTable* table,
                       int64_t n, const void* vps, const void* dirs,
                       const void* depth, const void* inc, const uint8_t* lmask, const int32_t* scan_id,
                       const double* poses, int n_scans, int model_kind, int n_terms, const double* w, const double* e,
                       int want_exponent_grad, int want_pose_grad, void* grad_points, double* partials_ws,
                       double* grads_out, dcStream_t stream);

/* ---- masks: filters.py:85-113 within_bounds, :184-193 valid neighbours, :196-254 eigenvalue (ratio) bounds,
 *      depth_cloud.py:314-326 dir / vp dispersion, preproc.py:122-164 global_cloud_mask ------------------------- */
/* mask[i] &= lo <= num[i, num_index] (/ den[i, den_index]) <= hi; +-inf = unbounded; NaN fails. */
int dc_mask_bounds(const void* num, int num_stride, int num_index, const void* den, int den_stride, int den_index,
                   int dtype, int64_t n, double lo, double hi, uint8_t* mask, dcStream_t stream);
int dc_valid_count(const int32_t* nbr, int64_t n, int k, int32_t* count_out, dcStream_t stream);
/* All bounds on the columns of ONE array [n, stride] in one pass (preproc.py:57-62: eigenvalue_bounds then
 * eigenvalue_ratio_bounds on cloud.eigvals): mask = (init ? 1 : mask) & AND_b lo[b] <= v[i, num_index[b]] (/ v[i, den_index[b]] when
 * den_index[b] >= 0) <= hi[b]; the arrays of the n_bounds <= 8 bounds are HOST arrays; +-inf = unbounded; NaN fails. */
int dc_mask_bounds_multi(const void* values, int stride, int dtype, int64_t n, int n_bounds, const int32_t* num_index,
                         const int32_t* den_index, const double* lo, const double* hi, int init, uint8_t* mask, dcStream_t stream);
/* cloud[mask] (depth_cloud.py:126-134: every per-point field sliced by one boolean mask): the rows i with mask[i] != 0 of up to 8
 * arrays, in their order, in two launches.  src / dst / row_bytes: HOST arrays of n_fields device pointers and row sizes in bytes
 * (rows of a multiple of 4 bytes are copied by words and need 4-byte aligned arrays); every dst holds n rows.  index_out int32 [n]
 * (optional): the kept row numbers.  *count_out (device int64) = number of kept rows.  ws: dc_compact_rows_workspace_bytes(n). */
size_t dc_compact_rows_workspace_bytes(int64_t n);
int dc_compact_rows(const uint8_t* mask, int64_t n, int n_fields, const void* const* src, void* const* dst, const int32_t* row_bytes,
                    int32_t* index_out, int64_t* count_out, void* ws, size_t ws_bytes, dcStream_t stream);
/* DepthCloud.to_points outside autograd (depth_cloud.py:251-252): points[i] = vps[i or 0] + depth[i] * dirs[i], the product and the sum
 * rounded separately like the reference's two tensor operations (bit-equal to them).  vps_rows 1 or n. */
int dc_to_points(const void* vps, int vps_rows, const void* dirs, const void* depth, int dtype, int64_t n, void* points_out,
                 dcStream_t stream);
/* The online node's first stage in one call (scripts/depth_correction:31-58, preproc.py:44-47): DepthCloud.from_points of the raw rows
 * (no crop, no depth bounds: the node's input has passed them, scripts/depth_correction:42), points = vps + depth * dirs, the scan-shadow
 * mask over the direction neighbourhoods of chord radius shadow_r with angle bounds [shadow_lo, shadow_hi] (dc_shadow_filter), and
 * cloud[mask]: vps_out / dirs_out / points_out [m,3], depth_out [m] of the kept rays in their order (buffers of n rows), *count_out = m
 * (device int64).  The same kernels as the separate calls -- bit-equal results --, launched without the host in between. */
size_t dc_scan_prefilter_workspace_bytes(int64_t n);
int dc_scan_prefilter(const void* points, int stride, int in_dtype, const void* vps, int64_t n, int out_dtype, double shadow_r,
                      double shadow_lo, double shadow_hi, void* vps_out, void* dirs_out, void* depth_out, void* points_out,
                      int64_t* count_out, void* ws, size_t ws_bytes, dcStream_t stream);
/* weights = valid_neighbor_mask().float() (depth_cloud.py:341-343): weights_out[e] = nbr[e] >= 0 over `count` table entries. */
int dc_valid_weights(const int32_t* nbr, int64_t count, float* weights_out, dcStream_t stream);
/* ---- K17: inlier correspondences of a scan pair (train.py:186-193, 202-209; loss.py:440-452) ---------------------------------------
 * dist fp64 [n] / idx int32 [n]: the 1-NN of scan 1's points in scan 2 (dc_knn_build with k = 1 and a query).  threshold_out <-
 * np.quantile(dist[~isnan(dist)], ratio) (linear interpolation, numpy's lerp) found by a radix select on the device -- no sort --,
 * mask_out uint8 [n] <- dist <= threshold, idx_out int32 [n] <- idx[mask] in order (first *count_out valid; count_out device int64).
 * Nothing returns to the host in between.  ws: dc_nn1_corr_workspace_bytes(n). */
size_t dc_nn1_corr_workspace_bytes(int64_t n);
int dc_nn1_corr(const double* dist, const int32_t* idx, int64_t n, double ratio, uint8_t* mask_out, int32_t* idx_out, int64_t* count_out,
                double* threshold_out, void* ws, size_t ws_bytes, dcStream_t stream);
/* out[i] = trace of the weighted covariance of vec[nbr[i]]; weights [n,k] or NULL (validity). */
int dc_dispersion(const void* vec, int dtype, const int32_t* nbr, const void* weights, int64_t n, int k, void* out,
                  dcStream_t stream);

/* ---- pose corrections: eval.create_corrected_poses eval.py:68-82 = T0_s * xyz_axis_angle_to_matrix(delta_s)
 *      (transform.py:68-78; rotation by pytorch3d.transforms.axis_angle_to_matrix: axis-angle -> quaternion with the
 *      small-angle series -> matrix) and its backward, one launch each instead of ~50 tensor ops each way.
 * poses, poses_out, grad_poses: fp64 [n_poses, 16] row-major 4x4; deltas / grad_deltas: fp64 [n_deltas, 6] =
 * (xyz, axis-angle) with n_deltas = n_poses (PoseCorrection.pose) or 1 (common / sequence: one correction applied to
 * every pose, its gradient is the sum over the poses).  Conventions of the reference's autograd graph are kept: zero
 * subgradient of the norm at a zero rotation, gradient only through the taken branch of the small-angle switch. */
int dc_pose_correct_fwd(const double* poses, const double* deltas, int n_poses, int n_deltas, double* poses_out,
                        dcStream_t stream);
int dc_pose_correct_bwd(const double* poses, const double* deltas, int n_poses, int n_deltas, const double* grad_poses,
                        double* grad_deltas, dcStream_t stream);
/* The rest of a training iteration with pose corrections in ONE launch (train.py:300-322 after the loss: loss.backward() through
 * eval.create_corrected_poses, the first pose kept fixed -- train.py:309-311 --, optimizer.step()).  layout 0: `sums` fp64
 * [2 + 2 P + 12 S] is what dc_sequence_eval wrote for the corrected poses `poses_used` [S,16]; the mean loss sums[0] / sums[1] is
 * what train() back-propagates, so every gradient is scaled by 1 / sums[1].  layout 1: `sums` fp64 [1 + 2 P + 12 S] is the output of
 * dc_p2plane_sequence / dc_p2point_sequence (the ICP loss of the sequence and its gradients as they are).  torch.optim.Adam's single-tensor update (step = *step + 1, written
 * back) on the corrections `deltas` [n_deltas, 6] (n_deltas = S: PoseCorrection.pose; 1: sequence) with moments d_m / d_v, and on
 * the weights `w` [P] with w_m / w_v unless w is NULL (validation sequences: only the corrections move).  Then poses_next [S,16] /
 * poses12_next [S,12] <- poses0 corrected by the UPDATED deltas: the poses of the next evaluation (poses_next may be poses_used:
 * it is read first).  record (or NULL): a ring of ring_rows rows of fp64 [2 + 2 P + 12 S | P | 6 n_deltas | 12 S]; row (*step mod
 * ring_rows) <- sums, and the weights, corrections and corrected poses (rows [R|t]) this iteration used -- the slot follows the
 * device counter, so a captured iteration replays into the right one.  totals (or NULL): the loss runs over SEVERAL sequences
 * (eval.py:85-112 divides the sum of the sequences' sums by the sum of their counts; icp_loss averages their losses, loss.py:403) --
 * dc_pose_train_combine's {loss, divisor, dL/dw [P]} over all of them (all-reduced over the ranks when the sequences are sharded,
 * SURVEY 8e): gradients are then scaled by 1 / totals[1], the weights step with totals[2..] (pass w for ONE sequence of the group
 * only).  record_extra (or NULL): n_record_extra further doubles copied behind the other fields of the record row (the joint sums
 * of the iteration's training and validation losses: with sharded sequences every rank's log needs both).  All arrays device fp64. */
int dc_pose_train_finish(const double* sums, int layout, int n_terms, int n_scans, double* w, double* w_m, double* w_v, const double* poses0,
                         double* deltas, double* d_m, double* d_v, int n_deltas, int zero_first, int64_t* step, double lr_w, double lr_d,
                         double beta1, double beta2, double eps, const double* poses_used, double* record, int ring_rows, double* poses_next,
                         double* poses12_next, const double* totals, const double* record_extra, int n_record_extra, dcStream_t stream);
/* totals fp64 [2 + P] <- {sum of outs[i][0], layout 0: sum of outs[i][1] (the counts) / layout 1: n_seq, sum of the sequences'
 * dL/dw}; outs: HOST array of n_seq <= 16 device pointers (the `sums` of every sequence of the loss), fixed order. */
int dc_pose_train_combine(const double* const* outs, int n_seq, int layout, int n_terms, double* totals, dcStream_t stream);
/* The same for the TWO losses of an iteration in one launch (training and validation sequences, train.py:250-270): totals fp64
 * [2][2 + P] <- group a, group b; either group may be empty (zeros: a rank that owns no validation sequence still takes part in the
 * all-reduce of the totals). */
int dc_pose_train_combine2(const double* const* outs_a, int n_a, const double* const* outs_b, int n_b, int layout, int n_terms,
                           double* totals, dcStream_t stream);

/* Scan-shadow filter: filters.filter_shadow_points filters.py:257-309 on the direction neighbourhoods of
 * DepthCloud.update_dir_neighbors depth_cloud.py:217-224 (radius search on the unit directions: dc_radius_*).
 * mask_out[i] = 1 when the angles between (vps_i - x_i) and (x_j - x_i), j in dir_nbr[i, :], all lie in [lo, hi];
 * missing neighbours (-1) count as the angle `fill` (the reference: mean of the bounds); a NaN angle removes the point.
 * vps [n,3], or one row shared by all points (vps_rows = 1).  Arithmetic in `dtype`, torch's operation order. */
int dc_shadow_mask(const void* points, const void* vps, int vps_rows, int dtype, const int32_t* dir_nbr, int64_t n, int k,
                   double lo, double hi, double fill, uint8_t* mask_out, dcStream_t stream);

/* The same mask WITHOUT the direction-neighbour table (the online call pattern, scripts/depth_correction:44-47 ->
 * preproc.local_feature_cloud preproc.py:44-47: update_dir_neighbors + filter_shadow_points, after which the table is dropped):
 * grid over `dirs` with cell edge r (the chord length of the neighbourhood angle, nearest_neighbors.py:13-19) and one walk
 * that evaluates the angles as it meets the neighbours.  Identical mask: it depends on the set of neighbours only.
 * ws: dc_knn_workspace_bytes(n, 0). */
int dc_shadow_filter(const void* points, const void* vps, int vps_rows, const void* dirs, int dtype, int64_t n, double r,
                     double lo, double hi, uint8_t* mask_out, void* ws, size_t ws_bytes, dcStream_t stream);

/* The polynomial models applied to a cloud OUTSIDE the training loop (no gradients): Polynomial.correct_depth / inverse
 * model.py:181-215, ScaledPolynomial.correct_depth / inverse model.py:250-274, as the node calls them
 * (scripts/depth_correction:52).  depth_out[i] = mask[i] ? f(depth[i], sum_k w_k gamma[i]^e_k) : depth[i] with
 * op 0: d - b, 1: d + b, 2: d (1 - b), 3: d / (1 - b); fp64 arithmetic in torch's order, rounded to `dtype` at the end. */
int dc_correct_depth(const void* depth, const void* gamma, const uint8_t* mask, const double* w, const double* exponent,
                     int n_terms, int op, int dtype, int64_t n, void* depth_out, dcStream_t stream);

/* ---- point-to-plane ICP loss: loss.point_to_plane_dist loss.py:406-488 inside icp_loss :373-403 (model(c),
 *      c.transform(pose) :381-386) with precomputed correspondences (train.py:178-210) ------------------------------
 * One scan pair (A, B): idxA / idxB int32 [m] index the local points of scan A / B; poseA / poseB fp64 [12]
 * device [R|t]; normals are the LOCAL normals (rotated by the pose inside).  Points are rounded to fp32 before
 * the distances are formed (loss.py:436-437).  Forward and backward are one kernel:
 *   out fp64 [2 + 2 P + 24] = { sum |nA.(xB-xA)| |nA|, sum |nB.(xA-xB)| |nB|,
 *                               d(sum12+sum21)/dw [P], /dexponent [P], /d[R|t]_A [12], /d[R|t]_B [12] }.
 * partials_ws: fp64 [dc_p2plane_partial_count(m)]. */
int64_t dc_p2plane_partial_count(int64_t m);
int dc_p2plane_pair(const void* vpsA, const void* dirsA, const void* depthA, const void* incA, const uint8_t* lmaskA,
                    const void* normalsA, const void* vpsB, const void* dirsB, const void* depthB, const void* incB,
                    const uint8_t* lmaskB, const void* normalsB, int dtype, const double* poseA, const double* poseB,
                    int model_kind, int n_terms, const double* w, const double* e, const int32_t* idxA,
                    const int32_t* idxB, int64_t m, int want_exponent_grad, int want_pose_grad, double* partials_ws,
                    double* out, dcStream_t stream);

/* The same for a whole sequence of scans in ONE host call (icp_loss loss.py:373-403 loops over consecutive pairs,
 * :391-399, and averages them): pair p adds weight_p * (sum12 + sum21) and its gradients, so with
 * weight_p = 0.5 / (m_p * n_pairs) out[0] is the reference's loss of the sequence.
 *   scans / pairs: HOST arrays of descriptors holding device pointers (idx_a / idx_b index the local points of
 *                  scans scan_a / scan_b, scan_a != scan_b); poses fp64 [n_scans, 12] device;
 *   out fp64 [1 + 2 P + 12 n_scans] = { loss, dloss/dw [P], /dexponent [P], /d[R|t] [n_scans, 12] };
 *   partials_ws fp64 [dc_p2plane_sequence_partial_count(pairs, n_pairs)]: the pairs run 16 to a launch (their descriptors
 *                  travel as kernel arguments; one pair launch and one reduction launch per 16 pairs), the workspace holds the
 *                  block rows of the largest such group. */
typedef struct dcIcpScan {
  const void* vps;            /* [n,3] or NULL (sensor at the origin) */
  const void* dirs;           /* [n,3] */
  const void* depth;          /* [n,1] */
  const void* inc;            /* [n,1] incidence angles (NULL without a model) */
  const uint8_t* lmask;       /* [n] local mask gating the model, or NULL */
  const void* normals;        /* [n,3] local normals */
} dcIcpScan;
typedef struct dcIcpPair {
  int32_t scan_a, scan_b;
  const int32_t* idx_a;       /* [m] */
  const int32_t* idx_b;       /* [m] */
  int64_t m;
  double weight;
} dcIcpPair;
int64_t dc_p2plane_sequence_partial_count(const dcIcpPair* pairs, int n_pairs);
int dc_p2plane_sequence(const dcIcpScan* scans, int n_scans, const dcIcpPair* pairs, int n_pairs, int dtype,
                        const double* poses, int model_kind, int n_terms, const double* w, const double* e,
                        double* partials_ws, double* out, dcStream_t stream);
/* Point-to-point variant (loss.point_to_point_dist loss.py:491-565, what scripts/model_poses_learning_icp optimises and
 * model_poses_learning reports as map accuracy): every correspondence contributes |xB - xA| (points rounded to fp32
 * first, loss.py:524-525); no normals.  out of the pair call: fp64 [2 + 2 P + 24] = {sum |xB - xA|, 0, d/dw, d/dexponent,
 * d/d[R|t]_A, d/d[R|t]_B}; the sequence call as dc_p2plane_sequence (pair weight = 1 / (m * n_pairs), loss.py:553,563),
 * dcIcpScan.normals is ignored. */
int dc_p2point_pair(const void* vpsA, const void* dirsA, const void* depthA, const void* incA, const uint8_t* lmaskA,
                    const void* vpsB, const void* dirsB, const void* depthB, const void* incB, const uint8_t* lmaskB, int dtype,
                    const double* poseA, const double* poseB, int model_kind, int n_terms, const double* w, const double* e,
                    const int32_t* idxA, const int32_t* idxB, int64_t m, double* partials_ws, double* out, dcStream_t stream);
int dc_p2point_sequence(const dcIcpScan* scans, int n_scans, const dcIcpPair* pairs, int n_pairs, int dtype,
                        const double* poses, int model_kind, int n_terms, const double* w, const double* e,
                        double* partials_ws, double* out, dcStream_t stream);

/* ---- an ICP training iteration in ONE launch (round 5; train.py:300-312 with icp_loss, loss.py:373-488) --------------------------
 * dc_p2plane_sequence / dc_p2point_sequence (plane = 1 / 0) with the sums finished inside the same launch -- every block takes a
 * ticket, the one that draws the last sums the rows in fixed order -- and, with fin != NULL, dc_pose_train_finish (layout 1, no
 * totals) run by that same block: adjoint of the pose chain, both Adam updates, next poses, record.  For sequences of at most 16
 * pairs with at least one correspondence (DC_ERR_UNSUPPORTED otherwise: issue the separate calls).  ticket: device int32 [17], zero
 * before the first call; the launch leaves it zero.  partials_ws: dc_p2plane_sequence_partial_count doubles.  fin's fields are dc_pose_train_finish's arguments of the same names
 * (poses_used [S,16] may be poses_next: it is read first; poses12_next may be `poses`: every block has read it by then). */
typedef struct dcPoseTrainStep {
  double *w, *w_m, *w_v;
  const double* poses0;
  double *deltas, *d_m, *d_v;
  int32_t n_deltas, zero_first;
  int64_t* step;
  double lr_w, lr_d, beta1, beta2, eps;
  const double* poses_used;
  double* record;
  int32_t ring_rows, n_record_extra;
  const double* record_extra;
  double* poses_next;
  double* poses12_next;
} dcPoseTrainStep;
int dc_icp_sequence_step(int plane, const dcIcpScan* scans, int n_scans, const dcIcpPair* pairs, int n_pairs, int dtype,
                         const double* poses, int model_kind, int n_terms, const double* w, const double* e, double* partials_ws,
                         double* out, int32_t* ticket, const dcPoseTrainStep* fin, dcStream_t stream);


/* Quantile-inlier gating for the fused path (min_eigval_loss / trace_loss with inlier_ratio < 1 or inlier_max_loss,
 * loss.py:256-277).  raw_pointwise [n] (dtype): the forward's loss before relu / sqrt (dc_consistency_fwd with
 * loss_kind | DC_LOSS_RAW_POINTWISE); threshold: device fp64 scalar the caller derived from it (multiplier x quantile of
 * the masked values, min with the given maximum).  Masked centres with raw loss > threshold (or NaN) are dropped: the
 * coefficients of their record rec [n,8] (point_fmt) are zeroed, so dc_consistency_bwd passes nothing through them;
 * sums_out fp64 [2] = {sum of relu(raw) [sqrt] over the inliers, number of inliers}.  partials_ws as dc_consistency_fwd. */
int dc_consistency_gate(const void* raw_pointwise, int dtype, int point_fmt, const uint8_t* mask, int64_t n,
                        const double* threshold, int sqrt_, void* rec, double* partials_ws, double* sums_out,
                        dcStream_t stream);

/* ---- whole-sequence evaluation + optimiser step (train.py:220-312 per-iteration body for one sequence) ----------
 * The caller fills a descriptor with the device arrays of one sequence (SequencePlan in Python) once; every
 * iteration is then ONE host call that launches dc_points_fwd, dc_consistency_fwd and dc_consistency_bwd -- or, with
 * dcSequenceDesc.basis set and no pose / exponent gradient requested, the basis form: no pass over the points, and for up
 * to three weights a single kernel that returns the loss AND dL/dw (second sweep over each centre's own neighbours; no
 * backward record, no transposed table).
 * x [n,4] / rec [n,8] in point_fmt, partials fp64 [dc_sequence_partials_count(n, P, S)] are scratch. */
/* ---- pose mode in one pass (round 4; train.py:300-312 with pose corrections, eval.py:68-82) ---------------------------------
 * dcPoseTable: per 256-point block, derived once from the forward block table of a [rows, K] neighbour table (K = 4 / 8 / 10 / 16):
 * its distinct rows listed by (scan, id), the references' positions in that order, every point's own position, where each
 * scan's rows start and the scan of every listed row.  dc_pose_table_build fills caller-allocated arrays: ids int32
 * [blk_ptr[blocks]], loc uint16 [blocks * K * 256], own_pos uint16 [n], row_seg uint16 [blocks * (n_scans + 1)], row_scan uint8
 * [blk_ptr[blocks]]; info int32 [1] <- 1 when some block cannot take the pose kernel (more than 512 distinct rows, or a block
 * whose list misses one of its own rows). */
typedef struct dcPoseTable {
  const int32_t* blk_ptr;          /* the forward table's */
  const int32_t* ids;
  const uint16_t* loc;
  const uint16_t* own_pos;
  const uint16_t* row_seg;
  const uint8_t* row_scan;
} dcPoseTable;
int dc_pose_table_build(const dcBlockTable* fwd, const int32_t* scan_id, int64_t n, int n_scans, int k, int32_t* ids_out,
                        uint16_t* loc_out, uint16_t* own_pos, uint16_t* row_seg, uint8_t* row_scan, int32_t* info, dcStream_t stream);
/* Pose-independent rows of the rays (sensor frame, viewpoints at the sensor origin): {d0, dir, c_k = dd'/dw_k} as 6 float32 words
 * per row for float32 clouds and models of one or two weights (model.py:113-349 are affine in their weights); valid for the
 * exponents `e`.  Stored per block of `table`, in the order of its list: rows_out [blk_ptr[blocks], 6], block b's row t at
 * blk_ptr[b] + t -- the pose kernel stages them as one contiguous stream. */
int dc_points_local_basis(const void* dirs, const void* depth, const void* inc, const uint8_t* lmask, const int32_t* scan_id,
                          int model_kind, int n_terms, const double* e, int64_t n, int dtype, const dcPoseTable* table, void* rows_out,
                          dcStream_t stream);

typedef struct dcSequenceDesc {
  int64_t n;
  int32_t k, n_scans, dtype, point_fmt;
  double qparams[4];
  const void *vps, *dirs, *depth, *inc;
  const uint8_t* lmask;
  const int32_t* scan_id;
  const int32_t *nbr, *csr_ptr, *csr_src;
  const uint8_t* mask;
  const uint8_t* lane_perm;
  const int32_t* centre_idx; /* optional compact centre list (see dc_consistency_fwd); nbr / rec / mask then have n_centres rows */
  int64_t n_centres;
  void *x, *rec;
  double* partials;
  int32_t model_kind, n_terms, loss_kind, normalization, sqrt_, reserved;
  const dcBlockTable* fwd_table;   /* block table of nbr, or NULL */
  const dcBlockTable* bwd_table;   /* block table of (csr_ptr, csr_src), or NULL */
  int32_t* status;                 /* device int or NULL: bit 0 raised by dc_points_fwd when a DC_Q32 coordinate overflowed /
                                      was NaN, bit 1 by a chained launch whose wait for its weights ran out (dc_set_option 5);
                                      while it is non-zero the evaluation's loss (out[0]) is NaN */
  const void* basis;               /* basis rows of dc_points_basis, valid FOR THE POSES AND EXPONENTS OF THE CALL, or NULL:
                                      x = X0 + (sum_k w_k c_k) u, so an evaluation needs no pass over the points */
  int64_t partials_count;          /* doubles behind `partials`: at least dc_sequence_partials_count(n, n_terms, n_scans), checked by
                                      every call (DC_ERR_WORKSPACE) -- the chained steps keep their rows behind the ordinary columns */
  const uint16_t* scan_seg;        /* [ceil(n / 256), 2 n_scans + 1] or NULL.  Non-NULL promises that the points of every 256-point
                                      block are grouped: first those inside `mask`, by scan id, then those outside, by scan id;
                                      segment v of block b (v < S: inside, scan v; v >= S: outside, scan v - S) is its lanes
                                      scan_seg[b (2 S + 1) + v] .. scan_seg[b (2 S + 1) + v + 1].  Pose gradients (train.py:300-312
                                      with pose corrections) then sum per scan over those static ranges instead of counting-sorting
                                      the block's lanes at run time (n_scans <= 64); and the masked-out points of a block fill whole
                                      wavefronts at its end, which the one-pass kernels skip */
  const uint8_t* blk_skip;         /* [ceil(n / 256)] or NULL: 1 = none of the block's centres is inside `mask` (the plan's layout collects
                                      masked-out points into blocks of their own).  The one-pass evaluation skips such a block -- a
                                      centre outside the mask adds nothing to the loss, the count or dL/dw; its points remain
                                      neighbours of others like any other point */
  int32_t fwd_rows_active;         /* with blk_skip: the longest distinct-row list among the blocks that are NOT skipped (sizes the
                                      one-pass kernels' LDS tile; 0 = use fwd_table->max_rows) */
  int32_t reserved2;
  const dcPoseTable* pose_table;   /* with local_basis: evaluations that ask for pose gradients (and no exponent gradients) of a float32
                                      sequence run as ONE launch (consistency_step_pose_kernel) instead of dc_points_fwd + forward +
                                      backward; NULL: the three-kernel path */
  const void* local_basis;         /* dc_points_local_basis rows, valid FOR THE EXPONENTS OF THE CALL, or NULL */
  const dcBlockTable* fwd_table_loss; /* or NULL: block table of `nbr` in which the rows of the points OUTSIDE `mask` keep only their
                                      reference to themselves.  A centre outside the mask adds nothing to the loss, the count or
                                      dL/dw (loss.py:283-284 indexes the pointwise loss by the mask), so the one-pass evaluation
                                      -- which never produces per-point outputs -- stages only the rows the centres inside the mask
                                      gather: shorter lists per block.  Every other evaluation keeps reading fwd_table */
  int32_t fwd_rows_active_loss;    /* fwd_rows_active of fwd_table_loss */
  int32_t reserved3;
} dcSequenceDesc;

/* Doubles of dcSequenceDesc.partials for a sequence of n points evaluated with up to n_terms weights and n_scans poses:
 * dc_partial_rows(n) * (2 + 2 n_terms + 12 n_scans) for ordinary evaluations plus the two row buffers of chained steps. */
int64_t dc_sequence_partials_count(int64_t n, int n_terms, int n_scans);

/* out fp64 [2 + 2 P + 12 S] = {sum of pointwise loss over mask, mask count, d(sum)/dw, /dexponent, /d[R|t]};
 * w, e, poses: device fp64 (model weights [P], exponents [P], poses [S,12]). */
int dc_sequence_eval(const dcSequenceDesc* desc, const double* w, const double* e, const double* poses, int want_grad,
                     int want_exponent_grad, int want_pose_grad, double* out, dcStream_t stream);
/* The same followed by torch.optim.Adam's step on w (dc_adam_step semantics, grad = grad_scale * d(sum)/dw) inside the
 * final reduction kernel: one host call and two to four launches per optimisation step of a single sequence (train.py:220-312).
 * out as in dc_sequence_eval (exponent / pose gradients are not requested). */
int dc_sequence_step(const dcSequenceDesc* d, double* w, const double* e, const double* poses, double* exp_avg,
                     double* exp_avg_sq, int64_t step, double grad_scale, double lr, double beta1, double beta2, double eps,
                     double weight_decay, double* out, dcStream_t stream);

/* A CHAIN of optimisation steps of one sequence with one launch per step instead of two (basis form, up to three weights;
 * DC_ERR_UNSUPPORTED otherwise -- step with dc_sequence_step then).  The dependent reduction launch after
 * the evaluation kernel costs ~9 us, an eighth of a C2 step; here the launch of step t also FINISHES step t - 1: its first
 * blocks sum the previous evaluation's partial rows, write them to out_prev ({sum loss, count, dL/dw, 0...} of evaluation
 * t - 1) and take its Adam update, while the other blocks fetch what does not depend on the weights and then wait for them
 * (bounded; a wait that never ends yields NaN sums).  step: this evaluation's number (1, 2, ...; its parity selects the flag
 * and the partial-row buffer), has_prev: evaluation step - 1 is still unfinished (0 for the first launch and after a flush);
 * ready: device int32 [16] (64 bytes, 8-byte aligned), zeroed once -- the leading blocks publish the launch's weights there,
 * every word stamped with `step`, and the other blocks pick them up in one trip; dcSequenceDesc.partials must hold dc_partial_rows(n) * (2 + 2 P + 12 S + 2 (2 + P))
 * doubles (the chain's two row buffers sit behind the columns of ordinary evaluations).  dc_sequence_chain_flush finishes evaluation `step` with the ordinary reduction
 * launch (sums -> out, Adam update `step`): the chain's last call. */
int dc_sequence_step_chained(const dcSequenceDesc* d, double* w, const double* e, const double* poses, double* exp_avg,
                             double* exp_avg_sq, int64_t step, int has_prev, double grad_scale, double lr, double beta1, double beta2,
                             double eps, double weight_decay, int32_t* ready, double* out_prev, dcStream_t stream);
/* dc_sequence_step_chained that also records, next to the previous evaluation's sums, the weights that evaluation used
 * (w_used_prev fp64 [n_terms] or NULL; written only when has_prev): what a training log needs per iteration -- loss, gradient
 * and the parameters they belong to (train.py:219-244) -- without a copy launch between the steps of a chain. */
int dc_sequence_step_chained_rec(const dcSequenceDesc* d, double* w, const double* e, const double* poses, double* exp_avg,
                                 double* exp_avg_sq, int64_t step, int has_prev, double grad_scale, double lr, double beta1, double beta2,
                                 double eps, double weight_decay, int32_t* ready, double* out_prev, double* w_used_prev,
                                 dcStream_t stream);
int dc_sequence_chain_flush(const dcSequenceDesc* d, double* w, double* exp_avg, double* exp_avg_sq, int64_t step, double grad_scale,
                            double lr, double beta1, double beta2, double eps, double weight_decay, double* out, dcStream_t stream);
/* A chain over the SEVERAL sequences of one loss on one GPU (train.py:172-175 loops over them, eval.py:85-112 pools their sums and
 * counts): one launch per sequence and step and nothing else.  Launch (step, i) evaluates sequence i -- its rows go to buffer
 * `step & 1` of ITS descriptor -- and first finishes the launch before it by summing that launch's rows (d_prev, buffer prev_parity:
 * sequence i - 1 of this step, or the last sequence of step - 1) onto the running sums acc_in (device fp64 [2 + P]; NULL for the
 * step's second launch).  finish: 0 = nothing is pending (the chain's very first launch), 1 = the sums go on into out_prev (which may
 * be acc_in itself), 2 = they complete step - 1: out_prev <- its totals, torch.optim.Adam's update step - 1 on w, w_used_prev.
 * stamp: a number that grows with every launch of the chain (marks the published weights); ready as for dc_sequence_step_chained.
 * dc_sequence_chain_flush_linked finishes the chain's last launch on its own (one small launch): d_prev's rows + acc_in -> out, Adam
 * update `step`. */
int dc_sequence_step_linked(const dcSequenceDesc* d, const dcSequenceDesc* d_prev, int prev_parity, int finish, const double* acc_in,
                            double* w, const double* e, const double* poses, double* exp_avg, double* exp_avg_sq, int64_t step,
                            int64_t stamp, double grad_scale, double lr, double beta1, double beta2, double eps, double weight_decay,
                            int32_t* ready, double* out_prev, double* w_used_prev, dcStream_t stream);
int dc_sequence_chain_flush_linked(const dcSequenceDesc* d_prev, int prev_parity, const double* acc_in, double* w, double* exp_avg,
                                   double* exp_avg_sq, int64_t step, int64_t stamp, double grad_scale, double lr, double beta1, double beta2,
                                   double eps, double weight_decay, int32_t* ready, double* out, dcStream_t stream);
/* The same idea when several sequences / ranks share the weights (an all-reduce of the sums sits between an evaluation and
 * its update, so a launch cannot sum the previous rows itself): evaluation `step`, whose leading blocks first take Adam update
 * step - 1 from grad_sum (device fp64 [P]: the previous evaluation's dL/dw summed over all ranks; NULL for the first call),
 * followed by the ordinary reduction of THIS evaluation into out -- three launches per multi-rank step (evaluation,
 * reduction, all-reduce) instead of four (+ dc_adam_step).  The last update of a loop is a plain dc_adam_step.
 * w_used (device fp64 [n_terms] or NULL) <- the weights THIS evaluation uses, i.e. after update step - 1: with `out` the record a
 * training log keeps per iteration (train.py:219-244), written by the launches themselves. */
int dc_sequence_eval_after_update(const dcSequenceDesc* d, double* w, const double* e, const double* poses, double* exp_avg,
                                  double* exp_avg_sq, int64_t step, const double* grad_sum, double grad_scale, double lr, double beta1,
                                  double beta2, double eps, double weight_decay, int32_t* ready, double* out, double* w_used,
                                  dcStream_t stream);

/* torch.optim.Adam step (train.py:139-149,312) on a device fp64 vector; grad is multiplied by grad_scale first
 * (1 / number of masked points of all sequences = the reference's mean reduction, loss.py:205-213). */
int dc_adam_step(double* param, const double* grad, double* exp_avg, double* exp_avg_sq, int64_t n, int64_t step,
                 double grad_scale, double lr, double beta1, double beta2, double eps, double weight_decay,
                 dcStream_t stream);
/* The same with the step counter on the device: step (device int64, 0 before the first update) is read, t = step + 1 enters
 * the bias corrections, and t is written back -- capturable into a hipGraph (dc_adam_step / dc_sequence_step take the step
 * from the host: a captured launch would replay one and the same step number).  One block; meant for the small tensors of
 * this path (model weights, pose corrections). */
int dc_adam_step_device(double* param, const double* grad, double* exp_avg, double* exp_avg_sq, int64_t n, int64_t* step,
                        double grad_scale, double lr, double beta1, double beta2, double eps, double weight_decay,
                        dcStream_t stream);

/* ---- voxel-grid pre-filter (next row): filters.filter_grid filters.py:24-82 -------------------------------
 * One survivor per voxel of edge grid_res with the reference's dict semantics: points are offered in the sequence
 * seq (int32 [n], NULL = 0..n-1; reversed for keep='first', numpy's seeded shuffle for keep='random'), the LAST one
 * offered to a voxel survives; survivors come out in order of their voxel's FIRST appearance (preserve_order: by index).
 * out_idx int32 [n] (first *count_out valid), count_out / status_out device int32 (status != 0: voxel range exceeds
 * the 3 x 21-bit key, fall back to the host).  ws: dc_voxel_filter_workspace_bytes(n). */
size_t dc_voxel_filter_workspace_bytes(int64_t n);
int dc_voxel_filter(const void* points, int stride, int dtype, int64_t n, double grid_res, const int32_t* seq,
                    int preserve_order, int32_t* out_idx, int32_t* count_out, int32_t* status_out, void* ws, size_t ws_bytes,
                    dcStream_t stream);

/* ---- scan files -> DepthCloud source fields on the device ------------------------------------------
 * points: the uploaded raw rows [n, stride] (stride >= 3; KITTI-360 .bin: float32 [n,4] x,y,z,intensity,
 * datasets/kitti360.py:96-99; ASL / FEE-corridor arrays: [n,3]); vps [n,3] in the same dtype or NULL (sensor origin).
 * One flag kernel + a stable compaction apply, in the reference's order,
 *   the ego-vehicle crop  keep = x < -d | x > d | y < -d | y > d   (kitti360.py:101-105; ego_box <= 0: off),
 *   the depth pre-filter  min_depth <= |p - vp| <= max_depth in the raw dtype (filters.filter_depth filters.py:116-141;
 *                         NaN / -inf / +inf: unbounded),
 *   DepthCloud.from_points depth_cloud.py:592-638 in out_dtype: depth = |p - vp|, dirs = (p - vp) / depth, rays of zero
 *                         depth left un-normalised (:626-627),
 * and write dirs_out [m,3], depth_out [m], vps_out [m,3] (optional), index_out int32 [m] = kept source rows (optional),
 * *count_out = m (device int64).  Outputs must hold n rows.  ws: dc_cloud_from_points_workspace_bytes(n).  With neither crop nor
 * bounds every row is kept and one kernel writes the fields (vps_out of a NULL vps: zeros). */
size_t dc_cloud_from_points_workspace_bytes(int64_t n);
int dc_cloud_from_points(const void* points, int stride, int in_dtype, const void* vps, int64_t n, double ego_box,
                         double min_depth, double max_depth, int out_dtype, void* dirs_out, void* depth_out, void* vps_out,
                         int32_t* index_out, int64_t* count_out, void* ws, size_t ws_bytes, dcStream_t stream);

/* Tuning / ablation switches (process-wide atomics, read once per launch; results are identical either way).
 * option 0: value 1 makes the fused kernels ignore block tables and gather from global memory.
 * option 1: value 1 makes dc_consistency_fwd use the run-time slot loop instead of the kernels specialised for
 *           k = 4 / 8 / 10 / 16.
 * option 3: value 1 makes dc_sequence_eval / _step ignore dcSequenceDesc.basis (general path).
 * option 4: value 1 makes basis-form evaluations run the forward and the backward kernel separately instead of the one-pass
 *           loss + dL/dw kernel.
 * option 7: value 1 sends pose-gradient evaluations through the three-kernel general path even when dcSequenceDesc.pose_table is set.
 * option 8: value 1 makes every launch of a chain of fixed-K steps walk the blocks forwards; by default every other launch walks
 *           each XCD's share backwards, so that its first blocks find their rows in that XCD's L2 (same sums to rounding).
 * option 5: number of polls a chained launch's blocks make while they wait for the weights its leading blocks publish
 *           (default 2^22, negative restores it).  A wait that runs out yields NaN sums for that evaluation and raises bit 1 of
 *           dcSequenceDesc.status (bit 0: DC_Q32 overflow), so the two causes of a NaN loss can be told apart; tests force 0. */
/* NOT thread-safe against launches: the switches are process-wide (one value for every plan and stream).  They exist for A-B
 * measurements and tests in a process that does nothing else meanwhile; a thread flipping one while another thread evaluates gets
 * either variant for that launch (each launch reads them once; results are the same to the stated tolerances, timings are not).
 * Product code never calls dc_set_option -- and cannot: the switches (and dc_knn_set_shell_budget / dc_knn_set_fine_cell_count) return
 * DC_ERR_UNSUPPORTED unless the process had DC_ENABLE_ABLATIONS=1 in its environment when the library was first used (tests/conftest.py,
 * bench.py's ablation extras and the tools under tools/ set it); without it the library has no mutable process-wide state. */
int dc_set_option(int option, int value);

/* ---- kernel timer: when enabled, dc_points_fwd / dc_consistency_fwd / dc_consistency_bwd (kinds 0 / 1 / 2)
 * and dc_features_fwd (kind 3) bracket their main kernel with HIP events on the launch stream; dc_profiler_read waits and sums them.
 * every: 0 = off, N >= 1 = time every N-th launch of each kind (an event pair costs a few microseconds of idle GPU
 * around the kernel, so a timed production loop samples); `launches` counts the timed launches.  The timer state is
 * one per process behind a mutex: entry points may be called from several host threads on distinct streams. */
int dc_profiler_enable(int every);
int dc_profiler_reset(void);
int dc_profiler_read(int kind, double* total_ms, int64_t* launches);
/* Source-level name of the kernel instantiation the last launch of `kind` used (which variant the dispatch chose). */
int dc_profiler_kernel(int kind, char* buf, int len);

#ifdef __cplusplus
}
#endif
#endif /* DC_HIP_H */
