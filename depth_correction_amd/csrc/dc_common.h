// Shared definitions for the depth_correction_amd HIP library (gfx950 / MI355X only).
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DC_HD __host__ __device__ __forceinline__
#else
// Host-only build of the per-point math (csrc/dc_hostcheck.cpp): lets CPU tests exercise the very
// same inline functions the kernels call.  Never part of the product library.
#define DC_HD inline
#endif

// Status codes of the C ABI: 0 ok, <0 invalid argument, >0 hipError_t value.
#define DC_OK 0
#define DC_ERR_ARG (-1)
#define DC_ERR_DTYPE (-2)
#define DC_ERR_WORKSPACE (-3)
#define DC_ERR_UNSUPPORTED (-4)
#define DC_ERR_BACKWARD_TABLES (-5) /* dc_sequence_eval / _step: this evaluation needs dcSequenceDesc.csr_ptr / csr_src (and bwd_table) */

#define DC_F32 0
#define DC_F64 1
// Internal point format of the fused path: 3 x int32 fixed point (+1 pad) = 16 B rows, x = origin + q * scale.
// Same traffic as padded fp32, 2^8 finer resolution at the range limit (uniform absolute error).
#define DC_Q32 2

// Loss kinds / model kinds (mirrors the reference's names: loss.py:216,297 ; model.py:149,218).
#define DC_LOSS_MIN_EIGVAL 0
#define DC_LOSS_TRACE 1
#define DC_LOSS_RAW_POINTWISE 0x100 /* OR-ed into loss_kind of dc_consistency_fwd: `pointwise` receives the loss before relu / sqrt */
/* OR-ed into loss_kind (dc_consistency_fwd, dcSequenceDesc.loss_kind): the reduction drops NaN (skip_nans) / non-finite (only_finite)
 * pointwise losses -- neither the sum nor the count nor any gradient sees them (loss.py:125-137) */
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
