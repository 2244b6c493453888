"""ctypes binding of libdc_hip.so (include/dc_hip.h) on torch device tensors.

Thin and explicit: every wrapper checks shapes / dtypes / devices on the host (a kernel that reads
out of bounds can take the whole GPU down), passes raw device pointers plus torch's current HIP
stream, and raises on a non-zero status.  There is NO fallback: if the library cannot be loaded the
first call raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes
import functools
import os

import torch

__all__ = ['lib', 'lib_path', 'on_device', 'DC_F32', 'DC_F64', 'DC_Q32', 'LOSS_KINDS', 'MODEL_KINDS']

DC_F32, DC_F64, DC_Q32 = 0, 1, 2
DC_TABLE_SLOTS, DC_TABLE_RUNS = 0, 1
LOSS_KINDS = {'min_eigval_loss': 0, 'trace_loss': 1}
DC_LOSS_RAW_POINTWISE, DC_LOSS_SKIP_NANS, DC_LOSS_ONLY_FINITE = 0x100, 0x200, 0x400      # OR-ed into a loss kind (include/dc_hip.h)
LOSS_RAW_POINTWISE = 0x100
DC_ERR_ARG, DC_ERR_DTYPE, DC_ERR_WORKSPACE, DC_ERR_UNSUPPORTED = -1, -2, -3, -4       # include/dc_hip.h / csrc/dc_common.h
DC_ERR_BACKWARD_TABLES = -5
MODEL_KINDS = {None: 0, 'BaseModel': 0, 'Polynomial': 1, 'ScaledPolynomial': 2, 'Linear': 3, 'InvCos': 4, 'ScaledInvCos': 5}
MAX_MODEL_TERMS = 8

_LIB = None
_vp, _i32, _i64, _f64, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_size_t

_SIGNATURES = {
    'dc_version': (ctypes.c_int, []),
    'dc_knn_workspace_bytes': (_sz, [_i64, _i64]),
    'dc_knn_set_shell_budget': (_i32, [_i32]),
    'dc_knn_set_fine_cell_count': (_i32, [_i32]),
    'dc_knn_build': (_i32, [_vp, _i32, _i32, _i64, _vp, _i32, _i64, _i32, _f64, _f64, _vp, _vp, _vp, _sz, _vp]),
    'dc_knn_build_i64': (_i32, [_vp, _i32, _i32, _i64, _vp, _i32, _i64, _i32, _f64, _f64, _vp, _vp, _vp, _vp, _sz, _vp]),
    'dc_radius_count': (_i32, [_vp, _i32, _i32, _i64, _f64, _vp, _vp, _vp, _sz, _vp]),
    'dc_radius_fill': (_i32, [_i64, _f64, _i32, _vp, _vp, _sz, _vp]),
    'dc_radius_count_query': (_i32, [_vp, _i32, _i32, _i64, _vp, _i32, _i64, _f64, _vp, _vp, _vp, _sz, _vp]),
    'dc_radius_fill_query': (_i32, [_i64, _i64, _f64, _i32, _vp, _vp, _sz, _vp]),
    'dc_knn_transpose_workspace_bytes': (_sz, [_i64, _i32]),
    'dc_knn_transpose': (_i32, [_vp, _i64, _i32, _i64, _vp, _vp, _vp, _sz, _vp]),
    'dc_spatial_order_workspace_bytes': (_sz, [_i64]),
    'dc_spatial_order': (_i32, [_vp, _i32, _i32, _i64, _vp, _vp, _sz, _vp]),
    'dc_points_fwd': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _i64, _i32, _i32, _vp, _i32,
                             _vp, _vp, _vp, _vp, _vp, _vp]),
    'dc_consistency_gate': (_i32, [_vp, _i32, _i32, _vp, _i64, _vp, _i32, _vp, _vp, _vp, _vp]),
    'dc_points_basis': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    'dc_partial_rows': (_i64, [_i64]),
    'dc_param_grad_count': (_i32, [_i32, _i32]),
    'dc_sequence_partials_count': (_i64, [_i64, _i32, _i32]),
    'dc_points_bwd': (_i32, [_vp, _vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32,
                             _i32, _vp, _vp, _vp]),
    'dc_features_fwd': (_i32, [_vp, _i32, _i32, _vp, _i64, _i32, _vp, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp, _vp]),
    'dc_features_set_tiled': (_i32, [_i32]),
    'dc_features_bwd': (_i32, [_vp, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'dc_block_table_slots': (_i32, [_vp, _i64, _i32, _vp, _vp, _vp]),
    'dc_block_table_workspace_bytes': (_sz, [_i64]),
    'dc_block_table_build': (_i32, [_vp, _vp, _i64, _i32, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'dc_block_group': (_i32, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    'dc_pose_table_build': (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'dc_points_local_basis': (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i64, _i32, _vp, _vp, _vp]),
    'dc_pose_train_finish': (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _f64, _f64, _f64, _f64, _f64,
                                    _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp]),
    'dc_pose_train_combine': (_i32, [_vp, _i32, _i32, _i32, _vp, _vp]),
    'dc_pose_train_combine2': (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp]),
    'dc_table_permute': (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    'dc_gather_rows': (_i32, [_vp, _i32, _vp, _i64, _vp, _vp]),
    'dc_points_extent_workspace_bytes': (_sz, []),
    'dc_points_extent': (_i32, [_vp, _i32, _i32, _i64, _vp, _vp, _sz, _vp]),
    'dc_scan_ids': (_i32, [_vp, _i32, _i64, _vp, _vp]),
    'dc_scan_lattice_workspace_bytes': (_sz, [_i32]),
    'dc_scan_lattice_shift': (_i32, [_vp, _i32, _i64, _vp, _i32, _vp, _vp, _vp, _sz, _vp]),
    'dc_scan_lattice_localize': (_i32, [_vp, _i64, _i32, _vp, _i32, _vp, _vp]),
    'dc_block_table_set_lds_build': (_i32, [_i32]),
    'dc_block_table_slots_workspace_bytes': (_sz, [_i64, _i32]),
    'dc_block_table_own_base': (_i32, [_vp, _vp, _i64, _vp, _vp]),
    'dc_block_table_run_capacity': (_i64, [_i64, _i64]),
    'dc_block_table_build_runs': (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'dc_consistency_fwd': (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp,
                                  _vp, _vp, _vp, _vp]),
    'dc_consistency_bwd': (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                  _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    'dc_mask_bounds': (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _i64, _f64, _f64, _vp, _vp]),
    'dc_valid_count': (_i32, [_vp, _i64, _i32, _vp, _vp]),
    'dc_mask_bounds_multi': (_i32, [_vp, _i32, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    'dc_compact_rows_workspace_bytes': (_sz, [_i64]),
    'dc_compact_rows': (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'dc_to_points': (_i32, [_vp, _i32, _vp, _vp, _i32, _i64, _vp, _vp]),
    'dc_valid_weights': (_i32, [_vp, _i64, _vp, _vp]),
    'dc_scan_prefilter_workspace_bytes': (_sz, [_i64]),
    'dc_scan_prefilter': (_i32, [_vp, _i32, _i32, _vp, _i64, _i32, _f64, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'dc_nn1_corr_workspace_bytes': (_sz, [_i64]),
    'dc_nn1_corr': (_i32, [_vp, _vp, _i64, _f64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'dc_dispersion': (_i32, [_vp, _i32, _vp, _vp, _i64, _i32, _vp, _vp]),
    'dc_sequence_eval': (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    'dc_sequence_step': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _f64, _f64, _f64, _f64, _f64, _f64, _vp, _vp]),
    'dc_sequence_step_chained': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f64, _f64, _f64, _f64, _f64, _f64, _vp, _vp, _vp]),
    'dc_sequence_step_chained_rec': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f64, _f64, _f64, _f64, _f64, _f64, _vp, _vp, _vp, _vp]),
    'dc_sequence_eval_after_update': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _f64, _f64, _f64, _f64, _f64, _f64, _vp, _vp, _vp, _vp]),
    'dc_sequence_step_linked': (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _f64, _f64, _f64, _f64, _f64, _f64, _vp, _vp,
                                       _vp, _vp]),
    'dc_sequence_chain_flush_linked': (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _i64, _i64, _f64, _f64, _f64, _f64, _f64, _f64, _vp, _vp, _vp]),
    'dc_sequence_chain_flush': (_i32, [_vp, _vp, _vp, _vp, _i64, _f64, _f64, _f64, _f64, _f64, _f64, _vp, _vp]),
    'dc_adam_step': (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _f64, _f64, _f64, _f64, _f64, _f64, _vp]),
    'dc_adam_step_device': (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _f64, _f64, _f64, _f64, _f64, _f64, _vp]),
    'dc_voxel_filter_workspace_bytes': (_sz, [_i64]),
    'dc_voxel_filter': (_i32, [_vp, _i32, _i32, _i64, _f64, _vp, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    'dc_cloud_from_points_workspace_bytes': (_sz, [_i64]),
    'dc_cloud_from_points': (_i32, [_vp, _i32, _i32, _vp, _i64, _f64, _f64, _f64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'dc_set_option': (_i32, [_i32, _i32]),
    'dc_profiler_enable': (_i32, [_i32]),
    'dc_profiler_reset': (_i32, []),
    'dc_profiler_read': (_i32, [_i32, _vp, _vp]),
    'dc_profiler_kernel': (_i32, [_i32, _vp, _i32]),
    'dc_p2plane_partial_count': (_i64, [_i64]),
    'dc_p2plane_pair': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32,
                               _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    'dc_p2plane_sequence_partial_count': (_i64, [_vp, _i32]),
    'dc_p2plane_sequence': (_i32, [_vp, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'dc_p2point_pair': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp,
                               _i64, _vp, _vp, _vp]),
    'dc_p2point_sequence': (_i32, [_vp, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'dc_icp_sequence_step': (_i32, [_i32, _vp, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'dc_pose_correct_fwd': (_i32, [_vp, _vp, _i32, _i32, _vp, _vp]),
    'dc_pose_correct_bwd': (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    'dc_shadow_mask': (_i32, [_vp, _vp, _i32, _i32, _vp, _i64, _i32, _f64, _f64, _f64, _vp, _vp]),
    'dc_correct_depth': (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp]),
    'dc_shadow_filter': (_i32, [_vp, _vp, _i32, _vp, _i32, _i64, _f64, _f64, _f64, _vp, _vp, _sz, _vp]),
}


class BlockTableDesc(ctypes.Structure):
    """dcBlockTable of include/dc_hip.h."""
    _fields_ = [('blk_ptr', _vp), ('blk_ids', _vp), ('slot_ptr', _vp), ('loc', _vp), ('max_rows', ctypes.c_int32),
                ('layout', ctypes.c_int32), ('run_ptr', _vp), ('own_base', _vp), ('packed', ctypes.c_int32),
                ('reserved', ctypes.c_int32), ('row_ptr', _vp)]


class PoseTableDesc(ctypes.Structure):
    """dcPoseTable of include/dc_hip.h."""
    _fields_ = [('blk_ptr', _vp), ('ids', _vp), ('loc', _vp), ('own_pos', _vp), ('row_seg', _vp), ('row_scan', _vp)]


class IcpScan(ctypes.Structure):
    """dcIcpScan of include/dc_hip.h."""
    _fields_ = [('vps', _vp), ('dirs', _vp), ('depth', _vp), ('inc', _vp), ('lmask', _vp), ('normals', _vp)]


class IcpPair(ctypes.Structure):
    """dcIcpPair of include/dc_hip.h."""
    _fields_ = [('scan_a', ctypes.c_int32), ('scan_b', ctypes.c_int32), ('idx_a', _vp), ('idx_b', _vp),
                ('m', ctypes.c_int64), ('weight', ctypes.c_double)]


class PoseTrainStepDesc(ctypes.Structure):
    """dcPoseTrainStep of include/dc_hip.h."""
    _fields_ = [('w', _vp), ('w_m', _vp), ('w_v', _vp), ('poses0', _vp), ('deltas', _vp), ('d_m', _vp), ('d_v', _vp),
                ('n_deltas', ctypes.c_int32), ('zero_first', ctypes.c_int32), ('step', _vp), ('lr_w', _f64), ('lr_d', _f64),
                ('beta1', _f64), ('beta2', _f64), ('eps', _f64), ('poses_used', _vp), ('record', _vp), ('ring_rows', ctypes.c_int32),
                ('n_record_extra', ctypes.c_int32), ('record_extra', _vp), ('poses_next', _vp), ('poses12_next', _vp)]


class SequenceDesc(ctypes.Structure):
    """dcSequenceDesc of include/dc_hip.h."""
    _fields_ = [('n', ctypes.c_int64), ('k', ctypes.c_int32), ('n_scans', ctypes.c_int32), ('dtype', ctypes.c_int32),
                ('point_fmt', ctypes.c_int32), ('qparams', ctypes.c_double * 4),
                ('vps', _vp), ('dirs', _vp), ('depth', _vp), ('inc', _vp), ('lmask', _vp), ('scan_id', _vp),
                ('nbr', _vp), ('csr_ptr', _vp), ('csr_src', _vp), ('mask', _vp), ('lane_perm', _vp), ('centre_idx', _vp),
                ('n_centres', ctypes.c_int64), ('x', _vp), ('rec', _vp),
                ('partials', _vp), ('model_kind', ctypes.c_int32), ('n_terms', ctypes.c_int32),
                ('loss_kind', ctypes.c_int32), ('normalization', ctypes.c_int32), ('sqrt_', ctypes.c_int32),
                ('reserved', ctypes.c_int32), ('fwd_table', _vp), ('bwd_table', _vp), ('status', _vp), ('basis', _vp),
                ('partials_count', ctypes.c_int64), ('scan_seg', _vp), ('blk_skip', _vp), ('fwd_rows_active', ctypes.c_int32),
                ('reserved2', ctypes.c_int32), ('pose_table', _vp), ('local_basis', _vp), ('fwd_table_loss', _vp),
                ('fwd_rows_active_loss', ctypes.c_int32), ('reserved3', ctypes.c_int32)]


def lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libdc_hip.so')


def lib():
    """The loaded HIP library; raises (never falls back) when it is missing."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError('depth_correction_amd: %s is missing -- build it with `python -c "import '
                               '__graft_entry__ as g; g.build()"` (hipcc, gfx950). There is no CPU fallback.' % path)
        handle = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError = header / library mismatch: fail loudly
            fn.restype, fn.argtypes = res, args
        _LIB = handle
    return _LIB


def check(status, what):
    if status != 0:
        kind = {-1: 'invalid argument', -2: 'unsupported dtype', -3: 'workspace too small',
                -4: 'unsupported size', -5: 'backward tables missing'}.get(status, 'HIP error %d' % status)
        raise RuntimeError('%s failed: %s' % (what, kind))


def stream_ptr():
    """torch's current stream of the CURRENT device; launches run under ``on_device`` so that this is the operands' device."""
    raw = getattr(torch._C, '_cuda_getCurrentRawStream', None)
    if raw is not None:                         # the handle itself, without building a torch.cuda.Stream object per launch
        return ctypes.c_void_p(raw(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _device_of(args, kwargs):
    for a in tuple(args) + tuple(kwargs.values()):
        dev = a.device if isinstance(a, torch.Tensor) else getattr(a, 'device', None)
        if callable(dev):                       # DepthCloud.device() is a method (depth_cloud.py:170 of the reference)
            dev = dev()
        if isinstance(dev, torch.device) and dev.type == 'cuda':
            return dev
        if isinstance(a, dict):
            a = list(a.values())
        if isinstance(a, (list, tuple)) and a and not isinstance(a[0], (int, float, str)):
            dev = _device_of(a[:1], {})
            if dev is not None:
                return dev
    return None


def on_device(fn):
    """Run ``fn`` with the device of its first GPU operand (tensor, or object with a ``.device``) made current, so the
    raw pointers, torch's current stream and every workspace the wrapper allocates belong to the same GPU.  Without
    it a cloud on ``cuda:1`` in a process whose current device is 0 would launch device-0 kernels on device-1 pointers."""
    @functools.wraps(fn)
    def guarded(*args, **kwargs):
        dev = _device_of(args, kwargs)
        if dev is None or dev.index is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return guarded


def dtype_code(t):
    if t.dtype == torch.float32:
        return DC_F32
    if t.dtype == torch.float64:
        return DC_F64
    raise TypeError('depth_correction_amd kernels take float32 or float64 tensors, got %s' % t.dtype)


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def need(t, shape=None, dtype=None, name='tensor', device=None):
    """Host-side operand check before a raw pointer goes to a kernel."""
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor' % name)
    if not t.is_cuda:
        raise RuntimeError('%s must live on the GPU (depth_correction_amd has no CPU path)' % name)
    if device is not None and t.device != device:
        raise RuntimeError('%s is on %s, expected %s' % (name, t.device, device))
    if not t.is_contiguous():
        raise RuntimeError('%s must be contiguous' % name)
    if dtype is not None and t.dtype != dtype:
        raise TypeError('%s must be %s, got %s' % (name, dtype, t.dtype))
    if shape is not None:
        if len(shape) != t.dim() or any(s is not None and s != d for s, d in zip(shape, t.shape)):
            raise ValueError('%s has shape %s, expected %s' % (name, tuple(t.shape), shape))
    return t
