"""Pose parametrisation used by the pose-correction path (transform.py:68-91): 6-vectors (translation + axis-angle)
to 4x4 matrices.  ``axis_angle_to_matrix`` restates the published pytorch3d algorithm the reference imports
(axis-angle -> quaternion with the small-angle series -> rotation matrix); it is differentiable at zero, which the
optimisation needs because pose corrections start at zero (eval.py:53-59).  Tiny tensors: plain torch -- except the
composition ``poses @ xyz_axis_angle_to_matrix(deltas)`` of the training loop (eval.py:68-82), which on the GPU is ONE
kernel forward and one backward (``corrected_poses``: dc_pose_correct_fwd / _bwd) instead of ~50 tensor ops each way."""
from __future__ import annotations

import torch

__all__ = ['axis_angle_to_matrix', 'xyz_axis_angle_to_matrix', 'matrix_to_xyz_axis_angle', 'corrected_poses']


def axis_angle_to_matrix(axis_angle):
    angle = torch.norm(axis_angle, p=2, dim=-1, keepdim=True)
    small = angle.abs() < 1e-6
    denom = torch.where(small, torch.ones_like(angle), angle)
    k = torch.where(small, 0.5 - angle * angle / 48, torch.sin(0.5 * angle) / denom)     # sin(a/2)/a
    quat = torch.cat([torch.cos(0.5 * angle), axis_angle * k], dim=-1)
    r, i, j, k = torch.unbind(quat, -1)
    s = 2.0 / (quat * quat).sum(-1)
    rows = (1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r),
            s * (i * j + k * r), 1 - s * (i * i + k * k), s * (j * k - i * r),
            s * (i * k - j * r), s * (j * k + i * r), 1 - s * (i * i + j * j))
    return torch.stack(rows, -1).reshape(quat.shape[:-1] + (3, 3))


def xyz_axis_angle_to_matrix(xyz_axis_angle):
    assert isinstance(xyz_axis_angle, torch.Tensor) and xyz_axis_angle.shape[-1] == 6
    lead = xyz_axis_angle.shape[:-1]
    top = torch.cat([axis_angle_to_matrix(xyz_axis_angle[..., 3:]), xyz_axis_angle[..., :3, None]], dim=-1)
    bottom = torch.zeros(lead + (1, 4), dtype=xyz_axis_angle.dtype, device=xyz_axis_angle.device)
    bottom[..., 0, 3] = 1.
    return torch.cat([top, bottom], dim=-2)


def matrix_to_xyz_axis_angle(T):
    """Inverse of ``xyz_axis_angle_to_matrix`` for rotation angles below pi."""
    assert isinstance(T, torch.Tensor) and T.dim() == 3 and T.shape[1:] == (4, 4)
    R = T[:, :3, :3]
    cos = ((R.diagonal(dim1=-2, dim2=-1).sum(-1) - 1) / 2).clamp(-1, 1)
    angle = torch.arccos(cos)
    axis = torch.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], dim=-1)
    sin = torch.sin(angle)
    scale = torch.where(sin.abs() < 1e-9, torch.full_like(sin, 0.5), angle / (2 * torch.where(sin.abs() < 1e-9, torch.ones_like(sin), sin)))
    return torch.cat([T[:, :3, 3], axis * scale[:, None]], dim=1)


class _CorrectedPoses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, poses, deltas):
        from ._native import lib, check, ptr, stream_ptr
        n = poses.shape[0]
        p64 = poses.detach().to(torch.float64).reshape(n, 16).contiguous()
        d64 = deltas.detach().to(torch.float64).contiguous()
        out = torch.empty_like(p64)
        with torch.cuda.device(poses.device):
            check(lib().dc_pose_correct_fwd(ptr(p64), ptr(d64), n, d64.shape[0], ptr(out), stream_ptr()), 'dc_pose_correct_fwd')
        ctx.save_for_backward(p64, d64)
        ctx.meta = (deltas.dtype, poses.dtype)
        return out.reshape(n, 4, 4).to(poses.dtype)

    @staticmethod
    def backward(ctx, grad):
        from ._native import lib, check, ptr, stream_ptr
        p64, d64 = ctx.saved_tensors
        n = p64.shape[0]
        g64 = grad.to(torch.float64).reshape(n, 16).contiguous()
        gd = torch.empty_like(d64)
        with torch.cuda.device(p64.device):
            check(lib().dc_pose_correct_bwd(ptr(p64), ptr(d64), n, d64.shape[0], ptr(g64), ptr(gd), stream_ptr()),
                  'dc_pose_correct_bwd')
        return None, gd.to(ctx.meta[0])


def corrected_poses(poses, deltas):
    """``poses @ xyz_axis_angle_to_matrix(deltas)`` (eval.py:68-82) for poses [S,4,4] and corrections [S,6] or [1,6].

    GPU tensors (poses constant, corrections possibly requiring grad): one fused kernel each way; anything else: the
    tensor expressions."""
    if (poses.is_cuda and deltas.is_cuda and poses.dim() == 3 and deltas.dim() == 2 and deltas.shape[0] in (1, poses.shape[0])
            and not poses.requires_grad and poses.dtype in (torch.float32, torch.float64)):
        return _CorrectedPoses.apply(poses, deltas)
    return torch.matmul(poses, xyz_axis_angle_to_matrix(deltas))
