"""Pose parametrisation used by the pose-correction path (transform.py:68-91): 6-vectors (translation + axis-angle)
to 4x4 matrices.  ``axis_angle_to_matrix`` restates the published pytorch3d algorithm the reference imports
(axis-angle -> quaternion with the small-angle series -> rotation matrix); it is differentiable at zero, which the
optimisation needs because pose corrections start at zero (eval.py:53-59).  Tiny tensors: plain torch."""
from __future__ import annotations

import torch

__all__ = ['axis_angle_to_matrix', 'xyz_axis_angle_to_matrix', 'matrix_to_xyz_axis_angle']


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
