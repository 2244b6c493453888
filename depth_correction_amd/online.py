"""The online correction call pattern of the reference's ROS node without ROS (scripts/depth_correction:31-58):

    dc = local_feature_cloud(input_cloud, cfg)      # [shadow filter,] neighbourhoods, features, planarity mask
    dc = model(dc)                                  # depth correction on the masked points
    dc.update_points()

``correct_cloud`` is that sequence under ``torch.no_grad()``; the input is a structured / plain array as ``numpify`` of a
PointCloud2 yields, the uploaded rows [N, >=3] as a device tensor (the fastest way in: from_points, shadow filter and
``cloud[mask]`` are then one native call, dc_scan_prefilter), or a DepthCloud already on the device (scan_io.cloud_on_device).  Message conversion (ros_numpy)
and publishing stay outside (ROS is out of scope); ``to_structured_array`` on the result gives the fields the
node publishes.
"""
from __future__ import annotations

import torch

from .config import Config
from .preproc import local_feature_cloud

__all__ = ['correct_cloud']


def correct_cloud(input_cloud, model, cfg: Config):
    """Corrected DepthCloud of one incoming scan (depth and grid pre-filters are assumed to have run earlier, as the node
    assumes, scripts/depth_correction:42)."""
    with torch.no_grad():
        dc = local_feature_cloud(input_cloud, cfg)
        dc = model(dc)
        dc.update_points()
    return dc
