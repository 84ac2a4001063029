"""MI355X-native (gfx950) implementation of MVSTER's cost-volume + epipolar-attention hot path.

Drop-in for the hot-path symbols of the reference's ``models/mvs4net_utils.py`` (see hotpath.py);
the kernels live in csrc/ behind the C ABI of include/mvster_hip.h.
"""
from .hotpath import (homo_warping, stagenet, init_range, init_inverse_range, schedule_inverse_range,  # noqa: F401
                      stage_poses, relative_pose, fused_cost_volume, depth_regress, to_channels_last,
                      to_planar)

from .mvs4net import MVS4net  # noqa: F401,E402
from .losses import sinkhorn, MVS4net_loss, Blend_loss  # noqa: F401,E402
from .geofilter import check_geometric_consistency, geometric_filter  # noqa: F401,E402
from .io_formats import read_pfm, save_pfm, read_camera_parameters, write_cam, read_pair_file  # noqa: F401,E402

__all__ = ["read_pfm", "save_pfm", "read_camera_parameters", "write_cam", "read_pair_file", "MVS4net", "check_geometric_consistency", "geometric_filter", "MVS4net_loss", "Blend_loss", "sinkhorn", "homo_warping", "stagenet", "init_range", "init_inverse_range", "schedule_inverse_range",
           "stage_poses", "relative_pose", "fused_cost_volume", "depth_regress", "to_channels_last", "to_planar"]
