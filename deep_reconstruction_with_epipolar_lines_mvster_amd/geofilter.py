"""F4 -- geometric-consistency check of the depth-map filter on the GPU.

    check_geometric_consistency   test_mvs4.py:653-670  (+ reproject_with_depth :612-649)
    geometric_filter              the per-reference-view loop of filter_depth, test_mvs4.py:723-746

numpy in / numpy out like the reference's functions (they sit between PFM files and the PLY writer); the thresholds
the reference reads from its global ``args`` (--condmask_pixel, --condmask_depth, --geomask) are explicit arguments.
PARITY UNPINNED: the sampling restates ``cv2.remap(INTER_LINEAR)`` without OpenCV's 1/32-pixel quantisation (cv2 is
not available where the oracle was written), see oracle/geofilter_oracle.py.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .hotpath import _on, _stream

__all__ = ["check_geometric_consistency", "geometric_filter"]


def _mats(K_ref, E_ref, K_src, E_src):
    K_ref, E_ref = np.asarray(K_ref, dtype=np.float64), np.asarray(E_ref, dtype=np.float64)
    K_src, E_src = np.asarray(K_src, dtype=np.float64), np.asarray(E_src, dtype=np.float64)
    parts = [np.linalg.inv(K_ref), np.matmul(E_src, np.linalg.inv(E_ref)), K_src, np.linalg.inv(K_src),
             np.matmul(E_ref, np.linalg.inv(E_src)), K_ref]
    return np.ascontiguousarray(np.concatenate([p.reshape(-1) for p in parts]))          # 68 doubles, host


def _device_depth(a, device):
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return t.to(device=device, dtype=torch.float32).contiguous()


def _check(dref, dsrc, mats, pix, rel, want_xy, device):
    H, W = dref.shape
    Hs, Ws = dsrc.shape
    mask = torch.empty(H, W, dtype=torch.uint8, device=device)
    drep = torch.empty(H, W, dtype=torch.float32, device=device)
    xs = torch.empty(H, W, dtype=torch.float32, device=device) if want_xy else None
    ys = torch.empty(H, W, dtype=torch.float32, device=device) if want_xy else None
    with _on(device):
        _lib.call("mvster_geometric_consistency", dref.data_ptr(), dsrc.data_ptr(),
                  mats.ctypes.data_as(ctypes.c_void_p), mask.data_ptr(), drep.data_ptr(),
                  xs.data_ptr() if want_xy else None, ys.data_ptr() if want_xy else None, H, W, Hs, Ws, float(pix),
                  float(rel), _stream(device))
    return mask, drep, xs, ys


def check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src,
                                condmask_pixel=1.0, condmask_depth=0.01, device="cuda"):
    """-> (mask bool [H,W], depth_reprojected float32 [H,W] (0 where rejected), x2d_src, y2d_src) as numpy arrays."""
    device = torch.device(device)
    dref, dsrc = _device_depth(depth_ref, device), _device_depth(depth_src, device)
    mask, drep, xs, ys = _check(dref, dsrc, _mats(intrinsics_ref, extrinsics_ref, intrinsics_src, extrinsics_src),
                                condmask_pixel, condmask_depth, True, device)
    return mask.cpu().numpy().astype(bool), drep.cpu().numpy(), xs.cpu().numpy(), ys.cpu().numpy()


def geometric_filter(depth_ref, intrinsics_ref, extrinsics_ref, sources, condmask_pixel=1.0, condmask_depth=0.01,
                     geomask=3, device="cuda"):
    """The source-view loop of filter_depth for one reference view.  ``sources`` = [(depth, K, E), ...].
    -> (geo_mask bool, depth_est_averaged float32, geo_mask_sum int32) as numpy arrays."""
    device = torch.device(device)
    dref = _device_depth(depth_ref, device)
    total = torch.zeros_like(dref)
    count = torch.zeros(dref.shape, dtype=torch.int32, device=device)
    for depth_src, K_src, E_src in sources:
        mask, drep, _, _ = _check(dref, _device_depth(depth_src, device), _mats(intrinsics_ref, extrinsics_ref, K_src, E_src),
                                  condmask_pixel, condmask_depth, False, device)
        total += drep
        count += mask.to(torch.int32)
    averaged = (total + dref) / (count + 1).to(torch.float32)
    return (count >= geomask).cpu().numpy(), averaged.cpu().numpy(), count.cpu().numpy()
