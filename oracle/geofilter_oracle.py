"""TEST INFRASTRUCTURE -- numpy restatement of the reference's geometric-consistency check.

    reproject_with_depth         test_mvs4.py:612-649
    check_geometric_consistency  test_mvs4.py:653-670

PARITY UNPINNED.  The reference samples the source depth map with ``cv2.remap(..., INTER_LINEAR)``; cv2 is not
importable in the build container and the reference ships no fixtures for this step, so ``remap_linear`` below
restates the documented behaviour of that call (bilinear weights at the float32 sample position, constant border 0)
and does NOT reproduce OpenCV's internal 1/32-pixel coordinate quantisation.  Everything else (the float64 matrix
algebra, the float32 casts, the two thresholds) follows the reference line by line.
Only tests/ may import this file.
"""
import numpy as np


def remap_linear(img, map_x, map_y):
    H, W = img.shape
    xf, yf = np.floor(map_x), np.floor(map_y)
    ax, ay = (map_x - xf).astype(np.float32), (map_y - yf).astype(np.float32)
    out = np.zeros(map_x.shape, dtype=np.float32)
    for dy in (0, 1):
        for dx in (0, 1):
            tx, ty = xf + dx, yf + dy
            ok = (tx >= 0) & (tx <= W - 1) & (ty >= 0) & (ty <= H - 1)
            w = (ax if dx else np.float32(1) - ax) * (ay if dy else np.float32(1) - ay)
            vals = img[np.clip(ty, 0, H - 1).astype(np.int64), np.clip(tx, 0, W - 1).astype(np.int64)]
            out = out + np.where(ok, w * vals, np.float32(0)).astype(np.float32)
    return out


def reproject_with_depth(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src):
    width, height = depth_ref.shape[1], depth_ref.shape[0]
    x_ref, y_ref = np.meshgrid(np.arange(0, width), np.arange(0, height))
    x_ref, y_ref = x_ref.reshape([-1]), y_ref.reshape([-1])
    xyz_ref = np.matmul(np.linalg.inv(intrinsics_ref), np.vstack((x_ref, y_ref, np.ones_like(x_ref))) * depth_ref.reshape([-1]))
    xyz_src = np.matmul(np.matmul(extrinsics_src, np.linalg.inv(extrinsics_ref)), np.vstack((xyz_ref, np.ones_like(x_ref))))[:3]
    K_xyz_src = np.matmul(intrinsics_src, xyz_src)
    xy_src = K_xyz_src[:2] / K_xyz_src[2:3]
    x_src = xy_src[0].reshape([height, width]).astype(np.float32)
    y_src = xy_src[1].reshape([height, width]).astype(np.float32)
    sampled = remap_linear(depth_src, x_src, y_src)
    xyz_src = np.matmul(np.linalg.inv(intrinsics_src), np.vstack((xy_src, np.ones_like(x_ref))) * sampled.reshape([-1]))
    xyz_rep = np.matmul(np.matmul(extrinsics_ref, np.linalg.inv(extrinsics_src)), np.vstack((xyz_src, np.ones_like(x_ref))))[:3]
    depth_rep = xyz_rep[2].reshape([height, width]).astype(np.float32)
    K_xyz_rep = np.matmul(intrinsics_ref, xyz_rep)
    xy_rep = K_xyz_rep[:2] / K_xyz_rep[2:3]
    return (depth_rep, xy_rep[0].reshape([height, width]).astype(np.float32),
            xy_rep[1].reshape([height, width]).astype(np.float32), x_src, y_src)


def check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src,
                                condmask_pixel=1.0, condmask_depth=0.01):
    width, height = depth_ref.shape[1], depth_ref.shape[0]
    x_ref, y_ref = np.meshgrid(np.arange(0, width), np.arange(0, height))
    depth_rep, x_rep, y_rep, x_src, y_src = reproject_with_depth(depth_ref, intrinsics_ref, extrinsics_ref, depth_src,
                                                                 intrinsics_src, extrinsics_src)
    dist = np.sqrt((x_rep - x_ref) ** 2 + (y_rep - y_ref) ** 2)
    rel = np.abs(depth_rep - depth_ref) / depth_ref
    mask = np.logical_and(dist < condmask_pixel, rel < condmask_depth)
    depth_rep[~mask] = 0
    return mask, depth_rep, x_src, y_src
