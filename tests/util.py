"""Shared helpers for the test-suite (fixture loading, synthetic geometry)."""
import glob
import math
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    """Load tests/golden/<name>.npz as a dict of torch tensors (0-d arrays stay numpy scalars)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a.copy()) if a.ndim > 0 else a
    return out


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def stage_case(name):
    """Unpack a stage_*.npz fixture into (cfg dict, tensors dict)."""
    z = golden(name)
    C, G, D, H, W, N, group_cor, fuse_d, training, B = [int(v) for v in z["meta"]]
    cfg = dict(C=C, G=G, D=D, H=H, W=W, N=N, B=B, group_cor=bool(group_cor), attn_fuse_d=bool(fuse_d),
               training=bool(training), temp=float(z["temp"]), split_itv=float(z["split_itv"]))
    return cfg, z


def synth_views(B, N, H, W, scale=1.0, seed=0, yaw_deg=0.0):
    """DTU-shaped synthetic cameras, SURVEY.md section 8(d): f = 2890*(W_full/1600) px scaled per stage,
    principal point at the image centre, identity rotation (+ optional yaw) and translation
    (30 v, -10 v, 0) mm for view v.  ``scale`` = W / W_full (1/8, 1/4, 1/2, 1 per stage)."""
    mats = torch.zeros(B, N, 2, 4, 4)
    for v in range(N):
        K = torch.eye(4)
        K[0, 0] = K[1, 1] = 2890.0 * (W / scale) / 1600.0 * scale
        K[0, 2] = W / 2.0
        K[1, 2] = H / 2.0
        E = torch.eye(4)
        a = math.radians(yaw_deg * v)
        E[:3, :3] = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
        E[0, 3] = 30.0 * v
        E[1, 3] = -10.0 * v
        mats[:, v, 0] = E
        mats[:, v, 1] = K
    return mats
