"""On-disk formats either side of the hot path ("the wire contract").

    read_pfm / save_pfm           datasets/data_io.py:6-71      depth and confidence maps (``*.pfm``)
    read_camera_parameters        test_mvs4.py:143-151          ``*_cam.txt``: extrinsic 4x4, intrinsic 3x3
    write_cam                     test_mvs4.py:187-203          ``*_cam.txt`` incl. the depth-range line
    read_pair_file                test_mvs4.py:175-185          ``pair.txt``: reference view -> ranked source views

Host-side byte/text handling only (no GPU, no third-party dependency); same signatures, return values and error
behaviour (a malformed file raises ``Exception``) as the reference functions, so ``save_scene_depth`` / ``filter_depth``
can call them unchanged.  PFM: text header ``PF``/``Pf``, ``<width> <height>``, scale (negative = little endian), then
float32 rows stored BOTTOM-UP.
"""
import re
import sys

import numpy as np

__all__ = ["read_pfm", "save_pfm", "read_camera_parameters", "write_cam", "read_pair_file"]


def read_pfm(filename):
    """-> (float32 array [H,W] or [H,W,3], top row first; scale as a positive float)."""
    with open(filename, "rb") as f:
        magic = f.readline().decode("utf-8").rstrip()
        if magic not in ("PF", "Pf"):
            raise Exception("Not a PFM file.")
        dims = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if dims is None:
            raise Exception("Malformed PFM header.")
        width, height = int(dims.group(1)), int(dims.group(2))
        scale = float(f.readline().rstrip())
        order = "<" if scale < 0 else ">"
        payload = np.fromfile(f, order + "f")
    shape = (height, width, 3) if magic == "PF" else (height, width)
    return np.flipud(payload.reshape(shape)), abs(scale)


def save_pfm(filename, image, scale=1):
    """float32 [H,W], [H,W,1] (grey, ``Pf``) or [H,W,3] (colour, ``PF``), written bottom row first in the array's byte
    order (the sign of the scale line records it)."""
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    if image.ndim == 3 and image.shape[2] == 3:
        magic = "PF"
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        magic = "Pf"
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    order = image.dtype.byteorder
    little = order == "<" or (order == "=" and sys.byteorder == "little")
    with open(filename, "wb") as f:
        f.write(("%s\n%d %d\n%f\n" % (magic, image.shape[1], image.shape[0], -scale if little else scale)).encode("utf-8"))
        np.flipud(image).tofile(f)


def read_camera_parameters(filename):
    """-> (intrinsics float64 [3,3], extrinsics float64 [4,4]); lines 1-4 and 7-9 of the file (0-based)."""
    with open(filename) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float64).reshape(4, 4)
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float64).reshape(3, 3)
    return intrinsics, extrinsics


def write_cam(file, cam):
    """cam[0] = extrinsic 4x4; cam[1] = 4x4 whose top-left 3x3 is the intrinsic and whose last row is
    (depth_min, depth_interval, depth_num, depth_max)."""
    rows = ["extrinsic"]
    rows += ["".join(str(cam[0][i][j]) + " " for j in range(4)) for i in range(4)]
    rows += ["", "intrinsic"]
    rows += ["".join(str(cam[1][i][j]) + " " for j in range(3)) for i in range(3)]
    rows += ["", " ".join(str(cam[1][3][j]) for j in range(4))]
    with open(file, "w") as f:
        f.write("\n".join(rows) + "\n")


def read_pair_file(filename):
    """-> [(ref_view, [src views, best first])]; views without source views are dropped.  Per view the file holds the
    view id on one line and ``<n> <id0> <score0> <id1> <score1> ...`` on the next."""
    pairs = []
    with open(filename) as f:
        for _ in range(int(f.readline())):
            ref_view = int(f.readline().rstrip())
            src_views = [int(tok) for tok in f.readline().rstrip().split()[1::2]]
            if src_views:
                pairs.append((ref_view, src_views))
    return pairs
