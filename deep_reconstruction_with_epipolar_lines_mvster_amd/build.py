"""Build libmvster_hip.so (gfx950) in-tree with hipcc.

    python -m deep_reconstruction_with_epipolar_lines_mvster_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects go to csrc/build/, the shared library next to the
sources (git-ignored; it travels to the GPU box with the gpurun snapshot).
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")
LIB = os.path.join(CSRC, "libmvster_hip.so")
SOURCES = ["pose.hip", "layout.hip", "warp.hip", "cost_volume_fwd.hip", "cost_volume_bwd.hip", "cost_volume_scatter.hip", "cost_volume_binned.hip", "cost_volume_generic.hip", "regress.hip", "ranges.hip", "sinkhorn.hip", "batchnorm.hip", "fpn_merge.hip", "conv_direct.hip", "conv_mfma.hip", "geofilter.hip"]
HEADERS = ["mvster_device.h", "mvster_host.h", "cost_volume_common.h"]
# -ffp-contract=off: fused multiply-adds are written explicitly where wanted, everything else rounds
# like the reference's separate torch operators.
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libmvster_hip.so cannot be built")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every HIP translation unit for gfx950 and link the shared library. Returns its path."""
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    common = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(INCLUDE, "mvster_hip.h"), os.path.abspath(__file__)]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + common):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + ["-I", INCLUDE, "-I", CSRC, "-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return o

    if jobs:
        if verbose:
            print("[mvster build] compiling", ", ".join(os.path.basename(j[0]) for j in jobs), flush=True)
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) - 1))) as pool:
            list(pool.map(compile_one, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "--offload-arch=gfx950", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr)
        if verbose:
            print("[mvster build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
