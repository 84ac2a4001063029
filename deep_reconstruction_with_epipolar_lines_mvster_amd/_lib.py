"""ctypes binding of libmvster_hip.so (the C ABI declared in include/mvster_hip.h).

There is NO fallback: if the shared library is missing or does not export the full ABI, loading
raises, and every operator of the package raises with it.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVSTER_LIB_PATH: load another build of the same library (tuning sweeps); never a different implementation
LIB_PATH = os.environ.get("MVSTER_LIB_PATH") or os.path.join(_HERE, "csrc", "libmvster_hip.so")
ABI_VERSION = 3

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_vpp = ctypes.POINTER(ctypes.c_void_p)
_ll = ctypes.c_longlong

# name -> argtypes, exactly the prototypes of include/mvster_hip.h
PROTOTYPES = {
    "mvster_relative_pose": [_vp, _vp, _vp, _i, _vp],
    "mvster_stage_poses": [_vp, _vp, _i, _i, _vp],
    "mvster_warp_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvster_warp_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvster_nchw_to_nhwc": [_vp, _vp, _i, _i, _i, _i, _vp],
    "mvster_nhwc_to_nchw": [_vp, _vp, _i, _i, _i, _i, _vp],
    "mvster_nchw_to_nhwc_many": [_vpp, _vpp, _i, _i, _i, _i, _i, _vp],
    "mvster_nhwc_to_nchw_many": [_vpp, _vpp, _i, _i, _i, _i, _i, _vp],
    "mvster_cost_volume_fwd": [_vp, _vpp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    "mvster_cost_volume_fwd_counted": [_vp, _vpp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp],
    "mvster_cost_volume_bin_counts": [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i],
    "mvster_cost_volume_generic_fwd": [_vp, _vpp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp],
    "mvster_cost_volume_generic_bwd": [_vp, _vp, _vp, _vp, _vpp, _i, _vp, _vp, _vp, _vpp, _i, _i, _i, _i, _i, _i, _i, _i, _f,
                                       _i, _i, _vp],
    "mvster_cost_volume_bwd": [_vp, _vp, _vp, _vp, _vpp, _i, _vp, _vp, _vp, _vpp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i,
                               _vp],
    "mvster_cost_volume_bwd_ws": [_vp, _vp, _vp, _vp, _vpp, _i, _vp, _vp, _vp, _vpp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i,
                                  _vp, _ll, _vp, _vp],
    "mvster_cost_volume_bwd_workspace": [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i],
    "mvster_cost_volume_bwd_route_workspace": [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i],
    "mvster_cost_volume_bwd_route": [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _ll],
    "mvster_cost_volume_bwd_route_name": [_i, _i, _i, _i, ctypes.c_char_p, _i],
    "mvster_depth_regress_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "mvster_softmax_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvster_sinkhorn_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "mvster_sinkhorn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "mvster_sinkhorn_generic_workspace": [_i, _i, _i, _i, _i, _i, _i],
    "mvster_sinkhorn_generic_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _f, _i, _vp],
    "mvster_sinkhorn_generic_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _f, _i, _vp],
    "mvster_bn_train_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp],
    "mvster_bn_train_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvster_bn_train_fwd_grouped": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _i, _vp],
    "mvster_bn_train_bwd_grouped": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "mvster_bn_train_fwd_nhwc": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _f, _f, _i, _vp],
    "mvster_bn_train_bwd_nhwc": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp],
    "mvster_bn_eval_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp],
    "mvster_bn_eval_fwd_nhwc": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _f, _i, _vp],
    "mvster_fpn_merge_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "mvster_fpn_merge_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "mvster_fpn_head_gather_fwd": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "mvster_fpn_head_gather_fwd_layout": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mvster_fpn_head_gather_bwd": [_vp, _vp, _i, _i, _i, _vp],
    "mvster_conv_mfma_supported": [_i, _i, _i, _i],
    "mvster_conv_mfma_workspace": [_i, _i, _i, _i, _i, _i, _i, _i],
    "mvster_conv_mfma_packed_floats": [_i, _i, _i, _i, _i],
    "mvster_conv_mfma_fwd": [_vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvster_conv_mfma_wgrad": [_vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvster_conv_mfma_depth3_fwd": [_vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvster_conv_mfma_depth3_wgrad": [_vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvster_conv_mfma_transposed": [_vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvster_conv_direct_supported": [_i, _i, _i],
    "mvster_conv_direct_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "mvster_conv_direct_fwd_layout": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mvster_conv_direct_fwd_ep": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "mvster_bn_eval_prepare": [_vp, _vp, _vp, _vp, _vp, _i, _f, _vp],
    "mvster_bn_eval_norm_floats": [_i],
    "mvster_conv_mfma_pack": [_vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _vp],
    "mvster_conv_mfma_fwd_ep": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "mvster_conv_mfma_transposed_ep": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "mvster_conv_direct_wgrad_workspace": [_i, _i, _i, _i, _i, _i],
    "mvster_conv_direct_wgrad": [_vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _vp],
    "mvster_geometric_consistency": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp],
    "mvster_init_range": [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    "mvster_schedule_inverse_range": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
}

RESTYPES = {"mvster_cost_volume_bwd_workspace": _ll, "mvster_cost_volume_bwd_route_workspace": _ll,
            "mvster_cost_volume_bin_counts": _ll, "mvster_conv_direct_wgrad_workspace": _ll,
            "mvster_conv_mfma_workspace": _ll, "mvster_sinkhorn_generic_workspace": _ll,
            "mvster_conv_mfma_packed_floats": _ll, "mvster_bn_eval_norm_floats": _ll}      # everything else returns the int status

MVSTER_OK = 0
LAYOUT_NCHW = 0
LAYOUT_NHWC = 1
GRAD_PLANAR = 16
MVSTER_ERR_UNSUPPORTED = 3

_lock = threading.Lock()
_lib = None


class MvsterError(RuntimeError):
    """A C-ABI call returned a non-zero status."""

    def __init__(self, fn, code, text):
        super().__init__("%s failed: [%d] %s" % (fn, code, text))
        self.code = code


def load():
    """Load (once) and return the ctypes handle; raises if the library is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libmvster_hip.so is not built (%s). Build it with "
                "`python -m deep_reconstruction_with_epipolar_lines_mvster_amd.build`; "
                "this package has no CPU or PyTorch fallback." % LIB_PATH)
        # PyTorch-ROCm ships its own libamdhip64.so.7; importing torch first makes the dynamic loader
        # resolve our DT_NEEDED entry to that already-loaded runtime (one HIP runtime per process, so
        # torch's streams and device pointers are valid in our launches).
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in PROTOTYPES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is missing
            fn.argtypes = argtypes
            fn.restype = RESTYPES.get(name, ctypes.c_int)
        lib.mvster_abi_version.restype = ctypes.c_int
        lib.mvster_error_string.restype = ctypes.c_char_p
        lib.mvster_error_string.argtypes = [ctypes.c_int]
        got = lib.mvster_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError("libmvster_hip.so ABI version %d, binding expects %d" % (got, ABI_VERSION))
        _lib = lib
    return _lib


def call(name, *args):
    """Invoke an ABI function; non-zero status -> MvsterError."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != MVSTER_OK:
        raise MvsterError(name, rc, lib.mvster_error_string(rc).decode())


def try_call(name, *args):
    """Like call() but returns False on MVSTER_ERR_UNSUPPORTED (caller picks another HIP route)."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc == MVSTER_ERR_UNSUPPORTED:
        return False
    if rc != MVSTER_OK:
        raise MvsterError(name, rc, lib.mvster_error_string(rc).decode())
    return True


BWD_ROUTES = {0: "none", 1: "planar", 2: "dense", 3: "split_window", 4: "tile_scatter", 5: "binned"}


ROUTE_IDS = {v: k for k, v in BWD_ROUTES.items()}


def bwd_workspace(S, B, C, G, D, H, W, Hs, Ws, layout, route=None):
    """4-byte words of scratch the recommended backward route of this shape wants (0: none); with ``route`` (a label of
    BWD_ROUTES) the words that route needs, -1 when it does not exist for the shape."""
    if route is None:
        return load().mvster_cost_volume_bwd_workspace(S, B, C, G, D, H, W, Hs, Ws, layout)
    return load().mvster_cost_volume_bwd_route_workspace(ROUTE_IDS[route], S, B, C, G, D, H, W, Hs, Ws, layout)


def bwd_route(S, B, C, G, D, H, W, Hs, Ws, layout, workspace_floats):
    """(route id, route label, kernel symbols as rocprofv3 prints them: dominant first, ';'-separated) of the route the
    backward entry point picks."""
    lib = load()
    rid = lib.mvster_cost_volume_bwd_route(S, B, C, G, D, H, W, Hs, Ws, layout, workspace_floats)
    buf = ctypes.create_string_buffer(192)
    lib.mvster_cost_volume_bwd_route_name(rid, C, G, D, buf, len(buf))
    return rid, BWD_ROUTES.get(rid, "route%d" % rid), buf.value.decode()


def ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return ctypes.cast(arr, _vpp), arr   # keep `arr` alive for the duration of the call
