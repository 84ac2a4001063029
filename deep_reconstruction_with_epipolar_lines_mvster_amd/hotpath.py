"""Drop-in replacements for the hot-path symbols of the reference's ``models/mvs4net_utils.py``.

Same names, argument meaning and return structure as upstream, executed by the gfx950 HIP kernels
of ``libmvster_hip.so`` through its C ABI (``include/mvster_hip.h``):

    homo_warping            models/mvs4net_utils.py:21-67
    stagenet                models/mvs4net_utils.py:1017-1162
    init_range              models/mvs4net_utils.py:69-77
    init_inverse_range      models/mvs4net_utils.py:79-85
    schedule_inverse_range  models/mvs4net_utils.py:87-94

Host code is plain Python on PyTorch-ROCm: torch supplies device memory, the current HIP stream and
autograd bookkeeping.  Inputs must be fp32 tensors on a HIP device -- there is no CPU path and no
PyTorch re-implementation behind these functions; anything else raises.
"""
import torch
import torch.nn as nn

from . import _lib

__all__ = ["homo_warping", "stagenet", "init_range", "init_inverse_range", "schedule_inverse_range",
           "stage_poses", "relative_pose", "fused_cost_volume", "depth_regress", "to_channels_last", "to_planar"]


# ----------------------------------------------------------------------------------------------
# plumbing
# ----------------------------------------------------------------------------------------------
def _need(t, name, ndim=None, keep_layout=False):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor, got %s" % (name, type(t).__name__))
    if not t.is_cuda:
        raise RuntimeError("%s is on %s: the MI355X hot path only runs on HIP device tensors "
                           "(no CPU fallback)" % (name, t.device))
    if t.dtype != torch.float32:
        raise RuntimeError("%s has dtype %s: the hot path computes in float32 like the reference" % (name, t.dtype))
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError("%s must have %d dimensions, got shape %s" % (name, ndim, tuple(t.shape)))
    if keep_layout and t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return t                      # channels_last feature maps are consumed in place
    return t.contiguous()


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


class _on:
    """Make ``device`` current for the launches inside the block (DataParallel calls us from one
    thread per GPU, so the device is taken from the tensors, never from global state)."""

    def __init__(self, device):
        self.guard = torch.cuda.device(device)

    def __enter__(self):
        self.guard.__enter__()

    def __exit__(self, *exc):
        return self.guard.__exit__(*exc)


# ----------------------------------------------------------------------------------------------
# poses
# ----------------------------------------------------------------------------------------------
def relative_pose(src_proj, ref_proj):
    """[B,4,4] x [B,4,4] -> [B,12] (rot row-major, trans) of src_proj @ inverse(ref_proj)."""
    src_proj = _need(src_proj, "src_proj", 3)
    ref_proj = _need(ref_proj, "ref_proj", 3)
    B = src_proj.shape[0]
    if tuple(src_proj.shape) != (B, 4, 4) or tuple(ref_proj.shape) != (B, 4, 4):
        raise RuntimeError("src_proj/ref_proj must be [B,4,4]")
    pose = torch.empty(B, 12, dtype=torch.float32, device=src_proj.device)
    with _on(src_proj.device):
        _lib.call("mvster_relative_pose", src_proj.data_ptr(), ref_proj.data_ptr(), pose.data_ptr(), B,
                  _stream(src_proj.device))
    return pose


def stage_poses(proj_matrices):
    """[B,N,2,4,4] dataset matrices -> [B,N-1,12] poses of every source view w.r.t. view 0."""
    pm = _need(proj_matrices, "proj_matrices", 5)
    B, N = pm.shape[0], pm.shape[1]
    if tuple(pm.shape[2:]) != (2, 4, 4) or N < 2:
        raise RuntimeError("proj_matrices must be [B,N>=2,2,4,4], got %s" % (tuple(pm.shape),))
    pose = torch.empty(B, N - 1, 12, dtype=torch.float32, device=pm.device)
    with _on(pm.device):
        _lib.call("mvster_stage_poses", pm.data_ptr(), pose.data_ptr(), B, N, _stream(pm.device))
    return pose


# ----------------------------------------------------------------------------------------------
# A1 homo_warping
# ----------------------------------------------------------------------------------------------
class _Warp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, pose, depth):
        B, C, Hs, Ws = src.shape
        _, D, Hr, Wr = depth.shape
        out = torch.empty(B, C, D, Hr, Wr, dtype=torch.float32, device=src.device)
        with _on(src.device):
            _lib.call("mvster_warp_fwd", src.data_ptr(), pose.data_ptr(), depth.data_ptr(), out.data_ptr(),
                      B, C, D, Hr, Wr, Hs, Ws, _stream(src.device))
        ctx.save_for_backward(pose, depth)
        ctx.src_shape = tuple(src.shape)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        pose, depth = ctx.saved_tensors
        B, C, Hs, Ws = ctx.src_shape
        _, D, Hr, Wr = depth.shape
        gout = gout.contiguous()
        gsrc = torch.empty(B, C, Hs, Ws, dtype=torch.float32, device=gout.device)
        with _on(gout.device):
            _lib.call("mvster_warp_bwd", gout.data_ptr(), pose.data_ptr(), depth.data_ptr(), gsrc.data_ptr(),
                      B, C, D, Hr, Wr, Hs, Ws, _stream(gout.device))
        return gsrc, None, None


def homo_warping(src_fea, src_proj, ref_proj, depth_values, vis_ETA=False, fn=None):
    """src_fea [B,C,Hs,Ws], src_proj/ref_proj [B,4,4], depth_values [B,D,Hr,Wr] -> [B,C,D,Hr,Wr].

    Gradient flows to ``src_fea`` only (the reference builds the grid under no_grad).
    """
    if vis_ETA:
        raise NotImplementedError("vis_ETA dumps the sampling grid to .npy (a debugging aid of the reference, "
                                  "models/mvs4net_utils.py:54-56); the fused HIP path never materialises that grid")
    if isinstance(src_fea, torch.Tensor) and src_fea.dim() == 5:
        raise NotImplementedError("5-D src_fea (models/mvs4net_utils.py:61-65) is unreachable from stagenet and "
                                  "not implemented")
    src = _need(src_fea, "src_fea", 4)
    depth = _need(depth_values, "depth_values", 4)
    if depth.shape[0] != src.shape[0]:
        raise RuntimeError("batch mismatch between src_fea and depth_values")
    pose = relative_pose(src_proj, ref_proj)
    return _Warp.apply(src, pose, depth.detach())


# ----------------------------------------------------------------------------------------------
# A2/A3 fused cost volume
# ----------------------------------------------------------------------------------------------
def _is_channels_last(t):
    """True when ``t`` [B,C,H,W] is stored as [B,H,W,C] (torch.channels_last) and NOT also plain contiguous."""
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()


def to_channels_last(t):
    """[B,C,H,W] fp32 HIP tensor -> same logical tensor stored [B,H,W,C]; zero-copy if it already is."""
    if t.is_contiguous(memory_format=torch.channels_last):
        return t
    t = t.contiguous()
    B, C, H, W = t.shape
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=t.device, memory_format=torch.channels_last)
    with _on(t.device):
        _lib.call("mvster_nchw_to_nhwc", t.data_ptr(), out.data_ptr(), B, C, H, W, _stream(t.device))
    return out


def _convert_many(tensors, to_nhwc):
    """Layout-convert several [B,C,H,W] fp32 tensors in as few launches as their shapes allow (one per distinct shape).
    to_nhwc: NCHW-contiguous inputs -> channels_last copies; else channels_last inputs -> NCHW-contiguous copies."""
    outs = [None] * len(tensors)
    groups = {}
    for i, t in enumerate(tensors):
        groups.setdefault(tuple(t.shape), []).append(i)
    for shape, idx in groups.items():
        B, C, H, W = shape
        for lo in range(0, len(idx), _MAX_SRC + 1):
            part = idx[lo:lo + _MAX_SRC + 1]
            srcs = [tensors[i] for i in part]
            fmt = torch.channels_last if to_nhwc else torch.contiguous_format
            dsts = [torch.empty(shape, dtype=torch.float32, device=srcs[0].device, memory_format=fmt) for _ in part]
            sp, skeep = _lib.ptr_array(srcs)
            dp, dkeep = _lib.ptr_array(dsts)
            with _on(srcs[0].device):
                _lib.call("mvster_nchw_to_nhwc_many" if to_nhwc else "mvster_nhwc_to_nchw_many", sp, dp, len(part), B, C, H, W,
                          _stream(srcs[0].device))
            del skeep, dkeep
            for i, d in zip(part, dsts):
                outs[i] = d
    return outs


def to_planar(t):
    """channels_last [B,C,H,W] tensor -> NCHW-contiguous copy."""
    if t.is_contiguous():
        return t
    B, C, H, W = t.shape
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=t.device)
    with _on(t.device):
        _lib.call("mvster_nhwc_to_nchw", t.data_ptr(), out.data_ptr(), B, C, H, W, _stream(t.device))
    return out


# Scratch the backward may use (see mvster_hip.h, A3): "full" = what the library recommends for the shape (the binned,
# atomics-free scatter for C >= 16), "binned" = the binned route's scratch even where it is not the recommendation,
# "coefficients" = only the per-hypothesis coefficients (round-1 tiled scatter where it applies), "none" = the plain
# atomic routes.  Tests and tools switch it to compare the routes; product code leaves it alone.
BWD_WORKSPACE = "full"
PLANAR_GRADS = True         # NCHW leaves + binned route: gradients are written as planes by the kernels, no conversion pass
COUNT_IN_FORWARD = True     # training forward also counts the samples per destination tile for the binned backward


class _FusedCostVolume(torch.autograd.Function):
    """Feature maps may arrive NCHW-contiguous (the reference's FPN output) or channels_last.  The kernels
    gather from channels-last storage: NCHW inputs are converted once per call (and their gradients converted
    back), channels_last inputs are used in place; channels_last LEAVES receive channels_last gradients (their .grad
    matches them), channels_last maps produced by another operator (FPN4's output layers) receive NCHW-contiguous
    gradients on the binned route -- written as planes by the kernels themselves, the layout every upstream backward
    kernel of this package reads."""

    @staticmethod
    def forward(ctx, pose, hypo, attn_temp, G, layout, ref, *srcs):
        B, C, H, W = ref.shape
        D = hypo.shape[1]
        S = len(srcs)
        Hs, Ws = srcs[0].shape[-2:]
        feats = (ref,) + tuple(srcs)
        if layout == _lib.LAYOUT_NHWC:
            was_cl = [_is_channels_last(f) for f in feats]
            ctx.cl_leaf = any(cl and f.is_leaf for cl, f in zip(was_cl, (ref,) + tuple(srcs)))
            todo = [i for i, f in enumerate(feats) if not f.is_contiguous(memory_format=torch.channels_last)]
            if todo:                                        # all planar maps of the stage in one launch
                done = _convert_many([feats[i].contiguous() for i in todo], to_nhwc=True)
                feats = list(feats)
                for i, d in zip(todo, done):
                    feats[i] = d
                feats = tuple(feats)
        else:
            was_cl = [False] * len(feats)
            ctx.cl_leaf = False
            feats = tuple(f.contiguous() for f in feats)
        vol = torch.empty(B, G, D, H, W, dtype=torch.float32, device=ref.device)
        wsum = torch.empty(B, D, H, W, dtype=torch.float32, device=ref.device)
        # training: the forward kernel also takes the count pass of the binned backward's counting sort (it has every
        # projection at hand); the counters travel to the backward through ctx
        counts = None
        if COUNT_IN_FORWARD and BWD_WORKSPACE in ("full", "binned") and any(ctx.needs_input_grad[5:]):
            nb = _lib.load().mvster_cost_volume_bin_counts(S, B, C, G, D, H, W, Hs, Ws, layout)
            if nb > 0 and (BWD_WORKSPACE == "binned" or _lib.bwd_workspace(S, B, C, G, D, H, W, Hs, Ws, layout) ==
                           _lib.bwd_workspace(S, B, C, G, D, H, W, Hs, Ws, layout, route="binned")):
                counts = torch.empty(nb, dtype=torch.int32, device=ref.device)
        pp, keep = _lib.ptr_array(feats[1:])
        with _on(ref.device):
            _lib.call("mvster_cost_volume_fwd_counted", feats[0].data_ptr(), pp, S, pose.data_ptr(), hypo.data_ptr(),
                      vol.data_ptr(), wsum.data_ptr(), B, C, G, D, H, W, Hs, Ws, float(attn_temp), layout,
                      counts.data_ptr() if counts is not None else None, _stream(ref.device))
        del keep
        ctx.save_for_backward(pose, hypo, vol, wsum, *feats)
        ctx.bin_counts = counts
        ctx.cfg = (float(attn_temp), int(G), int(layout), was_cl)
        return vol

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gvol):
        pose, hypo, vol, wsum = ctx.saved_tensors[:4]
        feats = ctx.saved_tensors[4:]
        attn_temp, G, layout, was_cl = ctx.cfg
        ref, srcs = feats[0], feats[1:]
        B, C, H, W = ref.shape
        D = hypo.shape[1]
        S = len(srcs)
        Hs, Ws = srcs[0].shape[-2:]
        gvol = gvol.contiguous()
        grads = [torch.empty_like(f) for f in feats]        # same storage layout as the kernel-side features
        pp, keep = _lib.ptr_array(srcs)
        # scratch for the multi-kernel routes (0 words when none applies); torch's caching allocator owns it, the library
        # never allocates
        need = 0
        if BWD_WORKSPACE == "full":
            need = _lib.bwd_workspace(S, B, C, G, D, H, W, Hs, Ws, layout)
        elif BWD_WORKSPACE == "binned":
            need = max(0, _lib.bwd_workspace(S, B, C, G, D, H, W, Hs, Ws, layout, route="binned"))
        elif BWD_WORKSPACE == "coefficients" and layout == _lib.LAYOUT_NHWC:
            need = B * S * H * W * D * G
        ws = torch.empty(need, dtype=torch.float32, device=ref.device) if need > 0 else None
        # every leaf NCHW and the owner kernels run: they (and the coefficient kernel) write the gradients as planes directly
        planar_grads = (PLANAR_GRADS and layout == _lib.LAYOUT_NHWC and not ctx.cl_leaf and
                        _lib.bwd_route(S, B, C, G, D, H, W, Hs, Ws, layout, need)[1] == "binned")
        if planar_grads:
            grads = [torch.empty(f.shape, dtype=torch.float32, device=f.device) for f in feats]
            layout = layout | _lib.GRAD_PLANAR
        gp, gkeep = _lib.ptr_array(grads[1:])
        with _on(ref.device):
            _lib.call("mvster_cost_volume_bwd_ws", gvol.data_ptr(), vol.data_ptr(), wsum.data_ptr(), ref.data_ptr(), pp, S,
                      pose.data_ptr(), hypo.data_ptr(), grads[0].data_ptr(), gp, B, C, G, D, H, W, Hs, Ws, attn_temp,
                      layout, ws.data_ptr() if ws is not None else None, need,
                      ctx.bin_counts.data_ptr() if ctx.bin_counts is not None else None, _stream(ref.device))
        del keep, gkeep, ws
        if layout == _lib.LAYOUT_NHWC:                      # (not taken with planar_grads: the flag is part of `layout`)
            todo = [i for i, cl in enumerate(was_cl) if not cl and not grads[i].is_contiguous()]
            if todo:                                        # gradients of the planar leaves, one launch
                for i, d in zip(todo, _convert_many([grads[i] for i in todo], to_nhwc=False)):
                    grads[i] = d
        return (None, None, None, None, None) + tuple(grads)


class _GenericCostVolume(torch.autograd.Function):
    """The fusion for everything the specialised kernels do not instantiate (csrc/cost_volume_generic.hip): any channel
    count, ``group_cor=False`` (squared difference), ``attn_fuse_d=False`` (per-pixel maximum of the softmax as the view
    weight), any D <= 64.  Planar feature maps and gradients."""

    @staticmethod
    def forward(ctx, pose, hypo, attn_temp, G, group_cor, fuse_d, ref, *srcs):
        B, C, H, W = ref.shape
        D = hypo.shape[1]
        S = len(srcs)
        Hs, Ws = srcs[0].shape[-2:]
        feats = tuple(f.contiguous() for f in (ref,) + tuple(srcs))
        vol = torch.empty(B, G if group_cor else C, D, H, W, dtype=torch.float32, device=ref.device)
        wsum = torch.empty(B, D, H, W, dtype=torch.float32, device=ref.device)
        pp, keep = _lib.ptr_array(feats[1:])
        with _on(ref.device):
            _lib.call("mvster_cost_volume_generic_fwd", feats[0].data_ptr(), pp, S, pose.data_ptr(), hypo.data_ptr(),
                      vol.data_ptr(), wsum.data_ptr(), B, C, int(G), D, H, W, Hs, Ws, float(attn_temp), int(bool(group_cor)),
                      int(bool(fuse_d)), _stream(ref.device))
        del keep
        ctx.save_for_backward(pose, hypo, vol, wsum, *feats)
        ctx.cfg = (float(attn_temp), int(G), bool(group_cor), bool(fuse_d))
        return vol

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gvol):
        pose, hypo, vol, wsum = ctx.saved_tensors[:4]
        feats = ctx.saved_tensors[4:]
        attn_temp, G, group_cor, fuse_d = ctx.cfg
        ref, srcs = feats[0], feats[1:]
        B, C, H, W = ref.shape
        D = hypo.shape[1]
        S = len(srcs)
        Hs, Ws = srcs[0].shape[-2:]
        gvol = gvol.contiguous()
        grads = [torch.empty_like(f) for f in feats]
        pp, keep = _lib.ptr_array(srcs)
        gp, gkeep = _lib.ptr_array(grads[1:])
        with _on(ref.device):
            _lib.call("mvster_cost_volume_generic_bwd", gvol.data_ptr(), vol.data_ptr(), wsum.data_ptr(), ref.data_ptr(), pp, S,
                      pose.data_ptr(), hypo.data_ptr(), grads[0].data_ptr(), gp, B, C, G, D, H, W, Hs, Ws, attn_temp,
                      int(group_cor), int(fuse_d), _stream(ref.device))
        del keep, gkeep
        return (None, None, None, None, None, None) + tuple(grads)


_FUSED_CG = {(8, 4), (16, 4), (32, 4), (64, 4), (8, 8), (16, 8), (32, 8), (64, 8)}
_FUSED_D = {4, 8, 16}
_MAX_SRC = 32


def fused_supported(C, G, D, S):
    """Shapes instantiated in the fused kernels (mirrors check_volume() in csrc/cost_volume_common.h)."""
    return (C, G) in _FUSED_CG and D in _FUSED_D and 1 <= S <= _MAX_SRC


def fused_cost_volume(features, pose, depth_hypo, group_cor_dim, attn_temp, layout="nhwc"):
    """Fused warp + group correlation + softmax-over-depth fusion -> [B,G,D,H,W] (differentiable
    w.r.t. every feature map).  ``pose`` = stage_poses(proj_matrices).  ``layout`` selects the kernel family:
    "nhwc" (default; gathers from channels-last storage, converting NCHW inputs once) or "nchw" (planar gathers)."""
    ref = _need(features[0], "features[0]", 4, keep_layout=True)
    srcs = [_need(f, "features[%d]" % (i + 1), 4, keep_layout=True) for i, f in enumerate(features[1:])]
    hypo = _need(depth_hypo, "depth_hypo", 4).detach()
    B, C, H, W = ref.shape
    for s in srcs:
        if s.shape[:2] != ref.shape[:2] or s.shape[2:] != srcs[0].shape[2:]:
            raise RuntimeError("all source feature maps must share [B,C] with the reference and one [Hs,Ws]")
    if hypo.shape[0] != B or tuple(hypo.shape[2:]) != (H, W):
        raise RuntimeError("depth_hypo must be [B,D,H,W] at the reference feature resolution")
    lay = {"nhwc": _lib.LAYOUT_NHWC, "nchw": _lib.LAYOUT_NCHW}[layout]
    return _FusedCostVolume.apply(pose, hypo, float(attn_temp), int(group_cor_dim), lay, ref, *srcs)


# ----------------------------------------------------------------------------------------------
# A4 depth regression
# ----------------------------------------------------------------------------------------------
class _DepthRegress(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, hypo, want_conf, want_range, split_itv):
        B, D, H, W = logits.shape
        dev = logits.device
        attn = torch.empty_like(logits)
        depth = torch.empty(B, H, W, dtype=torch.float32, device=dev)
        conf = torch.empty(B, H, W, dtype=torch.float32, device=dev) if want_conf else None
        inv_min = torch.empty(B, H, W, dtype=torch.float32, device=dev) if want_range else None
        inv_max = torch.empty(B, H, W, dtype=torch.float32, device=dev) if want_range else None
        index = torch.empty(B, H, W, dtype=torch.int32, device=dev)
        with _on(dev):
            _lib.call("mvster_depth_regress_fwd", logits.data_ptr(), hypo.data_ptr(), attn.data_ptr(), depth.data_ptr(),
                      index.data_ptr(), conf.data_ptr() if want_conf else None,
                      inv_min.data_ptr() if want_range else None, inv_max.data_ptr() if want_range else None,
                      B, D, H, W, float(split_itv), _stream(dev))
        ctx.save_for_backward(attn)
        empty = logits.new_empty(0)
        outs = (attn, depth, index, conf if want_conf else empty, inv_min if want_range else empty,
                inv_max if want_range else empty)
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gattn, *_unused):
        (attn,) = ctx.saved_tensors
        B, D, H, W = attn.shape
        gattn = gattn.contiguous()
        glogits = torch.empty_like(attn)
        with _on(attn.device):
            _lib.call("mvster_softmax_bwd", attn.data_ptr(), gattn.data_ptr(), glogits.data_ptr(), B, D, H, W,
                      _stream(attn.device))
        return glogits, None, None, None, None


def depth_regress(logits, depth_hypo, training, inverse_depth=True, split_itv=1.0):
    """Post-regnet half of the stage.  Returns dict(attn_weight, depth, index[, photometric_confidence]
    [, inverse_min_depth, inverse_max_depth]); only ``attn_weight`` carries gradient."""
    logits = _need(logits, "regnet output", 4)
    hypo = _need(depth_hypo, "depth_hypo", 4)
    if logits.shape != hypo.shape:
        raise RuntimeError("regnet output %s and depth_hypo %s must both be [B,D,H,W]"
                           % (tuple(logits.shape), tuple(hypo.shape)))
    if inverse_depth and logits.shape[1] < 3:
        raise IndexError("inverse_depth needs at least 3 hypotheses (reads depth_hypo[:,2], "
                         "models/mvs4net_utils.py:1152)")
    attn, depth, index, conf, inv_min, inv_max = _DepthRegress.apply(logits, hypo.detach(), not training,
                                                                    bool(inverse_depth), float(split_itv))
    out = {"attn_weight": attn, "depth": depth, "index": index}
    if not training:
        out["photometric_confidence"] = conf
    if inverse_depth:
        out["inverse_min_depth"] = inv_min
        out["inverse_max_depth"] = inv_max
    return out


# ----------------------------------------------------------------------------------------------
# A5/A6 hypothesis scheduling
# ----------------------------------------------------------------------------------------------
def _init(cur_depth, ndepths, device, dtype, H, W, inverse):
    if dtype is not None and dtype != torch.float32:
        raise RuntimeError("depth hypotheses are float32 (got dtype %s)" % dtype)
    device = torch.device(device) if device is not None else cur_depth.device
    if device.type != "cuda":
        raise RuntimeError("init range requested on %s: the MI355X hot path has no CPU fallback" % device)
    if not isinstance(cur_depth, torch.Tensor) or cur_depth.dim() != 2:
        raise RuntimeError("cur_depth must be a [B,K] tensor (depth_values)")
    cur = cur_depth.to(device=device, dtype=torch.float32).contiguous()
    B, K = cur.shape
    out = torch.empty(B, int(ndepths), int(H), int(W), dtype=torch.float32, device=device)
    with _on(device):
        _lib.call("mvster_init_range", cur.data_ptr(), K, out.data_ptr(), B, int(ndepths), int(H), int(W),
                  1 if inverse else 0, _stream(device))
    return out


def init_range(cur_depth, ndepths, device, dtype, H, W):
    """Linear first-stage hypotheses [B,D,H,W] from cur_depth[:,0] .. cur_depth[:,-1]."""
    return _init(cur_depth, ndepths, device, dtype, H, W, inverse=False)


def init_inverse_range(cur_depth, ndepths, device, dtype, H, W):
    """Inverse-depth first-stage hypotheses [B,D,H,W]; plane 0 is the far plane."""
    return _init(cur_depth, ndepths, device, dtype, H, W, inverse=True)


def schedule_inverse_range(inverse_min_depth, inverse_max_depth, ndepths, H, W):
    """Next-stage hypotheses [B,D,H,W] from the previous stage's per-pixel inverse-depth range [B,H//2,W//2]."""
    imin = _need(inverse_min_depth, "inverse_min_depth", 3)
    imax = _need(inverse_max_depth, "inverse_max_depth", 3)
    if imin.shape != imax.shape:
        raise RuntimeError("inverse_min_depth and inverse_max_depth must have the same shape")
    B, Hp, Wp = imin.shape
    if (Hp, Wp) != (int(H) // 2, int(W) // 2):
        # the reference builds its ramp at (H//2, W//2) and broadcasts against the inputs (:90-92)
        raise RuntimeError("previous-stage maps are %dx%d, expected %dx%d (H//2, W//2)" % (Hp, Wp, H // 2, W // 2))
    out = torch.empty(B, int(ndepths), int(H), int(W), dtype=torch.float32, device=imin.device)
    with _on(imin.device):
        _lib.call("mvster_schedule_inverse_range", imin.data_ptr(), imax.data_ptr(), out.data_ptr(), B, int(ndepths),
                  Hp, Wp, int(H), int(W), _stream(imin.device))
    return out


# ----------------------------------------------------------------------------------------------
# A2+A4 stagenet
# ----------------------------------------------------------------------------------------------
def _debug_bits(n):
    return {p for p, v in enumerate(bin(int(n))[:1:-1]) if int(v)}


class stagenet(nn.Module):
    """One coarse-to-fine stage: cost volume over the source views, ``regnet`` callback, depth pick.

    Constructor and ``forward`` signatures are the reference's (models/mvs4net_utils.py:1018,1027).
    The module owns no parameters or buffers, so checkpoints are unaffected.
    """

    def __init__(self, inverse_depth=False, mono=False, attn_fuse_d=True, vis_ETA=False, attn_temp=1, debug=0):
        super(stagenet, self).__init__()
        self.inverse_depth = inverse_depth
        self.mono = mono
        self.attn_fuse_d = attn_fuse_d
        self.vis_ETA = vis_ETA
        self.attn_temp = attn_temp
        self.debug = debug

    def forward(self, features, proj_matrices, depth_hypo, regnet, stage_idx, group_cor=False, group_cor_dim=8,
                split_itv=1, fn=None):
        if self.vis_ETA:
            raise NotImplementedError("vis_ETA (debug dump of the epipolar sampling grid / view weights, "
                                      "models/mvs4net_utils.py:1041-1046,1073-1075) is not part of the HIP path")
        if _debug_bits(self.debug) & {5, 6, 7}:
            raise NotImplementedError("debug bits 5-7 open interactive cv2 windows in the reference "
                                      "(models/mvs4net_utils.py:1054,1088,1117) and are not supported")
        ref_feature = features[0]
        ref = _need(ref_feature, "features[0]", 4, keep_layout=True)
        srcs = [_need(f, "features[%d]" % (i + 1), 4, keep_layout=True) for i, f in enumerate(features[1:])]
        hypo = _need(depth_hypo, "depth_hypo", 4)
        B, D, H, W = hypo.shape
        C = ref.shape[1]
        S = len(srcs)
        if S < 1:
            raise RuntimeError("stagenet needs at least one source view")
        if proj_matrices.shape[1] != S + 1:
            raise RuntimeError("proj_matrices has %d views, features has %d" % (proj_matrices.shape[1], S + 1))
        pose = stage_poses(proj_matrices)

        if group_cor and self.attn_fuse_d and fused_supported(C, int(group_cor_dim), D, S):
            vol = fused_cost_volume([ref] + srcs, pose, hypo, group_cor_dim, self.attn_temp)
        else:
            # modes / shapes outside the specialised kernels: the generic HIP kernels (planar maps, any C, G, D <= 64)
            if S > _MAX_SRC or D > 64:
                raise RuntimeError("at most %d source views and 64 depth hypotheses per stage (got %d, %d)" % (_MAX_SRC, S, D))
            if group_cor and C % int(group_cor_dim) != 0:
                raise RuntimeError("group_cor_dim %d does not divide the %d feature channels" % (int(group_cor_dim), C))
            for t in srcs:
                if t.shape[:2] != ref.shape[:2] or t.shape[2:] != srcs[0].shape[2:]:
                    raise RuntimeError("all source feature maps must share [B,C] with the reference and one [Hs,Ws]")
            vol = _GenericCostVolume.apply(pose, hypo.detach(), float(self.attn_temp), int(group_cor_dim), bool(group_cor),
                                           bool(self.attn_fuse_d), ref, *srcs)

        attn_logits = regnet(vol)                                     # B D H W
        picked = depth_regress(attn_logits, hypo, self.training, self.inverse_depth, split_itv)
        if self.training:
            conf = torch.zeros((), dtype=torch.float32, device=ref.device)
        else:
            conf = picked["photometric_confidence"]
        ret_dict = {"depth": picked["depth"],
                    "photometric_confidence": conf,
                    "hypo_depth": depth_hypo,
                    "attn_weight": picked["attn_weight"]}
        if self.inverse_depth:
            ret_dict["inverse_min_depth"] = picked["inverse_min_depth"]
            ret_dict["inverse_max_depth"] = picked["inverse_max_depth"]
        if self.mono:
            ret_dict["mono_feat"] = ref_feature
        return ret_dict
