"""F1 -- drop-ins for the reference's optimal-transport loss.

    sinkhorn       models/mvs4net_utils.py:1164-1210
    MVS4net_loss   models/MVS4Net.py:195-240
    Blend_loss     models/MVS4Net.py:243-291

``sinkhorn`` runs as one fused per-pixel HIP kernel (csrc/sinkhorn.hip) instead of ~25 elementwise /
logsumexp operators per iteration over materialised [B,HW,D,D] tensors; the two ``*_loss`` functions are the
reference's per-stage bookkeeping around it (same keyword arguments and return tuples).
"""
import torch
import torch.nn.functional as F

from . import _lib
from .hotpath import _need, _on, _stream

__all__ = ["sinkhorn", "MVS4net_loss", "Blend_loss"]

_FUSED_D = (4, 8, 16)
_MAX_ITERS = 16


class _SinkhornPixelLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attn, gt, hypo, mask_u8, iters, eps, continuous, want_tmap):
        B, D, H, W = attn.shape
        pix = torch.empty(B, H, W, dtype=torch.float32, device=attn.device)
        tmap = None
        if want_tmap:
            tmap = torch.empty(B, H * W, D, D + (1 if continuous else 0), dtype=torch.float32, device=attn.device)
        with _on(attn.device):
            _lib.call("mvster_sinkhorn_fwd", gt.data_ptr(), hypo.data_ptr(), attn.data_ptr(), mask_u8.data_ptr(),
                      pix.data_ptr(), tmap.data_ptr() if want_tmap else None, B, D, H, W, int(iters), float(eps),
                      1 if continuous else 0, _stream(attn.device))
        ctx.save_for_backward(attn, gt, hypo, mask_u8)
        ctx.cfg = (int(iters), float(eps), bool(continuous))
        if want_tmap:
            ctx.mark_non_differentiable(tmap)
            return pix, tmap
        return pix, attn.new_empty(0)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gpix, _gtmap):
        attn, gt, hypo, mask_u8 = ctx.saved_tensors
        iters, eps, continuous = ctx.cfg
        B, D, H, W = attn.shape
        gpix = gpix.contiguous()
        gattn = torch.empty_like(attn)
        with _on(attn.device):
            _lib.call("mvster_sinkhorn_bwd", gt.data_ptr(), hypo.data_ptr(), attn.data_ptr(), mask_u8.data_ptr(),
                      gpix.data_ptr(), gattn.data_ptr(), B, D, H, W, iters, eps, 1 if continuous else 0,
                      _stream(attn.device))
        return gattn, None, None, None, None, None, None, None


class _SinkhornPixelLossGeneric(torch.autograd.Function):
    """Shapes the register-resident kernels do not instantiate (D not in 4/8/16 up to 64, more than 16 sweeps): the generic
    HIP kernels of csrc/sinkhorn.hip (plain loops, vectors and per-sweep history in a workspace from torch's allocator)."""

    @staticmethod
    def forward(ctx, attn, gt, hypo, mask_u8, iters, eps, continuous, want_tmap):
        B, D, H, W = attn.shape
        dev = attn.device
        pix = torch.empty(B, H, W, dtype=torch.float32, device=dev)
        tmap = torch.empty(B, H * W, D, D + (1 if continuous else 0), dtype=torch.float32, device=dev) if want_tmap else None
        need = _lib.load().mvster_sinkhorn_generic_workspace(B, D, H, W, int(iters), 1 if continuous else 0, 0)
        ws = torch.empty(need, dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.call("mvster_sinkhorn_generic_fwd", gt.data_ptr(), hypo.data_ptr(), attn.data_ptr(), mask_u8.data_ptr(),
                      pix.data_ptr(), tmap.data_ptr() if want_tmap else None, ws.data_ptr(), need, B, D, H, W, int(iters),
                      float(eps), 1 if continuous else 0, _stream(dev))
        ctx.save_for_backward(attn, gt, hypo, mask_u8)
        ctx.cfg = (int(iters), float(eps), bool(continuous))
        if want_tmap:
            ctx.mark_non_differentiable(tmap)
            return pix, tmap
        return pix, attn.new_empty(0)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gpix, _gtmap):
        attn, gt, hypo, mask_u8 = ctx.saved_tensors
        iters, eps, continuous = ctx.cfg
        B, D, H, W = attn.shape
        dev = attn.device
        gpix = gpix.contiguous()
        gattn = torch.empty_like(attn)
        need = _lib.load().mvster_sinkhorn_generic_workspace(B, D, H, W, iters, 1 if continuous else 0, 1)
        ws = torch.empty(need, dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.call("mvster_sinkhorn_generic_bwd", gt.data_ptr(), hypo.data_ptr(), attn.data_ptr(), mask_u8.data_ptr(),
                      gpix.data_ptr(), gattn.data_ptr(), ws.data_ptr(), need, B, D, H, W, iters, eps, 1 if continuous else 0,
                      _stream(dev))
        return gattn, None, None, None, None, None, None, None


_GENERIC_MAX_D = 64


def _pixel_loss(gt_depth, hypo_depth, attn_weight, mask, iters, eps, continuous, want_tmap):
    attn = _need(attn_weight, "attn_weight", 4)
    hypo = _need(hypo_depth, "hypo_depth", 4).detach()
    gt = _need(gt_depth, "gt_depth", 3).detach()
    if not (isinstance(mask, torch.Tensor) and mask.is_cuda and mask.dtype == torch.bool):
        raise RuntimeError("mask must be a bool tensor on the HIP device (the reference indexes with it)")
    B, D, H, W = attn.shape
    if hypo.shape != attn.shape or tuple(gt.shape) != (B, H, W) or tuple(mask.shape) != (B, H, W):
        raise RuntimeError("sinkhorn: gt_depth/mask must be [B,H,W] and hypo_depth/attn_weight [B,D,H,W]")
    if continuous and D < 3:
        raise IndexError("continuous bins need at least 3 hypotheses (reads hypo_depth[:,2], models/mvs4net_utils.py:1183)")
    if int(iters) < 0:
        raise ValueError("iters must be >= 0")
    mask_u8 = mask.contiguous().view(torch.uint8)
    if D in _FUSED_D and int(iters) <= _MAX_ITERS:
        fn = _SinkhornPixelLoss
    elif D <= _GENERIC_MAX_D:
        fn = _SinkhornPixelLossGeneric
    else:
        raise RuntimeError("sinkhorn: at most %d depth hypotheses (got %d)" % (_GENERIC_MAX_D, D))
    pix, tmap = fn.apply(attn, gt, hypo, mask_u8, int(iters), float(eps), bool(continuous), bool(want_tmap))
    return pix, (tmap if want_tmap else None)


def sinkhorn(gt_depth, hypo_depth, attn_weight, mask, iters, eps=1, continuous=False):
    """gt_depth [B,H,W], hypo_depth/attn_weight [B,D,H,W], mask [B,H,W] bool -> (T_map [B,HW,D,D(+1)], loss).

    ``loss`` = mean over the masked pixels of sum(T * cost), differentiable w.r.t. ``attn_weight``.  ``T_map`` is
    returned for signature parity; it carries no gradient and its rows are NaN where ``mask`` is False (the
    reference computes those rows but never uses them)."""
    pix, tmap = _pixel_loss(gt_depth, hypo_depth, attn_weight, mask, iters, eps, continuous, want_tmap=True)
    return tmap, pix.sum() / mask.sum()


def _stage_loss(gt_depth, hypo_depth, attn_weight, mask, iters, eps, continuous):
    pix, _ = _pixel_loss(gt_depth, hypo_depth, attn_weight, mask, iters, eps, continuous, want_tmap=False)
    return pix.sum() / mask.sum()


def _range_miss(hypo_depth, depth_gt, inverse):
    """Fraction helper of models/MVS4Net.py:220-226: True where no hypothesis lies within one interval of the GT."""
    if inverse:
        itv = (1 / hypo_depth[:, 2] - 1 / hypo_depth[:, 1]).abs()
        near = (1 / hypo_depth - 1 / depth_gt.unsqueeze(1)).abs() <= itv.unsqueeze(1)
    else:
        itv = (hypo_depth[:, 2] - hypo_depth[:, 1]).abs()
        near = (hypo_depth - depth_gt.unsqueeze(1)).abs() <= itv.unsqueeze(1)
    return near.sum(1) == 0


def _masked_mean(values, mask):
    """``values[mask].mean()`` without the boolean gather: no data-dependent shape, hence no host synchronisation (the
    step stays capturable in a HIP graph); an empty mask gives NaN either way.  Entries outside the mask are dropped, not
    multiplied by zero: a NaN / Inf there (invalid ground truth, an overflowed prediction) neither poisons the mean nor gets
    a gradient -- exactly like the reference's indexing."""
    return torch.where(mask, values, values.new_zeros(())).sum() / mask.sum()


def _stage_items(inputs):
    return [(k, inputs[k]) for k in inputs.keys() if "stage" in k]


def _losses(inputs, depth_gt_ms, mask_ms, kwargs):
    stage_lw = kwargs.get("stage_lw", [1, 1, 1, 1])
    l1ot_lw = kwargs.get("l1ot_lw", [0, 1])
    inverse = kwargs.get("inverse_depth", False)
    ot_iter = kwargs.get("ot_iter", 3)
    ot_eps = kwargs.get("ot_eps", 1)
    ot_continous = kwargs.get("ot_continous", False)
    mono = kwargs.get("mono", False)
    dev = mask_ms["stage1"].device
    total = torch.zeros((), dtype=torch.float32, device=dev)
    l1s, ots, misses = [], [], []
    last = None
    for stage_idx, (key, out) in enumerate(_stage_items(inputs)):
        mask = mask_ms[key] > 0.5
        gt = depth_gt_ms[key]
        if mono and stage_idx != 0:
            l1 = _masked_mean((out["mono_depth"] - gt).abs(), mask)          # F.l1_loss(mono[mask], gt[mask])
        else:
            l1 = torch.zeros((), dtype=torch.float32, device=dev)
        misses.append(_masked_mean(_range_miss(out["hypo_depth"], gt, inverse).float(), mask))
        ot = _stage_loss(gt, out["hypo_depth"], out["attn_weight"], mask, ot_iter, ot_eps, ot_continous)
        l1s.append(l1)
        ots.append(ot)
        total = total + stage_lw[stage_idx] * (l1ot_lw[0] * l1 + l1ot_lw[1] * ot)
        last = (out["depth"], gt, mask)
    return total, l1s, ots, misses, last


def MVS4net_loss(inputs, depth_gt_ms, mask_ms, **kwargs):
    """-> (total_loss, stage_l1_loss[], stage_ot_loss[], range_err_ratio[])   (models/MVS4Net.py:195-240)"""
    total, l1s, ots, misses, _ = _losses(inputs, depth_gt_ms, mask_ms, kwargs)
    return total, l1s, ots, misses


def Blend_loss(inputs, depth_gt_ms, mask_ms, **kwargs):
    """-> (total_loss, stage_l1_loss[], stage_ot_loss[], range_err_ratio[], epe, err3, err1)  (models/MVS4Net.py:243-291)"""
    depth_max = kwargs.get("depth_max", 100)
    depth_min = kwargs.get("depth_min", 1)
    total, l1s, ots, misses, (pred, gt, mask) = _losses(inputs, depth_gt_ms, mask_ms, kwargs)
    scale = 128 / (depth_max - depth_min)[:, None, None]
    abs_err = (pred * scale - gt * scale).abs()
    return (total, l1s, ots, misses, _masked_mean(abs_err, mask), _masked_mean((abs_err <= 3).float(), mask) * 100,
            _masked_mean((abs_err <= 1).float(), mask) * 100)
