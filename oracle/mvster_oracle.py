"""TEST INFRASTRUCTURE -- CPU restatement ("oracle") of the MVSTER cost-volume hot path.

This file is NOT part of the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker / the timed CPU
baseline.  The product path (``deep_reconstruction_with_epipolar_lines_mvster_amd``) never routes
through it and has no CPU fallback.

What it restates (all citations relative to the upstream reference tree):

* ``homo_warping``                    models/mvs4net_utils.py:21-67
* ``stagenet.forward`` pre-regnet     models/mvs4net_utils.py:1030-1102
* ``stagenet.forward`` post-regnet    models/mvs4net_utils.py:1105-1162
* ``init_range``                      models/mvs4net_utils.py:69-77
* ``init_inverse_range``              models/mvs4net_utils.py:79-85
* ``schedule_inverse_range``          models/mvs4net_utils.py:87-94
* ``sinkhorn``                        models/mvs4net_utils.py:1164-1210   
* ``MVS4net_loss``                    models/MVS4Net.py:195-240

It deliberately uses the same torch operator *sequence* where the operator decides the fp32
rounding (``torch.inverse``, ``matmul``, the [-1,1] grid round trip through ``F.grid_sample``,
``torch.softmax``, ``max``/``gather``, trilinear ``F.interpolate``), because that sequence IS the
reference's CPU PyTorch path whose time ``bench.py`` reports as ``cpu_baseline`` (kind "port").

Parity pin: ``tests/test_oracle_golden.py`` checks every function here against
``tests/golden/*.npz``, which were produced by ``tests/golden/make_golden.py`` importing the
reference itself on CPU (the reference has no tests or fixtures of its own).
"""
import math

import torch
import torch.nn.functional as F

__all__ = [
    "compose_projection", "relative_pose", "sampling_grid", "homo_warping",
    "fuse_source_views", "regress_depth", "stage_forward",
    "init_range", "init_inverse_range", "schedule_inverse_range",
    "sinkhorn", "mvs4net_loss",
]


# --------------------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------------------
def compose_projection(view_mats):
    """[B,2,4,4] (extrinsic, intrinsic) -> 4x4 with rows 0..2 = K[:3,:3] @ E[:3,:4].

    Follows models/mvs4net_utils.py:1047-1050 (clone of the extrinsic, top 3x4 overwritten).
    """
    extr = view_mats[:, 0]
    intr = view_mats[:, 1]
    full = extr.clone()
    full[:, :3, :4] = torch.matmul(intr[:, :3, :3], extr[:, :3, :4])
    return full


def relative_pose(src_proj, ref_proj):
    """rot [B,3,3], trans [B,3,1] of src_proj @ inverse(ref_proj) (mvs4net_utils.py:32-34)."""
    rel = torch.matmul(src_proj, torch.inverse(ref_proj))
    return rel[:, :3, :3], rel[:, :3, 3:4]


def sampling_grid(rot, trans, depth_values, Hs, Ws):
    """Normalised [-1,1] sampling grid [B, D, Hr*Wr, 2] (mvs4net_utils.py:36-53).

    x = column, y = row of the reference pixel; p = (rot @ [x,y,1]) * depth + trans;
    z == 0 is replaced by 1e-9; the grid is normalised with the SOURCE map size.
    """
    B, D, Hr, Wr = depth_values.shape
    dev = depth_values.device
    # float32 in the reference (:36-37); the dtype follows the inputs so that the same restatement run
    # in float64 serves as the exact arbiter when the fp32 paths disagree at rounding level
    fdt = depth_values.dtype
    rows = torch.arange(0, Hr, dtype=fdt, device=dev)
    cols = torch.arange(0, Wr, dtype=fdt, device=dev)
    yy, xx = torch.meshgrid(rows, cols, indexing="ij")
    homog = torch.stack((xx.reshape(-1), yy.reshape(-1), torch.ones(Hr * Wr, dtype=fdt, device=dev)))
    homog = homog.unsqueeze(0).expand(B, 3, Hr * Wr)
    rotated = torch.matmul(rot, homog)                                   # [B,3,HW]
    pts = rotated.unsqueeze(2) * depth_values.reshape(B, 1, D, Hr * Wr)   # [B,3,D,HW]
    pts = pts + trans.reshape(B, 3, 1, 1)
    z = pts[:, 2:3]
    z = torch.where(z == 0, torch.full_like(z, 1e-9), z)
    uv = pts[:, :2] / z                                                  # [B,2,D,HW]
    gx = uv[:, 0] / ((Ws - 1) / 2) - 1
    gy = uv[:, 1] / ((Hs - 1) / 2) - 1
    return torch.stack((gx, gy), dim=3)


def homo_warping(src_fea, src_proj, ref_proj, depth_values):
    """Warp src features onto the D hypotheses of the reference view -> [B,C,D,Hr,Wr].

    models/mvs4net_utils.py:21-67 (4-D ``src_fea`` branch; grid under no_grad, bilinear,
    zero padding, align_corners=True).
    """
    B, C, Hs, Ws = src_fea.shape
    _, D, Hr, Wr = depth_values.shape
    with torch.no_grad():
        rot, trans = relative_pose(src_proj, ref_proj)
        grid = sampling_grid(rot, trans, depth_values, Hs, Ws)
    out = F.grid_sample(src_fea, grid.reshape(B, D * Hr, Wr, 2), mode="bilinear",
                        padding_mode="zeros", align_corners=True)
    return out.reshape(B, C, D, Hr, Wr)


# --------------------------------------------------------------------------------------
# stage body
# --------------------------------------------------------------------------------------
def fuse_source_views(features, proj_matrices, depth_hypo, group_cor=False, group_cor_dim=8,
                      attn_temp=1.0, attn_fuse_d=True, return_aux=False):
    """Pre-regnet half of ``stagenet.forward`` (mvs4net_utils.py:1030-1102).

    features: list of N [B,C,H,W] (0 = reference); proj_matrices [B,N,2,4,4];
    depth_hypo [B,D,H,W].  Returns the fused correlation volume [B,G|C,D,H,W]
    (and, with ``return_aux``, the weight sum that normalised it).
    """
    views = torch.unbind(proj_matrices, 1)
    ref_fea, ref_mats = features[0], views[0]
    B, D, H, W = depth_hypo.shape
    C = ref_fea.shape[1]
    ref_vol = ref_fea.unsqueeze(2).repeat(1, 1, D, 1, 1)
    weight_sum = 1e-8
    fused = 0
    for src_fea, src_mats in zip(features[1:], views[1:]):
        src_p = compose_projection(src_mats)
        ref_p = compose_projection(ref_mats)
        warped = homo_warping(src_fea, src_p, ref_p, depth_hypo)
        if group_cor:
            G = group_cor_dim
            cor = (warped.reshape(B, G, C // G, D, H, W) * ref_vol.reshape(B, G, C // G, D, H, W)).mean(2)
        else:
            cor = (ref_vol - warped) ** 2
        if attn_fuse_d:
            wgt = torch.softmax(cor.sum(1) / attn_temp, 1) / math.sqrt(C)   # [B,D,H,W]
            weight_sum = weight_sum + wgt
            fused = fused + wgt.unsqueeze(1) * cor
        else:
            wgt = torch.softmax(cor.sum(1), 1).max(1)[0]                    # [B,H,W]
            weight_sum = weight_sum + wgt
            fused = fused + wgt.unsqueeze(1).unsqueeze(1) * cor
    if attn_fuse_d:
        fused = fused / weight_sum.unsqueeze(1)
    else:
        fused = fused / weight_sum.unsqueeze(1).unsqueeze(1)
    if return_aux:
        return fused, weight_sum
    return fused


def regress_depth(logits, depth_hypo, training, inverse_depth=True, split_itv=1.0):
    """Post-regnet half of ``stagenet.forward`` (mvs4net_utils.py:1105-1162).

    Eval mode: confidence = max raw logit / sum of raw logits (:1109-1113,1135-1138).
    depth = depth_hypo gathered at the FIRST arg-max of softmax(logits) (:1126-1130).
    """
    out = {}
    if not training:
        with torch.no_grad():
            raw_idx = logits.max(1, keepdim=True)[1]
            raw_max = torch.gather(logits, 1, raw_idx).squeeze(1)
            raw_sum = logits.sum(1)
    attn = F.softmax(logits, dim=1)
    idx = attn.max(1, keepdim=True)[1]
    depth = torch.gather(depth_hypo, 1, idx).squeeze(1)
    if not training:
        with torch.no_grad():
            conf = raw_max / raw_sum
    else:
        conf = torch.tensor(0.0, dtype=torch.float32, device=logits.device)   # (:1144)
    out["depth"] = depth
    out["photometric_confidence"] = conf
    out["hypo_depth"] = depth_hypo
    out["attn_weight"] = attn
    out["argmax_index"] = idx.squeeze(1)          # oracle-only extra, for bit-exact index checks
    if inverse_depth:
        step = 1.0 / depth_hypo[:, 2] - 1.0 / depth_hypo[:, 1]
        out["inverse_min_depth"] = 1 / depth + split_itv * step
        out["inverse_max_depth"] = 1 / depth - split_itv * step
    return out


def stage_forward(features, proj_matrices, depth_hypo, regnet, training, group_cor=False,
                  group_cor_dim=8, split_itv=1.0, inverse_depth=True, mono=False,
                  attn_fuse_d=True, attn_temp=1.0):
    """Whole ``stagenet.forward`` (mvs4net_utils.py:1027-1162) with the regnet callback."""
    vol = fuse_source_views(features, proj_matrices, depth_hypo, group_cor, group_cor_dim,
                            attn_temp, attn_fuse_d)
    logits = regnet(vol)
    out = regress_depth(logits, depth_hypo, training, inverse_depth, split_itv)
    if mono:
        out["mono_feat"] = features[0]
    return out


# --------------------------------------------------------------------------------------
# hypothesis scheduling
# --------------------------------------------------------------------------------------
def init_range(cur_depth, ndepths, device, dtype, H, W):
    """Linear stage-1 hypotheses (mvs4net_utils.py:69-77)."""
    dmin = cur_depth[:, 0]
    dmax = cur_depth[:, -1]
    step = (dmax - dmin) / (ndepths - 1)
    ks = torch.arange(0, ndepths, device=device, dtype=dtype).reshape(1, -1)
    planes = dmin.unsqueeze(1) + ks * step.unsqueeze(1)
    return planes[:, :, None, None].repeat(1, 1, H, W)


def init_inverse_range(cur_depth, ndepths, device, dtype, H, W):
    """Inverse-depth stage-1 hypotheses; plane 0 is the FAR plane (mvs4net_utils.py:79-85)."""
    inv_min = 1.0 / cur_depth[:, 0]
    inv_max = 1.0 / cur_depth[:, -1]
    frac = torch.arange(0, ndepths, device=device, dtype=dtype).reshape(1, -1, 1, 1).repeat(1, 1, H, W) / (ndepths - 1)
    inv = inv_max[:, None, None, None] + (inv_min - inv_max)[:, None, None, None] * frac
    return 1.0 / inv


def schedule_inverse_range(inverse_min_depth, inverse_max_depth, ndepths, H, W):
    """Next-stage hypotheses from the previous stage's inverse range (mvs4net_utils.py:87-94)."""
    frac = torch.arange(0, ndepths, device=inverse_min_depth.device,
                        dtype=inverse_min_depth.dtype).reshape(1, -1, 1, 1).repeat(1, 1, H // 2, W // 2) / (ndepths - 1)
    inv = inverse_max_depth[:, None] + (inverse_min_depth - inverse_max_depth)[:, None] * frac
    inv = F.interpolate(inv.unsqueeze(1), [ndepths, H, W], mode="trilinear", align_corners=True).squeeze(1)
    return 1.0 / inv


# --------------------------------------------------------------------------------------
# optimal-transport loss
# --------------------------------------------------------------------------------------
def sinkhorn(gt_depth, hypo_depth, attn_weight, mask, iters, eps=1, continuous=False):
    """Log-domain Sinkhorn between the depth distribution and the GT bin (mvs4net_utils.py:1164-1210).

    Returns (T_map [B,HW,D,D(+1)], loss) with loss = mean over masked pixels of sum(T * cost)."""
    B, D, H, W = attn_weight.shape
    dev = gt_depth.device
    ramp = torch.stack([torch.arange(-i, D - i, 1, dtype=torch.float32, device=dev) for i in range(D)], dim=1).abs()
    if not continuous:
        cost = ramp[None, None].repeat(B, H * W, 1, 1)
        gt_idx = torch.abs(hypo_depth - gt_depth[:, None]).min(1)[1].reshape(B * H * W, 1)
        target = torch.zeros_like(hypo_depth).permute(0, 2, 3, 1).reshape(B * H * W, D)
        target.scatter_add_(1, gt_idx, torch.ones([target.shape[0], 1], dtype=target.dtype, device=dev))
        target = target.reshape(B, H * W, D)
    else:
        target = torch.zeros((B, H * W, D + 1), dtype=torch.float32, device=dev)
        target[:, :, -1] = 1
        cost = torch.zeros((B, D, D + 1), dtype=torch.float32, device=dev)
        cost[:, :D, :D] = ramp.unsqueeze(0)
        cost = cost[:, None, None].repeat(1, H, W, 1, 1)
        step = 1 / hypo_depth[:, 2] - 1 / hypo_depth[:, 1]
        gt_bin = (1 / gt_depth - 1 / hypo_depth[:, 0]) / step
        gt_bin[~mask] = 10
        cost[..., -1] = torch.stack([(gt_bin - i).abs() for i in range(D)], dim=1).permute(0, 2, 3, 1)
        cost = cost.reshape(B, H * W, D, D + 1)
    pred = attn_weight.permute(0, 2, 3, 1).reshape(B, H * W, D)
    log_mu = (target + 1e-12).log()
    log_nu = (pred + 1e-12).log()
    u, v = torch.zeros_like(log_nu), torch.zeros_like(log_mu)
    for _ in range(iters):
        v = log_mu - torch.logsumexp(cost / eps + u.unsqueeze(3), dim=2)
        u = log_nu - torch.logsumexp(cost / eps + v.unsqueeze(2), dim=3)
    plan = (cost / eps + u.unsqueeze(3) + v.unsqueeze(2)).exp()
    loss = (plan * cost).reshape(B * H * W, -1)[mask.reshape(-1)].sum(-1).mean()
    return plan, loss


def mvs4net_loss(inputs, depth_gt_ms, mask_ms, **kwargs):
    """Per-stage OT (+ mono L1) loss (models/MVS4Net.py:195-240) -> (total, l1[], ot[], range_err_ratio[])."""
    stage_lw = kwargs.get("stage_lw", [1, 1, 1, 1])
    l1ot_lw = kwargs.get("l1ot_lw", [0, 1])
    inverse = kwargs.get("inverse_depth", False)
    iters = kwargs.get("ot_iter", 3)
    eps = kwargs.get("ot_eps", 1)
    continuous = kwargs.get("ot_continous", False)
    mono = kwargs.get("mono", False)
    total = torch.tensor(0.0, dtype=torch.float32, device=mask_ms["stage1"].device)
    l1s, ots, misses = [], [], []
    for n, key in enumerate([k for k in inputs.keys() if "stage" in k]):
        out = inputs[key]
        hypo, attn = out["hypo_depth"], out["attn_weight"]
        mask = mask_ms[key] > 0.5
        gt = depth_gt_ms[key]
        if mono and n != 0:
            l1 = F.l1_loss(out["mono_depth"][mask], gt[mask], reduction="mean")
        else:
            l1 = torch.tensor(0.0, dtype=torch.float32, device=total.device)
        if inverse:
            itv = (1 / hypo[:, 2] - 1 / hypo[:, 1]).abs()
            miss = ((1 / hypo - 1 / gt.unsqueeze(1)).abs() <= itv.unsqueeze(1)).sum(1) == 0
        else:
            itv = (hypo[:, 2] - hypo[:, 1]).abs()
            miss = ((hypo - gt.unsqueeze(1)).abs() <= itv.unsqueeze(1)).sum(1) == 0
        misses.append(miss[mask].float().mean())
        ot = sinkhorn(gt, hypo, attn, mask, iters=iters, eps=eps, continuous=continuous)[1]
        l1s.append(l1)
        ots.append(ot)
        total = total + stage_lw[n] * (l1ot_lw[0] * l1 + l1ot_lw[1] * ot)
    return total, l1s, ots, misses
