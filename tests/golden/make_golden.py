#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UPSTREAM REFERENCE on CPU.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference is imported read-only from /root/reference with an empty ``cv2`` stub (cv2 is only
used by its interactive debug blocks).  The fixtures hold DATA ONLY: seeded inputs and the
reference's outputs / input-gradients for the hot-path functions of SURVEY.md section 8(a):

  warp_*.npz      homo_warping                       (models/mvs4net_utils.py:21-67)
  stage_*.npz     stagenet.forward, stub regnets     (models/mvs4net_utils.py:1017-1162)
  range_*.npz     init_range / init_inverse_range / schedule_inverse_range (:69-94)
  ties.npz        arg-max tie behaviour of the post-regnet half (:1126-1130)
  sinkhorn_*.npz, loss_mvs4net.npz  sinkhorn (:1164-1210) and MVS4net_loss (models/MVS4Net.py:195-240)
  model_e2e_64.npz  MVS4net (models/MVS4Net.py:16-193) seeded construction checksums, eval + train forward, 64x64
  pfm_io.npz      save_pfm / read_pfm (datasets/data_io.py:6-71): file bytes written and arrays read back
"""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, "/root/reference")

from models.mvs4net_utils import (homo_warping, stagenet, init_range, init_inverse_range,  # noqa: E402
                                  schedule_inverse_range)

torch.set_num_threads(4)


def pinhole_views(B, N, H, W, gen, focal=None, baseline=30.0, yaw_deg=1.5, z_shift=4.0):
    """[B,N,2,4,4] synthetic (extrinsic, intrinsic) pairs: DTU-like pinhole cameras looking down +z."""
    f = focal if focal is not None else 1.8 * W
    mats = torch.zeros(B, N, 2, 4, 4)
    for b in range(B):
        for v in range(N):
            K = torch.eye(4)
            K[0, 0] = f * (1 + 0.01 * v)
            K[1, 1] = f * (1 - 0.005 * v)
            K[0, 2] = (W - 1) / 2 + 0.3 * v
            K[1, 2] = (H - 1) / 2 - 0.2 * v
            a = math.radians(yaw_deg * v * (1 if b % 2 == 0 else -1))
            c = math.radians(0.7 * v)
            Ry = torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
            Rx = torch.tensor([[1, 0, 0], [0, math.cos(c), -math.sin(c)], [0, math.sin(c), math.cos(c)]])
            E = torch.eye(4)
            E[:3, :3] = Ry @ Rx
            E[:3, 3] = torch.tensor([baseline * v, -baseline / 3 * v, z_shift * v]) + 0.5 * torch.randn(3, generator=gen)
            mats[b, v, 0] = E
            mats[b, v, 1] = K
    return mats


def compose(view):  # same arithmetic the reference does inline before calling homo_warping
    P = view[:, 0].clone()
    P[:, :3, :4] = torch.matmul(view[:, 1, :3, :3], view[:, 0, :3, :4])
    return P


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


# ------------------------------------------------------------------ (i) homo_warping
def make_warp():
    g = torch.Generator().manual_seed(101)
    B, C, D, H, W = 2, 16, 4, 12, 16
    # (a) normal geometry, some taps out of bounds; src map is a different size than the ref map
    Hs, Ws = 10, 18
    mats = pinhole_views(B, 2, H, W, g, baseline=45.0)
    src = torch.randn(B, C, Hs, Ws, generator=g, requires_grad=True)
    depth = 400 + 500 * torch.rand(B, D, H, W, generator=g)
    refP, srcP = compose(mats[:, 0]), compose(mats[:, 1])
    out = homo_warping(src, srcP, refP, depth)
    gout = torch.randn(out.shape, generator=g)
    (out * gout).sum().backward()
    save("warp_oob", src=src, src_proj=srcP, ref_proj=refP, depth=depth, out=out, gout=gout, gsrc=src.grad)
    # (b) geometry with z <= 0 for part of the volume (camera behind the plane) and exact z == 0
    src2 = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    mats2 = pinhole_views(B, 2, H, W, g, baseline=20.0, z_shift=-600.0)
    depth2 = 300 + 600 * torch.rand(B, D, H, W, generator=g)
    refP2, srcP2 = compose(mats2[:, 0]), compose(mats2[:, 1])
    out2 = homo_warping(src2, srcP2, refP2, depth2)
    gout2 = torch.randn(out2.shape, generator=g)
    (out2 * gout2).sum().backward()
    save("warp_negz", src=src2, src_proj=srcP2, ref_proj=refP2, depth=depth2, out=out2, gout=gout2, gsrc=src2.grad)
    # (c) identity pose: proj == I so the sample lands exactly on the pixel grid; z==0 plane included
    src3 = torch.randn(1, 4, 6, 7, generator=g)
    eye = torch.eye(4).unsqueeze(0)
    depth3 = torch.rand(1, 3, 6, 7, generator=g) + 0.5
    depth3[0, 1] = 0.0      # rot*[x,y,1]*0 + 0 -> z == 0 -> replaced by 1e-9 (x=y=0 -> 0/1e-9)
    out3 = homo_warping(src3, eye, eye, depth3)
    save("warp_identity", src=src3, src_proj=eye, ref_proj=eye, depth=depth3, out=out3)


# ------------------------------------------------------------------ (ii)+(iii) stagenet
class Recorder:
    """Stub regnet: records the fused volume and reduces it with a fixed per-channel weighting."""

    def __init__(self, chan_weights):
        self.w = chan_weights
        self.seen = None

    def __call__(self, vol):
        self.seen = vol
        return (vol * self.w.view(1, -1, 1, 1, 1)).sum(1)


def make_stage(name, C, G, D, H, W, N, group_cor, attn_fuse_d, training, temp, seed, B=1, split_itv=0.5, plain_sum=False):
    g = torch.Generator().manual_seed(seed)
    mats = pinhole_views(B, N, H, W, g)
    feats = [(0.5 * torch.randn(B, C, H, W, generator=g)).requires_grad_(True) for _ in range(N)]
    # per-pixel inverse-depth hypotheses around a smooth surface (what the cascade produces)
    base = 600 + 120 * torch.rand(B, 1, H, W, generator=g)
    inv = 1.0 / base + (torch.arange(D).view(1, D, 1, 1) - (D - 1) / 2) * 6e-5
    hypo = 1.0 / inv
    nchan = G if group_cor else C
    # x12 so the stub logits spread enough for a peaked (non-uniform) softmax over depth
    cw = torch.ones(nchan) if plain_sum else 12.0 * (0.5 + torch.rand(nchan, generator=g))
    reg = Recorder(cw)
    net = stagenet(inverse_depth=True, mono=True, attn_fuse_d=attn_fuse_d, vis_ETA=False, attn_temp=temp)
    net.train(training)
    out = net(feats, mats, hypo, reg, 0, group_cor=group_cor, group_cor_dim=G, split_itv=split_itv)
    lw = torch.randn(out["attn_weight"].shape, generator=g)
    (out["attn_weight"] * lw).sum().backward()
    arrays = dict(proj=mats, hypo=hypo, chan_w=cw, loss_w=lw, cor_feats=reg.seen,
                  attn_weight=out["attn_weight"], depth=out["depth"],
                  conf=out["photometric_confidence"], inv_min=out["inverse_min_depth"],
                  inv_max=out["inverse_max_depth"],
                  meta=np.array([C, G, D, H, W, N, int(group_cor), int(attn_fuse_d), int(training), B]),
                  temp=np.float32(temp), split_itv=np.float32(split_itv))
    for i, f in enumerate(feats):
        arrays["feat%d" % i] = f
        arrays["gfeat%d" % i] = f.grad
    save(name, **arrays)


def make_stages():
    # the four DTU stage shapes (C,G,D) at tiny resolution, group correlation, train + eval
    make_stage("stage_c64g8d8_train", 64, 8, 8, 8, 10, 3, True, True, True, 2.0, 201)
    make_stage("stage_c32g8d8_eval", 32, 8, 8, 10, 12, 3, True, True, False, 2.0, 202)
    make_stage("stage_c16g4d4_train", 16, 4, 4, 12, 16, 4, True, True, True, 2.0, 203, B=2)
    make_stage("stage_c8g4d4_eval", 8, 4, 4, 14, 16, 5, True, True, False, 2.0, 204, split_itv=1.0)
    make_stage("stage_c8g4d4_train", 8, 4, 4, 14, 16, 5, True, True, True, 2.0, 214, split_itv=1.0, plain_sum=True)
    # hi-res config's first stage: D = 16
    make_stage("stage_c64g8d16_train", 64, 8, 16, 6, 8, 3, True, True, True, 2.0, 205)
    # no group correlation (squared difference volume, C channels out)
    make_stage("stage_c16nogroup_train", 16, 16, 4, 10, 12, 3, False, True, True, 1.0, 206)
    make_stage("stage_c8nogroup_eval", 8, 8, 8, 10, 12, 3, False, True, False, 2.0, 207)
    # attn_fuse_d = False (per-pixel view weight)
    make_stage("stage_c16g4d4_nofuse_train", 16, 4, 4, 10, 12, 3, True, False, True, 2.0, 208)
    make_stage("stage_c8nogroup_nofuse_eval", 8, 8, 4, 10, 12, 3, False, False, False, 2.0, 209)


# ------------------------------------------------------------------ (iv) ranges
def make_ranges():
    g = torch.Generator().manual_seed(301)
    dv = torch.tensor([[425.0, 935.0], [300.0, 1200.0]])
    for D in (4, 8, 16):
        save("range_init_d%d" % D, depth_values=dv,
             inverse=init_inverse_range(dv, D, dv.device, dv.dtype, 5, 6),
             linear=init_range(dv, D, dv.device, dv.dtype, 5, 6))
    dv192 = torch.linspace(425.0, 935.0, 192).unsqueeze(0)
    save("range_init_192", depth_values=dv192, inverse=init_inverse_range(dv192, 8, dv192.device, dv192.dtype, 3, 4),
         linear=init_range(dv192, 8, dv192.device, dv192.dtype, 3, 4))
    for (D, Hp, Wp) in ((8, 9, 11), (4, 16, 20), (16, 5, 7)):
        base = 1.0 / (500 + 300 * torch.rand(2, Hp, Wp, generator=g))
        half = 2e-5 + 1e-4 * torch.rand(2, Hp, Wp, generator=g)
        imin, imax = base + half, base - half
        save("range_sched_d%d" % D, inv_min=imin, inv_max=imax,
             out=schedule_inverse_range(imin, imax, D, 2 * Hp, 2 * Wp))


# ------------------------------------------------------------------ (v) ties
def make_ties():
    g = torch.Generator().manual_seed(401)
    B, D, H, W = 1, 8, 4, 8
    logits = torch.randn(B, D, H, W, generator=g)
    logits[0, :, 0, 0] = 0.25                      # all equal -> index 0
    logits[0, :, 0, 1] = -1.0
    logits[0, 3, 0, 1] = 2.0
    logits[0, 6, 0, 1] = 2.0                       # two-way tie -> first (3)
    logits[0, :, 0, 2] = torch.tensor([0.1, 0.7, 0.7, 0.7, -3, 0.2, 0.7, 0.0])
    logits[0, :, 0, 3] = 50.0 + torch.arange(D) * 1e-30   # ties after rounding
    logits[0, 7, 0, 4] = 9.0                       # max at the last index
    logits[0, 0, 0, 5] = 9.0                       # max at the first index
    logits[0, :, 1, 0] = -40.0                     # large negative common offset
    logits[0, 2, 1, 0] = -39.5
    hypo = 1.0 / (1.0 / 900 + torch.arange(D).view(1, D, 1, 1) * 1.5e-4 + 1e-5 * torch.rand(B, 1, H, W, generator=g))

    feats = [torch.zeros(B, 8, H, W), torch.zeros(B, 8, H, W)]
    mats = pinhole_views(B, 2, H, W, g)
    for training in (True, False):
        net = stagenet(inverse_depth=True, mono=False, attn_fuse_d=True, attn_temp=2)
        net.train(training)
        out = net(feats, mats, hypo, lambda vol: logits, 0, group_cor=True, group_cor_dim=4, split_itv=0.5)
        idx = out["attn_weight"].max(1)[1]
        save("ties_%s" % ("train" if training else "eval"), logits=logits, hypo=hypo,
             attn_weight=out["attn_weight"], depth=out["depth"], index=idx,
             conf=out["photometric_confidence"], inv_min=out["inverse_min_depth"], inv_max=out["inverse_max_depth"])


# ------------------------------------------------------------------ sinkhorn / MVS4net_loss
def make_sinkhorn():
    from models.mvs4net_utils import sinkhorn
    from models.MVS4Net import MVS4net_loss
    g = torch.Generator().manual_seed(501)
    cases = [("d4_it10", 4, 10, 1.0, False), ("d8_it10", 8, 10, 1.0, False), ("d16_it3", 16, 3, 1.0, False),
             ("d8_eps01", 8, 5, 0.1, False), ("d4_cont", 4, 10, 1.0, True), ("d8_cont", 8, 3, 1.0, True),
             ("d5_it4", 5, 4, 1.0, False)]
    for name, D, iters, eps, cont in cases:
        B, H, W = 2, 9, 11
        base = 500 + 300 * torch.rand(B, 1, H, W, generator=g)
        inv = 1.0 / base + (torch.arange(D).view(1, D, 1, 1) - (D - 1) / 2) * 5e-5
        hypo = 1.0 / inv                                                     # decreasing depth, like the cascade
        gt = 1.0 / (1.0 / base[:, 0] + (torch.rand(B, H, W, generator=g) - 0.5) * 5e-5 * (D + 1))
        mask = torch.rand(B, H, W, generator=g) < 0.7
        logits = 2 * torch.randn(B, D, H, W, generator=g)
        attn = torch.softmax(logits, 1).requires_grad_(True)
        T, loss = sinkhorn(gt, hypo, attn, mask, iters, eps=eps, continuous=cont)
        loss.backward()
        save("sinkhorn_" + name, gt=gt, hypo=hypo, attn=attn, mask=mask, T=T, loss=loss, gattn=attn.grad,
             meta=np.array([D, iters, int(cont)]), eps=np.float32(eps))
    # MVS4net_loss over a 4-stage output dictionary (inverse depth, mono L1 on stages 2-4)
    outs, gts, masks, arrays = {}, {}, {}, {}
    for k, (D, H, W) in enumerate(((8, 6, 8), (8, 12, 16), (4, 24, 32), (4, 48, 64))):
        key = "stage%d" % (k + 1)
        base = 500 + 300 * torch.rand(1, 1, H, W, generator=g)
        hypo = 1.0 / (1.0 / base + (torch.arange(D).view(1, D, 1, 1) - (D - 1) / 2) * 4e-5)
        attn = torch.softmax(2 * torch.randn(1, D, H, W, generator=g), 1).requires_grad_(True)
        gts[key] = 1.0 / (1.0 / base[:, 0] + (torch.rand(1, H, W, generator=g) - 0.5) * 3e-4)
        masks[key] = (torch.rand(1, H, W, generator=g) < 0.7).float()
        outs[key] = {"depth": torch.gather(hypo, 1, attn.max(1, keepdim=True)[1]).squeeze(1), "hypo_depth": hypo,
                     "attn_weight": attn, "mono_depth": (gts[key] + torch.randn(1, H, W, generator=g)).requires_grad_(True)}
        arrays.update({key + "_hypo": hypo, key + "_attn": attn, key + "_gt": gts[key], key + "_mask": masks[key],
                       key + "_mono": outs[key]["mono_depth"], key + "_depth": outs[key]["depth"]})
    total, l1s, ots, miss = MVS4net_loss(outs, gts, masks, stage_lw=[1, 1, 1, 1], l1ot_lw=[0.5, 1], inverse_depth=True,
                                         ot_iter=10, ot_eps=1, ot_continous=False, mono=True)
    total.backward()
    arrays.update(total=total, l1=torch.stack(l1s), ot=torch.stack(ots), miss=torch.stack(miss))
    for key in outs:
        arrays[key + "_gattn"] = outs[key]["attn_weight"].grad
    save("loss_mvs4net", **arrays)


# ------------------------------------------------------------------ (vi) end-to-end MVS4net
def model_inputs(B, N, H, W, gen):
    """imgs list, proj_matrices dict stage1..4 (intrinsics scaled 1/8,1/4,1/2,1 like datasets/dataloader_eval.py:281-287),
    depth_values [B,2]."""
    imgs = [torch.rand(B, 3, H, W, generator=gen) for _ in range(N)]
    full = pinhole_views(B, N, H, W, gen, baseline=25.0)
    proj = {}
    for k, sc in enumerate((0.125, 0.25, 0.5, 1.0)):
        m = full.clone()
        m[:, :, 1, :2, :] = full[:, :, 1, :2, :] * sc
        proj["stage%d" % (k + 1)] = m
    dv = torch.tensor([[425.0, 935.0]] * B)
    return imgs, proj, dv


def state_checksums(model):
    sd = model.state_dict()
    keys = list(sd.keys())
    sums = torch.tensor([float(v.double().sum()) for v in sd.values()], dtype=torch.float64)
    abss = torch.tensor([float(v.double().abs().sum()) for v in sd.values()], dtype=torch.float64)
    return keys, sums, abss


def make_model():
    from models.MVS4Net import MVS4net
    g = torch.Generator().manual_seed(601)
    B, N, H, W = 1, 3, 64, 64
    imgs, proj, dv = model_inputs(B, N, H, W, g)
    kw = dict(arch_mode="fpn", reg_net="reg2d", num_stage=4, fpn_base_channel=8, reg_channel=8,
              stage_splits=[8, 8, 4, 4], depth_interals_ratio=[0.5, 0.5, 0.5, 1], group_cor=True,
              group_cor_dim=[8, 8, 4, 4], inverse_depth=True, agg_type="ConvBnReLU3D", dcn=False, pos_enc=0,
              mono=True, asff=False, attn_temp=2, attn_fuse_d=True)
    torch.manual_seed(7)
    model = MVS4net(**kw)
    keys, sums, abss = state_checksums(model)
    arrays = dict(depth_values=dv, state_sum=sums, state_abs=abss,
                  state_keys=np.array("\n".join(keys)), n_params=np.int64(sum(p.numel() for p in model.parameters())))
    for i, im in enumerate(imgs):
        arrays["img%d" % i] = im
    for k, v in proj.items():
        arrays["proj_" + k] = v
    # An untrained network in eval mode is degenerate (BatchNorm running stats are still 0/1 and the activations
    # decay to a uniform attention).  Warm the running statistics with 25 train-mode passes and ship them (12 KB):
    # the test loads them into its own seeded model.
    model.train()
    with torch.no_grad():
        for _ in range(25):
            model(imgs, proj, dv)
    bufs = {k: v for k, v in model.state_dict().items() if "running_" in k}
    arrays["bn_keys"] = np.array("\n".join(bufs.keys()))
    arrays["bn_values"] = torch.cat([v.reshape(-1).float() for v in bufs.values()])
    arrays["bn_sizes"] = np.array([v.numel() for v in bufs.values()], dtype=np.int64)
    model.eval()
    with torch.no_grad():
        out = model(imgs, proj, dv)
    for k in ("stage1", "stage2", "stage3", "stage4"):
        arrays["eval_%s_depth" % k] = out[k]["depth"]
        arrays["eval_%s_conf" % k] = out[k]["photometric_confidence"]
        arrays["eval_%s_attn" % k] = out[k]["attn_weight"]
        arrays["eval_%s_hypo" % k] = out[k]["hypo_depth"]
    model.train()
    out = model(imgs, proj, dv)
    lw = {k: torch.randn(out[k]["attn_weight"].shape, generator=g) for k in ("stage1", "stage2", "stage3", "stage4")}
    loss = sum((out[k]["attn_weight"] * lw[k]).sum() for k in lw) + sum(out[k]["mono_depth"].mean() * 1e-3 for k in ("stage2", "stage3", "stage4"))
    loss.backward()
    for k in lw:
        arrays["train_%s_depth" % k] = out[k]["depth"]
        arrays["train_%s_attn" % k] = out[k]["attn_weight"]
        arrays["loss_w_%s" % k] = lw[k]
        if "mono_depth" in out[k]:
            arrays["train_%s_mono" % k] = out[k]["mono_depth"]
    arrays["train_loss"] = loss.detach()
    gkeys = [n for n, p in model.named_parameters() if p.grad is not None]
    arrays["grad_keys"] = np.array("\n".join(gkeys))
    arrays["grad_abs"] = torch.tensor([float(dict(model.named_parameters())[n].grad.double().abs().sum()) for n in gkeys],
                                      dtype=torch.float64)
    save("model_e2e_64", **arrays)


def make_pfm():
    """PFM files as the reference's datasets/data_io.py writes them (bytes) and reads them back (arrays).  The module's
    only other top-level import is cv2 (stubbed above); read_pfm / save_pfm use numpy alone."""
    import tempfile
    from datasets.data_io import read_pfm, save_pfm
    g = np.random.default_rng(7)
    cases = {"grey": (g.standard_normal((5, 7)) * 300 + 600).astype(np.float32),
             "grey1": g.standard_normal((3, 4, 1)).astype(np.float32),
             "colour": g.standard_normal((4, 3, 3)).astype(np.float32),
             "bigendian": (g.standard_normal((2, 6)) * 10).astype(">f4")}
    arrays = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, img in cases.items():
            path = os.path.join(tmp, name + ".pfm")
            save_pfm(path, img, scale=1 if name != "colour" else 2.5)
            back, scale = read_pfm(path)
            arrays[name + "_image"] = img.astype(np.float32) if name == "bigendian" else img
            arrays[name + "_bytes"] = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
            arrays[name + "_read"] = np.ascontiguousarray(back).astype(np.float32)
            arrays[name + "_scale"] = np.float64(scale)
    np.savez_compressed(os.path.join(HERE, "pfm_io.npz"), **arrays)
    print("pfm_io.npz", {k: v.shape for k, v in arrays.items()})


if __name__ == "__main__":
    if "--pfm-only" in sys.argv:
        make_pfm()
        sys.exit(0)
    if "--model-only" in sys.argv:
        make_model()
        sys.exit(0)
    if "--loss-only" in sys.argv:
        make_sinkhorn()
        sys.exit(0)
    make_warp()
    make_stages()
    make_ranges()
    make_ties()
    make_sinkhorn()
    make_model()
    make_pfm()
