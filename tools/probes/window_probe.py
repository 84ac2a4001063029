#!/usr/bin/env python3
"""Probe: how compact are the landing positions of one tile of reference pixels in a source view?  For the bench's stage
inputs (noise / scene features, hypotheses from the real cascade) prints, per stage, tile shape and window size, the
fraction of (pixel, hypothesis, view) samples whose 2x2 footprint lies inside a fixed window of source texels anchored at
the tile's median landing position -- what an LDS-staged source window would serve without touching the gather path.
Run on the GPU box:  python tools/probes/window_probe.py [--data scene]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                                             # noqa: E402
import deep_reconstruction_with_epipolar_lines_mvster_amd as mv          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default="noise")
    ap.add_argument("--batch", type=int, default=2)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.batch
    stages, dv = bench.make_inputs(B, 1234, dev, data=args.data)
    net = mv.stagenet(inverse_depth=True, mono=True, attn_temp=2.0).train()
    prev = None
    for k, st in enumerate(stages):
        H, W, D = st["H"], st["W"], st["D"]
        hypo = (mv.init_inverse_range(dv, D, dev, torch.float32, H, W) if k == 0 else
                mv.schedule_inverse_range(prev["inverse_min_depth"], prev["inverse_max_depth"], D, H, W))
        with torch.no_grad():
            prev = net(st["feats"], st["mats"], hypo, bench.stub_regnet, k, group_cor=True, group_cor_dim=st["G"], split_itv=st["itv"])
        pose = mv.stage_poses(st["mats"])                                # [B,S,12]
        S = pose.shape[1]
        ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
        for (tw, th) in ((16, 16), (32, 8), (32, 16), (16, 8), (8, 8)):
            if H % th or W % tw:
                Hc, Wc = H - H % th, W - W % tw
            else:
                Hc, Wc = H, W
            line = "stage %d C%-2d D%d tile %2dx%-2d:" % (k + 1, st["C"], D, tw, th)
            res = {}
            for s in range(S):
                R = pose[:, s, :9].view(B, 3, 3)
                t = pose[:, s, 9:].view(B, 3)
                rx = R[:, 0, 0, None, None] * xs + R[:, 0, 1, None, None] * ys + R[:, 0, 2, None, None]
                ry = R[:, 1, 0, None, None] * xs + R[:, 1, 1, None, None] * ys + R[:, 1, 2, None, None]
                rz = R[:, 2, 0, None, None] * xs + R[:, 2, 1, None, None] * ys + R[:, 2, 2, None, None]
                px = rx[:, None] * hypo + t[:, 0, None, None, None]
                py = ry[:, None] * hypo + t[:, 1, None, None, None]
                pz = rz[:, None] * hypo + t[:, 2, None, None, None]
                ix = torch.floor(px / pz)[:, :, :Hc, :Wc]                # [B,D,H,W]
                iy = torch.floor(py / pz)[:, :, :Hc, :Wc]
                # tiles: [B, D, Hc/th, th, Wc/tw, tw] -> [B, ty, tx, D*th*tw]
                def tiles(v):
                    return v.view(B, D, Hc // th, th, Wc // tw, tw).permute(0, 2, 4, 1, 3, 5).reshape(B, Hc // th, Wc // tw, -1)
                tx_, ty_ = tiles(ix), tiles(iy)
                mx, my = tx_.median(dim=-1, keepdim=True).values, ty_.median(dim=-1, keepdim=True).values
                for (ww, wh) in ((32, 24), (48, 24), (64, 24), (96, 24), (64, 32), (128, 32)):
                    inside = ((tx_ >= mx - ww // 2) & (tx_ + 1 < mx + ww // 2) & (ty_ >= my - wh // 2) & (ty_ + 1 < my + wh // 2))
                    res.setdefault((ww, wh), []).append(float(inside.float().mean()))
            for (ww, wh), v in res.items():
                line += "  %dx%d %.3f" % (ww, wh, sum(v) / len(v))
            print(line, flush=True)
            print("      window KB at this C: " + "  ".join("%dx%d %.0f" % (ww, wh, ww * wh * st["C"] * 4 / 1024) for (ww, wh) in res), flush=True)


if __name__ == "__main__":
    main()
