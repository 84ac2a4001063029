#!/usr/bin/env python3
"""Kernel micro-benchmark: times the fused cost-volume ABI calls per DTU stage on bench-like inputs (noise features,
hypotheses from the real cascade), HIP events on the launch stream.  python tools/kbench.py [--batch 4] [--reps 10]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                             # noqa: E402
import deep_reconstruction_with_epipolar_lines_mvster_amd as mv          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--layout", default="nhwc")
    ap.add_argument("--data", default="noise", choices=["noise", "scene"])
    ap.add_argument("--workspace", default="full", choices=["full", "binned", "coefficients", "none"],
                    help="scratch offered to the backward = its scatter route (hotpath.BWD_WORKSPACE)")
    ap.add_argument("--count-in-backward", action="store_true", help="stand-alone count / fill kernels instead of the fused ones")
    args = ap.parse_args()
    mv.hotpath.BWD_WORKSPACE = args.workspace
    mv.hotpath.COUNT_IN_FORWARD = not args.count_in_backward
    dev = torch.device("cuda:0")
    stages, dv = bench.make_inputs(args.batch, 1234, dev, data=args.data)
    net = mv.stagenet(inverse_depth=True, mono=True, attn_temp=2.0).train()
    prev = None
    for k, st in enumerate(stages):
        hypo = (mv.init_inverse_range(dv, st["D"], dev, torch.float32, st["H"], st["W"]) if k == 0 else
                mv.schedule_inverse_range(prev["inverse_min_depth"], prev["inverse_max_depth"], st["D"], st["H"], st["W"]))
        with torch.no_grad():
            prev = net(st["feats"], st["mats"], hypo, bench.stub_regnet, k, group_cor=True, group_cor_dim=st["G"], split_itv=st["itv"])
        feats = [mv.to_channels_last(f.detach()) if args.layout == "nhwc" else f.detach() for f in st["feats"]]
        feats = [f.requires_grad_(True) for f in feats]
        pose = mv.stage_poses(st["mats"])
        gvol = torch.randn(args.batch, st["G"], st["D"], st["H"], st["W"], device=dev)

        def fwd():
            return mv.fused_cost_volume(feats, pose, hypo, st["G"], 2.0, layout=args.layout)

        vol = fwd()
        vol.backward(gvol)
        torch.cuda.synchronize()
        tf = tb = 0.0
        for _ in range(args.reps):
            for f in feats:
                f.grad = None
            a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            a.record()
            vol = fwd()
            b.record()
            vol.backward(gvol)
            c.record()
            torch.cuda.synchronize()
            tf += a.elapsed_time(b)
            tb += b.elapsed_time(c)
        px = st["H"] * st["W"] * args.batch
        fb = bench.algorithmic_bytes("mvster_cost_volume_fwd", 1, st["C"], st["G"], st["D"], px, 4)
        bb = bench.algorithmic_bytes("mvster_cost_volume_bwd", 1, st["C"], st["G"], st["D"], px, 4)
        if k > 0:
            d = prev["depth"]
            rough = float((d[:, :, 1:] - d[:, :, :-1]).abs().mean())
        else:
            rough = 0.0
        print("stage %d  C%-2d D%d  mean|ddepth/dx| %6.2f mm  fwd %7.3f ms (%6.0f GB/s)   bwd %7.3f ms (%6.0f GB/s)" % (
            k + 1, st["C"], st["D"], rough, tf / args.reps, fb / (tf / args.reps) / 1e6, tb / args.reps, bb / (tb / args.reps) / 1e6), flush=True)


if __name__ == "__main__":
    main()
