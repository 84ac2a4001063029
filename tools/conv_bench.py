#!/usr/bin/env python3
"""Convolution micro-benchmark: every FPN4 / reg2d layer shape at DTU resolution (1600x1152; FPN4 over the five views in
one batch, reg2d stage 4 over its four depth slices), its three operations on the implementation mvs4net._conv_route picks,
HIP events on the launch stream.  Also the command the matrix-core counters are collected on:
  python tools/conv_bench.py [--reps 5]
  tools/pmc_pass.sh conv_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INSTS_VALU -- tools/conv_bench.py --reps 2"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_reconstruction_with_epipolar_lines_mvster_amd import mvs4net      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=5)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--only", default="", help="substring of the layer names to run")
args = ap.parse_args()
dev = torch.device("cuda:0")
H, W, V = 1152, 1600, args.views
LAYERS = [  # name, N, Cin, Cout, k, stride, h, w (input size); stride 0 = transposed convolution x2
    ("fpn conv0.0 3->8", V, 3, 8, 3, 1, H, W), ("fpn conv0.1 8->8", V, 8, 8, 3, 1, H, W),
    ("fpn conv1.0 8->16 5x5 s2", V, 8, 16, 5, 2, H, W), ("fpn conv1.x 16->16", V, 16, 16, 3, 1, H // 2, W // 2),
    ("fpn conv2.0 16->32 5x5 s2", V, 16, 32, 5, 2, H // 2, W // 2), ("fpn conv2.x 32->32", V, 32, 32, 3, 1, H // 4, W // 4),
    ("fpn conv3.0 32->64 5x5 s2", V, 32, 64, 5, 2, H // 4, W // 4), ("fpn conv3.x 64->64", V, 64, 64, 3, 1, H // 8, W // 8),
    ("fpn out1 64->64 1x1", V, 64, 64, 1, 1, H // 8, W // 8), ("fpn out2 64->32", V, 64, 32, 3, 1, H // 4, W // 4),
    ("fpn out3 64->16", V, 64, 16, 3, 1, H // 2, W // 2), ("fpn out4 64->8 (unfused head)", V, 64, 8, 3, 1, H, W),
    ("fpn head 64->72 1x1, low res", V, 64, 72, 1, 1, H // 2, W // 2),
    ("reg s4 conv0 4->8", 4, 4, 8, 3, 1, H, W), ("reg s4 conv1 8->16 s2", 4, 8, 16, 3, 2, H, W),
    ("reg s4 conv2 48->16", 4, 48, 16, 3, 1, H // 2, W // 2), ("reg s4 conv3 16->32 s2", 4, 16, 32, 3, 2, H // 2, W // 2),
    ("reg s4 conv4 96->32", 4, 96, 32, 3, 1, H // 4, W // 4), ("reg s4 conv5 32->64 s2", 4, 32, 64, 3, 2, H // 4, W // 4),
    ("reg s4 conv6 192->64", 4, 192, 64, 3, 1, H // 8, W // 8), ("reg s4 up7 64->32 T", 4, 64, 32, 3, 0, H // 8, W // 8),
    ("reg s4 up9 32->16 T", 4, 32, 16, 3, 0, H // 4, W // 4), ("reg s4 up11 16->8 T", 4, 16, 8, 3, 0, H // 2, W // 2),
    ("reg s4 prob 8->1 1x1", 4, 8, 1, 1, 1, H, W),
]


def timed(fn):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / args.reps


tot = [0.0, 0.0, 0.0]
print("%-28s %7s | %-22s | %-22s | %-22s" % ("layer", "GFLOP", "forward", "input gradient", "weight gradient"))
for name, N, ci, co, k, s, h, w in LAYERS:
    if args.only and not any(o in name for o in args.only.split(",")):
        continue
    x = torch.randn(N, ci, h, w, device=dev)
    if s == 0:
        wt = torch.randn(ci, co, k, k, device=dev)
        route = ("mfma", "mfma", "mfma") if mvs4net._MFMA_CONV else ("miopen",) * 3
        op = lambda xx, ww: mvs4net._conv_transpose2d_x2(xx, ww)
    else:
        wt = torch.randn(co, ci, k, k, device=dev)
        route = mvs4net._conv_route(ci, co, k, s) or ("miopen",) * 3
        op = lambda xx, ww: mvs4net._conv2d(xx, ww, None, s, k // 2)
    with torch.no_grad():
        tf = timed(lambda: op(x, wt))
        y = op(x, wt)
    flop = 2.0 * (x.numel() if s == 0 else y.numel() // co * ci) * co * k * k
    gy = torch.randn_like(y)
    cells = ["%-6s %6.3f ms %5.1f TF" % (route[0], tf, flop / tf / 1e9)]
    tot[0] += tf
    for i, which in ((1, "x"), (2, "w")):
        xx = x.clone().requires_grad_(which == "x")
        ww = wt.clone().requires_grad_(which == "w")
        yy = op(xx, ww)
        t = timed(lambda: torch.autograd.grad(yy, xx if which == "x" else ww, gy, retain_graph=True))
        cells.append("%-6s %6.3f ms %5.1f TF" % (route[i], t, flop / t / 1e9))
        tot[i] += t
    print("%-28s %7.1f | %s" % (name, flop / 1e9, " | ".join(cells)), flush=True)
print("sum (one call each): forward %.2f ms, input gradient %.2f ms, weight gradient %.2f ms" % tuple(tot))
