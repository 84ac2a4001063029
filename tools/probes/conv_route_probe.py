#!/usr/bin/env python3
"""Probe: every convolution layer shape of FPN4 (all views in one batch) / reg2d (stage 4) at DTU resolution, each of its
three operations (forward, input gradient, weight gradient) timed per implementation -- direct (csrc/conv_direct.hip),
mfma (csrc/conv_mfma.hip), miopen -- to set mvs4net._conv_route.  python tools/probes/conv_route_probe.py [--views 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deep_reconstruction_with_epipolar_lines_mvster_amd import mvs4net      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=5)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
H, W, V = 1152, 1600, args.views
layers = [  # name, N, Cin, Cout, k, stride, h, w (input size)
    ("fpn conv0.0 3->8", V, 3, 8, 3, 1, H, W), ("fpn conv0.1 8->8", V, 8, 8, 3, 1, H, W), ("reg s4 conv0 4->8", 4, 4, 8, 3, 1, H, W), ("fpn conv1.0 8->16 5x5s2", V, 8, 16, 5, 2, H, W),
    ("fpn conv1.x 16->16", V, 16, 16, 3, 1, H // 2, W // 2), ("fpn conv2.0 16->32 5x5s2", V, 16, 32, 5, 2, H // 2, W // 2),
    ("fpn conv2.x 32->32", V, 32, 32, 3, 1, H // 4, W // 4), ("fpn conv3.0 32->64 5x5s2", V, 32, 64, 5, 2, H // 4, W // 4),
    ("fpn conv3.x 64->64", V, 64, 64, 3, 1, H // 8, W // 8), ("fpn out1 64->64 1x1", V, 64, 64, 1, 1, H // 8, W // 8),
    ("fpn out2 64->32", V, 64, 32, 3, 1, H // 4, W // 4), ("fpn out3 64->16", V, 64, 16, 3, 1, H // 2, W // 2),
    ("fpn out4 64->8", V, 64, 8, 3, 1, H, W),
    ("reg s4 conv1 8->16 s2", 4, 8, 16, 3, 2, H, W), ("reg s4 conv2 48->16", 4, 48, 16, 3, 1, H // 2, W // 2),
    ("reg s4 conv3 16->32 s2", 4, 16, 32, 3, 2, H // 2, W // 2), ("reg s4 conv4 96->32", 4, 96, 32, 3, 1, H // 4, W // 4),
    ("reg s4 conv5 32->64 s2", 4, 32, 64, 3, 2, H // 4, W // 4), ("reg s4 conv6 192->64", 4, 192, 64, 3, 1, H // 8, W // 8),
    ("reg s2 conv2 48->16 (8 sl)", 8, 48, 16, 3, 1, H // 8, W // 8), ("reg s2 conv4 96->32 (8 sl)", 8, 96, 32, 3, 1, H // 16, W // 16),
]
lib = mvs4net._lib.load()


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


for name, N, ci, co, k, s, h, w in layers:
    x = torch.randn(N, ci, h, w, device=dev)
    wt = torch.randn(co, ci, k, k, device=dev)
    impls = ["miopen"]
    if lib.mvster_conv_mfma_supported(ci, co, k, s):
        impls.append("mfma")
    if s == 1 and k in (1, 3) and co in (1, 3, 4, 8, 16):
        impls.append("direct")
    flop = None
    cells = []
    for op in ("fwd", "dgrad", "wgrad"):
        row = []
        for impl in impls:
            if op == "dgrad" and ((impl == "mfma" and s != 1) or (impl == "direct" and not lib.mvster_conv_direct_supported(co, ci, k))):
                row.append("%s -" % impl)
                continue
            if op == "wgrad" and impl == "direct" and co not in (1, 8, 16):
                row.append("%s -" % impl)
                continue
            route = {"fwd": (impl, "miopen", "miopen"), "dgrad": ("miopen", impl, "miopen"), "wgrad": ("miopen", "miopen", impl)}[op]
            try:
                if op == "fwd":
                    with torch.no_grad():
                        t = timed(lambda: mvs4net._DirectConv2d.apply(x, wt, None, s, route))
                    y = mvs4net._DirectConv2d.apply(x, wt, None, s, route)
                    flop = 2.0 * y.numel() * ci * k * k
                else:
                    xx = x.clone().requires_grad_(op == "dgrad")
                    ww = wt.clone().requires_grad_(op == "wgrad")
                    y = mvs4net._DirectConv2d.apply(xx, ww, None, s, route)
                    gy = torch.randn_like(y)
                    t = timed(lambda: torch.autograd.grad(y, xx if op == "dgrad" else ww, gy, retain_graph=True))
                row.append("%s %.3f" % (impl, t))
            except Exception as exc:                       # noqa: BLE001 -- a probe: report and go on
                row.append("%s ERR(%s)" % (impl, str(exc)[:40]))
        cells.append("%s: %s" % (op, "  ".join(row)))
    print("%-28s %5.1f GF | %s" % (name, flop / 1e9, " | ".join(cells)), flush=True)
