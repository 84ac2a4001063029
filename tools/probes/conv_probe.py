#!/usr/bin/env python3
"""Probe: how fast does MIOpen run each convolution layer shape of FPN4 / reg2d at DTU resolution (1600x1152, fp32,
NCHW)?  Prints per layer: forward and backward (dgrad + wgrad) time and the achieved TFLOP/s, to see where a
hand-written kernel has room.  python tools/probes/conv_probe.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deep_reconstruction_with_epipolar_lines_mvster_amd import mvs4net      # noqa: E402

dev = torch.device("cuda:0")
H, W = 1152, 1600
layers = [  # name, N, Cin, Cout, k, stride, h, w (input size), transposed
    ("fpn conv0.0 3->8", 1, 3, 8, 3, 1, H, W, False), ("fpn conv0.1 8->8", 1, 8, 8, 3, 1, H, W, False),
    ("fpn conv1.0 8->16 5x5s2", 1, 8, 16, 5, 2, H, W, False), ("fpn conv1.x 16->16", 1, 16, 16, 3, 1, H // 2, W // 2, False),
    ("fpn conv2.0 16->32 5x5s2", 1, 16, 32, 5, 2, H // 2, W // 2, False), ("fpn conv2.x 32->32", 1, 32, 32, 3, 1, H // 4, W // 4, False),
    ("fpn conv3.0 32->64 5x5s2", 1, 32, 64, 5, 2, H // 4, W // 4, False), ("fpn conv3.x 64->64", 1, 64, 64, 3, 1, H // 8, W // 8, False),
    ("fpn out2 64->32", 1, 64, 32, 3, 1, H // 4, W // 4, False), ("fpn out3 64->16", 1, 64, 16, 3, 1, H // 2, W // 2, False),
    ("fpn out4 64->8", 1, 64, 8, 3, 1, H, W, False),
    ("reg s4 conv0 4->8", 4, 4, 8, 3, 1, H, W, False), ("reg s4 conv1 8->16 s2", 4, 8, 16, 3, 2, H, W, False),
    ("reg s4 conv2 48->16", 4, 48, 16, 3, 1, H // 2, W // 2, False), ("reg s4 conv3 16->32 s2", 4, 16, 32, 3, 2, H // 2, W // 2, False),
    ("reg s4 conv4 96->32", 4, 96, 32, 3, 1, H // 4, W // 4, False), ("reg s4 conv5 32->64 s2", 4, 32, 64, 3, 2, H // 4, W // 4, False),
    ("reg s4 conv6 192->64", 4, 192, 64, 3, 1, H // 8, W // 8, False),
    ("reg s4 up7 64->32 T", 4, 64, 32, 3, 2, H // 8, W // 8, True), ("reg s4 up9 32->16 T", 4, 32, 16, 3, 2, H // 4, W // 4, True),
    ("reg s4 up11 16->8 T", 4, 16, 8, 3, 2, H // 2, W // 2, True), ("reg s4 prob 8->1 1x1", 4, 8, 1, 1, 1, H, W, False),
]
tot_f = tot_b = 0.0
for name, N, ci, co, k, s, h, w, tr in layers:
    x = torch.randn(N, ci, h, w, device=dev, requires_grad=True)
    wt = torch.randn((ci, co, k, k) if tr else (co, ci, k, k), device=dev, requires_grad=True)
    op = (lambda: F.conv_transpose2d(x, wt, None, stride=s, padding=k // 2, output_padding=1)) if tr else \
        (lambda: F.conv2d(x, wt, None, stride=s, padding=k // 2))
    y = op()
    gy = torch.randn_like(y)
    y.backward(gy)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    reps = 5
    for _ in range(reps):
        x.grad = wt.grad = None
        ev[0].record()
        y = op()
        ev[1].record()
        y.backward(gy)
        ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1]) / reps
        tb += ev[1].elapsed_time(ev[2]) / reps
    out_px = y.shape[0] * y.shape[2] * y.shape[3]
    flop = 2.0 * out_px * co * ci * k * k if not tr else 2.0 * N * h * w * co * ci * k * k
    mb = (x.numel() + y.numel()) * 4 / 1e6
    line = "%-28s fwd %7.3f ms %6.1f TF/s | bwd %7.3f ms %6.1f TF/s | act %7.1f MB (%.3f ms at 5 TB/s)" % (
        name, tf, flop / tf / 1e9, tb, 2 * flop / tb / 1e9, mb, mb / 5e3)
    if not tr and s == 1 and k in (1, 3) and co in (1, 3, 4, 8, 16):
        # the direct HIP kernels (csrc/conv_direct.hip) on the same layer
        dop = lambda: mvs4net._DirectConv2d.apply(x, wt, None)
        y = dop()
        y.backward(gy)
        torch.cuda.synchronize()
        df = db = 0.0
        for _ in range(reps):
            x.grad = wt.grad = None
            ev[0].record()
            y = dop()
            ev[1].record()
            y.backward(gy)
            ev[2].record()
            torch.cuda.synchronize()
            df += ev[0].elapsed_time(ev[1]) / reps
            db += ev[1].elapsed_time(ev[2]) / reps
        line += " || direct fwd %7.3f ms bwd %7.3f ms" % (df, db)
    print(line, flush=True)
    tot_f += tf
    tot_b += tb
print("sum fwd %.2f ms, bwd %.2f ms (one call each)" % (tot_f, tot_b))
