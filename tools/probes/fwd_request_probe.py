#!/usr/bin/env python3
"""Probe: is the stage-4 forward bound by the number of L1->L2 requests?  Times mvster_cost_volume_fwd at the DTU stage-4
shape (B=4, C8 G4 D4 S4) on hypotheses that differ only in how many distinct source lines one wave-instruction touches:
  noise4   per-pixel random depth, 4 distinct hypotheses per pixel (bench-like)
  noise1   same random depth, the 4 hypotheses of a pixel identical (their lanes request the same addresses)
  smooth4  smooth depth, 4 distinct hypotheses        smooth1  smooth depth, identical hypotheses
Run on the GPU box:  python tools/probes/fwd_request_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                                             # noqa: E402
import deep_reconstruction_with_epipolar_lines_mvster_amd as mv          # noqa: E402

dev = torch.device("cuda:0")
H, W, C, G, D, scale, itv = bench.DTU_STAGES[3]
B, S = 4, 4
g = torch.Generator().manual_seed(0)
feats = [mv.to_channels_last(torch.randn(B, C, H, W, generator=g).to(dev)) for _ in range(S + 1)]
pose = mv.stage_poses(bench.synth_views(B, S + 1, H, W).to(dev))
step = (1 / 425.0 - 1 / 935.0) / 7 / 7 / 3 * 2 / 3          # stage-4 hypothesis spacing of the DTU cascade (inverse depth)
yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
smooth = (1.0 / (600 + 150 * torch.sin(6 * xx) * torch.cos(4 * yy)))[None, None].expand(B, 1, H, W)
noise = 1 / 935.0 + (1 / 425.0 - 1 / 935.0) * (0.45 + 0.1 * torch.rand(B, 1, H, W, generator=g))      # +-28 px at view 4
ramp = (torch.arange(D).view(1, D, 1, 1) - (D - 1) / 2) * step
cases = {"noise4": noise + ramp, "noise1": noise + 0 * ramp, "smooth4": smooth + ramp, "smooth1": smooth + 0 * ramp}
for name, inv in cases.items():
    hypo = (1.0 / inv).contiguous().to(dev)
    for _ in range(2):
        mv.fused_cost_volume(feats, pose, hypo, G, 2.0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        mv.fused_cost_volume(feats, pose, hypo, G, 2.0)
    b.record()
    torch.cuda.synchronize()
    print("%-8s forward %.3f ms" % (name, a.elapsed_time(b) / 10), flush=True)
