#!/usr/bin/env python3
"""Probe: every matrix-core convolution launch of one e2e training step with its shape, HIP-event time, workgroup count
and how the workgroups quantise onto the chip (256 CUs x workgroups per CU): where the forward / transposed kernels lose
time to partially filled last rounds.   python tools/probes/conv_calls.py [--eval]"""
import os
import sys
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                                             # noqa: E402
import deep_reconstruction_with_epipolar_lines_mvster_amd as mv          # noqa: E402
from deep_reconstruction_with_epipolar_lines_mvster_amd import _lib     # noqa: E402

evalm = "--eval" in sys.argv
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, N, H, W = 1, 5, 1152, 1600
model = mv.MVS4net(arch_mode="fpn", reg_net="reg2d", num_stage=4, fpn_base_channel=8, reg_channel=8, stage_splits=[8, 8, 4, 4],
                   depth_interals_ratio=[0.5, 0.5, 0.5, 1], group_cor=True, group_cor_dim=[8, 8, 4, 4], inverse_depth=True,
                   mono=True, attn_temp=2).to(dev)
imgs = [torch.rand(B, 3, H, W, device=dev) for _ in range(N)]
proj = {"stage%d" % (k + 1): bench.synth_views(B, N, H // sc, W // sc).to(dev) for k, sc in enumerate((8, 4, 2, 1))}
dv = torch.tensor([[425.0, 935.0]] * B, device=dev)
model.train(not evalm)


def step():
    if evalm:
        with torch.no_grad():
            model(imgs, proj, dv)
    else:
        out = model(imgs, proj, dv)
        sum(o["attn_weight"].sum() for k, o in out.items() if k.startswith("stage")).backward()


step(); step()
torch.cuda.synchronize()
real = _lib.call
rec = []


def timed(name, *a):
    if "conv_mfma" not in name or "workspace" in name or "pack" in name:
        return real(name, *a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = real(name, *a)
    e1.record()
    rec.append((name, a, e0, e1))
    return r


_lib.call = timed
mv.hotpath._lib.call = timed
step()
torch.cuda.synchronize()
_lib.call = real
agg = defaultdict(lambda: [0, 0.0])
for name, a, e0, e1 in rec:
    ints = tuple(v for v in a if isinstance(v, int) and not isinstance(v, bool) and abs(v) < 100000)
    k = (name.replace("mvster_conv_mfma_", ""), ints)
    agg[k][0] += 1
    agg[k][1] += e0.elapsed_time(e1)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print("matrix-core convolution launches: %d, %.2f ms" % (len(rec), tot))
for (name, ints), (cnt, ms) in rows[:60]:
    print("%-16s %-52s x%2d  %7.3f ms total  %7.1f us each" % (name, str(ints), cnt, ms, 1e3 * ms / cnt))
