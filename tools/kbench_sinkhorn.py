#!/usr/bin/env python3
"""Times the fused Sinkhorn kernel against the same sweeps issued as device tensor ops (what the reference's
sinkhorn does on a GPU), per DTU stage shape, fwd+bwd, 10 sweeps."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                        # noqa: E402
import deep_reconstruction_with_epipolar_lines_mvster_amd as mv                     # noqa: E402
from deep_reconstruction_with_epipolar_lines_mvster_amd import losses               # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    for k, (H, W, C, G, D, scale, itv) in enumerate(bench.DTU_STAGES):
        base = 500 + 300 * torch.rand(B, 1, H, W, generator=g)
        hypo = (1.0 / (1.0 / base + (torch.arange(D).view(1, D, 1, 1) - (D - 1) / 2) * 3e-6)).to(dev)
        gt = (1.0 / (1.0 / base[:, 0] + (torch.rand(B, H, W, generator=g) - 0.5) * 2e-5)).to(dev)
        mask = (torch.rand(B, H, W, generator=g) < 0.7).to(dev)
        attn = torch.softmax(torch.randn(B, D, H, W, generator=g), 1).to(dev).requires_grad_(True)

        def fused():
            attn.grad = None
            mv.sinkhorn(gt, hypo, attn, mask, 10)[1].backward()

        def fused_noplan():
            attn.grad = None
            losses._stage_loss(gt, hypo, attn, mask, 10, 1, False).backward()

        def tensor_ops():            # the generic HIP kernels (any D <= 64, any sweep count): the same problem through loops + a workspace
            attn.grad = None
            pix, _ = losses._SinkhornPixelLossGeneric.apply(attn, gt, hypo, mask.view(torch.uint8), 10, 1.0, False, False)
            (pix.sum() / mask.sum()).backward()

        tf, tn, tt = timeit(fused, 5), timeit(fused_noplan, 5), timeit(tensor_ops, 2)
        print("stage %d  %4dx%-4d D=%d B=%d  fused kernel (loss only) %7.3f ms | with T_map %7.3f ms | generic HIP kernels %8.2f ms  (x%.0f)"
              % (k + 1, H, W, D, B, tn, tf, tt, tt / tn), flush=True)


if __name__ == "__main__":
    main()
