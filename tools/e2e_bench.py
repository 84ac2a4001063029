#!/usr/bin/env python3
"""End-to-end training step of the drop-in MVS4net on one GPU (informational; the headline bench is bench.py, which also
reports this step in its `e2e` object):
FPN4 over the N views (HIP convolutions, BatchNorm+ReLU, fused top-down merge) -> 4 x [HIP hot path + reg2d (HIP
convolutions on the fp32 matrix cores / direct kernels)] -> mono head -> MVS4net_loss (HIP sinkhorn, 10 sweeps) ->
backward -> Adam.  1600x1152 (1184 is illegal for reg2d), N = 5 views, inverse depth, group correlation.

    python tools/e2e_bench.py [--batch 1] [--steps 3] [--height 1152] [--width 1600] [--channels-last]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deep_reconstruction_with_epipolar_lines_mvster_amd as mv            # noqa: E402
import bench                                                               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--height", type=int, default=1152)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--eval", action="store_true")
    ap.add_argument("--benchmark", action="store_true", help="torch.backends.cudnn.benchmark = True (slow MIOpen find)")
    ap.add_argument("--bn", default="hip", choices=["hip", "native", "miopen"])
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole step (forward, loss, backward, Adam) in one HIP graph and time its replays: the "
                         "step without the ~1600 kernel launches' host-side cost")
    ap.add_argument("--ddp", action="store_true",
                    help="wrap the model in DistributedDataParallel over RCCL like train_mvs4.py:562-570 (one process per GPU; "
                         "start with torch.distributed.run, with --gpus N, or alone for a one-rank group)")
    ap.add_argument("--fpn-nchw", action="store_true", help="FPN4's output layers store NCHW maps (round-2 behaviour: one "
                    "layout-conversion launch per stage in the hot path)")
    ap.add_argument("--gpus", type=int, default=1,
                    help="with --ddp and no launcher: start this many ranks (one per GPU) through bench.launch_ranks")
    args = ap.parse_args()
    if args.ddp and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one (nothing above has touched the GPU); rank 0's output is relayed, a failing rank's is shown
        raise SystemExit(bench.launch_ranks(args.gpus, sys.argv[1:], program=os.path.abspath(__file__)))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.ddp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        dist.init_process_group(backend="nccl", init_method="env://", rank=int(os.environ.get("RANK", "0")),
                                world_size=int(os.environ.get("WORLD_SIZE", "1")), device_id=dev)
    # the reference drivers set cudnn.benchmark = True (train_mvs4.py:16, test_mvs4.py:25); on ROCm that makes MIOpen run
    # its exhaustive find for every convolution configuration -- the first step then takes more than 7 minutes
    torch.backends.cudnn.benchmark = args.benchmark
    from deep_reconstruction_with_epipolar_lines_mvster_amd import mvs4net as _m
    _m._BN_IMPL = args.bn
    _m._FPN_EMITS_CHANNELS_LAST = not args.fpn_nchw
    torch.manual_seed(0)
    B, N, H, W = args.batch, args.views, args.height, args.width
    model = mv.MVS4net(arch_mode="fpn", reg_net="reg2d", num_stage=4, fpn_base_channel=8, reg_channel=8,
                       stage_splits=[8, 8, 4, 4], depth_interals_ratio=[0.5, 0.5, 0.5, 1], group_cor=True,
                       group_cor_dim=[8, 8, 4, 4], inverse_depth=True, mono=True, attn_temp=2).to(dev)
    if args.channels_last:
        model.to_channels_last(True)
    if args.ddp:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=args.graph)
    imgs = [torch.rand(B, 3, H, W, device=dev) for _ in range(N)]
    proj = {}
    for k, sc in enumerate((8, 4, 2, 1)):
        proj["stage%d" % (k + 1)] = bench.synth_views(B, N, H // sc, W // sc).to(dev)
    dv = torch.tensor([[425.0, 935.0]] * B, device=dev)
    gts = {"stage%d" % (k + 1): 480 + 400 * torch.rand(B, H // sc, W // sc, device=dev) for k, sc in enumerate((8, 4, 2, 1))}
    masks = {k: (torch.rand_like(v) < 0.7).float() for k, v in gts.items()}
    kw = dict(stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10, ot_eps=1, ot_continous=False, mono=True)

    def train_step():
        opt.zero_grad(set_to_none=True)
        out = model(imgs, proj, dv)
        loss = mv.MVS4net_loss(out, gts, masks, **kw)[0]
        loss.backward()
        opt.step()
        return loss

    def eval_step():
        with torch.no_grad():
            return model(imgs, proj, dv)["stage4"]["depth"]

    if args.eval:
        model.eval()
    step = eval_step if args.eval else train_step
    step()
    torch.cuda.synchronize()
    if args.graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # allocator and MIOpen warm-up on the capture side stream
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        if not args.eval:
            opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            step()
        step = graph.replay
        step()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    if args.ddp:
        import torch.distributed as dist
        print("DDP over %s, %d rank(s)" % (dist.get_backend(), dist.get_world_size()), flush=True)
        dist.destroy_process_group()
    if int(os.environ.get("RANK", "0")) != 0:
        return
    world = int(os.environ.get("WORLD_SIZE", "1")) if args.ddp else 1
    print("e2e %s step%s (bn=%s): %dx%d, N=%d, B=%d per rank x %d rank(s), channels_last=%s: %.1f ms/step = %.2f ref-views/s, peak mem %.1f GB"
          % ("eval" if args.eval else "train", " as one HIP graph" if args.graph else "", args.bn, W, H, N, B, world, args.channels_last, dt * 1e3, world * B / dt,
             torch.cuda.max_memory_allocated() / 1e9), flush=True)


if __name__ == "__main__":
    main()
