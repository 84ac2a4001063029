import sys, torch
sys.path.insert(0, '/root/repo')
import bench
import deep_reconstruction_with_epipolar_lines_mvster_amd as mv
from deep_reconstruction_with_epipolar_lines_mvster_amd import hotpath
dev = torch.device("cuda:0")
for (H, W, C, G, D, S, B) in ((768, 1024, 16, 4, 8, 9, 1), (384, 512, 32, 8, 8, 9, 1), (768, 1024, 16, 4, 8, 4, 2)):
    g = torch.Generator().manual_seed(0)
    feats = [torch.randn(B, C, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) for _ in range(S + 1)]
    mats = bench.synth_views(B, S + 1, H, W).to(dev)
    inv = 1 / 935.0 + (1 / 425.0 - 1 / 935.0) * torch.rand(B, 1, H, W, generator=g)
    hypo = (1.0 / (inv + (torch.arange(D).view(1, D, 1, 1) - (D - 1) / 2) * (1 / 425.0 - 1 / 935.0) / 128)).contiguous().to(dev)
    pose = mv.stage_poses(mats)
    gvol = torch.randn(B, G, D, H, W, device=dev)
    res = {}
    for mode in (True, False):
        hotpath.TILE_SCATTER = mode
        ts = []
        for it in range(6):
            for f in feats: f.grad = None
            vol = mv.fused_cost_volume(feats, pose, hypo, G, 2.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); vol.backward(gvol); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res[mode] = (sorted(ts)[len(ts) // 2], [f.grad.clone() for f in feats])
    err = max(float((x - y).abs().max()) for x, y in zip(res[True][1], res[False][1]))
    print("H%d W%d C%d D%d S%d B%d: bwd tiled %.3f ms, direct %.3f ms, max |diff| %.2e" % (H, W, C, D, S, B, res[True][0], res[False][0], err))
