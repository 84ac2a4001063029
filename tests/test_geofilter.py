"""F4 geometric-consistency filter.  PARITY UNPINNED (see oracle/geofilter_oracle.py: cv2.remap is restated, cv2 is not
available): the CPU tests check the numpy restatement on a synthetic multi-view-consistent scene where the answer is
known; the GPU tests check the HIP kernel against that restatement."""
import numpy as np
import pytest
import torch

from oracle import geofilter_oracle as gfo


def scene(H, W, n_views, f=None, noise=0.0, seed=0):
    """Depth maps of a smooth surface Z(X,Y) seen by n pinhole cameras (identity rotation, baseline along x/y)."""
    rng = np.random.default_rng(seed)
    f = f or 1.8 * W
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1.0]])
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    dx, dy = (xs - W / 2.0) / f, (ys - H / 2.0) / f
    surf = lambda X, Y: 650.0 + 90.0 * np.sin(X / 150.0) * np.cos(Y / 120.0)
    views = []
    for v in range(n_views):
        E = np.eye(4)
        E[0, 3], E[1, 3] = 25.0 * v, -8.0 * v
        z = np.full_like(dx, 650.0)
        for _ in range(30):
            z = surf(z * dx - E[0, 3], z * dy - E[1, 3])
        depth = (z + noise * rng.standard_normal(z.shape)).astype(np.float32)
        views.append((depth, K.copy(), E))
    return views


def test_oracle_accepts_a_consistent_scene_and_rejects_a_wrong_depth():
    views = scene(48, 64, 3)
    (d0, K0, E0), (d1, K1, E1) = views[0], views[1]
    mask, drep, xs, ys = gfo.check_geometric_consistency(d0, K0, E0, d1, K1, E1)
    inside = (xs > 1) & (xs < 62) & (ys > 1) & (ys < 46)
    assert mask[inside].mean() > 0.99                          # consistent geometry passes wherever it is observed
    assert np.abs(drep[mask] - d0[mask]).max() < 0.02 * 6.5      # reprojected depth == reference depth (1e-2 relative gate)
    assert (drep[~mask] == 0).all()
    wrong = d0 * 1.05                                          # 5 % depth error > 1 % gate
    mask2, _, _, _ = gfo.check_geometric_consistency(wrong, K0, E0, d1, K1, E1)
    assert mask2.mean() < 0.05
    same, drep3, xs3, ys3 = gfo.check_geometric_consistency(d0, K0, E0, d0, K0, E0)    # a view against itself
    assert same.all() and np.allclose(drep3, d0, rtol=1e-6) and np.allclose(xs3, np.arange(64)[None].repeat(48, 0), atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,noise", [(48, 64, 0.0), (120, 160, 1.5), (300, 400, 4.0)])
def test_hip_check_matches_restatement(H, W, noise):
    import deep_reconstruction_with_epipolar_lines_mvster_amd as mv
    views = scene(H, W, 4, noise=noise, seed=H)
    d0, K0, E0 = views[0]
    for (d1, K1, E1) in views[1:]:
        want = gfo.check_geometric_consistency(d0, K0, E0, d1, K1, E1)
        got = mv.check_geometric_consistency(d0, K0, E0, d1, K1, E1)
        assert np.abs(got[2] - want[2]).max() <= 1e-4 * W and np.abs(got[3] - want[3]).max() <= 1e-4 * H   # x_src, y_src
        differ = got[0] != want[0]
        # float64 matrix products are summed in a different order than BLAS: only pixels ON a threshold may flip
        assert differ.mean() < 2e-3
        both = got[0] & want[0]
        assert np.abs(got[1][both] - want[1][both]).max() < 1e-3
    want_sum = sum(gfo.check_geometric_consistency(d0, K0, E0, d, K, E)[0].astype(np.int32) for d, K, E in views[1:])
    geo, averaged, count = mv.geometric_filter(d0, K0, E0, views[1:], geomask=2)
    assert (count != want_sum).mean() < 5e-3
    want_avg = (sum(gfo.check_geometric_consistency(d0, K0, E0, d, K, E)[1] for d, K, E in views[1:]) + d0) / (want_sum + 1)
    agree = count == want_sum
    assert np.abs(averaged[agree] - want_avg[agree]).max() < 1e-3
    assert (geo == (count >= 2)).all()


@pytest.mark.gpu
def test_hip_check_full_resolution_timing():
    import time
    import deep_reconstruction_with_epipolar_lines_mvster_amd as mv
    views = scene(1184, 1600, 3, noise=1.0)
    d0, K0, E0 = views[0]
    d1, K1, E1 = views[1]
    t0 = time.perf_counter()
    want = gfo.check_geometric_consistency(d0, K0, E0, d1, K1, E1)
    t_cpu = time.perf_counter() - t0
    dev = torch.device("cuda:0")
    a, b = torch.from_numpy(d0).to(dev), torch.from_numpy(d1).to(dev)
    mv.check_geometric_consistency(a, K0, E0, b, K1, E1)
    torch.cuda.synchronize()
    from deep_reconstruction_with_epipolar_lines_mvster_amd import geofilter as gf
    mats = gf._mats(K0, E0, K1, E1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        out = gf._check(a, b, mats, 1.0, 0.01, True, dev)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    px = 1184 * 1600
    print("    [F4] 1600x1184 pair: numpy restatement %.0f ms, HIP kernel %.3f ms (%.0f GB/s algorithmic: %d B/px)"
          % (t_cpu * 1e3, ms, px * 25 / ms / 1e6, 25))
    assert (out[0].cpu().numpy().astype(bool) != want[0]).mean() < 2e-3
