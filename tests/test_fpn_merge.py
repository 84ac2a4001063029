"""F3 (GPU): the fused top-down merge of the feature pyramid, y = bilinear_x2(feat, align_corners=True) +
conv1x1(skip; weight, bias) (csrc/fpn_merge.hip), against the reference's operator sequence
(models/mvs4net_utils.py:490-505: F.interpolate + nn.Conv2d(1x1) + add) evaluated in float64 on the CPU:
forward and all four gradients, odd sizes, with and without bias, and the module-level switch inside FPN4."""
import pytest
import torch
import torch.nn.functional as F

from deep_reconstruction_with_epipolar_lines_mvster_amd import mvs4net

pytestmark = pytest.mark.gpu


def _close(got, want, what, rel=2e-5):
    want = want.detach().cpu().double()
    err = float((got.detach().cpu().double() - want).abs().max())
    scale = float(want.abs().max())
    print("    [parity] %-34s max|err| %.3e (max|ref| %.3e)" % (what, err, scale))
    assert err <= rel * scale + 1e-6, "%s: %.3e" % (what, err)


@pytest.mark.parametrize("N,Ct,Cs,h,w,bias", [(2, 64, 8, 13, 17, True), (1, 64, 16, 9, 20, True), (3, 64, 32, 6, 5, True),
                                              (2, 32, 8, 7, 11, False), (1, 64, 8, 2, 2, True)])
def test_top_down_merge_matches_torch_operators(N, Ct, Cs, h, w, bias):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(N * 100 + Cs + h)
    feat = torch.randn(N, Ct, h, w, generator=g)
    skip = torch.randn(N, Cs, 2 * h, 2 * w, generator=g)
    weight = 0.3 * torch.randn(Ct, Cs, 1, 1, generator=g)
    b = torch.randn(Ct, generator=g) if bias else None
    gy = torch.randn(N, Ct, 2 * h, 2 * w, generator=g)

    ref_in = [t.double().requires_grad_(True) if t is not None else None for t in (feat, skip, weight, b)]
    want = F.interpolate(ref_in[0], scale_factor=2, mode="bilinear", align_corners=True) + F.conv2d(ref_in[1], ref_in[2], ref_in[3])
    want.backward(gy.double())

    got_in = [t.to(dev).requires_grad_(True) if t is not None else None for t in (feat, skip, weight, b)]
    got = mvs4net._TopDownMerge.apply(*got_in)
    got.backward(gy.to(dev))
    tag = "N%d Ct%d Cs%d %dx%d" % (N, Ct, Cs, h, w)
    _close(got, want.detach(), tag + " y")
    for name, a, r in zip(("dfeat", "dskip", "dweight", "dbias"), got_in, ref_in):
        if a is not None:
            _close(a.grad, r.grad, tag + " " + name, rel=5e-5)


def test_fpn4_uses_the_fused_merge_and_matches_the_operator_path():
    """FPN4 forward + backward with the fused merge vs the torch operator sequence (same module, switch flipped)."""
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    net = mvs4net.FPN4(8).to(dev).train()
    x = torch.rand(2, 3, 64, 96, device=dev)
    outs = {}
    for fused in (True, False):
        mvs4net._FUSED_MERGE = fused
        try:
            net.zero_grad(set_to_none=True)
            xin = x.clone().requires_grad_(True)
            y = net(xin)
            loss = sum((v * v).mean() for v in y.values())
            loss.backward()
            outs[fused] = ({k: v.detach() for k, v in y.items()}, xin.grad, {n: p.grad.clone() for n, p in net.named_parameters()})
        finally:
            mvs4net._FUSED_MERGE = True
    for k in outs[True][0]:
        _close(outs[True][0][k], outs[False][0][k], "FPN4 " + k, rel=1e-4)
    _close(outs[True][1], outs[False][1], "FPN4 d/d image", rel=2e-3)
    for n in outs[True][2]:
        if "inner" in n:
            _close(outs[True][2][n], outs[False][2][n], "FPN4 grad " + n, rel=2e-3)


def test_fpn4_one_pass_over_all_views_equals_one_pass_per_view():
    """Training mode: FPN4 over the stacked views with per-view BatchNorm statistics (`groups`) vs the reference's loop,
    one view at a time (models/MVS4Net.py:88-93): features, image / parameter gradients, running statistics."""
    import copy
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    views, B = 3, 2
    net_a = mvs4net.FPN4(8).to(dev).train()
    net_b = copy.deepcopy(net_a)
    imgs = [torch.rand(B, 3, 64, 96, device=dev) * (1.0 + v) for v in range(views)]
    wts = None
    res = []
    for net, batched in ((net_a, True), (net_b, False)):
        xs = [i.clone().requires_grad_(True) for i in imgs]
        if batched:
            pyr = {k: v.split(B, 0) for k, v in net(torch.cat(xs, 0), groups=views).items()}
            feats = [{k: v[i] for k, v in pyr.items()} for i in range(views)]
        else:
            feats = [net(x) for x in xs]
        if wts is None:
            wts = [{k: torch.randn_like(v) for k, v in f.items()} for f in feats]
        loss = sum((f[k] * w[k]).sum() for f, w in zip(feats, wts) for k in f)
        loss.backward()
        res.append((feats, [x.grad for x in xs], net))
    (fa, ga, na), (fb, gb, nb) = res
    for i in range(views):
        for k in fa[i]:
            _close(fa[i][k], fb[i][k], "view %d %s" % (i, k), rel=1e-4)
        _close(ga[i], gb[i], "view %d d/d image" % i, rel=2e-3)
    for (n, p), (_, q) in zip(na.named_parameters(), nb.named_parameters()):
        _close(p.grad, q.grad, "grad " + n, rel=2e-3)
    for (n, p), (_, q) in zip(na.named_buffers(), nb.named_buffers()):
        if n.endswith("num_batches_tracked"):
            assert int(p) == int(q) == views, n
        else:
            _close(p, q, "buffer " + n, rel=1e-5)


@pytest.mark.parametrize("N,h,w", [(2, 13, 17), (1, 24, 40), (3, 9, 33)])
def test_top_level_head_matches_the_operator_sequence(N, h, w):
    """FPN4's last level computed without the 64-channel full-resolution map (channel contraction at low resolution +
    gather of nine shifted upsampled planes, csrc/fpn_merge.hip) vs the reference's operator sequence
    (models/mvs4net_utils.py:500-505) in float64: output and the gradients of feat, conv0, inner3.weight/bias, out4.weight."""
    import torch.nn as nn
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(N * 100 + h + w)
    feat, skip = torch.randn(N, 64, h, w, generator=g), torch.randn(N, 8, 2 * h, 2 * w, generator=g)
    lateral, head = nn.Conv2d(8, 64, 1, bias=True), nn.Conv2d(64, 8, 3, padding=1, bias=False)
    with torch.no_grad():
        lateral.bias.copy_(torch.randn(64, generator=g))
    gy = torch.randn(N, 8, 2 * h, 2 * w, generator=g)
    lat64, head64 = nn.Conv2d(8, 64, 1).double(), nn.Conv2d(64, 8, 3, padding=1, bias=False).double()
    lat64.load_state_dict({k: v.double() for k, v in lateral.state_dict().items()})
    head64.load_state_dict({k: v.double() for k, v in head.state_dict().items()})
    fr, sr = feat.double().requires_grad_(True), skip.double().requires_grad_(True)
    want = head64(F.interpolate(fr, scale_factor=2, mode="bilinear", align_corners=True) + lat64(sr))
    (want * gy.double()).sum().backward()
    lateral, head = lateral.to(dev), head.to(dev)
    fg, sg = feat.to(dev).requires_grad_(True), skip.to(dev).requires_grad_(True)
    got = mvs4net._top_level_head(fg, sg, lateral, head)
    (got * gy.to(dev)).sum().backward()
    tag = "head N%d %dx%d" % (N, h, w)
    _close(got, want, tag + " out4", rel=5e-6)
    _close(fg.grad, fr.grad, tag + " d/d feat", rel=1e-5)
    _close(sg.grad, sr.grad, tag + " d/d conv0", rel=1e-5)
    _close(lateral.weight.grad, lat64.weight.grad, tag + " d/d inner3.weight", rel=2e-5)
    _close(lateral.bias.grad, lat64.bias.grad, tag + " d/d inner3.bias", rel=2e-5)
    _close(head.weight.grad, head64.weight.grad, tag + " d/d out4.weight", rel=2e-5)


def test_output_layers_store_channels_last_with_identical_values():
    """out1 .. out4 write their maps as torch.channels_last from the kernels' own epilogues (matrix-core 1x1 /
    3x3, direct 3x3, the head gather): same bits as the NCHW stores, the layout the cost-volume kernels gather from, and
    the backward (which receives planes) is unchanged."""
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    net = mvs4net.FPN4(8).to(dev).train()
    x = torch.rand(3, 3, 72, 104, device=dev)
    res = {}
    real_call = mvs4net._lib.call
    for cl in (True, False):
        mvs4net._FPN_EMITS_CHANNELS_LAST = cl
        calls = []
        mvs4net._lib.call = lambda name, *a: (calls.append((name, a)), real_call(name, *a))[1]
        try:
            net.zero_grad(set_to_none=True)
            y = net(x)
            gy = {k: torch.randn(v.shape, generator=torch.Generator().manual_seed(len(k) + v.shape[1])).to(dev) for k, v in y.items()}
            sum((y[k] * gy[k]).sum() for k in y).backward()
            res[cl] = ({k: v.detach() for k, v in y.items()}, {n: p.grad.clone() for n, p in net.named_parameters()}, calls)
        finally:
            mvs4net._FPN_EMITS_CHANNELS_LAST = True
            mvs4net._lib.call = real_call
    for k, v in res[True][0].items():
        assert v.is_contiguous(memory_format=torch.channels_last) and not v.is_contiguous(), k
        assert res[False][0][k].is_contiguous(), k
        assert torch.equal(v, res[False][0][k]), k
    # the four producers ran with the channels-last flag: two matrix-core layers (out1, out2), one direct layer (out3), the gather
    names = [(n, a) for n, a in res[True][2]]
    assert sum(1 for n, a in names if n == "mvster_conv_mfma_fwd" and a[-2] == 2) == 2
    assert sum(1 for n, a in names if n == "mvster_conv_direct_fwd_layout" and a[-2] == mvs4net._lib.LAYOUT_NHWC) == 1
    assert sum(1 for n, a in names if n == "mvster_fpn_head_gather_fwd_layout" and a[-2] == mvs4net._lib.LAYOUT_NHWC) == 1
    for n in res[True][1]:
        _close(res[True][1][n], res[False][1][n], "grad " + n, rel=1e-6)


@pytest.mark.parametrize("N,Cin,Cout,K,H,W,impl", [(2, 64, 64, 1, 19, 45, "mfma"), (1, 64, 32, 3, 21, 70, "mfma"), (2, 24, 20, 3, 9, 33, "mfma"),
                                                   (2, 64, 16, 3, 18, 37, "direct"), (1, 8, 8, 3, 30, 65, "direct"), (3, 16, 4, 1, 7, 9, "direct")])
def test_convolution_kernels_write_channels_last(N, Cin, Cout, K, H, W, impl):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, generator=g).to(dev)
    w = (0.2 * torch.randn(Cout, Cin, K, K, generator=g)).to(dev)
    b = torch.randn(Cout, generator=g).to(dev)
    route = (impl, impl, impl)
    a = mvs4net._DirectConv2d.apply(x, w, b, 1, route, False)
    c = mvs4net._DirectConv2d.apply(x, w, b, 1, route, True)
    assert c.shape == a.shape and c.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(a, c)
    want = F.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), 1, K // 2)
    _close(c, want, "%s %d->%d k%d channels-last" % (impl, Cin, Cout, K), rel=2e-5)
