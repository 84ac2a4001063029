"""F2/F3 inference path (GPU): convolution -> eval-mode BatchNorm -> ReLU (-> + skip connection) as ONE kernel
(csrc/conv_epilogue.h; VERDICT r2 #4b) with the derived weight arrays built once per parameter version (#4d), against
  * the reference's operator sequence in float64 on the CPU (models/mvs4net_utils.py:123-130,230-256,901-912,921-923), and
  * the unfused HIP route (convolution kernel, then mvster_bn_eval_fwd), which it must equal bit for bit."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from deep_reconstruction_with_epipolar_lines_mvster_amd import mvs4net

pytestmark = pytest.mark.gpu


def _close(got, want, what, rel=2e-5):
    want = want.detach().cpu().double()
    err = float((got.detach().cpu().double() - want).abs().max())
    scale = float(want.abs().max())
    print("    [parity] %-44s max|err| %.3e (max|ref| %.3e)" % (what, err, scale))
    assert err <= rel * scale + 1e-6, "%s: %.3e" % (what, err)


def _randomise(bn, g):
    with torch.no_grad():
        bn.weight.copy_(0.5 + torch.rand(bn.num_features, generator=g))
        bn.bias.copy_(torch.randn(bn.num_features, generator=g))
        bn.running_mean.copy_(torch.randn(bn.num_features, generator=g))
        bn.running_var.copy_(0.3 + torch.rand(bn.num_features, generator=g))


def _both(fn):
    """fn() under no_grad with the fused path on and off, plus the names of the ABI calls of the fused run."""
    real = mvs4net._lib.call
    out = {}
    for fused in (True, False):
        mvs4net._FUSED_EVAL = fused
        names = []
        mvs4net._lib.call = lambda name, *a: (names.append(name), real(name, *a))[1]
        try:
            with torch.no_grad():
                out[fused] = (fn(), names)
        finally:
            mvs4net._FUSED_EVAL = True
            mvs4net._lib.call = real
    return out[True][0], out[False][0], out[True][1]


@pytest.mark.parametrize("N,Cin,Cout,K,stride,H,W", [(2, 3, 8, 3, 1, 37, 50), (1, 8, 8, 3, 1, 20, 33), (2, 8, 16, 5, 2, 38, 52),
                                                     (1, 16, 16, 3, 1, 19, 70), (2, 16, 32, 5, 2, 21, 45), (1, 32, 32, 3, 1, 18, 40),
                                                     (1, 32, 64, 5, 2, 17, 31), (2, 64, 64, 3, 1, 9, 35), (1, 64, 32, 3, 1, 12, 33)])
def test_conv_bn_relu_2d_is_one_kernel_in_inference(N, Cin, Cout, K, stride, H, W):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Cin * 7 + Cout + K)
    block = mvs4net._ConvBNReLU2d(Cin, Cout, K, stride, K // 2)
    _randomise(block.bn, g)
    x = torch.randn(N, Cin, H, W, generator=g)
    want = F.relu(F.batch_norm(F.conv2d(x.double(), block.conv.weight.double(), None, stride, K // 2), block.bn.running_mean.double(),
                               block.bn.running_var.double(), block.bn.weight.double(), block.bn.bias.double(), False, 0.1, block.bn.eps))
    block = block.to(dev).eval()
    xg = x.to(dev)
    fused, plain, names = _both(lambda: block(xg))
    tag = "2d %d->%d k%d s%d" % (Cin, Cout, K, stride)
    assert [n for n in names if "conv" in n or "bn" in n][-1] in ("mvster_conv_mfma_fwd_ep", "mvster_conv_direct_fwd_ep"), names
    assert not any(n.startswith("mvster_bn_eval_fwd") for n in names), names
    assert torch.equal(fused, plain), tag
    _close(fused, want, tag)
    # second call: nothing is rebuilt (one launch); an in-place weight update invalidates the cached arrays
    _, _, names2 = _both(lambda: block(xg))
    assert names2 in (["mvster_conv_mfma_fwd_ep"], ["mvster_conv_direct_fwd_ep"]), names2
    with torch.no_grad():
        block.conv.weight.mul_(0.5)
        block.bn.running_mean.add_(0.25)
    fused3, plain3, names3 = _both(lambda: block(xg))
    assert len(names3) > 1 and torch.equal(fused3, plain3) and not torch.equal(fused3, fused)


@pytest.mark.parametrize("B,D,Cin,Cout,kd,stride,H,W", [(1, 4, 4, 8, 3, 1, 24, 40), (2, 8, 8, 8, 3, 1, 9, 35), (1, 4, 8, 16, 1, 2, 24, 40),
                                                        (1, 4, 16, 16, 3, 1, 12, 20), (2, 3, 16, 32, 1, 2, 17, 33), (1, 8, 32, 32, 3, 1, 8, 32),
                                                        (1, 4, 32, 64, 1, 2, 9, 18), (1, 4, 64, 64, 3, 1, 6, 11)])
def test_conv_bn_relu_3d_over_depth_slices_is_one_kernel_in_inference(B, D, Cin, Cout, kd, stride, H, W):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Cin + 3 * Cout + kd)
    if kd == 3:
        block = mvs4net._ConvBNReLU3d(Cin, Cout)
    else:
        block = mvs4net._ConvBNReLU3d(Cin, Cout, kernel=(1, 3, 3), stride=(1, stride, stride), pad=(0, 1, 1))
    _randomise(block.bn, g)
    x = torch.randn(B, Cin, D, H, W, generator=g)
    ref = block.double().eval()
    with torch.no_grad():
        want = ref(x.double())
    block = block.float().to(dev).eval()
    x2d = x.permute(0, 2, 1, 3, 4).reshape(B * D, Cin, H, W).contiguous().to(dev)
    fused, plain, names = _both(lambda: block.forward_slices(x2d, B, D))
    tag = "3d %d->%d kd%d s%d D%d" % (Cin, Cout, kd, stride, D)
    assert names[-1] in ("mvster_conv_mfma_fwd_ep", "mvster_conv_direct_fwd_ep"), names
    assert torch.equal(fused, plain), tag
    got = fused.view(B, D, Cout, fused.shape[2], fused.shape[3]).permute(0, 2, 1, 3, 4)
    _close(got, want, tag)


@pytest.mark.parametrize("N,Cin,Cout,H,W,skip", [(4, 64, 32, 9, 13, True), (2, 32, 16, 20, 33, True), (4, 16, 8, 18, 40, True), (1, 64, 32, 7, 9, False)])
def test_up_convolution_bn_relu_skip_is_one_kernel_in_inference(N, Cin, Cout, H, W, skip):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Cin + Cout + H)
    seq = mvs4net._up3d(Cin, Cout)
    _randomise(seq[1], g)
    x = torch.randn(N, Cin, H, W, generator=g)
    e = torch.randn(N, Cout, 2 * H, 2 * W, generator=g) if skip else None
    bn = seq[1]
    want = F.relu(F.batch_norm(F.conv_transpose2d(x.double(), seq[0].weight[:, :, 0].double(), None, 2, 1, 1), bn.running_mean.double(),
                               bn.running_var.double(), bn.weight.double(), bn.bias.double(), False, 0.1, bn.eps))
    if skip:
        want = want + e.double()
    seq = seq.to(dev).eval()
    xg, eg = x.to(dev), (e.to(dev) if skip else None)
    fused, plain, names = _both(lambda: mvs4net.reg2d._up(seq, xg, eg))
    assert names[-1] == "mvster_conv_mfma_transposed_ep", names
    assert torch.equal(fused, plain)
    _close(fused, want, "up %d->%d skip %s" % (Cin, Cout, skip))


def test_training_and_autograd_keep_the_unfused_path():
    dev = torch.device("cuda:0")
    block = mvs4net._ConvBNReLU2d(8, 16, 3, 1, 1).to(dev)
    x = torch.randn(2, 8, 12, 20, device=dev)
    real = mvs4net._lib.call
    for mode, grad in (("train", True), ("eval", True)):
        block.train(mode == "train")
        names = []
        mvs4net._lib.call = lambda name, *a: (names.append(name), real(name, *a))[1]
        try:
            with torch.set_grad_enabled(grad):
                y = block(x)
        finally:
            mvs4net._lib.call = real
        assert not any(n.endswith("_ep") for n in names), (mode, names)
        assert y.requires_grad


def test_running_statistics_updated_by_the_training_kernels_reach_the_fused_path():
    """The HIP training kernels write running_mean / running_var through their pointers (no tensor version moves): the cached
    BatchNorm affine must still be rebuilt after a training pass."""
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    block = mvs4net._ConvBNReLU2d(8, 16, 3, 1, 1).to(dev)
    x = 3.0 + torch.randn(2, 8, 12, 20, device=dev)
    block.eval()
    with torch.no_grad():
        before = block(x)
    block.train()
    block(x)                                     # updates the running statistics
    block.eval()
    fused, plain, _ = _both(lambda: block(x))
    assert torch.equal(fused, plain) and not torch.equal(fused, before)
