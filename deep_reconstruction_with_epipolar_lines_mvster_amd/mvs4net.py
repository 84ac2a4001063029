"""A7 -- ``MVS4net`` API shell around the HIP hot path (reference: models/MVS4Net.py:16-193).

Same constructor keywords, ``forward(imgs, proj_matrices, depth_values, filename=None)`` signature, output
dictionary and sub-module / parameter names (``feature``, ``reg``, ``mono_depth_decoder``, ``stagenet`` ...), so
reference checkpoints load with ``strict=True`` and ``train_mvs4.py`` / ``test_mvs4.py`` run unchanged.

The stage body is ``hotpath.stagenet`` + the hypothesis-scheduling kernels.  The
feature pyramid, the 3-D regulariser and the monocular head (section 8f rows F2, F3) keep the reference's modules,
parameter names and layer shapes (models/mvs4net_utils.py:123-130,230-256,426-509,840-882,884-926) -- so that
checkpoints load -- while their operators run on HIP kernels behind the C ABI: BatchNorm+ReLU (csrc/batchnorm.hip),
the FPN top-down merge and top-level head (csrc/fpn_merge.hip), the convolutions (csrc/conv_direct.hip,
csrc/conv_mfma.hip; one operation of one layer stays on MIOpen on measured merit, see ``_conv_route``).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .hotpath import stagenet, init_range, init_inverse_range, schedule_inverse_range, to_planar, _on, _stream

__all__ = ["MVS4net", "FPN4", "reg2d", "mono_depth_decoder"]


# --------------------------------------------------------------------------------------------------
# building blocks: the reference's modules and parameter names, operators on the HIP kernels
# --------------------------------------------------------------------------------------------------
class _BNReLUTrain(torch.autograd.Function):
    """Training-mode BatchNorm (+ReLU) as two streaming HIP passes each way (csrc/batchnorm.hip).  ``groups`` > 1: the
    batch is that many runs of consecutive samples, each normalised with its own statistics (NCHW only)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu, groups=1, residual=None):
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C)
        dev = x.device
        nhwc = not x.is_contiguous()                     # _batch_norm admits only NCHW- or channels_last-dense inputs
        y = torch.empty_like(x)                          # preserve_format: same strides as x
        save_mean = torch.empty(groups, C, dtype=torch.float32, device=dev)
        save_invstd = torch.empty(groups, C, dtype=torch.float32, device=dev)
        affine = torch.empty(groups, C, 2, dtype=torch.float32, device=dev)
        sums = torch.empty(groups, C, 2, dtype=torch.float64, device=dev)
        opt = lambda t: t.data_ptr() if t is not None else None
        with _on(dev):
            if nhwc:
                _lib.call("mvster_bn_train_fwd_nhwc", x.data_ptr(), opt(weight), opt(bias), y.data_ptr(),
                          save_mean.data_ptr(), save_invstd.data_ptr(), opt(running_mean), opt(running_var),
                          sums.data_ptr(), affine.data_ptr(), N * HW, C, float(momentum), float(eps),
                          1 if relu else 0, _stream(dev))
            elif groups == 1 and residual is None:
                _lib.call("mvster_bn_train_fwd", x.data_ptr(), opt(weight), opt(bias), y.data_ptr(), save_mean.data_ptr(),
                          save_invstd.data_ptr(), opt(running_mean), opt(running_var), sums.data_ptr(), affine.data_ptr(),
                          N, C, HW, float(momentum), float(eps), 1 if relu else 0, _stream(dev))
            else:
                _lib.call("mvster_bn_train_fwd_grouped", x.data_ptr(), opt(weight), opt(bias), opt(residual), y.data_ptr(),
                          save_mean.data_ptr(), save_invstd.data_ptr(), opt(running_mean), opt(running_var), sums.data_ptr(),
                          affine.data_ptr(), N, C, HW, groups, float(momentum), float(eps), 1 if relu else 0, _stream(dev))
        ctx.save_for_backward(x, affine, save_mean, save_invstd)
        ctx.cfg = (bool(relu), weight is not None, bias is not None, nhwc, groups)
        ctx.has_residual = residual is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, affine, save_mean, save_invstd = ctx.saved_tensors
        relu, has_w, has_b, nhwc, groups = ctx.cfg
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C)
        dev = x.device
        dy = dy.contiguous(memory_format=torch.channels_last) if nhwc else dy.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        sums = torch.empty(groups, C, 2, dtype=torch.float64, device=dev)
        with _on(dev):
            if nhwc:
                _lib.call("mvster_bn_train_bwd_nhwc", dy.data_ptr(), x.data_ptr(), affine.data_ptr(),
                          save_mean.data_ptr(), save_invstd.data_ptr(), dx.data_ptr(), dgamma.data_ptr(),
                          dbeta.data_ptr(), sums.data_ptr(), N * HW, C, 1 if relu else 0, _stream(dev))
            elif groups == 1:
                _lib.call("mvster_bn_train_bwd", dy.data_ptr(), x.data_ptr(), affine.data_ptr(), save_mean.data_ptr(),
                          save_invstd.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), sums.data_ptr(),
                          N, C, HW, 1 if relu else 0, _stream(dev))
            else:
                _lib.call("mvster_bn_train_bwd_grouped", dy.data_ptr(), x.data_ptr(), affine.data_ptr(), save_mean.data_ptr(),
                          save_invstd.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), sums.data_ptr(),
                          N, C, HW, groups, 1 if relu else 0, _stream(dev))
        return (dx, (dgamma if has_w else None), (dbeta if has_b else None), None, None, None, None, None, None,
                dy if ctx.has_residual else None)


def _batch_norm(bn, x, relu=False, groups=1, residual=None):
    """Batch normalisation (+ optional ReLU, + optional ``residual`` added after it: the skip connections of reg2d, folded
    into the normalisation pass) with the module's parameters/buffers.  ``groups`` > 1 (training mode): the
    batch holds that many runs of consecutive samples that the reference would have pushed through one after the other
    (the views of MVS4net.forward) -- each run gets its own statistics and the running statistics one update per run.

    Training mode on a HIP fp32 tensor, NCHW-contiguous or channels_last-dense: our two-pass kernels
    (``_BN_IMPL = "hip"``, default; one kernel family per layout, csrc/batchnorm.hip).
    MIOpen's spatial BN runs ~10x below the HBM rate at these shapes (8-64 channels, up to 1.8 M pixels per plane:
    42 ms of a 128 ms training step); ATen's native kernels measured slower still (142 vs 128 ms per step).
    Eval mode without autograd (inference): one HIP pass, ``mvster_bn_eval_fwd`` (affine with the running statistics, ReLU and
    the skip connection fused).  Every other case goes through ``F.batch_norm`` (MIOpen on ROCm)."""
    training = bn.training or not bn.track_running_stats
    if groups > 1 and training:
        if x.shape[0] % groups:
            raise ValueError("batch of %d samples does not split into %d statistic groups" % (x.shape[0], groups))
        if not (_BN_IMPL == "hip" and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and bn.momentum is not None):
            y = torch.cat([_batch_norm(bn, part, relu) for part in x.chunk(groups, 0)], 0)
            return y if residual is None else y + residual
    else:
        groups = 1
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(groups)
    dense_cl = x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] in (4, 8, 16, 32, 64)
    fused = (_BN_IMPL == "hip" and training and x.is_cuda and x.dtype == torch.float32
             and (x.is_contiguous() or dense_cl) and bn.momentum is not None and x.dim() >= 3)
    if fused:
        bn.__dict__["_mvster_stats_gen"] = bn.__dict__.get("_mvster_stats_gen", 0) + 1      # see _bn_eval_norm
        rm = bn.running_mean if bn.track_running_stats else None
        rv = bn.running_var if bn.track_running_stats else None
        if residual is not None and (not x.is_contiguous() or not residual.is_contiguous() or residual.shape != x.shape):
            return _BNReLUTrain.apply(x, bn.weight, bn.bias, rm, rv, bn.momentum, bn.eps, relu, groups) + residual
        return _BNReLUTrain.apply(x, bn.weight, bn.bias, rm, rv, bn.momentum, bn.eps, relu, groups, residual)
    if (_BN_IMPL == "hip" and not training and x.is_cuda and x.dtype == torch.float32 and (x.is_contiguous() or dense_cl)
            and x.dim() >= 3 and not (torch.is_grad_enabled() and (x.requires_grad or (bn.weight is not None and bn.weight.requires_grad)))):
        # inference: a per-channel affine with the running statistics (+ReLU, + the skip connection) in ONE streaming pass
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C)
        y = torch.empty_like(x)
        scratch = torch.empty(3 * C, dtype=torch.float32, device=x.device)
        opt = lambda t: t.data_ptr() if t is not None else None
        fuse_res = residual is not None and x.is_contiguous() and residual.is_contiguous() and residual.shape == x.shape
        with _on(x.device):
            if x.is_contiguous():
                _lib.call("mvster_bn_eval_fwd", x.data_ptr(), opt(bn.weight), opt(bn.bias), bn.running_mean.data_ptr(),
                          bn.running_var.data_ptr(), residual.data_ptr() if fuse_res else None, y.data_ptr(), scratch.data_ptr(),
                          N, C, HW, float(bn.eps), 1 if relu else 0, _stream(x.device))
            else:
                _lib.call("mvster_bn_eval_fwd_nhwc", x.data_ptr(), opt(bn.weight), opt(bn.bias), bn.running_mean.data_ptr(),
                          bn.running_var.data_ptr(), y.data_ptr(), scratch.data_ptr(), N * HW, C, float(bn.eps),
                          1 if relu else 0, _stream(x.device))
        return y if (residual is None or fuse_res) else y + residual
    if x.is_cuda and _BN_IMPL == "native":
        y = torch.native_batch_norm(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.momentum, bn.eps)[0]
    else:
        y = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps)
    y = F.relu(y, inplace=True) if relu else y
    return y if residual is None else y + residual


_BN_IMPL = "hip"             # "hip" (csrc/batchnorm.hip for training mode), "miopen" (F.batch_norm) or "native" (ATen)
_FUSED_EVAL = True           # inference (eval mode under no_grad): convolution -> BatchNorm -> ReLU (-> + skip) as ONE kernel
#                            # (the normalisation in the convolution's epilogue, csrc/conv_epilogue.h) and every derived
#                            # weight array (matrix-core packing, direct-kernel permutation, BatchNorm affine) built once per
#                            # parameter version instead of once per call


def _cached(owner, tag, sources, build, generation=0):
    """``build()`` once per version of the ``sources`` tensors; the result lives on the ``owner`` module (freed with it).
    In-place updates (optimizer steps, load_state_dict) bump ``_version``, ``.to()`` / ``.data =`` change ``data_ptr``."""
    key = (generation,) + tuple((t.data_ptr(), t._version) for t in sources if t is not None)
    store = owner.__dict__.setdefault("_mvster_cache", {})
    hit = store.get(tag)
    if hit is None or hit[0] != key:
        hit = (key, build())
        store[tag] = hit
    return hit[1]


def _bn_eval_norm(bn):
    """The coefficient array of mvster_bn_eval_prepare for an eval-mode BatchNorm module (a | b | mean, padded)."""
    def build():
        C = bn.num_features
        norm = torch.empty(_lib.load().mvster_bn_eval_norm_floats(C), dtype=torch.float32, device=bn.running_mean.device)
        opt = lambda t: t.data_ptr() if t is not None else None
        with _on(norm.device):
            _lib.call("mvster_bn_eval_prepare", opt(bn.weight), opt(bn.bias), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                      norm.data_ptr(), C, float(bn.eps), _stream(norm.device))
        return norm
    # (the training kernels update the running statistics through their pointers: no tensor version moves, hence the counter)
    return _cached(bn, "norm", (bn.weight, bn.bias, bn.running_mean, bn.running_var), build, bn.__dict__.get("_mvster_stats_gen", 0))


def _inference(bn, x):
    """The fused inference path applies: eval-mode BatchNorm with running statistics, no autograd, an NCHW fp32 HIP tensor."""
    return (_FUSED_EVAL and _BN_IMPL == "hip" and not bn.training and bn.track_running_stats and not torch.is_grad_enabled()
            and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous())


def _conv_bn_eval(x, owner, tag, weight2d, bias, stride, bn, relu, residual=None, depth=0):
    """relu?(bn(conv2d(x, weight2d(), bias, stride, padding K // 2))) [+ residual] as one kernel, or None when the layer has
    no HIP forward route (the caller then takes the unfused path).  ``weight2d``: () -> [Cout, Cin, K, K] view of
    ``owner.weight`` (only evaluated when the derived array has to be rebuilt); ``depth`` > 0: a (3,k,k) layer over depth
    slices stacked into the batch, weight2d() its folded [Cout, 3*C, k, k] weight, x [B*depth, C, H, W] read in place."""
    w = owner.weight
    N, Cx, H, W = x.shape
    if depth > 0:
        Cout, K = w.shape[0], w.shape[3]
        Cin = 3 * Cx
        impl = "mfma" if (_MFMA_CONV and _DEPTH_IN_PLACE and K == 3 and stride == 1 and Cout >= 16 and Cin <= _MFMA_MAX_WGRAD_CIN) else None
    else:
        Cout, K = w.shape[0], w.shape[-1]
        Cin = Cx
        route = _conv_route(Cin, Cout, K, stride)
        impl = route[0] if route is not None else None
    if impl not in ("mfma", "direct") or (residual is not None and not residual.is_contiguous()):
        return None
    dev = x.device
    norm = _bn_eval_norm(bn)
    P = K // 2
    Ho, Wo = (H + 2 * P - K) // stride + 1, (W + 2 * P - K) // stride + 1
    y = torch.empty(N, Cout, Ho, Wo, dtype=torch.float32, device=dev)
    if residual is not None and residual.shape != y.shape:
        return None
    opt = lambda t: t.data_ptr() if t is not None else None
    if impl == "mfma":
        def pack():
            w2 = weight2d().contiguous()
            need = _lib.load().mvster_conv_mfma_packed_floats(Cin, Cout, K, stride, 0)
            packed = torch.empty(need, dtype=torch.float32, device=dev)
            with _on(dev):
                _lib.call("mvster_conv_mfma_pack", w2.data_ptr(), packed.data_ptr(), need, Cin, Cout, K, stride, 0, 0, _stream(dev))
            return packed
        packed = _cached(owner, tag + ":mfma", (w,), pack)
        with _on(dev):
            _lib.call("mvster_conv_mfma_fwd_ep", x.data_ptr(), packed.data_ptr(), opt(bias), y.data_ptr(), N, Cin, Cout, H, W, K,
                      stride, 0, depth, norm.data_ptr(), opt(residual), 1 if relu else 0, _stream(dev))
    else:
        wt = _cached(owner, tag + ":direct", (w,), lambda: weight2d().permute(1, 2, 3, 0).contiguous())
        with _on(dev):
            _lib.call("mvster_conv_direct_fwd_ep", x.data_ptr(), wt.data_ptr(), opt(bias), y.data_ptr(), N, Cin, Cout, H, W, K,
                      _lib.LAYOUT_NCHW, norm.data_ptr(), opt(residual), 1 if relu else 0, _stream(dev))
    return y


class _ConvBNReLU2d(nn.Module):
    """conv -> BatchNorm2d -> ReLU; attributes ``conv``/``bn`` (models/mvs4net_utils.py:230-256, gn=False)."""

    def __init__(self, cin, cout, kernel, stride, padding):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(cout, momentum=0.1)

    def forward(self, x, groups=1):
        conv = self.conv
        if (_inference(self.bn, x) and conv.stride[0] == conv.stride[1] and conv.kernel_size[0] == conv.kernel_size[1]
                and conv.padding == (conv.kernel_size[0] // 2,) * 2):
            y = _conv_bn_eval(x, conv, "w", lambda: conv.weight, conv.bias, conv.stride[0], self.bn, True)
            if y is not None:
                return y
        return _batch_norm(self.bn, _conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding), relu=True, groups=groups)


def _conv3d_as_2d(conv, x2d, B, D):
    """A Conv3d whose kernel is (1,k,k) or (3,k,k) with unit depth stride, evaluated as ONE conv2d on depth slices
    stacked into the batch dimension, x2d = [B*D, Cin, H, W].  (3,k,k): the three depth-shifted copies of the input are
    concatenated along channels and the weight is folded to [Cout, 3*Cin, k, k].  MIOpen's conv3d weight-gradient path
    for these shapes is a naive kernel (79 ms per layer at DTU resolution); its 2-D paths are Winograd/implicit GEMM.
    On the matrix-core kernels the concatenation is virtual (``_MfmaDepthConv``: the three depth-shifted planes are read in
    place, forward and both gradients), which removes the copies and their backward."""
    w = conv.weight
    kd = w.shape[2]
    stride, pad = conv.stride[1:], conv.padding[1:]
    if kd == 1:
        return _conv2d(x2d, w[:, :, 0], conv.bias, stride, pad)
    assert kd == 3 and conv.stride[0] == 1 and conv.padding[0] == 1
    C, H, W = x2d.shape[1:]
    w2 = w.permute(0, 2, 1, 3, 4).reshape(w.shape[0], 3 * w.shape[1], w.shape[3], w.shape[4])
    k = w.shape[3]
    if (_MFMA_CONV and _DEPTH_IN_PLACE and x2d.is_cuda and x2d.dtype == torch.float32 and x2d.is_contiguous() and conv.bias is None
            and w.shape[3] == w.shape[4] and k == 3 and tuple(stride) == (1, 1) and tuple(pad) == (1, 1)
            and w.shape[0] >= 16 and 3 * C <= _MFMA_MAX_WGRAD_CIN):
        return _MfmaDepthConv.apply(x2d, w2, D)
    x5 = x2d.view(B, D, C, H, W)
    pad_slice = x5.new_zeros(B, 1, C, H, W)
    stacked = torch.cat([torch.cat([pad_slice, x5[:, :-1]], 1), x5, torch.cat([x5[:, 1:], pad_slice], 1)], 2)
    return _conv2d(stacked.view(B * D, 3 * C, H, W), w2, conv.bias, stride, pad)


class _ConvBNReLU3d(nn.Module):
    """conv3d -> BatchNorm3d -> ReLU; attributes ``conv``/``bn`` (models/mvs4net_utils.py:123-130)."""

    def __init__(self, cin, cout, kernel=3, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, kernel, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm3d(cout)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)

    def forward_slices(self, x2d, B, D):
        # BatchNorm3d over [B,C,D,H,W] == batch norm over [B*D,C,H,W]: same per-channel statistics, same parameters
        y = self._forward_slices_inference(x2d, B, D) if _inference(self.bn, x2d) else None
        return y if y is not None else _batch_norm(self.bn, _conv3d_as_2d(self.conv, x2d, B, D), relu=True)

    def _forward_slices_inference(self, x2d, B, D):
        """The layer as one kernel (BatchNorm + ReLU in the convolution's epilogue), the same case split as _conv3d_as_2d."""
        conv = self.conv
        w = conv.weight
        kd, k = w.shape[2], w.shape[3]
        stride, pad = conv.stride[1:], conv.padding[1:]
        if w.shape[3] != w.shape[4] or stride[0] != stride[1] or tuple(pad) != (k // 2, k // 2):
            return None
        if kd == 1:
            return _conv_bn_eval(x2d, conv, "w2d", lambda: w[:, :, 0], conv.bias, stride[0], self.bn, True)
        if not (kd == 3 and conv.stride[0] == 1 and conv.padding[0] == 1 and stride[0] == 1):
            return None
        fold = lambda: w.permute(0, 2, 1, 3, 4).reshape(w.shape[0], 3 * w.shape[1], k, k)
        y = _conv_bn_eval(x2d, conv, "w3", fold, conv.bias, 1, self.bn, True, depth=D) if conv.bias is None else None
        if y is not None:
            return y
        C, H, W = x2d.shape[1:]                # few output channels (conv0): the three depth-shifted copies, concatenated
        x5 = x2d.view(B, D, C, H, W)
        pad_slice = x5.new_zeros(B, 1, C, H, W)
        stacked = torch.cat([torch.cat([pad_slice, x5[:, :-1]], 1), x5, torch.cat([x5[:, 1:], pad_slice], 1)], 2)
        return _conv_bn_eval(stacked.view(B * D, 3 * C, H, W), conv, "w3cat", fold, conv.bias, 1, self.bn, True)


def _miopen_backward(dy, x, weight, stride, mask, bias_sizes=None):
    K = weight.shape[2]
    return torch.ops.aten.convolution_backward(dy, x, weight, bias_sizes, [stride, stride], [K // 2, K // 2], [1, 1], False,
                                               [0, 0], 1, mask)


def _mfma_workspace(lib, dev, N, Cin, Cout, H, W, K, stride, op):
    need = lib.mvster_conv_mfma_workspace(N, Cin, Cout, H, W, K, stride, op)
    return torch.empty(need, dtype=torch.float32, device=dev), need


class _DirectConv2d(torch.autograd.Function):
    """"Same"-padded convolution (padding K // 2) whose three operations are routed separately (``route`` = implementation
    of forward, input gradient, weight gradient; each "direct", "mfma" or "miopen"):
    direct -- csrc/conv_direct.hip: stride 1, K = 1 or 3, few channels; forward, input gradient (the same kernel on dy with
              the flipped, transposed weight) and weight / bias gradient (tile partial sums, no atomics);
    mfma   -- csrc/conv_mfma.hip: implicit GEMMs on the fp32 matrix cores, (K, stride) in (1,1), (3,1), (3,2), (5,2);
              forward, input gradient of the stride-1 layers, weight gradient;
    miopen -- ATen's convolution / convolution_backward.
    ``route`` None: the direct kernels with their own per-operation policy (the few-channel stride-1 layers)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride=1, route=None, out_channels_last=False):
        """``out_channels_last``: y [N,Cout,Ho,Wo] is stored as [N,Ho,Wo,Cout] (torch.channels_last) by the kernel itself --
        FPN4's output layers hand the cost-volume kernels the layout they gather from (HIP routes, Cout % 4 == 0)."""
        N, Cin, H, W = x.shape
        Cout, _, K, _ = weight.shape
        impl = "direct" if route is None else route[0]
        dev = x.device
        fmt = torch.channels_last if out_channels_last else torch.contiguous_format
        if impl == "miopen":
            y = F.conv2d(x, weight, bias, stride, K // 2).contiguous(memory_format=fmt)
        elif impl == "mfma":
            Ho, Wo = (H + 2 * (K // 2) - K) // stride + 1, (W + 2 * (K // 2) - K) // stride + 1
            y = torch.empty((N, Cout, Ho, Wo), dtype=torch.float32, device=dev, memory_format=fmt)
            ws, need = _mfma_workspace(_lib.load(), dev, N, Cin, Cout, H, W, K, stride, 0)
            with _on(dev):
                _lib.call("mvster_conv_mfma_fwd", x.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                          y.data_ptr(), ws.data_ptr(), need, N, Cin, Cout, H, W, K, stride, 2 if out_channels_last else 0,
                          _stream(dev))
        else:
            y = torch.empty((N, Cout, H, W), dtype=torch.float32, device=dev, memory_format=fmt)
            wt = weight.detach().permute(1, 2, 3, 0).contiguous()                  # [Cin, K, K, Cout]
            with _on(dev):
                _lib.call("mvster_conv_direct_fwd_layout", x.data_ptr(), wt.data_ptr(),
                          bias.data_ptr() if bias is not None else None, y.data_ptr(), N, Cin, Cout, H, W, K,
                          _lib.LAYOUT_NHWC if out_channels_last else _lib.LAYOUT_NCHW, _stream(dev))
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.stride = stride
        ctx.route = route
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        N, Cin, H, W = x.shape
        Cout, _, K, _ = weight.shape
        stride, route = ctx.stride, ctx.route
        dy = dy.contiguous()
        dev = x.device
        dx = dweight = dbias = None
        lib = _lib.load()
        if route is None:
            dgrad = "direct" if Cin <= _DIRECT_MAX_DGRAD_CIN and lib.mvster_conv_direct_supported(Cout, Cin, K) else "miopen"
            wgrad = "direct" if (Cout <= 8 or Cin * Cout <= _DIRECT_MAX_WGRAD_PRODUCT) and Cout in (1, 8, 16) else "miopen"
        else:
            dgrad, wgrad = route[1], route[2]
        with _on(dev):
            if ctx.needs_input_grad[0]:
                if dgrad == "direct":
                    wt = weight.detach().flip(2, 3).permute(0, 2, 3, 1).contiguous()       # [Cout, K, K, Cin]
                    dx = torch.empty_like(x)
                    _lib.call("mvster_conv_direct_fwd", dy.data_ptr(), wt.data_ptr(), None, dx.data_ptr(), N, Cout, Cin, H, W, K,
                              _stream(dev))
                elif dgrad == "mfma" and stride == 1:
                    dx = torch.empty_like(x)
                    ws, need = _mfma_workspace(lib, dev, N, Cout, Cin, H, W, K, 1, 0)
                    _lib.call("mvster_conv_mfma_fwd", dy.data_ptr(), weight.data_ptr(), None, dx.data_ptr(), ws.data_ptr(), need,
                              N, Cout, Cin, H, W, K, 1, 1, _stream(dev))
                elif dgrad == "mfma":                      # stride 2: dy scattered back = a transposed convolution, as a gather
                    dx = torch.empty_like(x)
                    Hd, Wd = dy.shape[2], dy.shape[3]
                    ws, need = _mfma_workspace(lib, dev, N, Cout, Cin, Hd, Wd, K, 2, 2)
                    _lib.call("mvster_conv_mfma_transposed", dy.data_ptr(), weight.data_ptr(), dx.data_ptr(), ws.data_ptr(), need,
                              N, Cout, Cin, Hd, Wd, H, W, K, _stream(dev))
                else:
                    dx = _miopen_backward(dy, x, weight, stride, [True, False, False])[0]
            want_bias = ctx.has_bias and ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1] or want_bias:
                if wgrad == "direct":
                    need = lib.mvster_conv_direct_wgrad_workspace(N, Cin, Cout, H, W, K)
                    ws = torch.empty(need, dtype=torch.float32, device=dev)
                    dweight = torch.empty_like(weight)
                    dbias = torch.empty(Cout, dtype=torch.float32, device=dev) if ctx.has_bias else None
                    _lib.call("mvster_conv_direct_wgrad", x.data_ptr(), dy.data_ptr(), dweight.data_ptr(),
                              dbias.data_ptr() if dbias is not None else None, ws.data_ptr(), need, N, Cin, Cout, H, W, K,
                              _stream(dev))
                elif wgrad == "mfma":
                    dweight = torch.empty_like(weight)
                    ws, need = _mfma_workspace(lib, dev, N, Cin, Cout, H, W, K, stride, 1)
                    _lib.call("mvster_conv_mfma_wgrad", x.data_ptr(), dy.data_ptr(), dweight.data_ptr(), ws.data_ptr(), need,
                              N, Cin, Cout, H, W, K, stride, _stream(dev))
                    dbias = dy.sum((0, 2, 3)) if want_bias else None
                else:
                    _, dweight, dbias = _miopen_backward(dy, x, weight, stride, [False, True, ctx.has_bias],
                                                         [Cout] if ctx.has_bias else None)
        return dx, dweight, dbias, None, None, None


class _MfmaDepthConv(torch.autograd.Function):
    """A (3,k,k) Conv3d (unit strides, padding (1,k//2,k//2), no bias) on depth slices stacked into the batch,
    x [B*D,C,H,W], weight folded to [Cout,3C,k,k]: the matrix-core kernels read the depth-shifted planes in place
    (csrc/conv_mfma.hip ``input_plane``) -- forward, input gradient and weight gradient, no concatenated copy either way."""

    @staticmethod
    def forward(ctx, x, w2, depth):
        N, C, H, W = x.shape
        Cout, K = w2.shape[0], w2.shape[2]
        dev = x.device
        y = torch.empty(N, Cout, H, W, dtype=torch.float32, device=dev)
        ws, need = _mfma_workspace(_lib.load(), dev, N, 3 * C, Cout, H, W, K, 1, 0)
        with _on(dev):
            _lib.call("mvster_conv_mfma_depth3_fwd", x.data_ptr(), w2.data_ptr(), None, y.data_ptr(), ws.data_ptr(), need,
                      N, C, Cout, H, W, K, depth, 0, _stream(dev))
        ctx.save_for_backward(x, w2)
        ctx.depth = depth
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        N, C, H, W = x.shape
        Cout, K = w2.shape[0], w2.shape[2]
        dy = dy.contiguous()
        dev = x.device
        lib = _lib.load()
        dx = dw = None
        with _on(dev):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                ws, need = _mfma_workspace(lib, dev, N, 3 * Cout, C, H, W, K, 1, 0)
                _lib.call("mvster_conv_mfma_depth3_fwd", dy.data_ptr(), w2.data_ptr(), None, dx.data_ptr(), ws.data_ptr(), need,
                          N, C, Cout, H, W, K, ctx.depth, 1, _stream(dev))
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(w2)
                ws, need = _mfma_workspace(lib, dev, N, 3 * C, Cout, H, W, K, 1, 1)
                _lib.call("mvster_conv_mfma_depth3_wgrad", x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), need,
                          N, C, Cout, H, W, K, ctx.depth, _stream(dev))
        return dx, dw, None


class _MfmaConvTranspose2d(torch.autograd.Function):
    """conv_transpose2d(x, weight [Cin,Cout,3,3], stride 2, padding 1, output_padding 1) on the matrix cores
    (csrc/conv_mfma.hip): forward = the gather-form transposed kernel, input gradient = the stride-2 forward kernel on dy,
    weight gradient = the weight-gradient kernel with the roles of x and dy swapped."""

    @staticmethod
    def forward(ctx, x, weight):
        N, Cin, H, W = x.shape
        Cout, K = weight.shape[1], weight.shape[2]
        dev = x.device
        y = torch.empty(N, Cout, 2 * H, 2 * W, dtype=torch.float32, device=dev)
        ws, need = _mfma_workspace(_lib.load(), dev, N, Cin, Cout, H, W, K, 2, 2)
        with _on(dev):
            _lib.call("mvster_conv_mfma_transposed", x.data_ptr(), weight.data_ptr(), y.data_ptr(), ws.data_ptr(), need,
                      N, Cin, Cout, H, W, 2 * H, 2 * W, K, _stream(dev))
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        N, Cin, H, W = x.shape
        Cout, K = weight.shape[1], weight.shape[2]
        dy = dy.contiguous()
        dev = x.device
        lib = _lib.load()
        dx = dweight = None
        with _on(dev):
            if ctx.needs_input_grad[0]:                    # a stride-2 convolution of dy with the same [Cin,Cout,K,K] array
                dx = torch.empty_like(x)
                ws, need = _mfma_workspace(lib, dev, N, Cout, Cin, 2 * H, 2 * W, K, 2, 0)
                _lib.call("mvster_conv_mfma_fwd", dy.data_ptr(), weight.data_ptr(), None, dx.data_ptr(), ws.data_ptr(), need,
                          N, Cout, Cin, 2 * H, 2 * W, K, 2, 0, _stream(dev))
            if ctx.needs_input_grad[1]:
                dweight = torch.empty_like(weight)
                ws, need = _mfma_workspace(lib, dev, N, Cout, Cin, 2 * H, 2 * W, K, 2, 1)
                _lib.call("mvster_conv_mfma_wgrad", dy.data_ptr(), x.data_ptr(), dweight.data_ptr(), ws.data_ptr(), need,
                          N, Cout, Cin, 2 * H, 2 * W, K, 2, _stream(dev))
        return dx, dweight


def _conv_transpose2d_x2(x, weight):
    """F.conv_transpose2d(x, weight, stride=2, padding=1, output_padding=1) for a 3x3 kernel: reg2d's up-convolutions."""
    if (_MFMA_CONV and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and weight.is_contiguous()
            and tuple(weight.shape[2:]) == (3, 3)):
        return _MfmaConvTranspose2d.apply(x, weight)
    return F.conv_transpose2d(x, weight, None, stride=2, padding=1, output_padding=1)


_DIRECT_CONV = True          # few-channel stride-1 layers through csrc/conv_direct.hip; False: the other two
_MFMA_CONV = True            # many-channel layers through csrc/conv_mfma.hip; False: MIOpen
_DEPTH_IN_PLACE = True       # (3,k,k) layers read their depth-shifted planes in place (needs _MFMA_CONV); False: concatenate
# Routing policy: every layer's three operations are routed separately on measured merit (tools/probes/conv_route_probe.py:
# DTU resolution, the five views of FPN4 in one batch / the four depth slices of reg2d stage 4; ms direct / mfma / MIOpen):
#   forward   8->8 0.29/0.50/0.83, 16->16 0.17/0.24/0.30, 64->8 1.55/2.75/2.97, 64->16 0.68/0.72/0.86, 48->16 0.41/0.46/0.50;
#             32->32 -/0.13/0.15, 64->32 -/0.23/0.24, 96->32 -/0.25/0.28, 192->64 -/0.24/0.29, 1x1 64->64 -/0.04/0.07,
#             stride 2: 8->16 -/0.16/0.26, 32->64 -/0.06/0.12, 5x5 8->16 -/0.37/0.52, 5x5 32->64 -/0.19/0.22
#   input gradient  8->8 0.29/0.51/0.87, 16->16 0.19/0.25/0.34, 64->8 1.55/1.89/1.62, 64->16 0.97/0.58/0.62,
#             32->32 -/0.16/0.15-0.95, 64->32 -/0.26/0.32, 96->32 -/0.28/0.37 (stride 2: MIOpen, no HIP kernel yet)
#   weight gradient 3->8 0.56/0.36/0.87, 8->8 0.56/0.38/1.39, 64->8 3.27/1.78/5.64, 16->16 0.73/0.32/0.53, 64->16 2.47/0.89/1.54,
#             32->32 -/0.19/0.31, 64->32 -/0.31/1.15, 96->32 -/0.31/0.53, 5x5 s2 16->32 -/0.28/0.44; but 192->64 -/0.82/0.41
# direct: one lane per pixel, Cout accumulators (few channels: bound by memory, which it streams best); mfma: implicit GEMM
# on the fp32 matrix cores (80-110 TFLOP/s on the many-channel layers); MIOpen keeps what it still does better.
_DIRECT_MAX_COUT = 16                # forward: few output channels ...
_DIRECT_MAX_PRODUCT = 1024           # ... and Cin * Cout up to 64 x 16 (two pixels per lane from 16 accumulators up)
_DIRECT_MAX_DGRAD_CIN = 16           # input gradient: Cin accumulators per lane (more when the layer has <= 8 output channels)
_DIRECT_MAX_WGRAD_PRODUCT = 128      # weight gradient: Cout <= 8, or Cin * Cout up to this
_DIRECT_WGRAD_COUT = (1, 8, 16)      # output channel counts mvster_conv_direct_wgrad instantiates (csrc/conv_direct.hip)
_MFMA_MAX_WGRAD_CIN = 192            # weight gradient on the matrix cores up to reg2d's conv6 (3 x 64 virtual input channels): slower
#                                    # than MIOpen there (0.82 vs 0.39 ms) but the model then runs without a single MIOpen
#                                    # kernel, so the reference drivers' cudnn.benchmark = True triggers no MIOpen find


def _conv_route(Cin, Cout, K, stride):
    """(forward, input gradient, weight gradient) implementations for a "same"-padded layer, or None for plain F.conv2d."""
    lib = _lib.load()
    direct_ok = _DIRECT_CONV and stride == 1 and K in (1, 3) and Cout in (1, 3, 4, 8, 16)
    mfma_ok = _MFMA_CONV and bool(lib.mvster_conv_mfma_supported(Cin, Cout, K, stride)) and Cout >= 16 and Cin >= 8
    if direct_ok and Cin * Cout <= _DIRECT_MAX_PRODUCT:
        fwd = "direct"
    else:
        fwd = "mfma" if mfma_ok else "miopen"
    if direct_ok and (Cin <= _DIRECT_MAX_DGRAD_CIN or Cout <= 8) and lib.mvster_conv_direct_supported(Cout, Cin, K):
        dgrad = "direct"
    else:
        dgrad = "mfma" if mfma_ok and (stride == 1 or K in (3, 5)) else "miopen"
    if _MFMA_CONV and K == 3 and stride == 1 and Cout in (4, 8) and Cin >= 3:
        wgrad = "mfma"                 # <= 8 output channels: the tap rows packed next to them on the MFMA's column axis
    elif direct_ok and Cout in _DIRECT_WGRAD_COUT and (Cout <= 8 or Cin * Cout <= _DIRECT_MAX_WGRAD_PRODUCT):
        wgrad = "direct"
    else:
        wgrad = "mfma" if mfma_ok and Cin <= _MFMA_MAX_WGRAD_CIN else "miopen"
    route = (fwd, dgrad, wgrad)
    return None if route == ("miopen", "miopen", "miopen") else route


def _conv2d(x, weight, bias=None, stride=1, padding=0, out_channels_last=False):
    """F.conv2d, routed per operation to the HIP kernels for the layers they are built for.  ``out_channels_last``: the
    HIP forward kernels store the result as torch.channels_last (ignored on the stock route and when Cout % 4 != 0)."""
    stride = (stride, stride) if isinstance(stride, int) else tuple(stride)
    padding = (padding, padding) if isinstance(padding, int) else tuple(padding)
    K = weight.shape[2]
    if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and weight.is_contiguous()
            and weight.shape[2] == weight.shape[3] and stride[0] == stride[1] and padding == (K // 2, K // 2)):
        route = _conv_route(weight.shape[1], weight.shape[0], K, stride[0])
        if route is not None:
            return _DirectConv2d.apply(x, weight, bias, stride[0], route,
                                       bool(out_channels_last) and weight.shape[0] % 4 == 0 and route[0] != "miopen")
    return F.conv2d(x, weight, bias, stride, padding)


class _TopDownMerge(torch.autograd.Function):
    """y = bilinear_x2(feat, align_corners=True) + conv1x1(skip; weight, bias), one HIP kernel forward and three streaming
    passes backward (csrc/fpn_merge.hip) instead of ATen's upsample / atomic upsample backward, a MIOpen 1x1 convolution
    with its gradient kernels and transposes, a bias reduction and the add -- models/mvs4net_utils.py:490-505."""

    @staticmethod
    def forward(ctx, feat, skip, weight, bias):
        N, Ct, h, w = feat.shape
        Cs = skip.shape[1]
        y = torch.empty(N, Ct, 2 * h, 2 * w, dtype=torch.float32, device=feat.device)
        with _on(feat.device):
            _lib.call("mvster_fpn_merge_fwd", feat.data_ptr(), skip.data_ptr(), weight.data_ptr(),
                      bias.data_ptr() if bias is not None else None, y.data_ptr(), N, Ct, Cs, h, w, _stream(feat.device))
        ctx.save_for_backward(skip, weight)
        ctx.has_bias = bias is not None
        ctx.lowres = (h, w)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        skip, weight = ctx.saved_tensors
        dy = dy.contiguous()
        N, Ct = dy.shape[0], dy.shape[1]
        Cs = skip.shape[1]
        h, w = ctx.lowres
        dfeat = torch.empty(N, Ct, h, w, dtype=torch.float32, device=dy.device)
        dskip = torch.empty_like(skip)
        dweight = torch.empty_like(weight)
        dbias = torch.empty(Ct, dtype=torch.float32, device=dy.device) if ctx.has_bias else None
        with _on(dy.device):
            _lib.call("mvster_fpn_merge_bwd", dy.data_ptr(), skip.data_ptr(), weight.data_ptr(), dfeat.data_ptr(),
                      dskip.data_ptr(), dweight.data_ptr(), dbias.data_ptr() if dbias is not None else None, N, Ct, Cs, h, w,
                      _stream(dy.device))
        return dfeat, dskip, dweight, dbias


_FUSED_MERGE = True          # FPN4 top-down path through csrc/fpn_merge.hip (HIP fp32 NCHW tensors); False: torch operators


def _top_down_merge(feat, skip, lateral):
    """``F.interpolate(feat, x2, bilinear, align_corners=True) + lateral(skip)`` for a 1x1 ``lateral`` convolution."""
    fused = (_FUSED_MERGE and feat.is_cuda and feat.dtype == torch.float32 and feat.is_contiguous() and skip.is_contiguous()
             and feat.shape[1] <= 64 and skip.shape[1] in (8, 16, 32) and feat.shape[2] > 1 and feat.shape[3] > 1
             and tuple(skip.shape[2:]) == (2 * feat.shape[2], 2 * feat.shape[3]) and lateral.kernel_size == (1, 1))
    if not fused:
        return F.interpolate(feat, scale_factor=2, mode="bilinear", align_corners=True) + lateral(skip)
    return _TopDownMerge.apply(feat, skip, lateral.weight, lateral.bias)


class _HeadGather(torch.autograd.Function):
    """out = base + sum over the nine taps of the shifted, x2-upsampled planes of g (csrc/fpn_merge.hip, head gather)."""

    @staticmethod
    def forward(ctx, g, base, out_channels_last=False):
        N, _, h, w = g.shape
        out = torch.empty(base.shape, dtype=torch.float32, device=base.device,
                          memory_format=torch.channels_last if out_channels_last else torch.contiguous_format)
        with _on(g.device):
            _lib.call("mvster_fpn_head_gather_fwd_layout", g.data_ptr(), base.data_ptr(), out.data_ptr(), N, h, w,
                      _lib.LAYOUT_NHWC if out_channels_last else _lib.LAYOUT_NCHW, _stream(g.device))
        ctx.lowres = (N, h, w)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        N, h, w = ctx.lowres
        dy = dy.contiguous()
        dg = None
        if ctx.needs_input_grad[0]:
            dg = torch.empty(N, 72, h, w, dtype=torch.float32, device=dy.device)
            with _on(dy.device):
                _lib.call("mvster_fpn_head_gather_bwd", dy.data_ptr(), dg.data_ptr(), N, h, w, _stream(dy.device))
        return dg, dy, None


_FUSED_HEAD = True           # FPN4's last level without its 64-channel full-resolution map (needs the HIP convolutions)
_FPN_EMITS_CHANNELS_LAST = True   # out1 .. out4 store their maps as torch.channels_last (the kernels' own epilogues): the
#                                 # cost-volume kernels gather from them in place, no nchw_to_nhwc pass per stage; the
#                                 # gradients come back as planes (hotpath._FusedCostVolume), which is what the layers'
#                                 # backward kernels read


def _top_level_head(feat, skip, lateral, head, out_channels_last=False):
    """``head(F.interpolate(feat, x2, bilinear, align_corners=True) + lateral(skip))`` for a 1x1 ``lateral`` and a 3x3
    bias-free ``head`` with 8 output channels, when nothing else needs the merged map (FPN4's last level,
    models/mvs4net_utils.py:500-505).  Both steps are linear: the head's channel contraction is applied to ``feat`` at LOW
    resolution (a 1x1 layer with 8*9 outputs, the lateral bias folded in: upsampling preserves constants), to ``skip`` as
    one 8 -> 8 3x3 layer with the composed weights, and a gather kernel adds the nine shifted upsampled planes.  The
    [N,64,H,W] map (472 MB per view at DTU resolution) and the passes over it never exist; gradients reach the original
    parameters through the (tiny) weight compositions below."""
    w4 = head.weight                                                       # [8, C, 3, 3]
    wi = lateral.weight[:, :, 0, 0]                                        # [C, Cs]
    wg = w4.permute(0, 2, 3, 1).reshape(72, w4.shape[1], 1, 1)             # plane co*9 + ky*3 + kx
    bg = (w4 * lateral.bias[None, :, None, None]).sum(1).reshape(72) if lateral.bias is not None else None
    wc = torch.einsum("ocyx,ci->oiyx", w4, wi).contiguous()                # [8, Cs, 3, 3]
    g = _conv2d(feat, wg.contiguous(), bg, 1, 0)
    base = _conv2d(skip, wc, None, 1, 1)
    return _HeadGather.apply(g, base, out_channels_last)


def _pyramid_level(cin, cout, downsample):
    first = _ConvBNReLU2d(cin, cout, 5, 2, 2) if downsample else _ConvBNReLU2d(cin, cout, 3, 1, 1)
    rest = [_ConvBNReLU2d(cout, cout, 3, 1, 1) for _ in range(2 if downsample else 1)]
    return nn.Sequential(first, *rest)


class FPN4(nn.Module):
    """4-level feature pyramid: 8b/4b/2b/b channels at 1/8, 1/4, 1/2, 1/1 resolution
    (models/mvs4net_utils.py:426-509; gn=False, dcn=False -- the only reachable configuration)."""

    def __init__(self, base_channels, gn=False, dcn=False):
        super().__init__()
        if gn or dcn:
            raise NotImplementedError("FPN4 with GroupNorm / deformable convolution is not reachable from the reference "
                                      "scripts (DeformConvPack is not vendored upstream)")
        b = base_channels
        self.base_channels = b
        self.conv0 = _pyramid_level(3, b, downsample=False)
        self.conv1 = _pyramid_level(b, 2 * b, downsample=True)
        self.conv2 = _pyramid_level(2 * b, 4 * b, downsample=True)
        self.conv3 = _pyramid_level(4 * b, 8 * b, downsample=True)
        top = 8 * b
        self.inner1 = nn.Conv2d(4 * b, top, 1, bias=True)
        self.inner2 = nn.Conv2d(2 * b, top, 1, bias=True)
        self.inner3 = nn.Conv2d(b, top, 1, bias=True)
        self.out1 = nn.Conv2d(top, 8 * b, 1, bias=False)
        self.out2 = nn.Conv2d(top, 4 * b, 3, padding=1, bias=False)
        self.out3 = nn.Conv2d(top, 2 * b, 3, padding=1, bias=False)
        self.out4 = nn.Conv2d(top, b, 3, padding=1, bias=False)
        self.dcn = False
        self.out_channels = [8 * b, 4 * b, 2 * b, b]

    def forward(self, x, groups=1):
        """``groups`` > 1: ``x`` stacks that many views' batches along dim 0 and every BatchNorm normalises each view's
        run with its own statistics -- one pass equal to the reference's one pass per view (models/MVS4Net.py:88-93)."""
        skips = []
        for level in (self.conv0, self.conv1, self.conv2, self.conv3):
            for block in level:
                x = block(x, groups)
            skips.append(x)
        feat = skips[3]
        cl = _FPN_EMITS_CHANNELS_LAST and x.is_cuda
        head_conv = lambda m, t: _conv2d(t, m.weight, m.bias, m.stride, m.padding, out_channels_last=cl)
        outs = {"stage1": head_conv(self.out1, feat)}
        for k, (lateral, head) in enumerate(((self.inner1, self.out2), (self.inner2, self.out3), (self.inner3, self.out4))):
            skip = skips[2 - k]
            if (k == 2 and _FUSED_HEAD and _FUSED_MERGE and _MFMA_CONV and _DIRECT_CONV and feat.is_cuda
                    and feat.dtype == torch.float32 and feat.is_contiguous() and skip.is_contiguous() and head.bias is None
                    and tuple(head.weight.shape[:1] + head.weight.shape[2:]) == (8, 3, 3) and feat.shape[2] > 1 and feat.shape[3] > 1
                    and tuple(skip.shape[2:]) == (2 * feat.shape[2], 2 * feat.shape[3])):
                outs["stage4"] = _top_level_head(feat, skip, lateral, head, cl)  # the merged map is not needed any further
                break
            feat = _top_down_merge(feat, skip, lateral)
            outs["stage%d" % (k + 2)] = head_conv(head, feat)
        return outs


def _up3d(cin, cout):
    return nn.Sequential(
        nn.ConvTranspose3d(cin, cout, kernel_size=(1, 3, 3), padding=(0, 1, 1), output_padding=(0, 1, 1),
                           stride=(1, 2, 2), bias=False),
        nn.BatchNorm3d(cout),
        nn.ReLU(inplace=True))


class reg2d(nn.Module):
    """U-Net over [B,G,D,H,W] with (1,3,3) kernels -> logits [B,D,H,W] (models/mvs4net_utils.py:884-926)."""

    def __init__(self, input_channel=128, base_channel=32, conv_name="ConvBnReLU3D"):
        super().__init__()
        if conv_name != "ConvBnReLU3D":
            raise NotImplementedError("aggregation block %r (attention variants of the reference) is out of scope" % conv_name)
        b = base_channel
        flat = dict(kernel=(1, 3, 3), pad=(0, 1, 1))
        self.conv0 = _ConvBNReLU3d(input_channel, b, **flat)
        self.conv1 = _ConvBNReLU3d(b, 2 * b, stride=(1, 2, 2), **flat)
        self.conv2 = _ConvBNReLU3d(2 * b, 2 * b)
        self.conv3 = _ConvBNReLU3d(2 * b, 4 * b, stride=(1, 2, 2), **flat)
        self.conv4 = _ConvBNReLU3d(4 * b, 4 * b)
        self.conv5 = _ConvBNReLU3d(4 * b, 8 * b, stride=(1, 2, 2), **flat)
        self.conv6 = _ConvBNReLU3d(8 * b, 8 * b)
        self.conv7 = _up3d(8 * b, 4 * b)
        self.conv9 = _up3d(4 * b, 2 * b)
        self.conv11 = _up3d(2 * b, b)
        self.prob = nn.Conv3d(8, 1, 1, stride=1, padding=0)      # hard-wired to 8 input channels upstream (:914)
        # True: evaluate the (1,3,3)/(3,3,3) convolutions as conv2d over depth slices (same parameters, same maths)
        self.depth_slices_as_batch = True

    def forward(self, x):
        return self.forward_slices(x) if self.depth_slices_as_batch else self.forward_conv3d(x)

    def forward_conv3d(self, x):
        """The reference's operator sequence (conv3d / ConvTranspose3d / BatchNorm3d), models/mvs4net_utils.py:916-926."""
        e0 = self.conv0(x)
        e1 = self.conv2(self.conv1(e0))
        e2 = self.conv4(self.conv3(e1))
        y = self.conv6(self.conv5(e2))
        y = e2 + self.conv7(y)
        y = e1 + self.conv9(y)
        y = e0 + self.conv11(y)
        return self.prob(y).squeeze(1)

    @staticmethod
    def _up(seq, x2d, skip=None):
        """up-convolution -> BatchNorm -> ReLU (+ ``skip``: ``conv_k + self.conv_j(x)``, models/mvs4net_utils.py:921-923)"""
        deconv, bn = seq[0], seq[1]
        w = deconv.weight
        if (_inference(bn, x2d) and _MFMA_CONV and deconv.stride[1:] == (2, 2) and deconv.padding[1:] == (1, 1)
                and deconv.output_padding[1:] == (1, 1) and tuple(w.shape[2:]) == (1, 3, 3) and deconv.bias is None
                and (skip is None or skip.is_contiguous())):
            # inference: up-convolution, BatchNorm, ReLU and the skip connection in one kernel, weights packed once
            N, Cin, H, W = x2d.shape
            Cout, dev = w.shape[1], x2d.device
            y = torch.empty(N, Cout, 2 * H, 2 * W, dtype=torch.float32, device=dev)
            if skip is None or skip.shape == y.shape:
                def pack():
                    w2 = w[:, :, 0].contiguous()
                    need = _lib.load().mvster_conv_mfma_packed_floats(Cin, Cout, 3, 2, 2)
                    packed = torch.empty(need, dtype=torch.float32, device=dev)
                    with _on(dev):
                        _lib.call("mvster_conv_mfma_pack", w2.data_ptr(), packed.data_ptr(), need, Cin, Cout, 3, 2, 2, 2, _stream(dev))
                    return packed
                packed = _cached(deconv, "wT:mfma", (w,), pack)
                norm = _bn_eval_norm(bn)
                with _on(dev):
                    _lib.call("mvster_conv_mfma_transposed_ep", x2d.data_ptr(), packed.data_ptr(), y.data_ptr(), N, Cin, Cout, H, W,
                              2 * H, 2 * W, 3, norm.data_ptr(), skip.data_ptr() if skip is not None else None, 1, _stream(dev))
                return y
        if deconv.stride[1:] == (2, 2) and deconv.padding[1:] == (1, 1) and deconv.output_padding[1:] == (1, 1):
            y = _conv_transpose2d_x2(x2d, deconv.weight[:, :, 0].contiguous())
        else:
            y = F.conv_transpose2d(x2d, deconv.weight[:, :, 0], None, stride=deconv.stride[1:], padding=deconv.padding[1:],
                                   output_padding=deconv.output_padding[1:])
        return _batch_norm(bn, y, relu=True, residual=skip)

    def forward_slices(self, x):
        """Same network, same parameters, with the D depth slices stacked into the batch: every layer is a 2-D
        convolution on [B*D, C, H, W]."""
        B, C, D, H, W = x.shape
        # .contiguous(): with B = 1 the reshape is a strided VIEW, which would send conv0 down the generic F.conv2d path
        x2d = x.permute(0, 2, 1, 3, 4).reshape(B * D, C, H, W).contiguous()
        e0 = self.conv0.forward_slices(x2d, B, D)
        e1 = self.conv2.forward_slices(self.conv1.forward_slices(e0, B, D), B, D)
        e2 = self.conv4.forward_slices(self.conv3.forward_slices(e1, B, D), B, D)
        y = self.conv6.forward_slices(self.conv5.forward_slices(e2, B, D), B, D)
        y = self._up(self.conv7, y, e2)
        y = self._up(self.conv9, y, e1)
        y = self._up(self.conv11, y, e0)
        y = _conv2d(y, self.prob.weight[:, :, 0], self.prob.bias)
        return y.view(B, D, H, W)


class _PlanarCopy(torch.autograd.Function):
    """NCHW-contiguous copy of a channels_last map (csrc/layout.hip); the gradient passes through unchanged (autograd
    does not tie a gradient's strides to its tensor's).  The mono head reads the reference view's maps as planes while
    the cost-volume kernels gather from the channels-last originals: one view's copy instead of a conversion of all N."""

    @staticmethod
    def forward(ctx, t):
        return to_planar(t)

    @staticmethod
    def backward(ctx, dy):
        return dy


def _as_planes(t):
    if t.is_cuda and t.dim() == 4 and t.dtype == torch.float32 and not t.is_contiguous() \
            and t.is_contiguous(memory_format=torch.channels_last):
        return _PlanarCopy.apply(t)
    return t


class mono_depth_decoder(nn.Module):
    """Auxiliary monocular depth head on the reference-view features (models/mvs4net_utils.py:840-882)."""

    def __init__(self, mono_stg_itrpl):
        super().__init__()
        self.convblocks = nn.ModuleList([_ConvBNReLU2d(c, c // 2, 3, 1, 1) for c in (64, 32, 16)])
        self.conv3x3 = nn.ModuleList([nn.Conv2d(c, 1, 3, 1, 1) for c in (64, 32, 16)])
        self.sigmoid = nn.Sigmoid()
        self.interpol = mono_stg_itrpl

    def forward(self, outputs, d_min, d_max, debug=0):
        lo = (1 / d_max)[:, None, None, None]
        hi = (1 / d_min)[:, None, None, None]
        feats = {i: _as_planes(outputs["stage%d" % i]["mono_feat"]) for i in (1, 2, 3, 4)}
        for i in (1, 2, 3):
            coarse = self.convblocks[i - 1](feats[i])
            coarse = F.interpolate(coarse, scale_factor=2, mode=self.interpol)
            fine = feats[i + 1]
            last = self.conv3x3[i - 1]
            disp = self.sigmoid(_conv2d(torch.cat([coarse, fine], 1), last.weight, last.bias, last.stride, last.padding))
            outputs["stage%d" % (i + 1)]["mono_depth"] = (1 / (lo + (hi - lo) * disp)).squeeze(1)
        return outputs


# --------------------------------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------------------------------
class MVS4net(nn.Module):
    def __init__(self, arch_mode="fpn", reg_net='reg2d', num_stage=4, fpn_base_channel=8,
                 reg_channel=8, stage_splits=[8, 8, 4, 4], depth_interals_ratio=[0.5, 0.5, 0.5, 1],
                 group_cor=False, group_cor_dim=[8, 8, 8, 8],
                 inverse_depth=False,
                 agg_type='ConvBnReLU3D',
                 dcn=False,
                 pos_enc=0,
                 mono=False,
                 mono_stg_itrpl="nearest",
                 asff=False,
                 attn_temp=2,
                 attn_fuse_d=True,
                 vis_ETA=False,
                 vis_stg_features=False,
                 debug=0
                 ):
        super(MVS4net, self).__init__()
        if arch_mode != "fpn":
            raise NotImplementedError("only arch_mode='fpn' builds a feature net upstream (models/MVS4Net.py:45-46)")
        if reg_net != "reg2d":
            raise NotImplementedError("reg_net=%r: only reg2d (every reference script) is provided; reg3d is out of scope" % reg_net)
        if asff:
            raise NotImplementedError("ASFF feature fusion is out of scope (SURVEY.md section 2 row 6)")
        if debug:
            raise NotImplementedError("debug bit-masks open interactive cv2 windows upstream; not supported")
        self.arch_mode = arch_mode
        self.num_stage = num_stage
        self.depth_interals_ratio = depth_interals_ratio
        self.group_cor = group_cor
        self.group_cor_dim = group_cor_dim
        self.inverse_depth = inverse_depth
        self.asff = False
        # registration order and construction order follow models/MVS4Net.py:42-67 so that a seeded construction
        # reproduces the reference's initial weights and state_dict key order
        self.attn_ob = nn.ModuleList()
        self.feature = FPN4(base_channels=fpn_base_channel, gn=False, dcn=dcn)
        self.vis_stg_features = vis_stg_features
        self.stagenet = stagenet(inverse_depth, mono, attn_fuse_d, vis_ETA, attn_temp, debug=debug)
        self.stage_splits = stage_splits
        self.reg = nn.ModuleList()
        self.pos_enc = pos_enc
        self.pos_enc_func = nn.ModuleList()          # never populated upstream: --pos_enc is a no-op
        self.mono = mono
        self.debug = debug
        if self.mono:
            self.mono_depth_decoder = mono_depth_decoder(mono_stg_itrpl)
        for idx in range(num_stage):
            in_dim = group_cor_dim[idx] if self.group_cor else self.feature.out_channels[idx]
            self.reg.append(reg2d(input_channel=in_dim, base_channel=reg_channel, conv_name=agg_type))
        # torch.channels_last keeps the FPN outputs in the layout the fused kernels gather from (zero-copy);
        # off by default: identical numerics either way, the conversion kernels cost ~1 % of a step
        self.channels_last_features = False
        self.batch_views_in_eval = True
        self.batch_views_in_train = True

    def forward(self, imgs, proj_matrices, depth_values, filename=None):
        # step 1. feature extraction (stock PyTorch)
        features = []
        batched = self.batch_views_in_eval if not self.training else (self.batch_views_in_train and not self.channels_last_features)
        if batched and len({tuple(i.shape) for i in imgs}) == 1:
            # one FPN pass over [N*B,3,H,W] instead of N passes.  Eval mode: BatchNorm uses its running statistics, so the
            # views are independent samples of one batch.  Training mode: the reference normalises every view with its own
            # batch statistics (one pass per view, models/MVS4Net.py:88-93) -- the BatchNorm kernels do the same inside the
            # one pass (`groups`: statistics per run of B samples, running statistics updated view by view)
            stacked = torch.cat(list(imgs), 0)
            if self.channels_last_features:
                stacked = stacked.contiguous(memory_format=torch.channels_last)
            B = imgs[0].shape[0]
            pyramid = {k: v.split(B, 0) for k, v in self.feature(stacked, groups=len(imgs)).items()}
            features = [{k: v[i] for k, v in pyramid.items()} for i in range(len(imgs))]
        else:
            for img in imgs:
                if self.channels_last_features:
                    img = img.contiguous(memory_format=torch.channels_last)
                features.append(self.feature(img))
        device, dtype = imgs[0].device, imgs[0].dtype

        # step 2. coarse-to-fine stages (HIP hot path)
        outputs = {}
        outputs_stage = None
        for stage_idx in range(self.num_stage):
            key = "stage{}".format(stage_idx + 1)
            features_stage = [feat[key] for feat in features]
            proj_matrices_stage = proj_matrices[key]
            B, C, H, W = features_stage[0].shape
            D = self.stage_splits[stage_idx]
            if stage_idx == 0:
                if self.inverse_depth:
                    depth_hypo = init_inverse_range(depth_values, D, device, dtype, H, W)
                else:
                    depth_hypo = init_range(depth_values, D, device, dtype, H, W)
            else:
                if not self.inverse_depth:
                    # upstream calls schedule_range, which raises AttributeError (ndepth.device on an int,
                    # models/mvs4net_utils.py:102): only --inverse_depth gets past stage 1 in the reference
                    raise NotImplementedError("non-inverse depth scheduling is broken upstream "
                                              "(models/mvs4net_utils.py:96-106); use inverse_depth=True")
                depth_hypo = schedule_inverse_range(outputs_stage['inverse_min_depth'].detach(),
                                                    outputs_stage['inverse_max_depth'].detach(), D, H, W)
            outputs_stage = self.stagenet(features_stage, proj_matrices_stage, depth_hypo=depth_hypo,
                                          regnet=self.reg[stage_idx], stage_idx=stage_idx,
                                          group_cor=self.group_cor, group_cor_dim=self.group_cor_dim[stage_idx],
                                          split_itv=self.depth_interals_ratio[stage_idx], fn=filename)
            outputs[key] = outputs_stage

        if self.mono and self.training:
            outputs = self.mono_depth_decoder(outputs, depth_values[:, 0], depth_values[:, 1], self.debug)
        return outputs

    def to_channels_last(self, enabled=True):
        """Run the feature pyramid in torch.channels_last so that its outputs feed the fused kernels in place."""
        self.channels_last_features = bool(enabled)
        self.feature = self.feature.to(memory_format=torch.channels_last if enabled else torch.contiguous_format)
        return self
