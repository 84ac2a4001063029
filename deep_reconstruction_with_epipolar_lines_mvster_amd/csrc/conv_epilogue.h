// Inference epilogue of the forward convolution kernels (conv_mfma.hip, conv_direct.hip): eval-mode BatchNorm (+ ReLU, + the
// skip connection of reg2d's up path) applied to the accumulator before the one store -- the layer's BatchNorm pass over
// the activation (a read and a write of the whole tensor) disappears.
// replaces: F.batch_norm(running statistics) + F.relu after nn.Conv2d / nn.Conv3d / nn.ConvTranspose3d in eval mode
// (models/mvs4net_utils.py:123-130,230-256, the `+` of :921-923).  Same expression as bn_apply_kernel (batchnorm.hip):
//   y = fmaf(acc - mean[c], a[c], b[c]);  a = gamma / sqrt(running_var + eps) (fp64 inside), b = beta
// so the fused layer is bit-identical to convolution followed by mvster_bn_eval_fwd.
// The kernels are instantiated with and without the epilogue (template parameter EP): the training-time forward pays
// nothing for it.  norm = three arrays of Cpad = round_up(Cout, 32) floats, a | b | mean (zero padding): four consecutive
// channels are one 16-byte load per array, whole-array scalar loads where the channel index is uniform.
#pragma once
#include "mvster_device.h"

namespace mvster {

struct ConvEpilogue {
    const float* norm;       // [3][Cpad] from mvster_bn_eval_prepare; nullptr: no normalisation
    const float* residual;   // same shape and layout as y, added after the ReLU; nullptr: none
    int relu;
    int Cpad;
};

constexpr int conv_norm_pad(int C) { return (C + 31) / 32 * 32; }

struct NormCoef {
    float a, b, mean;
};
__device__ __forceinline__ float normalise(const NormCoef& k, float v, bool relu) {
    v = fmaf(v - k.mean, k.a, k.b);
    return relu ? fmaxf(v, 0.0f) : v;
}

}  // namespace mvster
