// F2/F3 support: training-mode BatchNorm (+ReLU) over NCHW planes, fused into two streaming passes each way.
//   reference: the BatchNorm2d/3d + ReLU that follows every convolution of FPN4 and reg2d
//   (models/mvs4net_utils.py:123-130, 230-256, 884-926).  torch routes these to MIOpen's spatial BN, which at the
//   shapes of this model (8-64 channels, up to 1.8 M pixels per plane, N = B*D slices) runs ~10x below the HBM rate:
//   42 ms of the 128 ms end-to-end training step on MI355X.  Here:
//     forward : pass 1 reads x once  -> per-channel shifted sums (double atomics, one pair per workgroup)
//               pass 2 reads x, writes y = max(0, (x - mean_c)*a_c + beta_c)
//     backward: pass 1 reads dy, x   -> sum(dy'), sum(dy' * xhat)   (dy' = dy where y > 0)
//               pass 2 reads dy, x, writes dx = a_c * (dy' - mean(dy') - xhat * mean(dy' * xhat))
//   Algorithmic bytes: 3T forward, 5T backward for a tensor of T bytes.
#include <math.h>

#include "conv_epilogue.h"
#include "mvster_device.h"
#include "mvster_host.h"

namespace mvster {

constexpr int kBnVec = 4;              // floats per lane per load (16 B)
constexpr int kBnIters = 8;            // loads per lane per workgroup: 256 * 4 * 8 = 8192 elements per workgroup

__device__ __forceinline__ void block_sum2(double& a, double& b, double* __restrict__ scratch) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        a += __shfl_xor(a, m, kWave);
        b += __shfl_xor(b, m, kWave);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { scratch[2 * wave] = a; scratch[2 * wave + 1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = scratch[0] + scratch[2] + scratch[4] + scratch[6];
        b = scratch[1] + scratch[3] + scratch[5] + scratch[7];
    }
}

// Statistic groups: the N samples are `groups` runs of Ng consecutive samples, each normalised with its own batch
// statistics (the reference pushes every view through the pyramid on its own, models/MVS4Net.py:88-93: one launch over
// all views here, same numbers).  Per-(group, channel) arrays are indexed gc = group * C + c.
// sums[gc] = { sum(x - K), sum((x - K)^2) } with K = the group's first element of the channel: shifting by a sample keeps
// E[(x-K)^2] - E[x-K]^2 free of the cancellation the raw moments suffer when |mean| >> std.
__global__ void __launch_bounds__(kBlock)
bn_stats_kernel(const float* __restrict__ x, double* __restrict__ sums, int C, int HW, int Ng) {
    __shared__ double scratch[8];
    const int c = blockIdx.y, n = blockIdx.z;
    const int g = n / Ng, gc = g * C + c;
    const float shift = x[((size_t)g * Ng * C + c) * HW];
    const float* __restrict__ plane = x + ((size_t)n * C + c) * HW;
    const int base = blockIdx.x * (kBlock * kBnVec * kBnIters);
    float s1 = 0.0f, s2 = 0.0f;
    const bool vec_ok = (HW % kBnVec) == 0;
#pragma unroll
    for (int it = 0; it < kBnIters; ++it) {
        const int i = base + (it * kBlock + threadIdx.x) * kBnVec;
        if (vec_ok && i + kBnVec <= HW) {
            const float4 v = *reinterpret_cast<const float4*>(plane + i);
            const float a = v.x - shift, b = v.y - shift, cc = v.z - shift, d = v.w - shift;
            s1 += (a + b) + (cc + d);
            s2 = fmaf(a, a, fmaf(b, b, fmaf(cc, cc, fmaf(d, d, s2))));
        } else {
            for (int k = 0; k < kBnVec; ++k)
                if (i + k < HW) {
                    const float a = plane[i + k] - shift;
                    s1 += a;
                    s2 = fmaf(a, a, s2);
                }
        }
    }
    double d1 = s1, d2 = s2;
    block_sum2(d1, d2, scratch);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[2 * gc], d1);
        atomicAdd(&sums[2 * gc + 1], d2);
    }
}

// per channel: mean, invstd, running statistics (PyTorch semantics: biased variance normalises, unbiased one is
// tracked), and the affine y = x * a + b that pass 2 applies.
__global__ void bn_finalize_kernel(const float* __restrict__ x, const double* __restrict__ sums,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ affine, int C, int HW, int Ng, int groups, double count,
                                   float momentum, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float g = gamma != nullptr ? gamma[c] : 1.0f, b = beta != nullptr ? beta[c] : 0.0f;
    for (int grp = 0; grp < groups; ++grp) {                      // in order: the running statistics see the groups one by one
        const int gc = grp * C + c;
        const double shift = (double)x[((size_t)grp * Ng * C + c) * HW];
        const double m1 = sums[2 * gc] / count, m2 = sums[2 * gc + 1] / count;
        const double mean = shift + m1;
        double var = m2 - m1 * m1;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        save_mean[gc] = (float)mean;
        save_invstd[gc] = invstd;
        if (running_mean != nullptr) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
        }
        affine[2 * gc] = g * invstd;   // y = (x - mean) * a + beta: subtracting the mean first keeps fp32 exact-ish for |mean| >> std
        affine[2 * gc + 1] = b;
    }
}

template <bool RELU>
__global__ void __launch_bounds__(kBlock)
bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ affine, const float* __restrict__ save_mean,
                const float* __restrict__ residual, float* __restrict__ y, int C, int HW, int Ng) {
    const int c = blockIdx.y, n = blockIdx.z;
    const int gc = (n / Ng) * C + c;
    const float a = affine[2 * gc], b = affine[2 * gc + 1], mean = save_mean[gc];
    const size_t off = ((size_t)n * C + c) * HW;
    const int base = blockIdx.x * (kBlock * kBnVec * kBnIters);
    const bool vec_ok = (HW % kBnVec) == 0;
#pragma unroll
    for (int it = 0; it < kBnIters; ++it) {
        const int i = base + (it * kBlock + threadIdx.x) * kBnVec;
        if (vec_ok && i + kBnVec <= HW) {
            float4 v = *reinterpret_cast<const float4*>(x + off + i);
            v.x = fmaf(v.x - mean, a, b); v.y = fmaf(v.y - mean, a, b); v.z = fmaf(v.z - mean, a, b); v.w = fmaf(v.w - mean, a, b);
            if (RELU) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
            if (residual != nullptr) {                             // y = relu(bn(x)) + residual: the skip connections of reg2d
                const float4 r = *reinterpret_cast<const float4*>(residual + off + i);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            *reinterpret_cast<float4*>(y + off + i) = v;
        } else {
            for (int k = 0; k < kBnVec; ++k)
                if (i + k < HW) {
                    float v = fmaf(x[off + i + k] - mean, a, b);
                    if (RELU) v = fmaxf(v, 0.0f);
                    y[off + i + k] = residual != nullptr ? v + residual[off + i + k] : v;
                }
        }
    }
}

// sums[gc] = { sum(dy'), sum(dy' * xhat) }
template <bool RELU>
__global__ void __launch_bounds__(kBlock)
bn_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ affine,
                    const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                    double* __restrict__ sums, int C, int HW, int Ng) {
    __shared__ double scratch[8];
    const int c = blockIdx.y, n = blockIdx.z;
    const int gc = (n / Ng) * C + c;
    const float a = affine[2 * gc], b = affine[2 * gc + 1], mean = save_mean[gc], invstd = save_invstd[gc];
    const size_t off = ((size_t)n * C + c) * HW;
    const int base = blockIdx.x * (kBlock * kBnVec * kBnIters);
    float s1 = 0.0f, s2 = 0.0f;
    auto take = [&](float xv, float g) {
        if (RELU && !(fmaf(xv - mean, a, b) > 0.0f)) g = 0.0f;
        s1 += g;
        s2 = fmaf(g, (xv - mean) * invstd, s2);
    };
    const bool vec_ok = (HW % kBnVec) == 0;
#pragma unroll
    for (int it = 0; it < kBnIters; ++it) {
        const int i = base + (it * kBlock + threadIdx.x) * kBnVec;
        if (vec_ok && i + kBnVec <= HW) {
            const float4 xv = *reinterpret_cast<const float4*>(x + off + i);
            const float4 gv = *reinterpret_cast<const float4*>(dy + off + i);
            take(xv.x, gv.x); take(xv.y, gv.y); take(xv.z, gv.z); take(xv.w, gv.w);
        } else {
            for (int k = 0; k < kBnVec; ++k)
                if (i + k < HW) take(x[off + i + k], dy[off + i + k]);
        }
    }
    double d1 = s1, d2 = s2;
    block_sum2(d1, d2, scratch);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[2 * gc], d1);
        atomicAdd(&sums[2 * gc + 1], d2);
    }
}

template <bool RELU>
__global__ void __launch_bounds__(kBlock)
bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ affine,
                    const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                    const double* __restrict__ sums, float* __restrict__ dx, float* __restrict__ dgamma,
                    float* __restrict__ dbeta, int C, int HW, int Ng, int groups, double count) {
    const int c = blockIdx.y, n = blockIdx.z;
    const int gc = (n / Ng) * C + c;
    const float a = affine[2 * gc], b = affine[2 * gc + 1], mean = save_mean[gc], invstd = save_invstd[gc];
    const float mean_dy = (float)(sums[2 * gc] / count), mean_dyx = (float)(sums[2 * gc + 1] / count);
    if (blockIdx.x == 0 && n == 0 && threadIdx.x == 0) {          // gamma and beta are shared by the groups: their gradients add
        double db = 0.0, dg = 0.0;
        for (int grp = 0; grp < groups; ++grp) {
            db += sums[2 * (grp * C + c)];
            dg += sums[2 * (grp * C + c) + 1];
        }
        if (dbeta != nullptr) dbeta[c] = (float)db;
        if (dgamma != nullptr) dgamma[c] = (float)dg;
    }
    const size_t off = ((size_t)n * C + c) * HW;
    const int base = blockIdx.x * (kBlock * kBnVec * kBnIters);
    auto grad = [&](float xv, float g) {
        if (RELU && !(fmaf(xv - mean, a, b) > 0.0f)) g = 0.0f;
        return a * (g - mean_dy - (xv - mean) * invstd * mean_dyx);
    };
    const bool vec_ok = (HW % kBnVec) == 0;
#pragma unroll
    for (int it = 0; it < kBnIters; ++it) {
        const int i = base + (it * kBlock + threadIdx.x) * kBnVec;
        if (vec_ok && i + kBnVec <= HW) {
            const float4 xv = *reinterpret_cast<const float4*>(x + off + i);
            const float4 gv = *reinterpret_cast<const float4*>(dy + off + i);
            *reinterpret_cast<float4*>(dx + off + i) = make_float4(grad(xv.x, gv.x), grad(xv.y, gv.y), grad(xv.z, gv.z), grad(xv.w, gv.w));
        } else {
            for (int k = 0; k < kBnVec; ++k)
                if (i + k < HW) dx[off + i + k] = grad(x[off + i + k], dy[off + i + k]);
        }
    }
}


// ---------------------------------------------------------------------------------------------- channels-last variant
// x is [M = N*H*W, C] (torch.channels_last storage), C a multiple of 4 and <= 64.  A lane owns one channel quad of a
// row (16-byte loads); the kBlock / (C/4) rows of a workgroup-iteration are consecutive in memory, so every load and
// store is a dense 4 KB burst.  Per-quad partial sums are reduced across the row-lanes through LDS.
constexpr int kBnRowsPerBlock = 2048;      // rows a workgroup walks (grid.x = ceil(M / 2048))

template <typename F>
__device__ __forceinline__ void nhwc_rows(int M, int C4, F&& body) {
    const int quad = threadIdx.x % C4, lane_row = threadIdx.x / C4, rows_per_iter = kBlock / C4;
    const int first = blockIdx.x * kBnRowsPerBlock;
    const int last = min(first + kBnRowsPerBlock, M);
    if (lane_row >= rows_per_iter) return;                       // C4 does not divide kBlock only for odd C4 (not used)
#pragma unroll 4
    for (int r = first + lane_row; r < last; r += rows_per_iter) body(r, quad);
}

__device__ __forceinline__ void nhwc_reduce_quads(float (&acc)[8], double* __restrict__ sums, int C4, float* __restrict__ lds) {
    // lanes of a wave with equal (lane % C4) own the same channel quad (C4 divides 64): butterfly over the strides >= C4,
    // then one LDS slot per (wave, quad) and C4 threads fold the kBlock/64 waves and issue the double atomics
    for (int off = 32; off >= C4; off >>= 1)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += __shfl_xor(acc[k], off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < C4)
#pragma unroll
        for (int k = 0; k < 8; ++k) lds[(wave * 16 + lane) * 8 + k] = acc[k];
    __syncthreads();
    if (threadIdx.x < C4) {
        double tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int w = 0; w < kBlock / 64; ++w)
#pragma unroll
            for (int k = 0; k < 8; ++k) tot[k] += (double)lds[(w * 16 + threadIdx.x) * 8 + k];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            atomicAdd(&sums[2 * (4 * threadIdx.x + i)], tot[i]);
            atomicAdd(&sums[2 * (4 * threadIdx.x + i) + 1], tot[4 + i]);
        }
    }
}

__global__ void __launch_bounds__(kBlock)
bn_stats_nhwc_kernel(const float* __restrict__ x, double* __restrict__ sums, int M, int C) {
    __shared__ float lds[(kBlock / 64) * 16 * 8];
    const int C4 = C / 4;
    const float4 shift = reinterpret_cast<const float4*>(x)[threadIdx.x % C4];      // row 0 of this quad
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    nhwc_rows(M, C4, [&](int r, int quad) {
        const float4 v = reinterpret_cast<const float4*>(x + (size_t)r * C)[quad];
        const float a = v.x - shift.x, b = v.y - shift.y, c = v.z - shift.z, d = v.w - shift.w;
        acc[0] += a; acc[1] += b; acc[2] += c; acc[3] += d;
        acc[4] = fmaf(a, a, acc[4]); acc[5] = fmaf(b, b, acc[5]); acc[6] = fmaf(c, c, acc[6]); acc[7] = fmaf(d, d, acc[7]);
    });
    nhwc_reduce_quads(acc, sums, C4, lds);
}

template <bool RELU>
__global__ void __launch_bounds__(kBlock)
bn_apply_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ affine, const float* __restrict__ save_mean,
                     float* __restrict__ y, int M, int C) {
    const int C4 = C / 4, quad = threadIdx.x % C4;
    float a[4], b[4], mu[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = affine[2 * (4 * quad + i)]; b[i] = affine[2 * (4 * quad + i) + 1]; mu[i] = save_mean[4 * quad + i]; }
    nhwc_rows(M, C4, [&](int r, int q) {
        float4 v = reinterpret_cast<const float4*>(x + (size_t)r * C)[q];
        v.x = fmaf(v.x - mu[0], a[0], b[0]); v.y = fmaf(v.y - mu[1], a[1], b[1]);
        v.z = fmaf(v.z - mu[2], a[2], b[2]); v.w = fmaf(v.w - mu[3], a[3], b[3]);
        if (RELU) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
        reinterpret_cast<float4*>(y + (size_t)r * C)[q] = v;
    });
}

template <bool RELU>
__global__ void __launch_bounds__(kBlock)
bn_bwd_stats_nhwc_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ affine,
                         const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                         double* __restrict__ sums, int M, int C) {
    __shared__ float lds[(kBlock / 64) * 16 * 8];
    const int C4 = C / 4, quad = threadIdx.x % C4;
    float a[4], b[4], mu[4], is[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = affine[2 * (4 * quad + i)]; b[i] = affine[2 * (4 * quad + i) + 1];
        mu[i] = save_mean[4 * quad + i]; is[i] = save_invstd[4 * quad + i];
    }
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    nhwc_rows(M, C4, [&](int r, int q) {
        const float4 xv = reinterpret_cast<const float4*>(x + (size_t)r * C)[q];
        const float4 gv = reinterpret_cast<const float4*>(dy + (size_t)r * C)[q];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        float gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (RELU && !(fmaf(xs[i] - mu[i], a[i], b[i]) > 0.0f)) gs[i] = 0.0f;
            acc[i] += gs[i];
            acc[4 + i] = fmaf(gs[i], (xs[i] - mu[i]) * is[i], acc[4 + i]);
        }
    });
    nhwc_reduce_quads(acc, sums, C4, lds);
}

template <bool RELU>
__global__ void __launch_bounds__(kBlock)
bn_bwd_apply_nhwc_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ affine,
                         const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                         const double* __restrict__ sums, float* __restrict__ dx, float* __restrict__ dgamma,
                         float* __restrict__ dbeta, int M, int C, double count) {
    const int C4 = C / 4, quad = threadIdx.x % C4;
    float a[4], b[4], mu[4], is[4], mdy[4], mdyx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = 4 * quad + i;
        a[i] = affine[2 * c]; b[i] = affine[2 * c + 1]; mu[i] = save_mean[c]; is[i] = save_invstd[c];
        mdy[i] = (float)(sums[2 * c] / count); mdyx[i] = (float)(sums[2 * c + 1] / count);
        if (blockIdx.x == 0 && threadIdx.x < C4) {
            if (dbeta != nullptr) dbeta[c] = (float)sums[2 * c];
            if (dgamma != nullptr) dgamma[c] = (float)sums[2 * c + 1];
        }
    }
    nhwc_rows(M, C4, [&](int r, int q) {
        const float4 xv = reinterpret_cast<const float4*>(x + (size_t)r * C)[q];
        const float4 gv = reinterpret_cast<const float4*>(dy + (size_t)r * C)[q];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        float gs[4] = {gv.x, gv.y, gv.z, gv.w}, o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (RELU && !(fmaf(xs[i] - mu[i], a[i], b[i]) > 0.0f)) gs[i] = 0.0f;
            o[i] = a[i] * (gs[i] - mdy[i] - (xs[i] - mu[i]) * is[i] * mdyx[i]);
        }
        reinterpret_cast<float4*>(dx + (size_t)r * C)[q] = make_float4(o[0], o[1], o[2], o[3]);
    });
}

// finalize for the channels-last layout: the shift sample of channel c is x[0*C + c]
__global__ void bn_finalize_nhwc_kernel(const float* __restrict__ x, const double* __restrict__ sums,
                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                        float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                        float* __restrict__ running_mean, float* __restrict__ running_var,
                                        float* __restrict__ affine, int C, double count, float momentum, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m1 = sums[2 * c] / count, m2 = sums[2 * c + 1] / count;
    const double mean = (double)x[c] + m1;
    double var = m2 - m1 * m1;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = (float)mean;
    save_invstd[c] = invstd;
    if (running_mean != nullptr) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
    const float g = gamma != nullptr ? gamma[c] : 1.0f, b = beta != nullptr ? beta[c] : 0.0f;
    affine[2 * c] = g * invstd;
    affine[2 * c + 1] = b;
}

// ---- inference mode ----------------------------------------------------------------------------------------------------
// y = (x - running_mean) / sqrt(running_var + eps) * gamma + beta (+ReLU, + residual): the statistics are constants, so the
// layer is ONE streaming pass -- a per-channel affine prepared by a C-thread launch, then the training path's apply kernel.
// (F.batch_norm + a separate F.relu are two passes on MIOpen / ATen; the reference's whole inference path runs in this mode,
// test_mvs4.py:393-411.)
__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                      float* __restrict__ affine, float* __restrict__ mean_out, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float g = gamma != nullptr ? gamma[c] : 1.0f, b = beta != nullptr ? beta[c] : 0.0f;
    affine[2 * c] = g * (float)(1.0 / sqrt((double)running_var[c] + (double)eps));
    affine[2 * c + 1] = b;
    mean_out[c] = running_mean[c];
}

// the same coefficients as three padded arrays a | b | mean for the convolution epilogues (conv_epilogue.h)
__global__ void bn_eval_norm_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                    float* __restrict__ norm, int C, int Cpad, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Cpad) return;
    const bool real = c < C;
    const float g = real && gamma != nullptr ? gamma[c] : 1.0f;
    norm[c] = real ? g * (float)(1.0 / sqrt((double)running_var[c] + (double)eps)) : 0.0f;
    norm[Cpad + c] = real && beta != nullptr ? beta[c] : 0.0f;
    norm[2 * Cpad + c] = real ? running_mean[c] : 0.0f;
}

static int check_bn_nhwc(long long M, int C) {
    MVSTER_REQUIRE_SHAPE(M > 0 && M <= (1LL << 31) - 1 && C > 0);
    if (C % 4 != 0 || C > 64 || 64 % (C / 4) != 0) return MVSTER_ERR_UNSUPPORTED;    // C in {4, 8, 16, 32, 64}
    return MVSTER_OK;
}

static int check_bn(int N, int C, int HW) {
    MVSTER_REQUIRE_SHAPE(N > 0 && N <= 65535 && C > 0 && C <= 65535 && HW > 0 && (long long)HW <= kMaxPlane);
    return MVSTER_OK;
}

}  // namespace mvster

extern "C" int mvster_bn_train_fwd_grouped(const float* x, const float* gamma, const float* beta, const float* residual, float* y,
                                           float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                                           double* sums, float* affine, int N, int C, int HW, int groups, float momentum,
                                           float eps, int relu, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(x); MVSTER_REQUIRE_PTR(y); MVSTER_REQUIRE_PTR(save_mean); MVSTER_REQUIRE_PTR(save_invstd);
    MVSTER_REQUIRE_PTR(sums); MVSTER_REQUIRE_PTR(affine);
    MVSTER_REQUIRE_SHAPE((running_mean == nullptr) == (running_var == nullptr));
    if (int rc = check_bn(N, C, HW)) return rc;
    MVSTER_REQUIRE_SHAPE(groups > 0 && N % groups == 0);
    const int Ng = N / groups;
    const hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * C * groups, st) != hipSuccess) { (void)hipGetLastError(); return MVSTER_ERR_LAUNCH; }
    dim3 grid(grid_for(HW, kBlock * kBnVec * kBnIters), C, N);
    hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(kBlock), 0, st, x, sums, C, HW, Ng);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(grid_for(C, 64)), dim3(64), 0, st, x, sums, gamma, beta, save_mean,
                       save_invstd, running_mean, running_var, affine, C, HW, Ng, groups, (double)Ng * HW, momentum, eps);
    if (relu) hipLaunchKernelGGL((bn_apply_kernel<true>), grid, dim3(kBlock), 0, st, x, affine, save_mean, residual, y, C, HW, Ng);
    else hipLaunchKernelGGL((bn_apply_kernel<false>), grid, dim3(kBlock), 0, st, x, affine, save_mean, residual, y, C, HW, Ng);
    return launch_status();
}

extern "C" int mvster_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* y, float* save_mean,
                                   float* save_invstd, float* running_mean, float* running_var, double* sums,
                                   float* affine, int N, int C, int HW, float momentum, float eps, int relu,
                                   mvster_stream_t stream) {
    return mvster_bn_train_fwd_grouped(x, gamma, beta, nullptr, y, save_mean, save_invstd, running_mean, running_var, sums, affine,
                                       N, C, HW, 1, momentum, eps, relu, stream);
}

extern "C" int mvster_bn_train_bwd_grouped(const float* dy, const float* x, const float* affine, const float* save_mean,
                                           const float* save_invstd, float* dx, float* dgamma, float* dbeta, double* sums,
                                           int N, int C, int HW, int groups, int relu, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(dy); MVSTER_REQUIRE_PTR(x); MVSTER_REQUIRE_PTR(affine); MVSTER_REQUIRE_PTR(save_mean);
    MVSTER_REQUIRE_PTR(save_invstd); MVSTER_REQUIRE_PTR(dx); MVSTER_REQUIRE_PTR(sums);
    if (int rc = check_bn(N, C, HW)) return rc;
    MVSTER_REQUIRE_SHAPE(groups > 0 && N % groups == 0);
    const int Ng = N / groups;
    const hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * C * groups, st) != hipSuccess) { (void)hipGetLastError(); return MVSTER_ERR_LAUNCH; }
    dim3 grid(grid_for(HW, kBlock * kBnVec * kBnIters), C, N);
    const double count = (double)Ng * HW;
    if (relu) {
        hipLaunchKernelGGL((bn_bwd_stats_kernel<true>), grid, dim3(kBlock), 0, st, dy, x, affine, save_mean, save_invstd, sums, C, HW, Ng);
        hipLaunchKernelGGL((bn_bwd_apply_kernel<true>), grid, dim3(kBlock), 0, st, dy, x, affine, save_mean, save_invstd, sums, dx, dgamma, dbeta, C, HW, Ng, groups, count);
    } else {
        hipLaunchKernelGGL((bn_bwd_stats_kernel<false>), grid, dim3(kBlock), 0, st, dy, x, affine, save_mean, save_invstd, sums, C, HW, Ng);
        hipLaunchKernelGGL((bn_bwd_apply_kernel<false>), grid, dim3(kBlock), 0, st, dy, x, affine, save_mean, save_invstd, sums, dx, dgamma, dbeta, C, HW, Ng, groups, count);
    }
    return launch_status();
}

extern "C" int mvster_bn_train_bwd(const float* dy, const float* x, const float* affine, const float* save_mean,
                                   const float* save_invstd, float* dx, float* dgamma, float* dbeta, double* sums,
                                   int N, int C, int HW, int relu, mvster_stream_t stream) {
    return mvster_bn_train_bwd_grouped(dy, x, affine, save_mean, save_invstd, dx, dgamma, dbeta, sums, N, C, HW, 1, relu, stream);
}

extern "C" int mvster_bn_train_fwd_nhwc(const float* x, const float* gamma, const float* beta, float* y, float* save_mean,
                                        float* save_invstd, float* running_mean, float* running_var, double* sums,
                                        float* affine, long long M, int C, float momentum, float eps, int relu,
                                        mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(x); MVSTER_REQUIRE_PTR(y); MVSTER_REQUIRE_PTR(save_mean); MVSTER_REQUIRE_PTR(save_invstd);
    MVSTER_REQUIRE_PTR(sums); MVSTER_REQUIRE_PTR(affine);
    MVSTER_REQUIRE_SHAPE((running_mean == nullptr) == (running_var == nullptr));
    if (int rc = check_bn_nhwc(M, C)) return rc;
    const hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, st) != hipSuccess) { (void)hipGetLastError(); return MVSTER_ERR_LAUNCH; }
    dim3 grid(grid_for(M, kBnRowsPerBlock));
    hipLaunchKernelGGL(bn_stats_nhwc_kernel, grid, dim3(kBlock), 0, st, x, sums, (int)M, C);
    hipLaunchKernelGGL(bn_finalize_nhwc_kernel, dim3(grid_for(C, 64)), dim3(64), 0, st, x, sums, gamma, beta, save_mean,
                       save_invstd, running_mean, running_var, affine, C, (double)M, momentum, eps);
    if (relu) hipLaunchKernelGGL((bn_apply_nhwc_kernel<true>), grid, dim3(kBlock), 0, st, x, affine, save_mean, y, (int)M, C);
    else hipLaunchKernelGGL((bn_apply_nhwc_kernel<false>), grid, dim3(kBlock), 0, st, x, affine, save_mean, y, (int)M, C);
    return launch_status();
}

extern "C" int mvster_bn_train_bwd_nhwc(const float* dy, const float* x, const float* affine, const float* save_mean,
                                        const float* save_invstd, float* dx, float* dgamma, float* dbeta, double* sums,
                                        long long M, int C, int relu, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(dy); MVSTER_REQUIRE_PTR(x); MVSTER_REQUIRE_PTR(affine); MVSTER_REQUIRE_PTR(save_mean);
    MVSTER_REQUIRE_PTR(save_invstd); MVSTER_REQUIRE_PTR(dx); MVSTER_REQUIRE_PTR(sums);
    if (int rc = check_bn_nhwc(M, C)) return rc;
    const hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, st) != hipSuccess) { (void)hipGetLastError(); return MVSTER_ERR_LAUNCH; }
    dim3 grid(grid_for(M, kBnRowsPerBlock));
    if (relu) {
        hipLaunchKernelGGL((bn_bwd_stats_nhwc_kernel<true>), grid, dim3(kBlock), 0, st, dy, x, affine, save_mean, save_invstd, sums, (int)M, C);
        hipLaunchKernelGGL((bn_bwd_apply_nhwc_kernel<true>), grid, dim3(kBlock), 0, st, dy, x, affine, save_mean, save_invstd, sums, dx, dgamma, dbeta, (int)M, C, (double)M);
    } else {
        hipLaunchKernelGGL((bn_bwd_stats_nhwc_kernel<false>), grid, dim3(kBlock), 0, st, dy, x, affine, save_mean, save_invstd, sums, (int)M, C);
        hipLaunchKernelGGL((bn_bwd_apply_nhwc_kernel<false>), grid, dim3(kBlock), 0, st, dy, x, affine, save_mean, save_invstd, sums, dx, dgamma, dbeta, (int)M, C, (double)M);
    }
    return launch_status();
}

// Inference-mode BatchNorm (+ReLU, + residual added after it) with the module's running statistics.  scratch: 3 * C floats
// (the per-channel affine and mean the apply pass reads); replaces F.batch_norm(training=False) + F.relu
// (models/mvs4net_utils.py:123-130,230-256 under model.eval()).
extern "C" int mvster_bn_eval_fwd(const float* x, const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, const float* residual, float* y, float* scratch, int N, int C,
                                  int HW, float eps, int relu, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(x); MVSTER_REQUIRE_PTR(y); MVSTER_REQUIRE_PTR(running_mean); MVSTER_REQUIRE_PTR(running_var);
    MVSTER_REQUIRE_PTR(scratch);
    if (int rc = check_bn(N, C, HW)) return rc;
    const hipStream_t st = as_stream(stream);
    float* const affine = scratch;
    float* const mean = scratch + 2 * (size_t)C;
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(grid_for(C, 64)), dim3(64), 0, st, gamma, beta, running_mean, running_var, affine,
                       mean, C, eps);
    dim3 grid(grid_for(HW, kBlock * kBnVec * kBnIters), C, N);
    if (relu) hipLaunchKernelGGL((bn_apply_kernel<true>), grid, dim3(kBlock), 0, st, x, affine, mean, residual, y, C, HW, N);
    else hipLaunchKernelGGL((bn_apply_kernel<false>), grid, dim3(kBlock), 0, st, x, affine, mean, residual, y, C, HW, N);
    return launch_status();
}

extern "C" long long mvster_bn_eval_norm_floats(int C) { return C > 0 ? 3LL * mvster::conv_norm_pad(C) : 0; }

extern "C" int mvster_bn_eval_prepare(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                      float* norm, int C, float eps, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(running_mean); MVSTER_REQUIRE_PTR(running_var); MVSTER_REQUIRE_PTR(norm);
    MVSTER_REQUIRE_SHAPE(C > 0 && C <= 65535);
    const int Cpad = conv_norm_pad(C);
    hipLaunchKernelGGL(bn_eval_norm_kernel, dim3(grid_for(Cpad, 64)), dim3(64), 0, as_stream(stream), gamma, beta, running_mean,
                       running_var, norm, C, Cpad, eps);
    return launch_status();
}

extern "C" int mvster_bn_eval_fwd_nhwc(const float* x, const float* gamma, const float* beta, const float* running_mean,
                                       const float* running_var, float* y, float* scratch, long long M, int C, float eps,
                                       int relu, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(x); MVSTER_REQUIRE_PTR(y); MVSTER_REQUIRE_PTR(running_mean); MVSTER_REQUIRE_PTR(running_var);
    MVSTER_REQUIRE_PTR(scratch);
    if (int rc = check_bn_nhwc(M, C)) return rc;
    const hipStream_t st = as_stream(stream);
    float* const affine = scratch;
    float* const mean = scratch + 2 * (size_t)C;
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(grid_for(C, 64)), dim3(64), 0, st, gamma, beta, running_mean, running_var, affine,
                       mean, C, eps);
    dim3 grid(grid_for(M, kBnRowsPerBlock));
    if (relu) hipLaunchKernelGGL((bn_apply_nhwc_kernel<true>), grid, dim3(kBlock), 0, st, x, affine, mean, y, (int)M, C);
    else hipLaunchKernelGGL((bn_apply_nhwc_kernel<false>), grid, dim3(kBlock), 0, st, x, affine, mean, y, (int)M, C);
    return launch_status();
}
