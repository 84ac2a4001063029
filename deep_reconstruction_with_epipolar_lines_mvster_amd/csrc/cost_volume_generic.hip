// A2 / A3 for the modes and shapes the specialised kernels do not instantiate: any channel count, group-wise correlation
// with any group count OR the per-channel squared difference of `group_cor=False`, both view-weight modes
// (`attn_fuse_d` True / False), any D <= MVSTER_MAX_DEPTHS.  Planar (NCHW) feature maps.
//   reference: models/mvs4net_utils.py:1066-1100 (+ autograd), operator for operator.
// No reference script uses these modes (scripts/*.sh: --group_cor, attn_fuse_d is not even exposed), so the code is
// written for clarity and constant register use, not speed: per source view a lane makes two passes over the channels
// (first the sum that feeds the softmax, then the weighted accumulation) and re-gathers in the second; the running
// numerator lives in the output volume itself.  Lanes: Dp = next power of two >= D lanes per pixel, d fastest.
#include "cost_volume_common.h"

namespace mvster {

struct GenericDims {
    VolumeDims dm;
    int C, Cc, CPG;        // Cc = output channels: G (group correlation, CPG = C/G) or C (squared difference, CPG = 1)
    int D, Dp;
    int variance;          // group_cor == False: cor[c] = (ref[c] - warped[c])^2
    int fuse_d;            // attn_fuse_d
};

__device__ __forceinline__ float lanes_max(float v, int n) {
    for (int m = 1; m < n; m <<= 1) v = fmaxf(v, __shfl_xor(v, m, kWave));
    return v;
}
__device__ __forceinline__ float lanes_sum(float v, int n) {
    for (int m = 1; m < n; m <<= 1) v += __shfl_xor(v, m, kWave);
    return v;
}
__device__ __forceinline__ int lanes_min(int v, int n) {
    for (int m = 1; m < n; m <<= 1) v = min(v, __shfl_xor(v, m, kWave));
    return v;
}

// correlation of output channel cc (a group or a single channel) for this lane's sample
__device__ __forceinline__ float correlate(const GenericDims& g, int cc, const float* __restrict__ refp,
                                           const float* __restrict__ srcb, const Taps& t, int HW, int HWs) {
    if (g.variance) {
        const float diff = refp[(size_t)cc * HW] - sample(srcb + (size_t)cc * HWs, t);
        return diff * diff;
    }
    float acc = 0.0f;
    for (int k = 0; k < g.CPG; ++k) {
        const int c = cc * g.CPG + k;
        acc = fmaf(sample(srcb + (size_t)c * HWs, t), refp[(size_t)c * HW], acc);
    }
    return acc * (1.0f / (float)g.CPG);
}

// softmax over the D live lanes of the pixel; returns this lane's probability (0 for padding lanes)
__device__ __forceinline__ float lane_softmax(float a, bool has_d, int Dp) {
    const float m = lanes_max(has_d ? a : -INFINITY, Dp);
    const float e = has_d ? expf(a - m) : 0.0f;
    return e / lanes_sum(e, Dp);
}

__global__ void __launch_bounds__(kBlock)
cost_volume_generic_fwd_kernel(const float* __restrict__ ref, SrcPtrs srcs, const float* __restrict__ pose,
                               const float* __restrict__ hypo, float* __restrict__ vol, float* __restrict__ wsum,
                               GenericDims g) {
    const VolumeDims& dm = g.dm;
    const int HW = dm.H * dm.W, HWs = dm.Hs * dm.Ws, b = blockIdx.y;
    const int d = threadIdx.x % g.Dp, pl = threadIdx.x / g.Dp;
    const int praw = blockIdx.x * (kBlock / g.Dp) + pl;
    const bool live = praw < HW, has_d = d < g.D, on = live && has_d;
    const int p = live ? praw : HW - 1, dd = has_d ? d : g.D - 1;
    const int y = p / dm.W, x = p - y * dm.W;
    const float* __restrict__ refp = ref + (size_t)b * g.C * HW + p;
    const float hyp = hypo[((size_t)b * g.D + dd) * HW + p];
    float* __restrict__ volp = vol + ((size_t)b * g.Cc * g.D + dd) * HW + p;          // + cc * D * HW
    float den = 1e-8f;                                                                 // (:1037)
    for (int s = 0; s < dm.S; ++s) {
        const float* __restrict__ srcb = srcs.p[s] + (size_t)b * g.C * HWs;
        const Pose q = load_pose(pose + ((size_t)b * dm.S + s) * 12);
        const Taps t = project_taps(q, rotate_pixel(q, (float)x, (float)y), hyp, dm.Hs, dm.Ws);
        float a = 0.0f;
        for (int cc = 0; cc < g.Cc; ++cc) {
            const float cor = correlate(g, cc, refp, srcb, t, HW, HWs);
            a = cc == 0 ? cor : a + cor;                                               // cor_feat.sum(1)
        }
        float w;
        if (g.fuse_d) w = lane_softmax(a / dm.temp, has_d, g.Dp) / dm.sqrt_c;          // (:1083)
        else w = lanes_max(lane_softmax(a, has_d, g.Dp), g.Dp);                        // (:1079): max over depth, per pixel
        den += w;
        for (int cc = 0; cc < g.Cc; ++cc) {
            const float term = w * correlate(g, cc, refp, srcb, t, HW, HWs);
            if (on) {
                float* __restrict__ o = volp + (size_t)cc * g.D * HW;
                *o = s == 0 ? term : *o + term;
            }
        }
    }
    if (!on) return;
    for (int cc = 0; cc < g.Cc; ++cc) {
        float* __restrict__ o = volp + (size_t)cc * g.D * HW;
        *o = *o / den;                                                                 // (:1098-1100)
    }
    if (wsum != nullptr) wsum[((size_t)b * g.D + d) * HW + p] = den;                   // per pixel in the no-fuse mode, stored per d
}

// With gr = dL/dvol, out = vol, den = 1e-8 + sum_s w_s, gd = gr / den:
//   attn_fuse_d:      dw_s[d] = sum_c gd[c,d] (cor_s[c,d] - out[c,d]);  da_s[d] = w^_s[d] (dw_s[d] - sum_d' w^_s[d'] dw_s[d']) / (T sqrt(C));
//                     dcor_s[c,d] = gd[c,d] w^_s[d] / sqrt(C) + da_s[d]
//   not attn_fuse_d:  dW_s = sum_{c,d} gd[c,d] (cor_s[c,d] - out[c,d]);  w_s = w^_s[d*], d* = first arg-max;
//                     da_s[d] = dW_s w_s ((d == d*) - w^_s[d]);           dcor_s[c,d] = gd[c,d] w_s + da_s[d]
//   group correlation: dwarped[c] = dcor[g(c)] ref[c] / CPG, dref[c] += dcor[g(c)] warped[c] / CPG
//   squared difference: dwarped[c] = -2 dcor[c] (ref[c] - warped[c]),   dref[c] += 2 dcor[c] (ref[c] - warped[c])
__global__ void __launch_bounds__(kBlock)
cost_volume_generic_bwd_kernel(const float* __restrict__ gvol, const float* __restrict__ vol, const float* __restrict__ wsum,
                               const float* __restrict__ ref, SrcPtrs srcs, const float* __restrict__ pose,
                               const float* __restrict__ hypo, float* __restrict__ gref, GradPtrs gsrcs, GenericDims g) {
    const VolumeDims& dm = g.dm;
    const int HW = dm.H * dm.W, HWs = dm.Hs * dm.Ws, b = blockIdx.y;
    const int d = threadIdx.x % g.Dp, pl = threadIdx.x / g.Dp;
    const int praw = blockIdx.x * (kBlock / g.Dp) + pl;
    const bool live = praw < HW, has_d = d < g.D, on = live && has_d;
    const int p = live ? praw : HW - 1, dd = has_d ? d : g.D - 1;
    const int y = p / dm.W, x = p - y * dm.W;
    const float* __restrict__ refp = ref + (size_t)b * g.C * HW + p;
    const float hyp = hypo[((size_t)b * g.D + dd) * HW + p];
    const size_t vbase = ((size_t)b * g.Cc * g.D + dd) * HW + p;                        // + cc * D * HW
    const float rden = 1.0f / wsum[((size_t)b * g.D + dd) * HW + p];
    float* __restrict__ grefp = gref + (size_t)b * g.C * HW + p;
    for (int s = 0; s < dm.S; ++s) {
        const float* __restrict__ srcb = srcs.p[s] + (size_t)b * g.C * HWs;
        float* __restrict__ gsrcb = gsrcs.p[s] + (size_t)b * g.C * HWs;
        const Pose q = load_pose(pose + ((size_t)b * dm.S + s) * 12);
        const Taps t = project_taps(q, rotate_pixel(q, (float)x, (float)y), hyp, dm.Hs, dm.Ws);
        float a = 0.0f, dw = 0.0f;
        for (int cc = 0; cc < g.Cc; ++cc) {
            const float cor = correlate(g, cc, refp, srcb, t, HW, HWs);
            a = cc == 0 ? cor : a + cor;
            const size_t at = vbase + (size_t)cc * g.D * HW;
            dw = fmaf(on ? gvol[at] * rden : 0.0f, cor - vol[at], dw);
        }
        float wv, da;
        if (g.fuse_d) {
            const float what = lane_softmax(a / dm.temp, has_d, g.Dp);
            const float sdot = lanes_sum(what * dw, g.Dp);
            da = what * (dw - sdot) * (1.0f / (dm.temp * dm.sqrt_c));
            wv = what / dm.sqrt_c;
        } else {
            const float what = lane_softmax(a, has_d, g.Dp);
            wv = lanes_max(what, g.Dp);
            const int dstar = lanes_min(has_d && what == wv ? d : kWave, g.Dp);        // torch.max: first index on ties
            const float dW = lanes_sum(dw, g.Dp);
            da = dW * wv * ((d == dstar ? 1.0f : 0.0f) - what);
        }
        for (int cc = 0; cc < g.Cc; ++cc) {
            const size_t at = vbase + (size_t)cc * g.D * HW;
            const float dcor = on ? fmaf(gvol[at] * rden, wv, da) : 0.0f;
            for (int k = 0; k < g.CPG; ++k) {
                const int c = cc * g.CPG + k;
                const float warped = sample(srcb + (size_t)c * HWs, t), r = refp[(size_t)c * HW];
                float dwarped, dref;
                if (g.variance) {
                    const float twice = 2.0f * (r - warped);
                    dwarped = -(dcor * twice);
                    dref = dcor * twice;
                } else {
                    dwarped = dcor * (1.0f / (float)g.CPG) * r;
                    dref = dcor * (1.0f / (float)g.CPG) * warped;
                }
                if (on) scatter(gsrcb + (size_t)c * HWs, t, dwarped);
                const float tot = lanes_sum(dref, g.Dp);                               // over the hypotheses of the pixel
                if (live && d == 0) grefp[(size_t)c * HW] = s == 0 ? tot : grefp[(size_t)c * HW] + tot;
            }
        }
    }
}

static int generic_dims(int S, int B, int C, int G, int D, int H, int W, int Hs, int Ws, float temp, int group_cor, int fuse_d,
                        GenericDims* out) {
    MVSTER_REQUIRE_SHAPE(B > 0 && B <= 65535 && C > 0 && D > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0 && S > 0 && temp != 0.0f);
    MVSTER_REQUIRE_SHAPE((long long)H * W <= kMaxPlane && (long long)Hs * Ws <= kMaxPlane);
    if (group_cor) MVSTER_REQUIRE_SHAPE(G > 0 && C % G == 0);
    if (S > MVSTER_MAX_SRC_VIEWS || D > MVSTER_MAX_DEPTHS) return MVSTER_ERR_UNSUPPORTED;
    GenericDims g{};
    g.dm = make_dims(H, W, Hs, Ws, S, C, temp);
    g.C = C;
    g.Cc = group_cor ? G : C;
    g.CPG = group_cor ? C / G : 1;
    g.D = D;
    g.Dp = 1;
    while (g.Dp < D) g.Dp <<= 1;
    g.variance = group_cor ? 0 : 1;
    g.fuse_d = fuse_d ? 1 : 0;
    *out = g;
    return MVSTER_OK;
}

}  // namespace mvster

extern "C" int mvster_cost_volume_generic_fwd(const float* ref, const float* const* srcs, int S, const float* pose,
                                              const float* hypo, float* vol, float* wsum, int B, int C, int G, int D,
                                              int H, int W, int Hs, int Ws, float attn_temp, int group_cor, int attn_fuse_d,
                                              mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(ref); MVSTER_REQUIRE_PTR(srcs); MVSTER_REQUIRE_PTR(pose); MVSTER_REQUIRE_PTR(hypo); MVSTER_REQUIRE_PTR(vol);
    GenericDims g;
    if (int rc = generic_dims(S, B, C, G, D, H, W, Hs, Ws, attn_temp, group_cor, attn_fuse_d, &g)) return rc;
    SrcPtrs sp{};
    for (int s = 0; s < S; ++s) {
        MVSTER_REQUIRE_PTR(srcs[s]);
        sp.p[s] = srcs[s];
    }
    const dim3 grid(grid_for((long long)H * W, kBlock / g.Dp), B);
    hipLaunchKernelGGL(cost_volume_generic_fwd_kernel, grid, dim3(kBlock), 0, as_stream(stream), ref, sp, pose, hypo, vol, wsum, g);
    return launch_status();
}

extern "C" int mvster_cost_volume_generic_bwd(const float* gvol, const float* vol, const float* wsum, const float* ref,
                                              const float* const* srcs, int S, const float* pose, const float* hypo,
                                              float* gref, float* const* gsrcs, int B, int C, int G, int D, int H, int W,
                                              int Hs, int Ws, float attn_temp, int group_cor, int attn_fuse_d,
                                              mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(gvol); MVSTER_REQUIRE_PTR(vol); MVSTER_REQUIRE_PTR(wsum); MVSTER_REQUIRE_PTR(ref);
    MVSTER_REQUIRE_PTR(srcs); MVSTER_REQUIRE_PTR(pose); MVSTER_REQUIRE_PTR(hypo); MVSTER_REQUIRE_PTR(gref); MVSTER_REQUIRE_PTR(gsrcs);
    GenericDims g;
    if (int rc = generic_dims(S, B, C, G, D, H, W, Hs, Ws, attn_temp, group_cor, attn_fuse_d, &g)) return rc;
    SrcPtrs sp{};
    GradPtrs gp{};
    const hipStream_t st = as_stream(stream);
    for (int s = 0; s < S; ++s) {
        MVSTER_REQUIRE_PTR(srcs[s]); MVSTER_REQUIRE_PTR(gsrcs[s]);
        sp.p[s] = srcs[s];
        gp.p[s] = gsrcs[s];
    }
    for (int s = 0; s < S; ++s) {
        if (hipMemsetAsync(gp.p[s], 0, sizeof(float) * (size_t)B * C * Hs * Ws, st) != hipSuccess) {
            (void)hipGetLastError();
            return MVSTER_ERR_LAUNCH;
        }
    }
    const dim3 grid(grid_for((long long)H * W, kBlock / g.Dp), B);
    hipLaunchKernelGGL(cost_volume_generic_bwd_kernel, grid, dim3(kBlock), 0, st, gvol, vol, wsum, ref, sp, pose, hypo, gref, gp, g);
    return launch_status();
}
