// A3  backward of the fused cost-volume kernel (see cost_volume_fwd.hip for the work decomposition).
//   reference: autograd of models/mvs4net_utils.py:1036-1102, exercised by loss.backward() (train_mvs4.py:344).
#include "cost_volume_common.h"

namespace mvster {

// --------------------------------------------------------------------------------------------- backward
// With gr = dL/dvol, out = vol, w^ = softmax_d(a/T), w = w^/sqrt(C), den = 1e-8 + sum_s w_s:
//   dw_s[d]      = sum_g gr[g,d] (cor_s[g,d] - out[g,d]) / den[d]
//   da_s[d]      = w^_s[d] (dw_s[d] - sum_d' w^_s[d'] dw_s[d']) / (T sqrt(C))
//   dcor_s[g,d]  = gr[g,d] w_s[d] / den[d] + da_s[d]
//   dwarped[c,d] = dcor_s[g(c),d] ref[c] / CPG      -> scattered onto the 4 taps of source view s
//   dref[c]     += sum_s sum_d dcor_s[g(c),d] warped_s[c,d] / CPG
// The per-view correlations are recomputed from the features (cheaper than saving [S,G,D,H,W]).
template <int CPG, int G, int D, int DPL>
__global__ void __launch_bounds__(kBlock)
cost_volume_bwd_kernel(const float* __restrict__ gvol, const float* __restrict__ vol, const float* __restrict__ wsum,
                       const float* __restrict__ ref, SrcPtrs srcs, const float* __restrict__ pose,
                       const float* __restrict__ hypo, float* __restrict__ gref, GradPtrs gsrcs, VolumeDims dm) {
    constexpr int C = CPG * G;
    constexpr int DL = D / DPL;
    constexpr int PXB = kBlock / DL;
    const int tile = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    if (tile >= dm.ntiles) return;
    const int b = blockIdx.y;
    const int HW = dm.H * dm.W, HWs = dm.Hs * dm.Ws;
    const int dl = threadIdx.x % DL, pl = threadIdx.x / DL;
    const int praw = tile * PXB + pl;
    const bool live = praw < HW;
    const int p = live ? praw : HW - 1;
    const int y = p / dm.W, x = p - y * dm.W;
    const float fx = (float)x, fy = (float)y;
    const float half_w = (float)((dm.Ws - 1) / 2.0), half_h = (float)((dm.Hs - 1) / 2.0);
    const float inv_tc = 1.0f / (dm.temp * dm.sqrt_c);

    const float* __restrict__ refp = ref + (size_t)b * C * HW + p;
    float hyp[DPL], gd[G][DPL], godot[DPL];
#pragma unroll
    for (int j = 0; j < DPL; ++j) {
        const int d = dl * DPL + j;
        hyp[j] = hypo[((size_t)b * D + d) * HW + p];
        const float rden = 1.0f / wsum[((size_t)b * D + d) * HW + p];
        godot[j] = 0.0f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const size_t at = (((size_t)b * G + g) * D + d) * HW + p;
            // a padding lane (beyond the last pixel) must contribute nothing to the atomics below
            gd[g][j] = live ? gvol[at] * rden : 0.0f;
            godot[j] = fmaf(gd[g][j], vol[at], godot[j]);
        }
    }
    float dref[C];
#pragma unroll
    for (int c = 0; c < C; ++c) dref[c] = 0.0f;

    for (int s = 0; s < dm.S; ++s) {
        const float* __restrict__ srcb = srcs.p[s] + (size_t)b * C * HWs;
        float* __restrict__ gsrcb = gsrcs.p[s] + (size_t)b * C * HWs;
        const Pose q = load_pose(pose + ((size_t)b * dm.S + s) * 12);
        const Ray ray = rotate_pixel(q, fx, fy);
        Taps taps[DPL];
        float warped[C][DPL], cor[G][DPL];
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
            taps[j] = project_taps(q, ray, hyp[j], dm.Hs, dm.Ws, half_w, half_h);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < CPG; ++k) {
                    const int c = g * CPG + k;
                    warped[c][j] = sample(srcb + (size_t)c * HWs, taps[j]);
                    acc = fmaf(warped[c][j], refp[(size_t)c * HW], acc);
                }
                cor[g][j] = acc * (1.0f / CPG);
            }
        }
        float a[DPL], m = -INFINITY;
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
            float sg = cor[0][j];
#pragma unroll
            for (int g = 1; g < G; ++g) sg += cor[g][j];
            a[j] = sg / dm.temp;
            m = fmaxf(m, a[j]);
        }
        m = group_max<DL>(m);
        float e[DPL], sum = 0.0f;
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
            e[j] = expf(a[j] - m);
            sum += e[j];
        }
        sum = group_sum<DL>(sum);
        float what[DPL], dw[DPL], sdot = 0.0f;
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
            what[j] = e[j] / sum;
            float acc = -godot[j];
#pragma unroll
            for (int g = 0; g < G; ++g) acc = fmaf(gd[g][j], cor[g][j], acc);
            dw[j] = acc;
            sdot = fmaf(what[j], dw[j], sdot);
        }
        sdot = group_sum<DL>(sdot);
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
            const float da = what[j] * (dw[j] - sdot) * inv_tc;
            const float w = what[j] / dm.sqrt_c;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float dcor = fmaf(gd[g][j], w, da) * (1.0f / CPG);
#pragma unroll
                for (int k = 0; k < CPG; ++k) {
                    const int c = g * CPG + k;
                    dref[c] = fmaf(dcor, warped[c][j], dref[c]);
                    if (live) scatter(gsrcb + (size_t)c * HWs, taps[j], dcor * refp[(size_t)c * HW]);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float v = group_sum<DL>(dref[c]);
        if (live && dl == 0) gref[((size_t)b * C + c) * HW + p] = v;
    }
}

template <int CPG, int G, int D, int DPL>
static int launch_bwd(const float* gvol, const float* vol, const float* wsum, const float* ref, const SrcPtrs& srcs,
                      const float* pose, const float* hypo, float* gref, const GradPtrs& gsrcs, int B, VolumeDims dm,
                      hipStream_t st) {
    constexpr int PXB = kBlock / (D / DPL);
    dm.ntiles = grid_for((long long)dm.H * dm.W, PXB);
    dim3 grid(((dm.ntiles + 7) / 8) * 8, B);
    hipLaunchKernelGGL((cost_volume_bwd_kernel<CPG, G, D, DPL>), grid, dim3(kBlock), 0, st, gvol, vol, wsum, ref, srcs,
                       pose, hypo, gref, gsrcs, dm);
    return launch_status();
}

}  // namespace mvster

extern "C" int mvster_cost_volume_bwd(const float* gvol, const float* vol, const float* wsum, const float* ref,
                                      const float* const* srcs, int S, const float* pose, const float* hypo,
                                      float* gref, float* const* gsrcs, int B, int C, int G, int D, int H, int W,
                                      int Hs, int Ws, float attn_temp, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(gvol); MVSTER_REQUIRE_PTR(vol); MVSTER_REQUIRE_PTR(wsum); MVSTER_REQUIRE_PTR(ref);
    MVSTER_REQUIRE_PTR(srcs); MVSTER_REQUIRE_PTR(pose); MVSTER_REQUIRE_PTR(hypo); MVSTER_REQUIRE_PTR(gref);
    MVSTER_REQUIRE_PTR(gsrcs);
    if (int rc = check_volume(S, B, C, G, D, H, W, Hs, Ws, attn_temp)) return rc;
    SrcPtrs sp{};
    GradPtrs gp{};
    for (int s = 0; s < S; ++s) {
        MVSTER_REQUIRE_PTR(srcs[s]); MVSTER_REQUIRE_PTR(gsrcs[s]);
        sp.p[s] = srcs[s];
        gp.p[s] = gsrcs[s];
    }
    const hipStream_t st = as_stream(stream);
    for (int s = 0; s < S; ++s) {
        if (hipMemsetAsync(gp.p[s], 0, sizeof(float) * (size_t)B * C * Hs * Ws, st) != hipSuccess) {
            (void)hipGetLastError();
            return MVSTER_ERR_LAUNCH;
        }
    }
    VolumeDims dm{H, W, Hs, Ws, S, 0, attn_temp, (float)sqrt((double)C)};
    const int cpg = C / G;
    const bool wide = pick_pixel_per_lane(B, H * W, C, D);
#define X(CPG_, G_)                                                                                                   \
    if (cpg == CPG_ && G == G_) {                                                                                     \
        if constexpr (CPG_ * G_ <= kMaxWideChannels) {                                                                \
            if (D == 4 && wide)                                                                                       \
                return launch_bwd<CPG_, G_, 4, 4>(gvol, vol, wsum, ref, sp, pose, hypo, gref, gp, B, dm, st);         \
        }                                                                                                             \
        if (D == 4) return launch_bwd<CPG_, G_, 4, 1>(gvol, vol, wsum, ref, sp, pose, hypo, gref, gp, B, dm, st);     \
        if (D == 8) return launch_bwd<CPG_, G_, 8, 1>(gvol, vol, wsum, ref, sp, pose, hypo, gref, gp, B, dm, st);     \
        if (D == 16) return launch_bwd<CPG_, G_, 16, 1>(gvol, vol, wsum, ref, sp, pose, hypo, gref, gp, B, dm, st);   \
    }
    MVSTER_FOR_SHAPES(X)
#undef X
    return MVSTER_ERR_UNSUPPORTED;
}
