// A2  fused homography warp + group-wise correlation + softmax-over-depth view weight + fusion over source views,
// forward (backward: cost_volume_bwd.hip).   reference: models/mvs4net_utils.py:1036-1102 (+ autograd).
//
// Work decomposition (gfx950, wave64): a workgroup of 256 lanes owns 256/DL consecutive reference
// pixels (row-major), DL = D/DPL lanes per pixel, each lane carrying DPL depth hypotheses.  Lanes of
// one pixel are adjacent (d fastest), so the softmax / arg reductions over depth are xor-shuffles
// inside an aligned lane group and the per-view state (G x DPL correlations, running numerator and
// denominator) lives in registers for the whole source-view loop.  The [B,C,D,H,W] warped volume of
// the reference is never written: HBM traffic is one read of every feature map, the hypotheses,
// and one write of the fused [B,G,D,H,W] volume.
//   DPL = 1  (D lanes per pixel): what the forward always uses -- it is gather-latency bound and this mapping runs
//             7 waves per SIMD at 65 VGPRs (1.55x faster than one lane per pixel at the DTU stage-3/4 shapes);
//   DPL = D  (one lane = one pixel): used by the backward at D = 4 (cost_volume_bwd.hip), where a lane must see all
//             hypotheses of its pixel to merge them before the atomic flush.
// Two kernel families: the planar one right below reads NCHW feature planes (for a fixed (channel, depth) the taps of
// consecutive pixels fall on consecutive addresses of at most two source rows -- fine for coherent depth); the
// channels-last one further down is what the training / inference path uses.
#include "cost_volume_common.h"

namespace mvster {

// --------------------------------------------------------------------------------------------- forward
template <int CPG, int G, int D, int DPL>
__global__ void __launch_bounds__(kBlock)
cost_volume_fwd_kernel(const float* __restrict__ ref, SrcPtrs srcs, const float* __restrict__ pose,
                       const float* __restrict__ hypo, float* __restrict__ vol, float* __restrict__ wsum,
                       VolumeDims dm) {
    constexpr int C = CPG * G;
    constexpr int DL = D / DPL;
    constexpr int PXB = kBlock / DL;
    constexpr bool kRefInRegs = (C <= 16);
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

    const float* __restrict__ refp = ref + (size_t)b * C * HW + p;
    float refv[kRefInRegs ? C : 1];
    if constexpr (kRefInRegs) {
#pragma unroll
        for (int c = 0; c < C; ++c) refv[c] = refp[(size_t)c * HW];
    }
    float hyp[DPL];
#pragma unroll
    for (int j = 0; j < DPL; ++j) hyp[j] = hypo[((size_t)b * D + dl * DPL + j) * HW + p];

    float den[DPL], num[G][DPL];
#pragma unroll
    for (int j = 0; j < DPL; ++j) {
        den[j] = 1e-8f;                                   // cor_weight_sum starts at 1e-8 (:1037)
#pragma unroll
        for (int g = 0; g < G; ++g) num[g][j] = 0.0f;
    }

    for (int s = 0; s < dm.S; ++s) {
        const float* __restrict__ srcb = srcs.p[s] + (size_t)b * C * HWs;
        const Pose q = load_pose(pose + ((size_t)b * dm.S + s) * 12);
        const Ray ray = rotate_pixel(q, fx, fy);
        float cor[G][DPL];
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
            const Taps t = project_taps(q, ray, hyp[j], dm.Hs, dm.Ws);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                // same association as the channels-last kernels (gather_correlate): even and odd channels of a group
                // accumulate separately, the reference feature carries the 1 / CPG of the group mean (:1069)
                float ax = 0.0f, ay = 0.0f;
#pragma unroll
                for (int k = 0; k < CPG; ++k) {
                    const int c = g * CPG + k;
                    const float r = (kRefInRegs ? refv[kRefInRegs ? c : 0] : refp[(size_t)c * HW]) * (1.0f / CPG);
                    const float v = sample(srcb + (size_t)c * HWs, t);
                    if (k == 0) ax = v * r;
                    else if (k == 1) ay = v * r;
                    else if (k % 2 == 0) ax = fmaf(v, r, ax);
                    else ay = fmaf(v, r, ay);
                }
                cor[g][j] = CPG == 1 ? ax : ax + ay;
            }
        }
        // view weight: softmax over ALL D hypotheses of sum_g cor / temp, then / sqrt(C)  (:1083)
        float a[DPL], m = -INFINITY;
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
            float sg = cor[0][j];
#pragma unroll
            for (int g = 1; g < G; ++g) sg += cor[g][j];
            a[j] = div_temp(sg, dm);
            m = fmaxf(m, a[j]);
        }
        m = group_max<DL>(m);
        float e[DPL], sum = 0.0f;
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
            e[j] = exp_nonpos(a[j] - m);
            sum += e[j];
        }
        sum = group_sum<DL>(sum);
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
            const float w = div_sqrt_c(div_fast(e[j], sum), dm);
            den[j] += w;                                   // (:1084)
#pragma unroll
            for (int g = 0; g < G; ++g) num[g][j] += w * cor[g][j];   // (:1085)
        }
    }
    if (!live) return;
#pragma unroll
    for (int j = 0; j < DPL; ++j) {
        const int d = dl * DPL + j;
#pragma unroll
        for (int g = 0; g < G; ++g) vol[(((size_t)b * G + g) * D + d) * HW + p] = div_fast(num[g][j], den[j]);   // (:1100)
        if (wsum != nullptr) wsum[((size_t)b * D + d) * HW + p] = den[j];
    }
}


// ------------------------------------------------------------------------------- forward, channels-last features
// Same decomposition, but every feature map is [B,H,W,C] (torch channels_last): one bilinear tap is C
// contiguous floats, fetched as C/4 16-byte loads.  The gather then costs one or two 64-byte sectors per
// tap pair however incoherent neighbouring pixels' depths are, instead of one sector per (tap, channel)
// as with NCHW planes -- this is the layout the training/inference path uses (hotpath.py converts once per
// stage, or takes channels_last tensors as they are).
//
// Round 3: written for VALU instruction count (the round-2 kernel issued ~415 wave-instructions per (sample, view), 95 %
// of its run time): lean projection (mvster_device.h land()), taps addressed by 32-bit byte offsets from the view's
// uniform base pointer, channel pairs interpolated and correlated with packed fp32 instructions, the 1 / CPG of the group
// mean folded into the reference feature (a power of two: exact), DPP reductions over the D lanes of a pixel, reciprocal
// + residual instead of IEEE divisions.

template <int C, int G, int D, int DPL>
__global__ void __launch_bounds__(kBlock)
cost_volume_fwd_nhwc_kernel(const float* __restrict__ ref, SrcPtrs srcs, const float* __restrict__ pose,
                            const float* __restrict__ hypo, float* __restrict__ vol, float* __restrict__ wsum,
                            VolumeDims dm, BinCount bc) {
    static_assert(DPL == 1 && C <= 16, "one lane per (pixel, hypothesis); C >= 32 runs the channel-split kernel");
    constexpr int CPG = C / G;
    constexpr int PXB = kBlock / D;
    const int tile = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    if (tile >= dm.ntiles) return;
    const int b = blockIdx.y;
    const int HW = dm.H * dm.W, HWs = dm.Hs * dm.Ws;
    const int lane = threadIdx.x & 63;
    const int d = threadIdx.x % D, pl = threadIdx.x / D;
    const OwnedPixel own = owned_pixel<PXB>(tile, pl, dm.H, dm.W);
    const bool live = own.live;
    const int p = own.p;
    const float fx = (float)own.x, fy = (float)own.y;

    f32x2 ref2[C / 2];
    {
        const float4* __restrict__ refp = reinterpret_cast<const float4*>(ref + ((size_t)b * HW + p) * C);
#pragma unroll
        for (int i = 0; i < C / 4; ++i) {
            const float4 r = refp[i];
            ref2[2 * i] = f32x2{r.x, r.y} * (1.0f / CPG);          // mean over the group's channels (:1069), exact scaling
            ref2[2 * i + 1] = f32x2{r.z, r.w} * (1.0f / CPG);
        }
    }
    const float hyp = hypo[((size_t)b * D + d) * HW + p];
    float den = 1e-8f, num[G];                                     // cor_weight_sum starts at 1e-8 (:1037)
#pragma unroll
    for (int g = 0; g < G; ++g) num[g] = 0.0f;
    const LaneMasks lm = lane_masks(lane);
    __shared__ BinAgg agg;
    const bool counting = bc.counters != nullptr;                  // training: first pass of the backward's counting sort
    if (counting) agg_reset(agg);

    for (int s = 0; s < dm.S; ++s) {
        const float* __restrict__ srcb = srcs.p[s] + (size_t)b * HWs * C;
        const Pose q = load_pose(pose + ((size_t)b * dm.S + s) * 12);
        const Landing l = land(q, rotate_pixel(q, fx, fy), hyp, dm.Hs, dm.Ws);
        if (counting) {                                            // wave-uniform
            const int v = s % kAggViews;
            agg_count<0>(agg, v, bc, bc.counters + ((size_t)b * dm.S + s) * bc.tiles_per_view * kBinCats, l, live, lane, lm);
            if (v == kAggViews - 1 || s == dm.S - 1) agg_flush_counts(agg, bc.counters, bc.tiles_per_view * kBinCats, b, dm.S, s - v);
        }
        unsigned o[4];
        float w[4];
        tap_addresses<C>(l, dm.Hs, dm.Ws, o, w);
        float cor[G];
        f32x2 unused[1];
        gather_correlate<C, CPG, false>(srcb, o, w, ref2, cor, unused);
        // view weight: softmax over ALL D hypotheses of sum_g cor / temp, then / sqrt(C)  (:1083)
        float sg = cor[0];
#pragma unroll
        for (int g = 1; g < G; ++g) sg += cor[g];
        const float a = div_temp(sg, dm);
        const float m = group_max<D>(a);
        const float e = exp_nonpos(a - m);
        const float sum = group_sum<D>(e);
        const float wv = div_sqrt_c(div_fast(e, sum), dm);
        den += wv;                                                 // (:1084)
#pragma unroll
        for (int g = 0; g < G; ++g) num[g] += wv * cor[g];         // (:1085)
    }
    if (!live) return;
    const float rden = rcp_approx(den);
    float* __restrict__ volb = vol + (size_t)b * G * D * HW;
    const unsigned at = (unsigned)d * (unsigned)HW + (unsigned)p;
#pragma unroll
    for (int g = 0; g < G; ++g) {                                  // num / den (:1100)
        const float qn = num[g] * rden;
        volb[at + (unsigned)(g * D) * (unsigned)HW] = fmaf(fmaf(-qn, den, num[g]), rden, qn);
    }
    if (wsum != nullptr) wsum[(size_t)b * D * HW + at] = den;
}

// ------------------------------------------------------------------------------- forward, C >= 32: channel-split lanes
// With 32 / 64 channels one lane per (pixel, hypothesis) keeps every tap's whole channel vector in flight: 160 / 262
// VGPRs, three / ONE wave per SIMD, and the kernel waits on its own gathers (0.85 / 0.50 ms at DTU stages 2 / 1 against
// a ~0.2 / 0.1 ms arithmetic floor).  Here a sample is shared by HS = C / 16 adjacent lanes of 16 channels each (whole
// correlation groups): a lane's four taps are one 64-byte line each, the lanes of a sample read adjacent lines, and
// only the sum over all groups (the softmax input) crosses the lanes (xor shuffles).  Lanes of a pixel: [d][h], h fastest.
template <int C, int G, int D, int HS>
__global__ void __launch_bounds__(kBlock)
cost_volume_fwd_split_kernel(const float* __restrict__ ref, SrcPtrs srcs, const float* __restrict__ pose,
                             const float* __restrict__ hypo, float* __restrict__ vol, float* __restrict__ wsum,
                             VolumeDims dm, BinCount bc) {
    constexpr int CPG = C / G, CL = C / HS, GL = G / HS;
    static_assert(CL == 16 && CL % CPG == 0 && GL >= 1, "16 channels = whole correlation groups per lane");
    constexpr int LPP = D * HS;                    // lanes per pixel
    static_assert(LPP <= kWave, "a pixel's lanes live in one wave");
    constexpr int PXB = kBlock / LPP;
    constexpr int HSLOG = HS == 2 ? 1 : 2;
    static_assert(HS == 2 || HS == 4, "C = 32 or 64");
    const int tile = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    if (tile >= dm.ntiles) return;
    const int b = blockIdx.y;
    const int HW = dm.H * dm.W, HWs = dm.Hs * dm.Ws;
    const int lane = threadIdx.x & 63;
    const int h = threadIdx.x % HS, d = (threadIdx.x / HS) % D, pl = threadIdx.x / LPP;
    const int praw = tile * PXB + pl;
    const bool live = praw < HW;
    const int p = live ? praw : HW - 1;
    const int y = p / dm.W, x = p - y * dm.W;

    f32x2 ref2[CL / 2];
    {
        const float4* __restrict__ refp = reinterpret_cast<const float4*>(ref + ((size_t)b * HW + p) * C + h * CL);
#pragma unroll
        for (int i = 0; i < CL / 4; ++i) {
            const float4 r = refp[i];
            ref2[2 * i] = f32x2{r.x, r.y} * (1.0f / CPG);
            ref2[2 * i + 1] = f32x2{r.z, r.w} * (1.0f / CPG);
        }
    }
    const float hyp = hypo[((size_t)b * D + d) * HW + p];
    float den = 1e-8f, num[GL];
#pragma unroll
    for (int g = 0; g < GL; ++g) num[g] = 0.0f;
    const LaneMasks lm = lane_masks(lane);
    __shared__ BinAgg agg;
    const bool counting = bc.counters != nullptr;
    if (counting) agg_reset(agg);

    for (int s = 0; s < dm.S; ++s) {
        const float* __restrict__ srcb = srcs.p[s] + (size_t)b * HWs * C + h * CL;
        const Pose q = load_pose(pose + ((size_t)b * dm.S + s) * 12);
        const Landing l = land(q, rotate_pixel(q, (float)x, (float)y), hyp, dm.Hs, dm.Ws);
        if (counting) {                                            // wave-uniform
            const int v = s % kAggViews;
            agg_count<HSLOG>(agg, v, bc, bc.counters + ((size_t)b * dm.S + s) * bc.tiles_per_view * kBinCats, l, live, lane, lm);
            if (v == kAggViews - 1 || s == dm.S - 1) agg_flush_counts(agg, bc.counters, bc.tiles_per_view * kBinCats, b, dm.S, s - v);
        }
        unsigned o[4];
        float w[4];
        tap_addresses<C>(l, dm.Hs, dm.Ws, o, w);
        float cor[GL];
        f32x2 unused[1];
        gather_correlate<CL, CPG, false>(srcb, o, w, ref2, cor, unused);
        float sg = cor[0];
#pragma unroll
        for (int g = 1; g < GL; ++g) sg += cor[g];
        sg = strided_sum<HS, 1>(sg);                                   // all G groups: every lane of the sample
        const float a = div_temp(sg, dm);
        const float m = strided_max<D, HS>(a);
        const float e = exp_nonpos(a - m);
        const float sum = strided_sum<D, HS>(e);
        const float wv = div_sqrt_c(div_fast(e, sum), dm);
        den += wv;
#pragma unroll
        for (int g = 0; g < GL; ++g) num[g] += wv * cor[g];
    }
    if (!live) return;
#pragma unroll
    for (int g = 0; g < GL; ++g) vol[(((size_t)b * G + h * GL + g) * D + d) * HW + p] = div_fast(num[g], den);
    if (wsum != nullptr && h == 0) wsum[((size_t)b * D + d) * HW + p] = den;
}

template <int CPG, int G, int D, int DPL>
static int launch_fwd(const float* ref, const SrcPtrs& srcs, const float* pose, const float* hypo, float* vol,
                      float* wsum, int B, VolumeDims dm, int layout, const BinCount& bc, hipStream_t st) {
    constexpr int PXB = kBlock / (D / DPL);
    dm.ntiles = layout == MVSTER_LAYOUT_NHWC ? PixelBlocks<PXB>::tiles(dm.H, dm.W) : grid_for((long long)dm.H * dm.W, PXB);
    dim3 grid(((dm.ntiles + 7) / 8) * 8, B);
    if constexpr (CPG * G >= 32 && DPL == 1) {
        if (layout == MVSTER_LAYOUT_NHWC) {
            constexpr int HS = CPG * G / 16;
            dm.ntiles = grid_for((long long)dm.H * dm.W, kBlock / (D * HS));
            dim3 grid2(((dm.ntiles + 7) / 8) * 8, B);
            hipLaunchKernelGGL((cost_volume_fwd_split_kernel<CPG * G, G, D, HS>), grid2, dim3(kBlock), 0, st, ref, srcs, pose,
                               hypo, vol, wsum, dm, bc);
            return launch_status();
        }
    }
    if (layout == MVSTER_LAYOUT_NHWC) {
        if constexpr (CPG * G <= 16)
            hipLaunchKernelGGL((cost_volume_fwd_nhwc_kernel<CPG * G, G, D, DPL>), grid, dim3(kBlock), 0, st, ref, srcs,
                               pose, hypo, vol, wsum, dm, bc);
        else
            return MVSTER_ERR_UNSUPPORTED;          // not reached: C >= 32 returned from the channel-split kernel above
    } else
        hipLaunchKernelGGL((cost_volume_fwd_kernel<CPG, G, D, DPL>), grid, dim3(kBlock), 0, st, ref, srcs, pose, hypo,
                           vol, wsum, dm);
    return launch_status();
}

}  // namespace mvster

extern "C" long long mvster_cost_volume_bin_counts(int S, int B, int C, int G, int D, int H, int W, int Hs, int Ws,
                                                   int layout) {
    mvster::BinnedLayout lay;
    if (layout != MVSTER_LAYOUT_NHWC || !mvster::binned_layout(S, B, C, G, D, H, W, Hs, Ws, &lay)) return 0;
    return lay.nbins;
}

extern "C" int mvster_cost_volume_fwd(const float* ref, const float* const* srcs, int S, const float* pose,
                                      const float* hypo, float* vol, float* wsum, int B, int C, int G, int D, int H,
                                      int W, int Hs, int Ws, float attn_temp, int layout, mvster_stream_t stream) {
    return mvster_cost_volume_fwd_counted(ref, srcs, S, pose, hypo, vol, wsum, B, C, G, D, H, W, Hs, Ws, attn_temp, layout,
                                          nullptr, stream);
}

extern "C" int mvster_cost_volume_fwd_counted(const float* ref, const float* const* srcs, int S, const float* pose,
                                              const float* hypo, float* vol, float* wsum, int B, int C, int G, int D,
                                              int H, int W, int Hs, int Ws, float attn_temp, int layout,
                                              int32_t* bin_counts, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(ref); MVSTER_REQUIRE_PTR(srcs); MVSTER_REQUIRE_PTR(pose); MVSTER_REQUIRE_PTR(hypo);
    MVSTER_REQUIRE_PTR(vol);
    if (int rc = check_volume(S, B, C, G, D, H, W, Hs, Ws, attn_temp, layout)) return rc;
    SrcPtrs sp{};
    for (int s = 0; s < S; ++s) {
        MVSTER_REQUIRE_PTR(srcs[s]);
        sp.p[s] = srcs[s];
    }
    VolumeDims dm = make_dims(H, W, Hs, Ws, S, C, attn_temp);
    const hipStream_t st = as_stream(stream);
    BinCount bc{};
    if (bin_counts != nullptr) {
        const long long nbins = mvster_cost_volume_bin_counts(S, B, C, G, D, H, W, Hs, Ws, layout);
        if (nbins <= 0) return MVSTER_ERR_UNSUPPORTED;
        if (hipMemsetAsync(bin_counts, 0, sizeof(int) * (size_t)nbins, st) != hipSuccess) {
            (void)hipGetLastError();
            return MVSTER_ERR_LAUNCH;
        }
        bc.counters = bin_counts;
        bc.bg = bin_grid_for(C, Hs, Ws);
        bc.tiles_per_view = bc.bg.tiles_x * bc.bg.tiles_y;
    }
    const int cpg = C / G;
#define X(CPG_, G_)                                                                                          \
    if (cpg == CPG_ && G == G_) {                                                                            \
        if (D == 4) return launch_fwd<CPG_, G_, 4, 1>(ref, sp, pose, hypo, vol, wsum, B, dm, layout, bc, st);        \
        if (D == 8) return launch_fwd<CPG_, G_, 8, 1>(ref, sp, pose, hypo, vol, wsum, B, dm, layout, bc, st);        \
        if (D == 16) return launch_fwd<CPG_, G_, 16, 1>(ref, sp, pose, hypo, vol, wsum, B, dm, layout, bc, st);      \
    }
    MVSTER_FOR_SHAPES(X)
#undef X
    return MVSTER_ERR_UNSUPPORTED;
}
