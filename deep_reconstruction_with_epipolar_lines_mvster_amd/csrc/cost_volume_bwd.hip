// A3  backward of the fused cost-volume kernel (see cost_volume_fwd.hip for the work decomposition).
//   reference: autograd of models/mvs4net_utils.py:1036-1102, exercised by loss.backward() (train_mvs4.py:344).
#include <stdio.h>

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
    const float inv_tc = dm.inv_tc;

    const float* __restrict__ refp = ref + (size_t)b * C * HW + p;
    float hyp[DPL], gd[G][DPL], godot[DPL];
#pragma unroll
    for (int j = 0; j < DPL; ++j) {
        const int d = dl * DPL + j;
        hyp[j] = hypo[((size_t)b * D + d) * HW + p];
        const float rden = div_fast(1.0f, wsum[((size_t)b * D + d) * HW + p]);
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
            taps[j] = project_taps(q, ray, hyp[j], dm.Hs, dm.Ws);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float ax = 0.0f, ay = 0.0f;                  // same association as gather_correlate() (channels-last kernels)
#pragma unroll
                for (int k = 0; k < CPG; ++k) {
                    const int c = g * CPG + k;
                    warped[c][j] = sample(srcb + (size_t)c * HWs, taps[j]);
                    const float r = refp[(size_t)c * HW] * (1.0f / CPG);
                    if (k == 0) ax = warped[c][j] * r;
                    else if (k == 1) ay = warped[c][j] * r;
                    else if (k % 2 == 0) ax = fmaf(warped[c][j], r, ax);
                    else ay = fmaf(warped[c][j], r, ay);
                }
                cor[g][j] = CPG == 1 ? ax : ax + ay;
            }
        }
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
        float what[DPL], dw[DPL], sdot = 0.0f;
#pragma unroll
        for (int j = 0; j < DPL; ++j) {
            what[j] = div_fast(e[j], sum);
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
            const float w = div_sqrt_c(what[j], dm);
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



// Scatter-add of one lane's per-tap gradients (4 taps x C channels) with DENSE atomic wave-instructions.
// Float atomics execute at the memory side in 64-byte requests and their rate is set by the number of
// requests, not by the dwords each carries: letting every lane add its own tap (one dword per request)
// costs ~16x more requests than necessary.  The wave therefore transposes through a private LDS strip:
// each lane deposits 32 values (C = 8: 4 taps x 8 channels; C >= 16: 2 taps x 16 channels per pass) plus
// its 4 tap offsets, then lane l of the wave adds element (l & 31) of source lane (l >> 5) + 2*iteration,
// so that 8/16 consecutive lanes cover one tap's contiguous channels (and a tap pair x0,x0+1 one 64-byte run).
constexpr int kStageRow = 36;   // 32 values + 4 offsets (dwords) per lane; 144 B keeps 16-byte alignment

template <int C>
__device__ __forceinline__ void dense_scatter(float (*strip)[kStageRow], int lane, const Taps& t,
                                              const float (&dwarp)[C], float* __restrict__ gsrcb, bool live) {
    static_assert(C == 8 || C % 16 == 0, "channel count not handled by dense_scatter");
    const float wt[4] = {live ? masked_weight(t, 0) : 0.0f, live ? masked_weight(t, 1) : 0.0f,
                         live ? masked_weight(t, 2) : 0.0f, live ? masked_weight(t, 3) : 0.0f};
    float4* __restrict__ mine = reinterpret_cast<float4*>(strip[lane]);
    reinterpret_cast<int4*>(strip[lane])[8] = make_int4(t.o00 * C, t.o01 * C, t.o10 * C, t.o11 * C);
    if constexpr (C == 8) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mine[2 * k + 0] = make_float4(dwarp[0] * wt[k], dwarp[1] * wt[k], dwarp[2] * wt[k], dwarp[3] * wt[k]);
            mine[2 * k + 1] = make_float4(dwarp[4] * wt[k], dwarp[5] * wt[k], dwarp[6] * wt[k], dwarp[7] * wt[k]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 4
        for (int it = 0; it < 32; ++it) {
            const int e = it * 64 + lane;
            const int sl = e >> 5, slot = e & 31;
            const float v = strip[sl][slot];
            const int off = reinterpret_cast<const int*>(strip[sl])[32 + (slot >> 3)];
            if (v != 0.0f) atomicAdd(gsrcb + off + (slot & 7), v);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
#pragma unroll
        for (int row = 0; row < 2; ++row) {
#pragma unroll
            for (int cb = 0; cb < C / 16; ++cb) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float wk = wt[2 * row + u];
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int c = 16 * cb + 4 * q4;
                        mine[4 * u + q4] = make_float4(dwarp[c] * wk, dwarp[c + 1] * wk, dwarp[c + 2] * wk, dwarp[c + 3] * wk);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 4
                for (int it = 0; it < 32; ++it) {
                    const int e = it * 64 + lane;
                    const int sl = e >> 5, slot = e & 31;
                    const float v = strip[sl][slot];
                    const int off = reinterpret_cast<const int*>(strip[sl])[32 + 2 * row + (slot >> 4)];
                    if (v != 0.0f) atomicAdd(gsrcb + off + 16 * cb + (slot & 15), v);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
}


// ---- C <= 16, one lane per pixel: merge the D hypotheses of a pixel before touching memory ---------------------
// At the fine stages the hypotheses of one pixel project a fraction of a pixel apart, so their 2x2 footprints share
// 64-byte lines; adding each hypothesis separately issues one atomic request per (hypothesis, row, line).
// Every lane therefore owns a 2-row x 4-pixel x C-channel window (whole 64-byte lines) in its
// LDS strip, accumulates the hypotheses that fall inside it with plain LDS read-modify-writes, and the wave flushes
// windows cooperatively: lane l adds dword l (+64, ...) of the window, so one wave-instruction covers whole lines of
// one source lane.  A hypothesis that leaves the window (other row, > 2 px away) first triggers a flush of that
// lane's window (wave-uniform branch on __any), then re-bases it.
template <int C>
struct Window {                                  // 2 rows x 4 pixels x C channels per lane
    static constexpr int kRowDwords = 4 * C;     // one window row: 128 B (C=8, two lines) / 256 B (C=16, four lines)
    static constexpr int kDwords = 2 * kRowDwords;
    static constexpr int kStrip = kDwords + 4;   // + base offset + flush flag (+2 pad, keeps 16-byte alignment)
    static constexpr int kAlign = 16 / C;        // pixels per 64-byte line: window base is line-aligned in x
};

template <int C>
__device__ __forceinline__ void window_zero(float* __restrict__ mine) {
    float4* __restrict__ w4 = reinterpret_cast<float4*>(mine);
#pragma unroll
    for (int q = 0; q < Window<C>::kDwords / 4; ++q) w4[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// accumulate one hypothesis (4 taps x C channels) into the lane's window at pixel offset k (0..2)
template <int C>
__device__ __forceinline__ void window_add(float* __restrict__ mine, int k, const Taps& t, const float (&dwarp)[C]) {
    float4* __restrict__ w4 = reinterpret_cast<float4*>(mine);
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
        const float wgt = masked_weight(t, tap);
        if (wgt != 0.0f) {
            const int q = (((tap >> 1) * 4) + k + (tap & 1)) * (C / 4);
#pragma unroll
            for (int i = 0; i < C / 4; ++i) {
                float4 a = w4[q + i];
                a.x = fmaf(dwarp[4 * i + 0], wgt, a.x);
                a.y = fmaf(dwarp[4 * i + 1], wgt, a.y);
                a.z = fmaf(dwarp[4 * i + 2], wgt, a.z);
                a.w = fmaf(dwarp[4 * i + 3], wgt, a.w);
                w4[q + i] = a;
            }
        }
    }
}


// flush of the several-lanes-per-pixel kernels: source lanes HS*k .. HS*k+HS-1 hold the HS channel parts of the same
// window; one wave-instruction adds 64 contiguous dwords of the pixel's window: one row (4 pixels x 16 channels) for
// C = 16, both rows (2 x 4 pixels x 8 channels) for C = 8.
template <int C, int HS>
__device__ __forceinline__ void split_window_flush(float (*strip)[Window<C / HS>::kStrip], int lane, float* __restrict__ gsrcb,
                                                   int row_stride) {
    static_assert(C == 16 || C == 8, "split flush: whole 64-byte lines per wave-instruction");
    constexpr int CL = C / HS;
    using Wn = Window<CL>;
    constexpr int RD = 4 * C;                 // dwords of one window row of the pixel in memory
    constexpr int RPI = kWave / RD;           // window rows per wave-instruction (1 or 2)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int in_row = lane % RD, sub_row = lane / RD;
    const int px = in_row / C, ch = in_row % C, part = ch / CL, c = ch % CL;
    for (int k = 0; k < kWave / HS; ++k) {
        const int2 meta = *reinterpret_cast<const int2*>(&strip[HS * k][Wn::kDwords]);   // same address in every lane
        if (__builtin_amdgcn_readfirstlane(meta.y) == 0) continue;                      // wave-uniform
#pragma unroll
        for (int step = 0; step < 2 / RPI; ++step) {
            const int row = step * RPI + sub_row;
            const float v = strip[HS * k + part][row * Wn::kRowDwords + px * CL + c];
            if (v != 0.0f) atomicAdd(gsrcb + (meta.x + row * row_stride + in_row), v);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------- backward, channels-last features: dense atomics
// Fallback route without workspace (MVSTER_BWD_ROUTE_DENSE): one lane per (pixel, hypothesis), the source-feature
// gradient leaves through dense_scatter().
template <int C, int G, int D, int DPL>
__global__ void __launch_bounds__(kBlock)
cost_volume_bwd_nhwc_kernel(const float* __restrict__ gvol, const float* __restrict__ vol,
                            const float* __restrict__ wsum, const float* __restrict__ ref, SrcPtrs srcs,
                            const float* __restrict__ pose, const float* __restrict__ hypo, float* __restrict__ gref,
                            GradPtrs gsrcs, VolumeDims dm) {
    constexpr int CPG = C / G;
    constexpr int PXB = kBlock / D;
    static_assert(DPL == 1, "channels-last: one lane per (pixel, hypothesis); D = 4 with C <= 16 runs the split kernel below");
    __shared__ __attribute__((aligned(16))) float stage[kBlock / kWave][kWave][kStageRow];   // wave-private strips
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
    float refv[C];
    {
        const float4* __restrict__ refp = reinterpret_cast<const float4*>(ref + ((size_t)b * HW + p) * C);
#pragma unroll
        for (int i = 0; i < C / 4; ++i) {
            const float4 r = refp[i];
            refv[4 * i] = r.x; refv[4 * i + 1] = r.y; refv[4 * i + 2] = r.z; refv[4 * i + 3] = r.w;
            ref2[2 * i] = f32x2{r.x, r.y} * (1.0f / CPG);
            ref2[2 * i + 1] = f32x2{r.z, r.w} * (1.0f / CPG);
        }
    }
    const float hyp = hypo[((size_t)b * D + d) * HW + p];
    const float rden = div_fast(1.0f, wsum[((size_t)b * D + d) * HW + p]);
    float gd[G], godot = 0.0f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const size_t at = (((size_t)b * G + g) * D + d) * HW + p;
        gd[g] = live ? gvol[at] * rden : 0.0f;            // a padding lane must contribute nothing to the atomics below
        godot = fmaf(gd[g], vol[at], godot);
    }
    float dref[C];
#pragma unroll
    for (int c = 0; c < C; ++c) dref[c] = 0.0f;

    for (int s = 0; s < dm.S; ++s) {
        const float* __restrict__ srcb = srcs.p[s] + (size_t)b * HWs * C;
        float* __restrict__ gsrcb = gsrcs.p[s] + (size_t)b * HWs * C;
        const Pose q = load_pose(pose + ((size_t)b * dm.S + s) * 12);
        const Landing l = land(q, rotate_pixel(q, fx, fy), hyp, dm.Hs, dm.Ws);
        const Taps t = taps_of(l, dm.Hs, dm.Ws);
        unsigned o[4];
        float w4[4];
        tap_addresses<C>(l, dm.Hs, dm.Ws, o, w4);
        float cor[G];
        f32x2 warped[C / 2];
        gather_correlate<C, CPG, true>(srcb, o, w4, ref2, cor, warped);
        float sg = cor[0];
#pragma unroll
        for (int g = 1; g < G; ++g) sg += cor[g];
        const float a = div_temp(sg, dm);
        const float m = group_max<D>(a);
        const float e = exp_nonpos(a - m);
        const float what = div_fast(e, group_sum<D>(e));
        float dw = -godot;
#pragma unroll
        for (int g = 0; g < G; ++g) dw = fmaf(gd[g], cor[g], dw);
        const float sdot = group_sum<D>(what * dw);
        const float da = what * (dw - sdot) * dm.inv_tc;
        const float w = div_sqrt_c(what, dm);
        float dwarp[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float dcor = fmaf(gd[c / CPG], w, da) * (1.0f / CPG);
            dref[c] = fmaf(dcor, c % 2 == 0 ? warped[c / 2].x : warped[c / 2].y, dref[c]);
            dwarp[c] = dcor * refv[c];
        }
        dense_scatter<C>(stage[threadIdx.x >> 6], lane, t, dwarp, gsrcb, live);
    }
    float4* __restrict__ grefp = reinterpret_cast<float4*>(gref + ((size_t)b * HW + p) * C);
#pragma unroll
    for (int c4 = 0; c4 < C / 4; ++c4) {
        float4 v;
        v.x = group_sum<D>(dref[4 * c4 + 0]);
        v.y = group_sum<D>(dref[4 * c4 + 1]);
        v.z = group_sum<D>(dref[4 * c4 + 2]);
        v.w = group_sum<D>(dref[4 * c4 + 3]);
        if (live && d == 0) grefp[c4] = v;
    }
}

// ------------------------------------------------------------------------------- coefficient kernel (deferred routes), C <= 16
// First half of the binned / tiled-scatter routes: everything that needs the feature gathers -- per-view correlations,
// softmax, the reference gradient (each lane owns its pixel: plain stores) -- and, per (view, pixel, hypothesis), the G
// group coefficients dcor (gradient w.r.t. the warped feature = dcor[group(c)] * ref[c]) into `dcor_out` [B,S,HW,D,G];
// the source-feature gradient is formed by the owner kernels of cost_volume_binned.hip (or tile_scatter_kernel) from
// them.  With bf.cursor the kernel also takes the FILL pass of the counting sort: the head of each run of equal bins
// claims the run's slots with one returning atomic, consumed one view later so that the atomic's round trip hides behind
// the next view's gathers.
// Round 3: ~170 VALU instructions per (sample, view) instead of ~500 (same measures as the forward kernel).
template <int C, int G, int D>
__global__ void __launch_bounds__(kBlock, C == 8 ? 5 : 1)          // C = 8: keep five waves per SIMD (<= 96 VGPRs)
cost_volume_bwd_coef_kernel(const float* __restrict__ gvol, const float* __restrict__ vol,
                            const float* __restrict__ wsum, const float* __restrict__ ref, SrcPtrs srcs,
                            const float* __restrict__ pose, const float* __restrict__ hypo, float* __restrict__ gref,
                            VolumeDims dm, float* __restrict__ dcor_out, BinFill bf) {
    static_assert(C <= 16, "C >= 32 runs cost_volume_bwd_coef_split_kernel");
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
            ref2[2 * i] = f32x2{r.x, r.y} * (1.0f / CPG);
            ref2[2 * i + 1] = f32x2{r.z, r.w} * (1.0f / CPG);
        }
    }
    const unsigned at = (unsigned)d * (unsigned)HW + (unsigned)p;       // inside one batch item of a [B,*,D,H,W] tensor
    const float hyp = hypo[(size_t)b * D * HW + at];
    const float rden = div_fast(1.0f, wsum[(size_t)b * D * HW + at]);
    float gd[G], godot = 0.0f;
    {
        const float* __restrict__ gvolb = gvol + (size_t)b * G * D * HW;
        const float* __restrict__ volb = vol + (size_t)b * G * D * HW;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const unsigned ag = at + (unsigned)(g * D) * (unsigned)HW;
            gd[g] = gvolb[ag] * rden;
            godot = fmaf(gd[g], volb[ag], godot);
        }
    }
    f32x2 dref[C / 2];
#pragma unroll
    for (int i = 0; i < C / 2; ++i) dref[i] = f32x2{0.0f, 0.0f};

    const LaneMasks lm = lane_masks(lane);
    const int item = pack_item(own.x, own.y, d);
    __shared__ BinAgg agg;
    __shared__ int places[kAggViews][kBlock];
    const bool filling = bf.cursor != nullptr;
    if (filling) agg_reset(agg);
    for (int s = 0; s < dm.S; ++s) {
        const float* __restrict__ srcb = srcs.p[s] + (size_t)b * HWs * C;
        const Pose q = load_pose(pose + ((size_t)b * dm.S + s) * 12);
        const Landing l = land(q, rotate_pixel(q, fx, fy), hyp, dm.Hs, dm.Ws);
        if (filling)                                                             // wave-uniform
            places[s % kAggViews][threadIdx.x] = agg_claim<0>(agg, s % kAggViews, bf, bf.cursor + ((size_t)b * dm.S + s) * bf.tiles_per_view * kBinCats,
                                                              l, live, lane, lm, item);
        unsigned o[4];
        float w4[4];
        tap_addresses<C>(l, dm.Hs, dm.Ws, o, w4);
        float cor[G];
        f32x2 warped[C / 2];
        gather_correlate<C, CPG, true>(srcb, o, w4, ref2, cor, warped);
        float sg = cor[0];
#pragma unroll
        for (int g = 1; g < G; ++g) sg += cor[g];
        const float a = div_temp(sg, dm);
        const float m = group_max<D>(a);
        const float e = exp_nonpos(a - m);
        const float what = div_fast(e, group_sum<D>(e));
        float dw = -godot;
#pragma unroll
        for (int g = 0; g < G; ++g) dw = fmaf(gd[g], cor[g], dw);
        const float sdot = group_sum<D>(what * dw);
        const float da = what * (dw - sdot) * dm.inv_tc;
        const float w = div_sqrt_c(what, dm);
        float coef[G];
#pragma unroll
        for (int g = 0; g < G; ++g) coef[g] = fmaf(gd[g], w, da) * (1.0f / CPG);
#pragma unroll
        for (int i = 0; i < C / 2; ++i) {
            const f32x2 cf = {coef[(2 * i) / CPG], coef[(2 * i + 1) / CPG]};
            dref[i] = __builtin_elementwise_fma(cf, warped[i], dref[i]);
        }
        if (live) {
            float4* __restrict__ out = reinterpret_cast<float4*>(dcor_out + ((((size_t)b * dm.S + s) * HW + p) * D + d) * G);
#pragma unroll
            for (int g4 = 0; g4 < G / 4; ++g4) out[g4] = make_float4(coef[4 * g4], coef[4 * g4 + 1], coef[4 * g4 + 2], coef[4 * g4 + 3]);
        }
        if (filling && (s % kAggViews == kAggViews - 1 || s == dm.S - 1)) {      // the batch's slots: one claim per occupied entry
            agg_flush_claims(agg, bf.cursor, bf.tiles_per_view * kBinCats, b, dm.S, s - s % kAggViews);
            for (int v = 0; v <= s % kAggViews; ++v) agg_write_item<0>(agg, v, bf, places[v][threadIdx.x], lane, item);
        }
    }
    // reference gradient: sum over the D hypothesis lanes of the pixel, lane d stores channels [d C/D', ...)
    float4* __restrict__ grefp = reinterpret_cast<float4*>(gref + ((size_t)b * HW + p) * C);
    float* __restrict__ gref_planes = gref + (size_t)b * C * HW + p;            // MVSTER_GRAD_PLANAR: [B,C,H,W]
#pragma unroll
    for (int c4 = 0; c4 < C / 4; ++c4) {
        float4 v;
        v.x = group_sum<D>(dref[2 * c4].x);
        v.y = group_sum<D>(dref[2 * c4].y);
        v.z = group_sum<D>(dref[2 * c4 + 1].x);
        v.w = group_sum<D>(dref[2 * c4 + 1].y);
        if (live && d == c4 % D) {                                              // spread the stores over the pixel's lanes
            if (dm.grad_planar) {                                                // wave-uniform
                gref_planes[(size_t)(4 * c4 + 0) * HW] = v.x;
                gref_planes[(size_t)(4 * c4 + 1) * HW] = v.y;
                gref_planes[(size_t)(4 * c4 + 2) * HW] = v.z;
                gref_planes[(size_t)(4 * c4 + 3) * HW] = v.w;
            } else {
                grefp[c4] = v;
            }
        }
    }
}


// ------------------------------------------------------------------------------- backward, several lanes per pixel (C = 16, 8)
// With 16 channels the one-lane-per-pixel kernel above needs a 128-dword window per lane (135 KB of LDS per workgroup)
// and 256 VGPRs: one wave per SIMD, and it falls 35 % short of the atomic request ceiling.  Here a pixel is shared by
// HS adjacent lanes, each owning C / HS channels (whole correlation groups): the per-lane window shrinks accordingly,
// several workgroups fit a CU, and only the sums over ALL groups (softmax input, dw) cross the lanes of a pixel (xor
// shuffles).  The flush walks the lane groups so that one wave-instruction still covers 256 contiguous bytes of whole
// 64-byte lines.  Measured (B = 4): C = 16 with two lanes per pixel 256 VGPRs, 8 waves per CU, 3.30 ms (17.2 G atomic
// requests/s); with four lanes 154 VGPRs, 12 waves per CU, 2.93 ms (19.3 G/s).  C = 8: two lanes (a lane loads float4s).
template <int C>
constexpr int kSplitLanes = (C == 16) ? 4 : 2;    // C = 16: four lanes per pixel (4 channels = one float4 per tap and lane)

template <int C, int G, int D, int HS>           // HS = lanes per pixel (2 or 4)
__global__ void __launch_bounds__(kBlock, 2)      // <= 36 KB of LDS per workgroup (HS = 4: 18 KB): registers decide, keep >= 2 workgroups per CU
cost_volume_bwd_nhwc_split_kernel(const float* __restrict__ gvol, const float* __restrict__ vol,
                                  const float* __restrict__ wsum, const float* __restrict__ ref, SrcPtrs srcs,
                                  const float* __restrict__ pose, const float* __restrict__ hypo,
                                  float* __restrict__ gref, GradPtrs gsrcs, VolumeDims dm) {
    static_assert(HS == 2 || HS == 4, "two or four lanes per pixel");
    constexpr int CL = C / HS;            // channels per lane
    constexpr int CPG = C / G;
    constexpr int GL = G / HS;            // correlation groups per lane
    constexpr int CL4 = CL / 4;
    static_assert((C == 16 || C == 8) && CL % CPG == 0 && GL >= 1, "split kernel: whole correlation groups per lane");
    constexpr int PXB = kBlock / HS;
    using Wn = Window<CL>;
    __shared__ __attribute__((aligned(16))) float stage[kBlock / kWave][kWave][Wn::kStrip];
    const int tile = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    if (tile >= dm.ntiles) return;
    const int b = blockIdx.y;
    const int HW = dm.H * dm.W, HWs = dm.Hs * dm.Ws;
    const int h = threadIdx.x % HS, pl = threadIdx.x / HS;
    const int praw = tile * PXB + pl;
    const bool live = praw < HW;
    const int p = live ? praw : HW - 1;
    const int y = p / dm.W, x = p - y * dm.W;
    const float fx = (float)x, fy = (float)y;
    const float inv_tc = dm.inv_tc;
    const int lane = threadIdx.x & 63;
    float (*strip)[Wn::kStrip] = stage[threadIdx.x >> 6];
    float* __restrict__ mine = strip[lane];
    const int row_stride = dm.Ws * C;

    const float4* __restrict__ refp = reinterpret_cast<const float4*>(ref + ((size_t)b * HW + p) * C + h * CL);
    float hyp[D], gd[GL][D], godot[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        hyp[d] = hypo[((size_t)b * D + d) * HW + p];
        const float rden = div_fast(1.0f, wsum[((size_t)b * D + d) * HW + p]);
        float acc = 0.0f;
#pragma unroll
        for (int g = 0; g < GL; ++g) {
            const size_t at = (((size_t)b * G + h * GL + g) * D + d) * HW + p;
            gd[g][d] = live ? gvol[at] * rden : 0.0f;
            acc = fmaf(gd[g][d], vol[at], acc);
        }
        godot[d] = group_sum<HS>(acc);                       // sum over ALL groups: every lane of the pixel
    }
    float dref[CL];
#pragma unroll
    for (int c = 0; c < CL; ++c) dref[c] = 0.0f;

    for (int s = 0; s < dm.S; ++s) {
        const float* __restrict__ srcb = srcs.p[s] + (size_t)b * HWs * C + h * CL;
        float* __restrict__ gsrcb = gsrcs.p[s] + (size_t)b * HWs * C;
        const Pose q = load_pose(pose + ((size_t)b * dm.S + s) * 12);
        const Ray ray = rotate_pixel(q, fx, fy);
        Taps taps[D];
        float warped[CL][D], cor[GL][D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            taps[d] = project_taps(q, ray, hyp[d], dm.Hs, dm.Ws);
            const Taps& t = taps[d];
            const float4* __restrict__ p00 = reinterpret_cast<const float4*>(srcb + (size_t)t.o00 * C);
            const float4* __restrict__ p01 = reinterpret_cast<const float4*>(srcb + (size_t)t.o01 * C);
            const float4* __restrict__ p10 = reinterpret_cast<const float4*>(srcb + (size_t)t.o10 * C);
            const float4* __restrict__ p11 = reinterpret_cast<const float4*>(srcb + (size_t)t.o11 * C);
            float cg[GL];
#pragma unroll
            for (int g = 0; g < GL; ++g) cg[g] = 0.0f;
#pragma unroll
            for (int c4 = 0; c4 < CL4; ++c4) {
                const float4 a = p00[c4], bq = p01[c4], cq = p10[c4], dq = p11[c4];
                const float4 r = refp[c4];
                warped[4 * c4 + 0][d] = fmaf(dq.x, t.w11, fmaf(cq.x, t.w10, fmaf(bq.x, t.w01, a.x * t.w00)));
                warped[4 * c4 + 1][d] = fmaf(dq.y, t.w11, fmaf(cq.y, t.w10, fmaf(bq.y, t.w01, a.y * t.w00)));
                warped[4 * c4 + 2][d] = fmaf(dq.z, t.w11, fmaf(cq.z, t.w10, fmaf(bq.z, t.w01, a.z * t.w00)));
                warped[4 * c4 + 3][d] = fmaf(dq.w, t.w11, fmaf(cq.w, t.w10, fmaf(bq.w, t.w01, a.w * t.w00)));
                cg[(4 * c4 + 0) / CPG] = fmaf(warped[4 * c4 + 0][d], r.x, cg[(4 * c4 + 0) / CPG]);
                cg[(4 * c4 + 1) / CPG] = fmaf(warped[4 * c4 + 1][d], r.y, cg[(4 * c4 + 1) / CPG]);
                cg[(4 * c4 + 2) / CPG] = fmaf(warped[4 * c4 + 2][d], r.z, cg[(4 * c4 + 2) / CPG]);
                cg[(4 * c4 + 3) / CPG] = fmaf(warped[4 * c4 + 3][d], r.w, cg[(4 * c4 + 3) / CPG]);
            }
#pragma unroll
            for (int g = 0; g < GL; ++g) cor[g][d] = cg[g] * (1.0f / CPG);
        }
        float a[D], m = -INFINITY;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float sg = cor[0][d];
#pragma unroll
            for (int g = 1; g < GL; ++g) sg += cor[g][d];
            // the reference sums the groups in order 0..G-1: lane h of the pixel holds the h-th part
            if constexpr (HS == 2) {
                const float other = __shfl_xor(sg, 1, kWave);
                sg = h == 0 ? sg + other : other + sg;
            } else {
                const int first = lane & ~(HS - 1);
                float tot = __shfl(sg, first, kWave);
#pragma unroll
                for (int i = 1; i < HS; ++i) tot += __shfl(sg, first + i, kWave);
                sg = tot;
            }
            a[d] = div_temp(sg, dm);
            m = fmaxf(m, a[d]);
        }
        float e[D], sum = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            e[d] = exp_nonpos(a[d] - m);
            sum += e[d];
        }
        float what[D], dw[D], sdot = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            what[d] = div_fast(e[d], sum);
            float acc = 0.0f;
#pragma unroll
            for (int g = 0; g < GL; ++g) acc = fmaf(gd[g][d], cor[g][d], acc);
            dw[d] = group_sum<HS>(acc) - godot[d];
            sdot = fmaf(what[d], dw[d], sdot);
        }
        bool have = false;
        int xb = 0, yb = 0;
        window_zero<CL>(mine);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float da = what[d] * (dw[d] - sdot) * inv_tc;
            const float w = div_sqrt_c(what[d], dm);
            float dwarp[CL];
#pragma unroll
            for (int c4 = 0; c4 < CL4; ++c4) {
                const float4 r = refp[c4];
                const float rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = 4 * c4 + i;
                    const float dcor = fmaf(gd[c / CPG][d], w, da) * (1.0f / CPG);
                    dref[c] = fmaf(dcor, warped[c][d], dref[c]);
                    dwarp[c] = dcor * rr[i];
                }
            }
            const Taps& t = taps[d];
            const bool active = live && t.mask != 0;
            const int k = t.ix - xb;
            const bool fits = have && t.iy == yb && k >= 0 && k <= 2;
            const bool spill = active && have && !fits;          // identical in all lanes of a pixel
            if (__any(spill)) {
                *reinterpret_cast<int2*>(&mine[Wn::kDwords]) = make_int2((yb * dm.Ws + xb) * C, spill ? 1 : 0);
                split_window_flush<C, HS>(strip, lane, gsrcb, row_stride);
                if (spill) {
                    window_zero<CL>(mine);
                    have = false;
                }
            }
            if (active) {
                if (!have) {
                    constexpr int A = 16 / C;              // pixels per 64-byte line in memory: the window base is line-aligned
                    xb = (taps[D - 1].ix < taps[0].ix) ? (((t.ix + 2 + (A - 1)) & ~(A - 1)) - 4) : (t.ix & ~(A - 1));
                    yb = t.iy;
                    have = true;
                }
                window_add<CL>(mine, t.ix - xb, t, dwarp);
            }
        }
        *reinterpret_cast<int2*>(&mine[Wn::kDwords]) = make_int2((yb * dm.Ws + xb) * C, have ? 1 : 0);
        split_window_flush<C, HS>(strip, lane, gsrcb, row_stride);
    }
    float4* __restrict__ grefp = reinterpret_cast<float4*>(gref + ((size_t)b * HW + p) * C + h * CL);
    if (live) {
#pragma unroll
        for (int c4 = 0; c4 < CL4; ++c4)
            grefp[c4] = make_float4(dref[4 * c4 + 0], dref[4 * c4 + 1], dref[4 * c4 + 2], dref[4 * c4 + 3]);
    }
}

// ------------------------------------------------------------------------------- coefficient kernel, C >= 32: channel-split lanes
// The coefficient kernel above with a sample shared by HS = C / 16 adjacent lanes of 16 channels each (see
// cost_volume_fwd_split_kernel): 284 -> ~130 VGPRs at C = 64.  Only the sums over ALL correlation groups (softmax input,
// dw, godot) cross the lanes of a sample; every lane stores the coefficients of its own groups and the reference gradient
// of its own channels.  Lanes of a pixel: [d][h], h fastest.  Always deferred (coefficients out, no scatter here).
template <int C, int G, int D, int HS>
__global__ void __launch_bounds__(kBlock)
cost_volume_bwd_coef_split_kernel(const float* __restrict__ gvol, const float* __restrict__ vol,
                                  const float* __restrict__ wsum, const float* __restrict__ ref, SrcPtrs srcs,
                                  const float* __restrict__ pose, const float* __restrict__ hypo, float* __restrict__ gref,
                                  VolumeDims dm, float* __restrict__ dcor_out, BinFill bf) {
    constexpr int CPG = C / G, CL = C / HS, GL = G / HS;
    static_assert(CL == 16 && CL % CPG == 0 && GL >= 1, "16 channels = whole correlation groups per lane");
    constexpr int LPP = D * HS;
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
    const float rden = div_fast(1.0f, wsum[((size_t)b * D + d) * HW + p]);
    float gd[GL], godot = 0.0f;
#pragma unroll
    for (int g = 0; g < GL; ++g) {
        const size_t at = (((size_t)b * G + h * GL + g) * D + d) * HW + p;
        gd[g] = gvol[at] * rden;
        godot = fmaf(gd[g], vol[at], godot);
    }
    godot = strided_sum<HS, 1>(godot);
    f32x2 dref[CL / 2];
#pragma unroll
    for (int i = 0; i < CL / 2; ++i) dref[i] = f32x2{0.0f, 0.0f};

    const LaneMasks lm = lane_masks(lane);
    const int item = pack_item(x, y, d);
    __shared__ BinAgg agg;
    __shared__ int places[kAggViews][kBlock];
    const bool filling = bf.cursor != nullptr;
    if (filling) agg_reset(agg);
    for (int s = 0; s < dm.S; ++s) {
        const float* __restrict__ srcb = srcs.p[s] + (size_t)b * HWs * C + h * CL;
        const Pose q = load_pose(pose + ((size_t)b * dm.S + s) * 12);
        const Landing l = land(q, rotate_pixel(q, (float)x, (float)y), hyp, dm.Hs, dm.Ws);
        if (filling)                                                             // wave-uniform
            places[s % kAggViews][threadIdx.x] = agg_claim<HSLOG>(agg, s % kAggViews, bf, bf.cursor + ((size_t)b * dm.S + s) * bf.tiles_per_view * kBinCats,
                                                                  l, live, lane, lm, item);
        unsigned o[4];
        float w4[4];
        tap_addresses<C>(l, dm.Hs, dm.Ws, o, w4);
        float cor[GL];
        f32x2 warped[CL / 2];
        gather_correlate<CL, CPG, true>(srcb, o, w4, ref2, cor, warped);
        float sg = cor[0], dwl = gd[0] * cor[0];
#pragma unroll
        for (int g = 1; g < GL; ++g) {
            sg += cor[g];
            dwl = fmaf(gd[g], cor[g], dwl);
        }
        sg = strided_sum<HS, 1>(sg);
        const float dw = strided_sum<HS, 1>(dwl) - godot;
        const float a = div_temp(sg, dm);
        const float m = strided_max<D, HS>(a);
        const float e = exp_nonpos(a - m);
        const float what = div_fast(e, strided_sum<D, HS>(e));
        const float sdot = strided_sum<D, HS>(what * dw);
        const float da = what * (dw - sdot) * dm.inv_tc;
        const float w = div_sqrt_c(what, dm);
        float coef[GL];
#pragma unroll
        for (int g = 0; g < GL; ++g) coef[g] = fmaf(gd[g], w, da) * (1.0f / CPG);
#pragma unroll
        for (int i = 0; i < CL / 2; ++i) {
            const f32x2 cf = {coef[(2 * i) / CPG], coef[(2 * i + 1) / CPG]};
            dref[i] = __builtin_elementwise_fma(cf, warped[i], dref[i]);
        }
        if (live) {
            float* __restrict__ out = dcor_out + ((((size_t)b * dm.S + s) * HW + p) * D + d) * G + h * GL;
            if constexpr (GL == 4) *reinterpret_cast<float4*>(out) = make_float4(coef[0], coef[1], coef[2], coef[3]);
            else if constexpr (GL == 2) *reinterpret_cast<float2*>(out) = make_float2(coef[0], coef[1]);
            else out[0] = coef[0];
        }
        if (filling && (s % kAggViews == kAggViews - 1 || s == dm.S - 1)) {
            agg_flush_claims(agg, bf.cursor, bf.tiles_per_view * kBinCats, b, dm.S, s - s % kAggViews);
            for (int v = 0; v <= s % kAggViews; ++v) agg_write_item<HSLOG>(agg, v, bf, places[v][threadIdx.x], lane, item);
        }
    }
    // reference gradient: sum over the D hypothesis lanes of this channel part
    float4* __restrict__ grefp = reinterpret_cast<float4*>(gref + ((size_t)b * HW + p) * C + h * CL);
    float* __restrict__ gref_planes = gref + ((size_t)b * C + h * CL) * HW + p;
#pragma unroll
    for (int c4 = 0; c4 < CL / 4; ++c4) {
        float4 v;
        v.x = strided_sum<D, HS>(dref[2 * c4].x);
        v.y = strided_sum<D, HS>(dref[2 * c4].y);
        v.z = strided_sum<D, HS>(dref[2 * c4 + 1].x);
        v.w = strided_sum<D, HS>(dref[2 * c4 + 1].y);
        if (live && d == c4 % D) {
            if (dm.grad_planar) {
                gref_planes[(size_t)(4 * c4 + 0) * HW] = v.x;
                gref_planes[(size_t)(4 * c4 + 1) * HW] = v.y;
                gref_planes[(size_t)(4 * c4 + 2) * HW] = v.z;
                gref_planes[(size_t)(4 * c4 + 3) * HW] = v.w;
            } else {
                grefp[c4] = v;
            }
        }
    }
}

// deferred routes: coefficients first, then either the tiled LDS-merged atomic scatter of
// cost_volume_scatter.hip or the binned, atomics-free scatter of cost_volume_binned.hip
template <int CPG, int G, int D>
static int launch_bwd_deferred(const float* gvol, const float* vol, const float* wsum, const float* ref, const SrcPtrs& srcs,
                               const float* pose, const float* hypo, float* gref, const GradPtrs& gsrcs, int B,
                               VolumeDims dm, int route, float* workspace, const BinnedLayout& lay, const int* fwd_counts,
                               hipStream_t st) {
    constexpr int PXB = kBlock / D;
    dm.ntiles = PixelBlocks<PXB>::tiles(dm.H, dm.W);
    dim3 grid(((dm.ntiles + 7) / 8) * 8, B);
    float* const dcor = workspace + (route == MVSTER_BWD_ROUTE_BINNED ? lay.coef : 0);
    BinFill bf{};
    const bool fill_here = route == MVSTER_BWD_ROUTE_BINNED && fwd_counts != nullptr;
    if (fill_here) {
        // the forward pass counted: scan its counts now, let the coefficient kernel claim the slots and write the items
        if (int rc = launch_bin_scan(fwd_counts, workspace, lay, st)) return rc;
        int* const words = reinterpret_cast<int*>(workspace);
        bf.cursor = words + lay.cursor;
        bf.items = words + lay.items;
        bf.bg = bin_grid_for(CPG * G, dm.Hs, dm.Ws);
        bf.tiles_per_view = bf.bg.tiles_x * bf.bg.tiles_y;
        bf.max_items = (int)lay.max_items;
    }
    if constexpr (CPG * G >= 32) {
        constexpr int HS = CPG * G / 16;
        dm.ntiles = grid_for((long long)dm.H * dm.W, kBlock / (D * HS));
        dim3 grid2(((dm.ntiles + 7) / 8) * 8, B);
        hipLaunchKernelGGL((cost_volume_bwd_coef_split_kernel<CPG * G, G, D, HS>), grid2, dim3(kBlock), 0, st, gvol, vol, wsum,
                           ref, srcs, pose, hypo, gref, dm, dcor, bf);
    } else {
        hipLaunchKernelGGL((cost_volume_bwd_coef_kernel<CPG * G, G, D>), grid, dim3(kBlock), 0, st, gvol, vol, wsum,
                           ref, srcs, pose, hypo, gref, dm, dcor, bf);
    }
    if (int rc = launch_status()) return rc;
    if (route == MVSTER_BWD_ROUTE_BINNED)
        return launch_binned_scatter(ref, pose, hypo, gsrcs, B, CPG * G, G, D, dm, workspace, lay, fill_here, st);
    return launch_tile_scatter(dcor, ref, pose, hypo, gsrcs, B, CPG * G, G, D, dm, st);
}

template <int CPG, int G, int D, int DPL>
static int launch_bwd(const float* gvol, const float* vol, const float* wsum, const float* ref, const SrcPtrs& srcs,
                      const float* pose, const float* hypo, float* gref, const GradPtrs& gsrcs, int B, VolumeDims dm,
                      int layout, hipStream_t st) {
    constexpr int PXB = kBlock / (D / DPL);
    dm.ntiles = grid_for((long long)dm.H * dm.W, PXB);
    dim3 grid(((dm.ntiles + 7) / 8) * 8, B);
    if constexpr ((CPG * G == 16 || CPG * G == 8) && DPL == 4 && D == 4) {
        if (layout == MVSTER_LAYOUT_NHWC) {
            constexpr int HS = kSplitLanes<CPG * G>;                            // lanes per pixel
            dm.ntiles = grid_for((long long)dm.H * dm.W, kBlock / HS);
            dim3 grid2(((dm.ntiles + 7) / 8) * 8, B);
            hipLaunchKernelGGL((cost_volume_bwd_nhwc_split_kernel<CPG * G, G, 4, HS>), grid2, dim3(kBlock), 0, st, gvol, vol,
                               wsum, ref, srcs, pose, hypo, gref, gsrcs, dm);
            return launch_status();
        }
    }
    if (layout == MVSTER_LAYOUT_NHWC) {
        // DPL > 1 is only dispatched for D = 4, C <= 16, which returned into the split kernel above
        dm.ntiles = PixelBlocks<PXB>::tiles(dm.H, dm.W);
        grid = dim3(((dm.ntiles + 7) / 8) * 8, B);
        if constexpr (DPL == 1)
            hipLaunchKernelGGL((cost_volume_bwd_nhwc_kernel<CPG * G, G, D, DPL>), grid, dim3(kBlock), 0, st, gvol, vol,
                               wsum, ref, srcs, pose, hypo, gref, gsrcs, dm);
        else
            return MVSTER_ERR_UNSUPPORTED;
    } else
        hipLaunchKernelGGL((cost_volume_bwd_kernel<CPG, G, D, DPL>), grid, dim3(kBlock), 0, st, gvol, vol, wsum, ref,
                           srcs, pose, hypo, gref, gsrcs, dm);
    return launch_status();
}

}  // namespace mvster

extern "C" long long mvster_cost_volume_bwd_route_workspace(int route, int S, int B, int C, int G, int D, int H, int W,
                                                            int Hs, int Ws, int layout) {
    using namespace mvster;
    if (S <= 0 || B <= 0 || G <= 0 || D <= 0 || H <= 0 || W <= 0 || Hs <= 0 || Ws <= 0) return -1;
    if (!shape_instantiated(C, G, D)) return -1;
    BinnedLayout lay;
    switch (route) {
        case MVSTER_BWD_ROUTE_PLANAR: return layout == MVSTER_LAYOUT_NCHW ? 0 : -1;
        case MVSTER_BWD_ROUTE_DENSE:
            return layout == MVSTER_LAYOUT_NHWC && !pick_pixel_per_lane(B, H * W, C, D) ? 0 : -1;
        case MVSTER_BWD_ROUTE_SPLIT_WINDOW:
            return layout == MVSTER_LAYOUT_NHWC && pick_pixel_per_lane(B, H * W, C, D) ? 0 : -1;
        case MVSTER_BWD_ROUTE_TILE_SCATTER:
            return tile_scatter_applies(C, G, D, layout) ? (long long)B * S * H * W * D * G : -1;
        case MVSTER_BWD_ROUTE_BINNED:
            return layout == MVSTER_LAYOUT_NHWC && binned_layout(S, B, C, G, D, H, W, Hs, Ws, &lay) ? lay.total : -1;
        default: return -1;
    }
}

extern "C" long long mvster_cost_volume_bwd_workspace(int S, int B, int C, int G, int D, int H, int W, int Hs, int Ws,
                                                      int layout) {
    using namespace mvster;
    if (layout != MVSTER_LAYOUT_NHWC) return 0;
    long long need = -1;
    if (C >= kMinBinnedChannels) need = mvster_cost_volume_bwd_route_workspace(MVSTER_BWD_ROUTE_BINNED, S, B, C, G, D, H, W, Hs, Ws, layout);
    if (need < 0 && pick_pixel_per_lane(B, H * W, C, D)) return 0;         // the lane-split window kernel needs none
    if (need < 0) need = mvster_cost_volume_bwd_route_workspace(MVSTER_BWD_ROUTE_TILE_SCATTER, S, B, C, G, D, H, W, Hs, Ws, layout);
    return need < 0 ? 0 : need;
}

extern "C" int mvster_cost_volume_bwd_route(int S, int B, int C, int G, int D, int H, int W, int Hs, int Ws, int layout,
                                            long long workspace_floats) {
    return mvster::pick_bwd_route(S, B, C, G, D, H, W, Hs, Ws, layout, workspace_floats);
}

extern "C" int mvster_cost_volume_bwd_route_name(int route, int C, int G, int D, char* buf, int cap) {
    char tmp[160];
    int n = 0;
    switch (route) {
        case MVSTER_BWD_ROUTE_PLANAR: {
            const bool wide = D == 4 && C <= mvster::kMaxWideChannels;      // only when pick_pixel_per_lane() held
            n = snprintf(tmp, sizeof tmp, "cost_volume_bwd_kernel<%d, %d, %d, %s>", G ? C / G : 0, G, D, wide ? "1|4" : "1");
            break;
        }
        case MVSTER_BWD_ROUTE_DENSE:
            n = snprintf(tmp, sizeof tmp, "cost_volume_bwd_nhwc_kernel<%d, %d, %d, 1>", C, G, D);
            break;
        case MVSTER_BWD_ROUTE_SPLIT_WINDOW:
            n = snprintf(tmp, sizeof tmp, "cost_volume_bwd_nhwc_split_kernel<%d, %d, 4, %d>", C, G, C == 16 ? 4 : 2);
            break;
        case MVSTER_BWD_ROUTE_TILE_SCATTER:
            n = snprintf(tmp, sizeof tmp, "tile_scatter_kernel<128, 8>");
            break;
        case MVSTER_BWD_ROUTE_BINNED: {
            // every kernel of the route that moves real data, dominant first, ';'-separated
            char owner[48], coef[64];
            if (C == 8) snprintf(owner, sizeof owner, "tile_owner_c8_kernel<%d, %d>", G, D);
            else snprintf(owner, sizeof owner, "tile_owner_kernel<%d, %d, %d>", C, G, D);
            if (C >= 32) snprintf(coef, sizeof coef, "cost_volume_bwd_coef_split_kernel<%d, %d, %d, %d>", C, G, D, C / 16);
            else snprintf(coef, sizeof coef, "cost_volume_bwd_coef_kernel<%d, %d, %d>", C, G, D);
            n = snprintf(tmp, sizeof tmp, "%s;%s", owner, coef);
            break;
        }
        default:
            tmp[0] = 0;
    }
    if (buf != nullptr && cap > 0) snprintf(buf, (size_t)cap, "%s", tmp);
    return n;
}

extern "C" int mvster_cost_volume_bwd(const float* gvol, const float* vol, const float* wsum, const float* ref,
                                      const float* const* srcs, int S, const float* pose, const float* hypo,
                                      float* gref, float* const* gsrcs, int B, int C, int G, int D, int H, int W,
                                      int Hs, int Ws, float attn_temp, int layout, mvster_stream_t stream) {
    return mvster_cost_volume_bwd_ws(gvol, vol, wsum, ref, srcs, S, pose, hypo, gref, gsrcs, B, C, G, D, H, W, Hs, Ws,
                                     attn_temp, layout, nullptr, 0, nullptr, stream);
}

extern "C" int mvster_cost_volume_bwd_ws(const float* gvol, const float* vol, const float* wsum, const float* ref,
                                         const float* const* srcs, int S, const float* pose, const float* hypo,
                                         float* gref, float* const* gsrcs, int B, int C, int G, int D, int H, int W,
                                         int Hs, int Ws, float attn_temp, int layout, float* workspace,
                                         long long workspace_floats, const int32_t* fwd_bin_counts,
                                         mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(gvol); MVSTER_REQUIRE_PTR(vol); MVSTER_REQUIRE_PTR(wsum); MVSTER_REQUIRE_PTR(ref);
    MVSTER_REQUIRE_PTR(srcs); MVSTER_REQUIRE_PTR(pose); MVSTER_REQUIRE_PTR(hypo); MVSTER_REQUIRE_PTR(gref);
    MVSTER_REQUIRE_PTR(gsrcs);
    const bool grad_planar = (layout & MVSTER_GRAD_PLANAR) != 0;
    layout &= ~MVSTER_GRAD_PLANAR;
    if (int rc = check_volume(S, B, C, G, D, H, W, Hs, Ws, attn_temp, layout)) return rc;
    SrcPtrs sp{};
    GradPtrs gp{};
    for (int s = 0; s < S; ++s) {
        MVSTER_REQUIRE_PTR(srcs[s]); MVSTER_REQUIRE_PTR(gsrcs[s]);
        sp.p[s] = srcs[s];
        gp.p[s] = gsrcs[s];
    }
    const hipStream_t st = as_stream(stream);
    const int route = pick_bwd_route(S, B, C, G, D, H, W, Hs, Ws, layout, workspace != nullptr ? workspace_floats : 0);
    BinnedLayout lay{};
    if (route == MVSTER_BWD_ROUTE_BINNED) binned_layout(S, B, C, G, D, H, W, Hs, Ws, &lay);
    if (grad_planar && route != MVSTER_BWD_ROUTE_BINNED) return MVSTER_ERR_UNSUPPORTED;   // only the owner kernels write planes
    if (route != MVSTER_BWD_ROUTE_BINNED) {          // the binned route writes every gradient element itself
        for (int s = 0; s < S; ++s) {
            if (hipMemsetAsync(gp.p[s], 0, sizeof(float) * (size_t)B * C * Hs * Ws, st) != hipSuccess) {
                (void)hipGetLastError();
                return MVSTER_ERR_LAUNCH;
            }
        }
    }
    VolumeDims dm = make_dims(H, W, Hs, Ws, S, C, attn_temp);
    dm.grad_planar = grad_planar ? 1 : 0;
    const int cpg = C / G;
    const bool deferred = route == MVSTER_BWD_ROUTE_BINNED || route == MVSTER_BWD_ROUTE_TILE_SCATTER;
    const bool wide = route == MVSTER_BWD_ROUTE_SPLIT_WINDOW || (route == MVSTER_BWD_ROUTE_PLANAR && pick_pixel_per_lane(B, H * W, C, D));
#define DEFERRED(CPG_, G_, D_) \
    return launch_bwd_deferred<CPG_, G_, D_>(gvol, vol, wsum, ref, sp, pose, hypo, gref, gp, B, dm, route, workspace, lay, \
                                             fwd_bin_counts, st)
#define X(CPG_, G_)                                                                                                   \
    if (cpg == CPG_ && G == G_) {                                                                                     \
        if (deferred) {                                                                                               \
            if (D == 4) DEFERRED(CPG_, G_, 4);                                                                        \
            if (D == 8) DEFERRED(CPG_, G_, 8);                                                                        \
            if (D == 16) DEFERRED(CPG_, G_, 16);                                                                      \
        }                                                                                                             \
        if constexpr (CPG_ * G_ <= kMaxWideChannels) {                                                                \
            if (D == 4 && wide)                                                                                       \
                return launch_bwd<CPG_, G_, 4, 4>(gvol, vol, wsum, ref, sp, pose, hypo, gref, gp, B, dm, layout, st);         \
        }                                                                                                             \
        if (D == 4) return launch_bwd<CPG_, G_, 4, 1>(gvol, vol, wsum, ref, sp, pose, hypo, gref, gp, B, dm, layout, st);     \
        if (D == 8) return launch_bwd<CPG_, G_, 8, 1>(gvol, vol, wsum, ref, sp, pose, hypo, gref, gp, B, dm, layout, st);     \
        if (D == 16) return launch_bwd<CPG_, G_, 16, 1>(gvol, vol, wsum, ref, sp, pose, hypo, gref, gp, B, dm, layout, st);   \
    }
    MVSTER_FOR_SHAPES(X)
#undef X
#undef DEFERRED
    return MVSTER_ERR_UNSUPPORTED;
}
