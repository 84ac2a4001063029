// Shared declarations of the fused cost-volume kernels (cost_volume_fwd.hip / cost_volume_bwd.hip).
#pragma once

#include <math.h>

#include "mvster_device.h"
#include "mvster_host.h"

namespace mvster {

struct SrcPtrs {
    const float* p[MVSTER_MAX_SRC_VIEWS];
};
struct GradPtrs {
    float* p[MVSTER_MAX_SRC_VIEWS];
};

struct VolumeDims {
    int H, W, Hs, Ws, S, ntiles;
    float temp;     // attn_temp
    float sqrt_c;   // (float)sqrt(C), the reference divides the view weight by math.sqrt(C)
};

// (channels per group, groups) pairs instantiated: C in {8,16,32,64} x G in {4,8}
#define MVSTER_FOR_SHAPES(X) \
    X(2, 4) X(4, 4) X(8, 4) X(16, 4) X(1, 8) X(2, 8) X(4, 8) X(8, 8)

inline bool shape_instantiated(int C, int G, int D) {
    if (G <= 0 || C % G != 0) return false;
    const int cpg = C / G;
    const bool cg = (G == 4 && (cpg == 2 || cpg == 4 || cpg == 8 || cpg == 16)) ||
                    (G == 8 && (cpg == 1 || cpg == 2 || cpg == 4 || cpg == 8));
    return cg && (D == 4 || D == 8 || D == 16);
}

// Lane mapping.  FORWARD: always D lanes per pixel (DPL = 1): the kernel is gather-latency bound (SQ_WAIT_ANY 75 % of
// wave cycles with one lane per pixel at 132 VGPRs / 3 waves per SIMD); D lanes per pixel need 65 VGPRs, run 7 waves
// per SIMD and measured 1.55x faster at DTU stages 3-4.  BACKWARD: one lane per pixel (DPL = D) whenever D == 4,
// C <= 16 and there are enough pixels to fill the chip, because only then can a lane merge the hypotheses of its
// pixel before the atomic flush (stage 4: 19.0 ms with D lanes per pixel vs 7.8 ms); otherwise D lanes per pixel.
constexpr long long kMinPixelWaves = 8192;
constexpr int kMaxWideChannels = 16;

inline bool pick_pixel_per_lane(int B, int HW, int C, int D) {
    return D == 4 && C <= kMaxWideChannels && ((long long)B * HW) / kWave >= kMinPixelWaves;
}

// Tiled scatter of the deferred backward (cost_volume_scatter.hip): used for the stages with D = 8 hypotheses spread
// along the whole epipolar segment (no per-pixel merging possible).  Measured against the direct kernels (B x S views):
// DTU stage 2 (C = 32) 6.72 -> 3.31 ms, stage 1 (C = 64) 2.17 -> 1.47 ms; hi-res config stage 3 (768 x 1024, C = 16,
// 1 x 9 views) 8.76 -> 6.94 ms, stage 2 (384 x 512, C = 32, 1 x 9) 4.83 -> 2.93 ms (tools/probes/hires_bwd_probe.py).
inline bool tile_scatter_applies(int C, int G, int D, int layout) {
    return layout == MVSTER_LAYOUT_NHWC && D == 8 && (C == 16 || C == 32 || C == 64) && shape_instantiated(C, G, D);
}

// Which kernel family the backward entry point runs for a shape (mvster_cost_volume_bwd_route in mvster_hip.h).
inline int pick_bwd_route(int S, int B, int C, int G, int D, int H, int W, int layout, long long workspace_floats) {
    if (S <= 0 || B <= 0 || H <= 0 || W <= 0 || !shape_instantiated(C, G, D)) return MVSTER_BWD_ROUTE_NONE;
    if (layout != MVSTER_LAYOUT_NHWC) return MVSTER_BWD_ROUTE_PLANAR;
    if (tile_scatter_applies(C, G, D, layout) && workspace_floats >= (long long)B * S * H * W * D * G)
        return MVSTER_BWD_ROUTE_TILE_SCATTER;
    if (pick_pixel_per_lane(B, H * W, C, D)) return MVSTER_BWD_ROUTE_SPLIT_WINDOW;
    return MVSTER_BWD_ROUTE_DENSE;
}

int launch_tile_scatter(const float* dcor, const float* ref, const float* pose, const float* hypo, const GradPtrs& gsrcs,
                        int B, int C, int G, int D, const VolumeDims& dm, hipStream_t st);

inline int check_volume(int S, int B, int C, int G, int D, int H, int W, int Hs, int Ws, float temp, int layout) {
    MVSTER_REQUIRE_SHAPE(layout == MVSTER_LAYOUT_NCHW || layout == MVSTER_LAYOUT_NHWC);
    MVSTER_REQUIRE_SHAPE(B > 0 && C > 0 && G > 0 && D > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0 && S > 0);
    MVSTER_REQUIRE_SHAPE(B <= 65535 && C % G == 0 && temp != 0.0f);
    MVSTER_REQUIRE_SHAPE((long long)H * W <= kMaxPlane && (long long)Hs * Ws <= kMaxPlane);
    if (S > MVSTER_MAX_SRC_VIEWS) return MVSTER_ERR_UNSUPPORTED;
    if (!shape_instantiated(C, G, D)) return MVSTER_ERR_UNSUPPORTED;
    return MVSTER_OK;
}

}  // namespace mvster
