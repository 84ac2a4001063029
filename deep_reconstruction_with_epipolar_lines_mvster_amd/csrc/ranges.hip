// A5/A6  depth-hypothesis scheduling of the coarse-to-fine cascade.
//   reference: models/mvs4net_utils.py:69-77 (init_range), :79-85 (init_inverse_range),
//              :87-94 (schedule_inverse_range).
// Pure streaming writes of a [B,D,H,W] tensor: one lane per output pixel, the D planes of a pixel
// are produced by the same lane (the bilinear footprint is shared by all planes), stores of one
// plane are coalesced along the row.
#include "mvster_device.h"
#include "mvster_host.h"

namespace mvster {

__global__ void __launch_bounds__(kBlock)
init_range_kernel(const float* __restrict__ depth_values, int K, float* __restrict__ out, int D, int HW, int inverse) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= HW) return;
    const float dmin = depth_values[(size_t)b * K];
    const float dmax = depth_values[(size_t)b * K + (K - 1)];
    float* __restrict__ o = out + (size_t)b * D * HW + p;
    if (inverse) {
        const float imin = 1.0f / dmin, imax = 1.0f / dmax;       // (:80-81)
        const float span = imin - imax;
        for (int d = 0; d < D; ++d) {
            const float frac = (float)d / (float)(D - 1);         // arange / (ndepths-1) (:82)
            o[(size_t)d * HW] = 1.0f / (imax + span * frac);      // (:83-85)
        }
    } else {
        const float step = (dmax - dmin) / (float)(D - 1);        // (:72)
        for (int d = 0; d < D; ++d) o[(size_t)d * HW] = dmin + (float)d * step;   // (:74-75)
    }
}

// x2 upsample with align_corners=True: src = dst * (in-1)/(out-1), like ATen's area_pixel_compute_scale.
__global__ void __launch_bounds__(kBlock)
schedule_inverse_range_kernel(const float* __restrict__ inv_min, const float* __restrict__ inv_max,
                              float* __restrict__ out, int D, int Hp, int Wp, int H, int W) {
    const int b = blockIdx.y;
    const int HW = H * W;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const float sy = H > 1 ? (float)(Hp - 1) / (float)(H - 1) : 0.0f;
    const float sx = W > 1 ? (float)(Wp - 1) / (float)(W - 1) : 0.0f;
    const float ry = sy * (float)y, rx = sx * (float)x;
    const int y0 = min((int)ry, Hp - 1), x0 = min((int)rx, Wp - 1);
    const int y1 = y0 + (y0 < Hp - 1 ? 1 : 0), x1 = x0 + (x0 < Wp - 1 ? 1 : 0);
    const float ly1 = ry - (float)y0, ly0 = 1.0f - ly1;
    const float lx1 = rx - (float)x0, lx0 = 1.0f - lx1;
    const float* __restrict__ mn = inv_min + (size_t)b * Hp * Wp;
    const float* __restrict__ mx = inv_max + (size_t)b * Hp * Wp;
    const float mn00 = mn[y0 * Wp + x0], mn01 = mn[y0 * Wp + x1], mn10 = mn[y1 * Wp + x0], mn11 = mn[y1 * Wp + x1];
    const float mx00 = mx[y0 * Wp + x0], mx01 = mx[y0 * Wp + x1], mx10 = mx[y1 * Wp + x0], mx11 = mx[y1 * Wp + x1];
    float* __restrict__ o = out + (size_t)b * D * HW + p;
    for (int d = 0; d < D; ++d) {
        const float frac = (float)d / (float)(D - 1);             // (:90)
        // ramp at the coarse resolution first (:92), then interpolate (:93), then reciprocal (:94)
        const float v00 = mx00 + (mn00 - mx00) * frac, v01 = mx01 + (mn01 - mx01) * frac;
        const float v10 = mx10 + (mn10 - mx10) * frac, v11 = mx11 + (mn11 - mx11) * frac;
        const float v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
        o[(size_t)d * HW] = 1.0f / v;
    }
}

}  // namespace mvster

extern "C" int mvster_init_range(const float* depth_values, int K, float* out, int B, int D, int H, int W,
                                 int inverse, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(depth_values); MVSTER_REQUIRE_PTR(out);
    MVSTER_REQUIRE_SHAPE(B > 0 && B <= 65535 && K >= 1 && D >= 1 && H > 0 && W > 0 && (long long)H * W <= kMaxPlane);
    dim3 grid(grid_for((long long)H * W, kBlock), B);
    hipLaunchKernelGGL(init_range_kernel, grid, dim3(kBlock), 0, as_stream(stream), depth_values, K, out, D, H * W,
                       inverse);
    return launch_status();
}

extern "C" int mvster_schedule_inverse_range(const float* inv_min, const float* inv_max, float* out, int B, int D,
                                             int Hp, int Wp, int H, int W, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(inv_min); MVSTER_REQUIRE_PTR(inv_max); MVSTER_REQUIRE_PTR(out);
    MVSTER_REQUIRE_SHAPE(B > 0 && B <= 65535 && D >= 1 && Hp > 0 && Wp > 0 && H > 0 && W > 0);
    MVSTER_REQUIRE_SHAPE((long long)H * W <= kMaxPlane);
    dim3 grid(grid_for((long long)H * W, kBlock), B);
    hipLaunchKernelGGL(schedule_inverse_range_kernel, grid, dim3(kBlock), 0, as_stream(stream), inv_min, inv_max, out,
                       D, Hp, Wp, H, W);
    return launch_status();
}
