// A1  homo_warping as a standalone operator: [B,C,Hs,Ws] -> [B,C,D,Hr,Wr]  (and its src gradient).
//   reference: models/mvs4net_utils.py:21-67.
// One lane per reference pixel, consecutive lanes = consecutive pixels of a row, so that for one
// depth plane the 64 taps of a wave fall on a few neighbouring cache lines of the source plane
// (the homography is close to a shift along the epipolar line).  Loops over depth and channel;
// the footprint of a (pixel, depth) is computed once and reused for every channel.
// This operator materialises the warped volume and exists for API parity and for the rarely used
// stagenet modes; the training/inference hot path uses the fused kernels in cost_volume.hip.
#include "mvster_device.h"
#include "mvster_host.h"

namespace mvster {

__global__ void __launch_bounds__(kBlock)
warp_fwd_kernel(const float* __restrict__ src, const float* __restrict__ pose, const float* __restrict__ depth,
                float* __restrict__ out, int C, int D, int Hr, int Wr, int Hs, int Ws) {
    const int b = blockIdx.y;
    const int HW = Hr * Wr, HWs = Hs * Ws;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= HW) return;
    const int y = p / Wr, x = p - y * Wr;
    const Pose q = load_pose(pose + (size_t)b * 12);
    const Ray ray = rotate_pixel(q, (float)x, (float)y);
    const float* __restrict__ srcb = src + (size_t)b * C * HWs;
    const float* __restrict__ depb = depth + (size_t)b * D * HW + p;
    float* __restrict__ outb = out + (size_t)b * C * D * HW + p;
    for (int d = 0; d < D; ++d) {
        const Taps t = project_taps(q, ray, depb[(size_t)d * HW], Hs, Ws);
        for (int c = 0; c < C; ++c) outb[((size_t)c * D + d) * HW] = sample(srcb + (size_t)c * HWs, t);
    }
}

__global__ void __launch_bounds__(kBlock)
warp_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ pose, const float* __restrict__ depth,
                float* __restrict__ gsrc, int C, int D, int Hr, int Wr, int Hs, int Ws) {
    const int b = blockIdx.y;
    const int HW = Hr * Wr, HWs = Hs * Ws;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= HW) return;
    const int y = p / Wr, x = p - y * Wr;
    const Pose q = load_pose(pose + (size_t)b * 12);
    const Ray ray = rotate_pixel(q, (float)x, (float)y);
    float* __restrict__ gsrcb = gsrc + (size_t)b * C * HWs;
    const float* __restrict__ depb = depth + (size_t)b * D * HW + p;
    const float* __restrict__ goutb = gout + (size_t)b * C * D * HW + p;
    for (int d = 0; d < D; ++d) {
        const Taps t = project_taps(q, ray, depb[(size_t)d * HW], Hs, Ws);
        for (int c = 0; c < C; ++c) scatter(gsrcb + (size_t)c * HWs, t, goutb[((size_t)c * D + d) * HW]);
    }
}

static int check_warp(const void* a, const void* b, const void* c, const void* d, int B, int C, int D, int Hr, int Wr,
                      int Hs, int Ws) {
    MVSTER_REQUIRE_PTR(a); MVSTER_REQUIRE_PTR(b); MVSTER_REQUIRE_PTR(c); MVSTER_REQUIRE_PTR(d);
    MVSTER_REQUIRE_SHAPE(B > 0 && C > 0 && D > 0 && Hr > 0 && Wr > 0 && Hs > 0 && Ws > 0);
    MVSTER_REQUIRE_SHAPE(B <= 65535);
    MVSTER_REQUIRE_SHAPE((long long)Hr * Wr <= kMaxPlane && (long long)Hs * Ws <= kMaxPlane);
    return MVSTER_OK;
}

}  // namespace mvster

extern "C" int mvster_warp_fwd(const float* src, const float* pose, const float* depth, float* out, int B, int C,
                               int D, int Hr, int Wr, int Hs, int Ws, mvster_stream_t stream) {
    using namespace mvster;
    if (int rc = check_warp(src, pose, depth, out, B, C, D, Hr, Wr, Hs, Ws)) return rc;
    dim3 grid(grid_for((long long)Hr * Wr, kBlock), B);
    hipLaunchKernelGGL(warp_fwd_kernel, grid, dim3(kBlock), 0, as_stream(stream), src, pose, depth, out, C, D, Hr, Wr,
                       Hs, Ws);
    return launch_status();
}

extern "C" int mvster_warp_bwd(const float* gout, const float* pose, const float* depth, float* gsrc, int B, int C,
                               int D, int Hr, int Wr, int Hs, int Ws, mvster_stream_t stream) {
    using namespace mvster;
    if (int rc = check_warp(gout, pose, depth, gsrc, B, C, D, Hr, Wr, Hs, Ws)) return rc;
    if (hipMemsetAsync(gsrc, 0, sizeof(float) * (size_t)B * C * Hs * Ws, as_stream(stream)) != hipSuccess) {
        (void)hipGetLastError();
        return MVSTER_ERR_LAUNCH;
    }
    dim3 grid(grid_for((long long)Hr * Wr, kBlock), B);
    hipLaunchKernelGGL(warp_bwd_kernel, grid, dim3(kBlock), 0, as_stream(stream), gout, pose, depth, gsrc, C, D, Hr,
                       Wr, Hs, Ws);
    return launch_status();
}
