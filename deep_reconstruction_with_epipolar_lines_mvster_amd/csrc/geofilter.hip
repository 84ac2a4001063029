// F4  geometric-consistency check of the depth-map filter: project every reference pixel with its
// estimated depth into a source view, sample the source depth map there, project back, compare.
//   reference: test_mvs4.py:612-649 (reproject_with_depth), :653-670 (check_geometric_consistency), numpy in float64
//   plus cv2.remap(INTER_LINEAR, constant border 0) for the sampling.
// One lane per reference pixel; the six small matrices the reference builds with np.linalg.inv / np.matmul are
// composed on the host in float64 exactly like upstream and passed by value; the per-pixel algebra runs in float64
// like numpy does.  HBM traffic: depth_ref + 4 gathers of depth_src in, mask + reprojected depth + 2 coordinate maps out.
// PARITY UNPINNED: cv2 is not importable in the build container, so the bilinear sampling restates cv2.remap's
// documented behaviour (weights at the float32 sample position, taps outside the map contribute 0) without its
// internal 1/32-pixel coordinate quantisation; see oracle/geofilter_oracle.py.
#include <math.h>

#include "mvster_device.h"
#include "mvster_host.h"

namespace mvster {

struct FilterMats {
    double Kr_inv[9];   // inv(intrinsics_ref)
    double T_rs[16];    // extrinsics_src @ inv(extrinsics_ref)
    double Ks[9];       // intrinsics_src
    double Ks_inv[9];   // inv(intrinsics_src)
    double T_sr[16];    // extrinsics_ref @ inv(extrinsics_src)
    double Kr[9];       // intrinsics_ref
};

__device__ __forceinline__ void mat3(const double* __restrict__ m, double x, double y, double z, double& ox, double& oy, double& oz) {
    ox = m[0] * x + m[1] * y + m[2] * z;
    oy = m[3] * x + m[4] * y + m[5] * z;
    oz = m[6] * x + m[7] * y + m[8] * z;
}

__device__ __forceinline__ void rigid(const double* __restrict__ m, double x, double y, double z, double& ox, double& oy, double& oz) {
    ox = m[0] * x + m[1] * y + m[2] * z + m[3];
    oy = m[4] * x + m[5] * y + m[6] * z + m[7];
    oz = m[8] * x + m[9] * y + m[10] * z + m[11];
}

__device__ __forceinline__ float remap_bilinear_zero(const float* __restrict__ img, int H, int W, float x, float y) {
    const float xf = floorf(x), yf = floorf(y);
    const float ax = x - xf, ay = y - yf;
    float acc = 0.0f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const float tx = xf + dx, ty = yf + dy;
            if (tx >= 0.0f && tx <= (float)(W - 1) && ty >= 0.0f && ty <= (float)(H - 1)) {
                const float w = (dx ? ax : 1.0f - ax) * (dy ? ay : 1.0f - ay);
                acc += w * img[(int)ty * W + (int)tx];
            }
        }
    }
    return acc;
}

__global__ void __launch_bounds__(kBlock)
geometric_consistency_kernel(const float* __restrict__ depth_ref, const float* __restrict__ depth_src, FilterMats m,
                             uint8_t* __restrict__ mask, float* __restrict__ depth_reproj, float* __restrict__ x_src_out,
                             float* __restrict__ y_src_out, int H, int W, int Hs, int Ws, float pix_thresh, float depth_thresh) {
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    const float dref = depth_ref[p];
    // step 1: reference pixel -> source view                                  (test_mvs4.py:619-626)
    double rx, ry, rz, sx, sy, sz, kx, ky, kz;
    mat3(m.Kr_inv, (double)x * dref, (double)y * dref, (double)dref, rx, ry, rz);
    rigid(m.T_rs, rx, ry, rz, sx, sy, sz);
    mat3(m.Ks, sx, sy, sz, kx, ky, kz);
    const double ux = kx / kz, uy = ky / kz;
    const float xs = (float)ux, ys = (float)uy;                              // (:630-631)
    // step 2: sample the source depth and come back                           (:632-648)
    const float sampled = remap_bilinear_zero(depth_src, Hs, Ws, xs, ys);
    mat3(m.Ks_inv, ux * sampled, uy * sampled, (double)sampled, sx, sy, sz);
    rigid(m.T_sr, sx, sy, sz, rx, ry, rz);
    float drep = (float)rz;
    mat3(m.Kr, rx, ry, rz, kx, ky, kz);
    const float xr = (float)(kx / kz), yr = (float)(ky / kz);
    // check |p_reproj - p| < pix_thresh and |d_reproj - d| / d < depth_thresh   (:660-667)
    const double ddx = (double)xr - (double)x, ddy = (double)yr - (double)y;
    const double dist = sqrt(ddx * ddx + ddy * ddy);
    const float rel = fabsf(drep - dref) / dref;
    const bool ok = (dist < (double)pix_thresh) && (rel < depth_thresh);
    mask[p] = ok ? 1 : 0;
    depth_reproj[p] = ok ? drep : 0.0f;                                      // depth_reprojected[~mask] = 0 (:668)
    if (x_src_out != nullptr) x_src_out[p] = xs;
    if (y_src_out != nullptr) y_src_out[p] = ys;
}

}  // namespace mvster

extern "C" int mvster_geometric_consistency(const float* depth_ref, const float* depth_src, const double* mats,
                                            uint8_t* mask, float* depth_reproj, float* x_src, float* y_src, int H, int W,
                                            int Hs, int Ws, float pix_thresh, float depth_thresh, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(depth_ref); MVSTER_REQUIRE_PTR(depth_src); MVSTER_REQUIRE_PTR(mats); MVSTER_REQUIRE_PTR(mask);
    MVSTER_REQUIRE_PTR(depth_reproj);
    MVSTER_REQUIRE_SHAPE(H > 0 && W > 0 && Hs > 0 && Ws > 0 && (long long)H * W <= kMaxPlane && (long long)Hs * Ws <= kMaxPlane);
    MVSTER_REQUIRE_SHAPE((x_src == nullptr) == (y_src == nullptr));
    FilterMats m;
    const double* q = mats;                      // HOST pointer: 9 + 16 + 9 + 9 + 16 + 9 = 68 doubles, row-major
    for (int i = 0; i < 9; ++i) m.Kr_inv[i] = *q++;
    for (int i = 0; i < 16; ++i) m.T_rs[i] = *q++;
    for (int i = 0; i < 9; ++i) m.Ks[i] = *q++;
    for (int i = 0; i < 9; ++i) m.Ks_inv[i] = *q++;
    for (int i = 0; i < 16; ++i) m.T_sr[i] = *q++;
    for (int i = 0; i < 9; ++i) m.Kr[i] = *q++;
    hipLaunchKernelGGL(geometric_consistency_kernel, dim3(grid_for((long long)H * W, kBlock)), dim3(kBlock), 0, as_stream(stream),
                       depth_ref, depth_src, m, mask, depth_reproj, x_src, y_src, H, W, Hs, Ws, pix_thresh, depth_thresh);
    return launch_status();
}
