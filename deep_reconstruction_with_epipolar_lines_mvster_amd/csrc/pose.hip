// Relative camera poses (SURVEY.md 8a rows A1/A2 preamble).
//   reference: models/mvs4net_utils.py:1047-1050 (P = E with rows 0..2 = K[:3,:3] @ E[:3,:4]) and
//              models/mvs4net_utils.py:32-34    (proj = src_proj @ inverse(ref_proj); rot, trans).
// One thread per (batch, source view); 4x4 Gauss-Jordan with partial pivoting in fp64, rounded to
// fp32 once.  A singular reference matrix yields inf/NaN like torch.inverse's result would poison
// the reference path; nothing traps.
#include "mvster_host.h"

namespace mvster {

__device__ void invert4(const double a[16], double inv[16]) {
    double m[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            m[i][j] = a[i * 4 + j];
            m[i][j + 4] = (i == j) ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        double best = fabs(m[col][col]);
        for (int r = col + 1; r < 4; ++r) {
            const double v = fabs(m[r][col]);
            if (v > best) { best = v; piv = r; }
        }
        if (piv != col)
            for (int j = 0; j < 8; ++j) { const double t = m[col][j]; m[col][j] = m[piv][j]; m[piv][j] = t; }
        const double d = 1.0 / m[col][col];
        for (int j = 0; j < 8; ++j) m[col][j] *= d;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const double f = m[r][col];
            for (int j = 0; j < 8; ++j) m[r][j] -= f * m[col][j];
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) inv[i * 4 + j] = m[i][j + 4];
}

__device__ void write_pose(const double src[16], const double ref[16], float* __restrict__ out) {
    double inv[16];
    invert4(ref, inv);
    for (int i = 0; i < 3; ++i) {
        double row[4];
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc += src[i * 4 + k] * inv[k * 4 + j];
            row[j] = acc;
        }
        out[i * 3 + 0] = (float)row[0];
        out[i * 3 + 1] = (float)row[1];
        out[i * 3 + 2] = (float)row[2];
        out[9 + i] = (float)row[3];
    }
}

// E [4,4], K [4,4] -> P (fp64): the fp32 product K[:3,:3] @ E[:3,:4] of the reference is evaluated
// exactly in fp64 (every term is a product of two fp32 numbers) and NOT re-rounded to fp32.
__device__ void compose(const float* __restrict__ view, double P[16]) {
    const float* E = view;
    const float* K = view + 16;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += (double)K[i * 4 + k] * (double)E[k * 4 + j];
            P[i * 4 + j] = acc;
        }
    for (int j = 0; j < 4; ++j) P[12 + j] = (double)E[12 + j];
}

__global__ void relative_pose_kernel(const float* __restrict__ src_proj, const float* __restrict__ ref_proj,
                                     float* __restrict__ pose, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s[16], r[16];
    for (int i = 0; i < 16; ++i) {
        s[i] = (double)src_proj[b * 16 + i];
        r[i] = (double)ref_proj[b * 16 + i];
    }
    write_pose(s, r, pose + (size_t)b * 12);
}

__global__ void stage_poses_kernel(const float* __restrict__ proj, float* __restrict__ pose, int B, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int S = N - 1;
    if (i >= B * S) return;
    const int b = i / S, v = i - b * S + 1;
    double s[16], r[16];
    compose(proj + ((size_t)b * N + v) * 32, s);
    compose(proj + ((size_t)b * N) * 32, r);
    write_pose(s, r, pose + (size_t)i * 12);
}

}  // namespace mvster

extern "C" int mvster_relative_pose(const float* src_proj, const float* ref_proj, float* pose, int B,
                                    mvster_stream_t stream) {
    MVSTER_REQUIRE_PTR(src_proj); MVSTER_REQUIRE_PTR(ref_proj); MVSTER_REQUIRE_PTR(pose);
    MVSTER_REQUIRE_SHAPE(B > 0);
    hipLaunchKernelGGL(mvster::relative_pose_kernel, dim3(mvster::grid_for(B, 64)), dim3(64), 0,
                       mvster::as_stream(stream), src_proj, ref_proj, pose, B);
    return mvster::launch_status();
}

extern "C" int mvster_stage_poses(const float* proj_matrices, float* pose, int B, int N, mvster_stream_t stream) {
    MVSTER_REQUIRE_PTR(proj_matrices); MVSTER_REQUIRE_PTR(pose);
    MVSTER_REQUIRE_SHAPE(B > 0 && N >= 2);
    hipLaunchKernelGGL(mvster::stage_poses_kernel, dim3(mvster::grid_for((long long)B * (N - 1), 64)), dim3(64), 0,
                       mvster::as_stream(stream), proj_matrices, pose, B, N);
    return mvster::launch_status();
}

extern "C" int mvster_abi_version(void) { return MVSTER_ABI_VERSION; }

extern "C" const char* mvster_error_string(int code) {
    switch (code) {
        case MVSTER_OK: return "ok";
        case MVSTER_ERR_NULL_POINTER: return "a required pointer argument is NULL";
        case MVSTER_ERR_BAD_SHAPE: return "a dimension is non-positive, inconsistent or too large";
        case MVSTER_ERR_UNSUPPORTED: return "shape/mode not instantiated in the fused gfx950 kernels";
        case MVSTER_ERR_LAUNCH: return "the HIP runtime rejected a kernel launch or memset";
        default: return "unknown mvster error code";
    }
}
