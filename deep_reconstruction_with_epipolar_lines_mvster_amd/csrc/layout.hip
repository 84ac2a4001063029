// Feature-map layout conversion between the reference's NCHW planes and the channels-last (NHWC) layout the
// fused cost-volume kernels gather from.  One lane per pixel: the planar side is read/written as coalesced
// dwords (consecutive lanes = consecutive pixels of one channel plane), the channels-last side as 16-byte
// vectors of the lane's own C contiguous floats.  Only used when a feature map does not already arrive as a
// torch channels_last tensor.
#include "mvster_device.h"
#include "mvster_host.h"

namespace mvster {

// Several maps of one shape in one launch (blockIdx.z picks the map): a stage converts its N feature maps -- and the N
// gradients on the way back -- with one launch each instead of N.
constexpr int kMaxMaps = MVSTER_MAX_SRC_VIEWS + 1;
struct MapPtrs {
    const float* src[kMaxMaps];
    float* dst[kMaxMaps];
};

template <int C4T>   // C4T > 0: C/4 known at compile time (registers); 0: runtime C (C % 4 == 0)
__global__ void __launch_bounds__(kBlock)
nchw_to_nhwc_kernel(MapPtrs maps, int C, int HW) {
    const float* __restrict__ src = maps.src[blockIdx.z];
    float* __restrict__ dst = maps.dst[blockIdx.z];
    const int b = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= HW) return;
    const float* __restrict__ s = src + (size_t)b * C * HW + p;
    float4* __restrict__ d = reinterpret_cast<float4*>(dst + ((size_t)b * HW + p) * C);
    const int C4 = C4T > 0 ? C4T : C / 4;
    if constexpr (C4T > 0) {
        float4 v[C4T];
#pragma unroll
        for (int q = 0; q < C4T; ++q) {
            v[q].x = s[(size_t)(4 * q + 0) * HW];
            v[q].y = s[(size_t)(4 * q + 1) * HW];
            v[q].z = s[(size_t)(4 * q + 2) * HW];
            v[q].w = s[(size_t)(4 * q + 3) * HW];
        }
#pragma unroll
        for (int q = 0; q < C4T; ++q) d[q] = v[q];
    } else {
        for (int q = 0; q < C4; ++q) {
            float4 v;
            v.x = s[(size_t)(4 * q + 0) * HW];
            v.y = s[(size_t)(4 * q + 1) * HW];
            v.z = s[(size_t)(4 * q + 2) * HW];
            v.w = s[(size_t)(4 * q + 3) * HW];
            d[q] = v;
        }
    }
}

template <int C4T>
__global__ void __launch_bounds__(kBlock)
nhwc_to_nchw_kernel(MapPtrs maps, int C, int HW) {
    const float* __restrict__ src = maps.src[blockIdx.z];
    float* __restrict__ dst = maps.dst[blockIdx.z];
    const int b = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= HW) return;
    const float4* __restrict__ s = reinterpret_cast<const float4*>(src + ((size_t)b * HW + p) * C);
    float* __restrict__ d = dst + (size_t)b * C * HW + p;
    const int C4 = C4T > 0 ? C4T : C / 4;
    if constexpr (C4T > 0) {
        float4 v[C4T];
#pragma unroll
        for (int q = 0; q < C4T; ++q) v[q] = s[q];
#pragma unroll
        for (int q = 0; q < C4T; ++q) {
            d[(size_t)(4 * q + 0) * HW] = v[q].x;
            d[(size_t)(4 * q + 1) * HW] = v[q].y;
            d[(size_t)(4 * q + 2) * HW] = v[q].z;
            d[(size_t)(4 * q + 3) * HW] = v[q].w;
        }
    } else {
        for (int q = 0; q < C4; ++q) {
            const float4 v = s[q];
            d[(size_t)(4 * q + 0) * HW] = v.x;
            d[(size_t)(4 * q + 1) * HW] = v.y;
            d[(size_t)(4 * q + 2) * HW] = v.z;
            d[(size_t)(4 * q + 3) * HW] = v.w;
        }
    }
}

static int check_layout(const void* a, const void* b, int B, int C, int H, int W) {
    MVSTER_REQUIRE_PTR(a); MVSTER_REQUIRE_PTR(b);
    MVSTER_REQUIRE_SHAPE(B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0 && (long long)H * W <= kMaxPlane);
    if (C % 4 != 0) return MVSTER_ERR_UNSUPPORTED;
    return MVSTER_OK;
}

}  // namespace mvster

#define MVSTER_LAYOUT_DISPATCH(KERNEL)                                                                 \
    switch (C) {                                                                                       \
        case 8: hipLaunchKernelGGL((KERNEL<2>), grid, dim3(kBlock), 0, st, maps, C, H * W); break;     \
        case 16: hipLaunchKernelGGL((KERNEL<4>), grid, dim3(kBlock), 0, st, maps, C, H * W); break;    \
        case 32: hipLaunchKernelGGL((KERNEL<8>), grid, dim3(kBlock), 0, st, maps, C, H * W); break;    \
        default: hipLaunchKernelGGL((KERNEL<0>), grid, dim3(kBlock), 0, st, maps, C, H * W); break;    \
    }

namespace mvster {
static int gather_maps(const float* const* srcs, float* const* dsts, int N, MapPtrs* maps) {
    MVSTER_REQUIRE_PTR(srcs); MVSTER_REQUIRE_PTR(dsts);
    MVSTER_REQUIRE_SHAPE(N > 0);
    if (N > kMaxMaps) return MVSTER_ERR_UNSUPPORTED;
    for (int i = 0; i < N; ++i) {
        MVSTER_REQUIRE_PTR(srcs[i]); MVSTER_REQUIRE_PTR(dsts[i]);
        maps->src[i] = srcs[i];
        maps->dst[i] = dsts[i];
    }
    return MVSTER_OK;
}
}  // namespace mvster

extern "C" int mvster_nchw_to_nhwc_many(const float* const* srcs, float* const* dsts, int N, int B, int C, int H, int W,
                                        mvster_stream_t stream) {
    using namespace mvster;
    MapPtrs maps{};
    if (int rc = gather_maps(srcs, dsts, N, &maps)) return rc;
    if (int rc = check_layout(maps.src[0], maps.dst[0], B, C, H, W)) return rc;
    dim3 grid(grid_for((long long)H * W, kBlock), B, N);
    const hipStream_t st = as_stream(stream);
    MVSTER_LAYOUT_DISPATCH(nchw_to_nhwc_kernel)
    return launch_status();
}

extern "C" int mvster_nhwc_to_nchw_many(const float* const* srcs, float* const* dsts, int N, int B, int C, int H, int W,
                                        mvster_stream_t stream) {
    using namespace mvster;
    MapPtrs maps{};
    if (int rc = gather_maps(srcs, dsts, N, &maps)) return rc;
    if (int rc = check_layout(maps.src[0], maps.dst[0], B, C, H, W)) return rc;
    dim3 grid(grid_for((long long)H * W, kBlock), B, N);
    const hipStream_t st = as_stream(stream);
    MVSTER_LAYOUT_DISPATCH(nhwc_to_nchw_kernel)
    return launch_status();
}

extern "C" int mvster_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, mvster_stream_t stream) {
    return mvster_nchw_to_nhwc_many(&src, &dst, 1, B, C, H, W, stream);
}

extern "C" int mvster_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, mvster_stream_t stream) {
    return mvster_nhwc_to_nchw_many(&src, &dst, 1, B, C, H, W, stream);
}
