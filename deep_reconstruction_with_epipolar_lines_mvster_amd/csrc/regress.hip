// A4  post-regnet half of the stage: softmax over depth, first arg-max, depth gather, confidence and
// the next stage's inverse-depth range -- one streaming pass over the [B,D,H,W] logits.
//   reference: models/mvs4net_utils.py:1109-1113 (raw-logit max/sum, eval), :1126 (softmax),
//              :1129-1130 (arg-max + gather), :1135-1138 (confidence), :1151-1156 (inverse range).
// One lane per pixel, consecutive lanes = consecutive pixels: every load/store is a fully coalesced
// 256-byte row segment of one depth plane.  D values per lane stay in registers (D templated) or are
// re-read from L1/L2 (generic D).
#include <math.h>

#include "mvster_device.h"
#include "mvster_host.h"

namespace mvster {

// torch.max semantics: the first maximal element wins; a NaN beats every number and the first NaN wins.
__device__ __forceinline__ bool beats(float cand, float best) {
    return (cand > best) | ((cand != cand) & (best == best));
}

template <int DT>   // DT > 0: compile-time D held in registers; DT == 0: runtime D, values re-read
__global__ void __launch_bounds__(kBlock)
depth_regress_kernel(const float* __restrict__ logits, const float* __restrict__ hypo, float* __restrict__ attn,
                     float* __restrict__ depth, int32_t* __restrict__ index, float* __restrict__ conf,
                     float* __restrict__ inv_min, float* __restrict__ inv_max, int Drt, int HW, float split_itv) {
    const int D = DT > 0 ? DT : Drt;
    const int b = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= HW) return;
    const float* __restrict__ lg = logits + (size_t)b * D * HW + p;
    const float* __restrict__ hy = hypo + (size_t)b * D * HW + p;
    float* __restrict__ at = attn + (size_t)b * D * HW + p;

    float v[DT > 0 ? DT : 1];
    float raw_max = lg[0], raw_sum = 0.0f;
    int raw_idx = 0;
    if constexpr (DT > 0) {
#pragma unroll
        for (int d = 0; d < DT; ++d) v[d] = lg[(size_t)d * HW];
    }
    for (int d = 0; d < D; ++d) {
        const float x = DT > 0 ? v[DT > 0 ? d : 0] : lg[(size_t)d * HW];
        raw_sum += x;                                   // attn_weight.sum(1) on the raw logits (:1113)
        if (d > 0 && beats(x, raw_max)) { raw_max = x; raw_idx = d; }
    }
    (void)raw_idx;
    // softmax: exp(x - max) / sum, like torch.softmax
    float esum = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float x = DT > 0 ? v[DT > 0 ? d : 0] : lg[(size_t)d * HW];
        const float e = expf(x - raw_max);
        if constexpr (DT > 0) v[DT > 0 ? d : 0] = e; else at[(size_t)d * HW] = e;
        esum += e;
    }
    float best = 0.0f;
    int best_idx = 0;
    for (int d = 0; d < D; ++d) {
        const float e = DT > 0 ? v[DT > 0 ? d : 0] : at[(size_t)d * HW];
        const float a = e / esum;
        at[(size_t)d * HW] = a;
        if (d == 0 || beats(a, best)) { best = a; best_idx = d; }   // arg-max of the softmax OUTPUT (:1129)
    }
    const float dsel = hy[(size_t)best_idx * HW];                    // gather (:1130)
    const size_t o = (size_t)b * HW + p;
    depth[o] = dsel;
    if (index != nullptr) index[o] = best_idx;
    if (conf != nullptr) conf[o] = raw_max / raw_sum;                // unnormalised on purpose (:1136)
    if (inv_min != nullptr) {
        const float step = 1.0f / hy[(size_t)2 * HW] - 1.0f / hy[(size_t)HW];   // (:1152)
        const float inv = 1.0f / dsel;
        inv_min[o] = inv + split_itv * step;                         // (:1153)
        inv_max[o] = inv - split_itv * step;                         // (:1154)
    }
}

// glogits = attn * (gattn - sum_d gattn*attn)
template <int DT>
__global__ void __launch_bounds__(kBlock)
softmax_bwd_kernel(const float* __restrict__ attn, const float* __restrict__ gattn, float* __restrict__ glogits,
                   int Drt, int HW) {
    const int D = DT > 0 ? DT : Drt;
    const int b = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= HW) return;
    const float* __restrict__ a = attn + (size_t)b * D * HW + p;
    const float* __restrict__ g = gattn + (size_t)b * D * HW + p;
    float* __restrict__ o = glogits + (size_t)b * D * HW + p;
    float av[DT > 0 ? DT : 1], gv[DT > 0 ? DT : 1];
    float dot = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float x = a[(size_t)d * HW], y = g[(size_t)d * HW];
        if constexpr (DT > 0) { av[DT > 0 ? d : 0] = x; gv[DT > 0 ? d : 0] = y; }
        dot = fmaf(x, y, dot);
    }
    for (int d = 0; d < D; ++d) {
        const float x = DT > 0 ? av[DT > 0 ? d : 0] : a[(size_t)d * HW];
        const float y = DT > 0 ? gv[DT > 0 ? d : 0] : g[(size_t)d * HW];
        o[(size_t)d * HW] = x * (y - dot);
    }
}

}  // namespace mvster

extern "C" int mvster_depth_regress_fwd(const float* logits, const float* hypo, float* attn, float* depth,
                                        int32_t* index, float* conf, float* inv_min, float* inv_max, int B, int D,
                                        int H, int W, float split_itv, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(logits); MVSTER_REQUIRE_PTR(hypo); MVSTER_REQUIRE_PTR(attn); MVSTER_REQUIRE_PTR(depth);
    MVSTER_REQUIRE_SHAPE(B > 0 && B <= 65535 && D > 0 && H > 0 && W > 0 && (long long)H * W <= kMaxPlane);
    MVSTER_REQUIRE_SHAPE((inv_min == nullptr) == (inv_max == nullptr));
    MVSTER_REQUIRE_SHAPE(inv_min == nullptr || D >= 3);
    const int HW = H * W;
    dim3 grid(grid_for(HW, kBlock), B), block(kBlock);
    const hipStream_t st = as_stream(stream);
#define LAUNCH(DT)                                                                                               \
    hipLaunchKernelGGL((depth_regress_kernel<DT>), grid, block, 0, st, logits, hypo, attn, depth, index, conf, \
                       inv_min, inv_max, D, HW, split_itv)
    switch (D) {
        case 4: LAUNCH(4); break;
        case 8: LAUNCH(8); break;
        case 16: LAUNCH(16); break;
        default: LAUNCH(0); break;
    }
#undef LAUNCH
    return launch_status();
}

extern "C" int mvster_softmax_bwd(const float* attn, const float* gattn, float* glogits, int B, int D, int H, int W,
                                  mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(attn); MVSTER_REQUIRE_PTR(gattn); MVSTER_REQUIRE_PTR(glogits);
    MVSTER_REQUIRE_SHAPE(B > 0 && B <= 65535 && D > 0 && H > 0 && W > 0 && (long long)H * W <= kMaxPlane);
    const int HW = H * W;
    dim3 grid(grid_for(HW, kBlock), B), block(kBlock);
    const hipStream_t st = as_stream(stream);
#define LAUNCH(DT) hipLaunchKernelGGL((softmax_bwd_kernel<DT>), grid, block, 0, st, attn, gattn, glogits, D, HW)
    switch (D) {
        case 4: LAUNCH(4); break;
        case 8: LAUNCH(8); break;
        case 16: LAUNCH(16); break;
        default: LAUNCH(0); break;
    }
#undef LAUNCH
    return launch_status();
}
