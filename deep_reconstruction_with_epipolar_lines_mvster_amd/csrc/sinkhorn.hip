// F1  Sinkhorn optimal-transport loss between the predicted depth distribution and the
// ground-truth depth bin, fused into one per-pixel kernel (forward) and one per-pixel reverse sweep (backward).
//   reference: models/mvs4net_utils.py:1164-1210 (sinkhorn), called per stage by MVS4net_loss (models/MVS4Net.py:229).
// The reference materialises [B,HW,D,D] tensors for every one of ~25 elementwise/logsumexp operators per
// iteration; here a lane owns one pixel, the D x (D|D+1) transport problem lives in its registers, HBM traffic is
// one read of hypo/attn/gt/mask and one write of the per-pixel loss (or the attn gradient).
//
//   cost[r][c] = |r - c|                       (c < D);   continuous mode adds column D: |gt_bin - r|      (:1171,:1179-1189)
//   log_mu[c]  = log(gt_dist[c] + 1e-12),  gt_dist = one-hot(arg-min_d |hypo_d - gt|)  (or column D)   (:1175-1178,:1195)
//   log_nu[r]  = log(attn[r] + 1e-12)                                                                  (:1196)
//   repeat iters:  v[c] = log_mu[c] - LSE_r(cost/eps + u[r]);   u[r] = log_nu[r] - LSE_c(cost/eps + v[c])  (:1199-1202)
//   T = exp(cost/eps + u[r] + v[c]);   pixel loss = sum T * cost  (mean over masked pixels on the host)  (:1205-1206)
#include <math.h>

#include "mvster_device.h"
#include "mvster_host.h"

namespace mvster {

constexpr int kSinkhornMaxIters = 16;

template <int D, bool CONT>
struct Transport {
    static constexpr int NC = CONT ? D + 1 : D;
    float log_nu[D];
    float log_mu[NC];
    float extra[CONT ? D : 1];     // continuous mode: cost of the extra column, |gt_bin - r|
    float eps;                     // entropic scale: the reference adds D_map/eps (a true division)

    __device__ __forceinline__ float cost(int r, int c) const {
        if (CONT && c == D) return extra[CONT ? r : 0];
        return fabsf((float)(r - c));
    }
};

// loads one pixel's problem; returns false when the pixel is masked out (nothing to compute)
template <int D, bool CONT>
__device__ __forceinline__ bool load_problem(Transport<D, CONT>& t, const float* __restrict__ gt,
                                             const float* __restrict__ hypo, const float* __restrict__ attn,
                                             const uint8_t* __restrict__ mask, size_t bp, size_t bdp, int HW, float eps,
                                             float (&a)[D]) {
    if (mask[bp] == 0) return false;
    const float g = gt[bp];
    float h[D];
    int gi = 0;
    float best = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        h[d] = hypo[bdp + (size_t)d * HW];
        a[d] = attn[bdp + (size_t)d * HW];
        t.log_nu[d] = logf(a[d] + 1e-12f);
        const float dist = fabsf(h[d] - g);
        if (d == 0 || dist < best) { best = dist; gi = d; }      // first arg-min, like torch.min (:1175)
    }
    t.eps = eps;
    if constexpr (CONT) {
#pragma unroll
        for (int c = 0; c < D; ++c) t.log_mu[c] = logf(1e-12f);
        t.log_mu[D] = logf(1.0f + 1e-12f);
        const float itv = 1.0f / h[2] - 1.0f / h[1];                                  // (:1183)
        const float bin = (1.0f / g - 1.0f / h[0]) / itv;                             // (:1184)
#pragma unroll
        for (int r = 0; r < D; ++r) t.extra[r] = fabsf(bin - (float)r);               // (:1188)
    } else {
#pragma unroll
        for (int c = 0; c < D; ++c) t.log_mu[c] = logf((c == gi ? 1.0f : 0.0f) + 1e-12f);
    }
    return true;
}

// one Sinkhorn sweep: v from u, then u from v  (torch.logsumexp = max + log(sum(exp(x - max))))
template <int D, bool CONT>
__device__ __forceinline__ void sweep(const Transport<D, CONT>& t, float (&u)[D], float (&v)[Transport<D, CONT>::NC]) {
    constexpr int NC = Transport<D, CONT>::NC;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < D; ++r) mx = fmaxf(mx, t.cost(r, c) / t.eps + u[r]);
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < D; ++r) s += expf(t.cost(r, c) / t.eps + u[r] - mx);
        v[c] = t.log_mu[c] - (mx + logf(s));
    }
#pragma unroll
    for (int r = 0; r < D; ++r) {
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < NC; ++c) mx = fmaxf(mx, t.cost(r, c) / t.eps + v[c]);
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < NC; ++c) s += expf(t.cost(r, c) / t.eps + v[c] - mx);
        u[r] = t.log_nu[r] - (mx + logf(s));
    }
}

template <int D, bool CONT>
__global__ void __launch_bounds__(kBlock)
sinkhorn_fwd_kernel(const float* __restrict__ gt, const float* __restrict__ hypo, const float* __restrict__ attn,
                    const uint8_t* __restrict__ mask, float* __restrict__ pix_loss, float* __restrict__ tmap, int HW,
                    int iters, float eps) {
    constexpr int NC = Transport<D, CONT>::NC;
    const int b = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= HW) return;
    const size_t bp = (size_t)b * HW + p, bdp = (size_t)b * D * HW + p;
    Transport<D, CONT> t;
    float a[D];
    float* __restrict__ tm = tmap != nullptr ? tmap + bp * (size_t)(D * NC) : nullptr;
    if (!load_problem<D, CONT>(t, gt, hypo, attn, mask, bp, bdp, HW, eps, a)) {
        pix_loss[bp] = 0.0f;
        if (tm != nullptr) {
            // the reference evaluates masked-out pixels too; their T is never used by its loss.  NaN marks them.
            for (int i = 0; i < D * NC; ++i) tm[i] = __builtin_nanf("");
        }
        return;
    }
    float u[D], v[NC];
#pragma unroll
    for (int r = 0; r < D; ++r) u[r] = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = 0.0f;
    for (int it = 0; it < iters; ++it) sweep<D, CONT>(t, u, v);
    float loss = 0.0f;
#pragma unroll
    for (int r = 0; r < D; ++r) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float T = expf(t.cost(r, c) / t.eps + u[r] + v[c]);
            loss += T * t.cost(r, c);
            if (tm != nullptr) tm[r * NC + c] = T;
        }
    }
    pix_loss[bp] = loss;
}

// Reverse sweep.  With P_k[r][c] = softmax_c(cost/eps + v_k[c]) = exp(cost/eps + v_k[c] + u_k[r] - log_nu[r]) and
// Q_k[r][c] = softmax_r(cost/eps + u_{k-1}[r]) = exp(cost/eps + u_{k-1}[r] + v_k[c] - log_mu[c]):
//   ubar_K = dL/du_K, vbar_K = dL/dv_K from T;  for k = K..1:  g_lognu += ubar_k;  vbar_k -= ubar_k^T P_k;
//   ubar_{k-1} = -Q_k vbar_k;  vbar_{k-1} = 0.      gattn[r] = gpix * g_lognu[r] / (attn[r] + 1e-12).
template <int D, bool CONT>
__global__ void __launch_bounds__(kBlock)
sinkhorn_bwd_kernel(const float* __restrict__ gt, const float* __restrict__ hypo, const float* __restrict__ attn,
                    const uint8_t* __restrict__ mask, const float* __restrict__ gpix, float* __restrict__ gattn, int HW,
                    int iters, float eps) {
    constexpr int NC = Transport<D, CONT>::NC;
    const int b = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= HW) return;
    const size_t bp = (size_t)b * HW + p, bdp = (size_t)b * D * HW + p;
    Transport<D, CONT> t;
    float a[D];
    if (!load_problem<D, CONT>(t, gt, hypo, attn, mask, bp, bdp, HW, eps, a)) {
#pragma unroll
        for (int d = 0; d < D; ++d) gattn[bdp + (size_t)d * HW] = 0.0f;
        return;
    }
    // forward with history (iters is small and runtime: the history lives in private scratch)
    float uh[kSinkhornMaxIters + 1][D], vh[kSinkhornMaxIters + 1][NC];
    float u[D], v[NC];
#pragma unroll
    for (int r = 0; r < D; ++r) { u[r] = 0.0f; uh[0][r] = 0.0f; }
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = 0.0f;
    for (int k = 1; k <= iters; ++k) {
        sweep<D, CONT>(t, u, v);
#pragma unroll
        for (int r = 0; r < D; ++r) uh[k][r] = u[r];
#pragma unroll
        for (int c = 0; c < NC; ++c) vh[k][c] = v[c];
    }
    float ubar[D], vbar[NC], glognu[D];
#pragma unroll
    for (int r = 0; r < D; ++r) { ubar[r] = 0.0f; glognu[r] = 0.0f; }
#pragma unroll
    for (int c = 0; c < NC; ++c) vbar[c] = 0.0f;
#pragma unroll
    for (int r = 0; r < D; ++r) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float w = expf(t.cost(r, c) / t.eps + u[r] + v[c]) * t.cost(r, c);
            ubar[r] += w;
            vbar[c] += w;
        }
    }
    for (int k = iters; k >= 1; --k) {
        float uk[D], um[D], vk[NC];
#pragma unroll
        for (int r = 0; r < D; ++r) { uk[r] = uh[k][r]; um[r] = uh[k - 1][r]; }
#pragma unroll
        for (int c = 0; c < NC; ++c) vk[c] = vh[k][c];
#pragma unroll
        for (int r = 0; r < D; ++r) glognu[r] += ubar[r];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float acc = 0.0f;
#pragma unroll
            for (int r = 0; r < D; ++r)
                acc = fmaf(ubar[r], expf(t.cost(r, c) / t.eps + vk[c] + uk[r] - t.log_nu[r]), acc);
            vbar[c] -= acc;
        }
#pragma unroll
        for (int r = 0; r < D; ++r) {
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < NC; ++c)
                acc = fmaf(vbar[c], expf(t.cost(r, c) / t.eps + um[r] + vk[c] - t.log_mu[c]), acc);
            ubar[r] = -acc;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) vbar[c] = 0.0f;
    }
    const float g = gpix[bp];
#pragma unroll
    for (int d = 0; d < D; ++d) gattn[bdp + (size_t)d * HW] = g * glognu[d] / (a[d] + 1e-12f);
}

static int check_sinkhorn(const void* a, const void* b, const void* c, const void* d, const void* e, int B, int D, int H,
                          int W, int iters, float eps, int continuous) {
    MVSTER_REQUIRE_PTR(a); MVSTER_REQUIRE_PTR(b); MVSTER_REQUIRE_PTR(c); MVSTER_REQUIRE_PTR(d); MVSTER_REQUIRE_PTR(e);
    MVSTER_REQUIRE_SHAPE(B > 0 && B <= 65535 && D > 0 && H > 0 && W > 0 && (long long)H * W <= kMaxPlane);
    MVSTER_REQUIRE_SHAPE(iters >= 0 && eps != 0.0f && (!continuous || D >= 3));
    if (!(D == 4 || D == 8 || D == 16) || iters > kSinkhornMaxIters) return MVSTER_ERR_UNSUPPORTED;
    return MVSTER_OK;
}

}  // namespace mvster

#define MVSTER_SINKHORN_DISPATCH(KERNEL, ...)                                                                     \
    do {                                                                                                          \
        if (continuous) {                                                                                         \
            if (D == 4) hipLaunchKernelGGL((KERNEL<4, true>), grid, dim3(kBlock), 0, st, __VA_ARGS__);            \
            else if (D == 8) hipLaunchKernelGGL((KERNEL<8, true>), grid, dim3(kBlock), 0, st, __VA_ARGS__);       \
            else hipLaunchKernelGGL((KERNEL<16, true>), grid, dim3(kBlock), 0, st, __VA_ARGS__);                  \
        } else {                                                                                                  \
            if (D == 4) hipLaunchKernelGGL((KERNEL<4, false>), grid, dim3(kBlock), 0, st, __VA_ARGS__);           \
            else if (D == 8) hipLaunchKernelGGL((KERNEL<8, false>), grid, dim3(kBlock), 0, st, __VA_ARGS__);      \
            else hipLaunchKernelGGL((KERNEL<16, false>), grid, dim3(kBlock), 0, st, __VA_ARGS__);                 \
        }                                                                                                         \
    } while (0)

extern "C" int mvster_sinkhorn_fwd(const float* gt_depth, const float* hypo, const float* attn, const uint8_t* mask,
                                   float* pix_loss, float* tmap, int B, int D, int H, int W, int iters, float eps,
                                   int continuous, mvster_stream_t stream) {
    using namespace mvster;
    if (int rc = check_sinkhorn(gt_depth, hypo, attn, mask, pix_loss, B, D, H, W, iters, eps, continuous)) return rc;
    const int HW = H * W;
    dim3 grid(grid_for(HW, kBlock), B);
    const hipStream_t st = as_stream(stream);
    MVSTER_SINKHORN_DISPATCH(sinkhorn_fwd_kernel, gt_depth, hypo, attn, mask, pix_loss, tmap, HW, iters, eps);
    return launch_status();
}

extern "C" int mvster_sinkhorn_bwd(const float* gt_depth, const float* hypo, const float* attn, const uint8_t* mask,
                                   const float* gpix, float* gattn, int B, int D, int H, int W, int iters, float eps,
                                   int continuous, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(gpix);
    if (int rc = check_sinkhorn(gt_depth, hypo, attn, mask, gattn, B, D, H, W, iters, eps, continuous)) return rc;
    const int HW = H * W;
    dim3 grid(grid_for(HW, kBlock), B);
    const hipStream_t st = as_stream(stream);
    MVSTER_SINKHORN_DISPATCH(sinkhorn_bwd_kernel, gt_depth, hypo, attn, mask, gpix, gattn, HW, iters, eps);
    return launch_status();
}
