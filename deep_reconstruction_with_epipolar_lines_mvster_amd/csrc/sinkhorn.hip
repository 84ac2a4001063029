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

// ---- any D <= 64, any number of sweeps: coverage, not speed --------------------------------------------------------------
// The specialised kernels above keep the whole transport problem of a pixel in registers (D in {4, 8, 16}, <= 16 sweeps).
// Every other shape (stage_splits such as 5, 6, 12, 24, 48; ot_iter > 16) runs here: the same sweeps as plain loops, the
// pixel's vectors in a caller-provided workspace laid out [slot][pixel] (consecutive lanes = consecutive pixels: every
// access is coalesced), the backward's per-sweep history included.  Slots, NC = D (+1 continuous):
//   log_nu [D] | log_mu [NC] | extra [D] | u_k [D], k = 0..K | v_k [NC], k = 0..K        K = sweeps kept: 0 forward, iters backward
struct GenericSinkhorn {
    int D, NC, continuous, iters;
    float eps;
    size_t stride;              // pixels in all (B * HW): distance between slots
    __device__ __forceinline__ size_t log_nu(int r) const { return (size_t)r * stride; }
    __device__ __forceinline__ size_t log_mu(int c) const { return (size_t)(D + c) * stride; }
    __device__ __forceinline__ size_t extra(int r) const { return (size_t)(D + NC + r) * stride; }
    __device__ __forceinline__ size_t u(int k, int r) const { return (size_t)(2 * D + NC + k * (D + NC) + r) * stride; }
    __device__ __forceinline__ size_t v(int k, int c) const { return (size_t)(2 * D + NC + k * (D + NC) + D + c) * stride; }
};

__device__ __forceinline__ float generic_cost(const GenericSinkhorn& g, const float* __restrict__ w, int r, int c) {
    return (g.continuous && c == g.D) ? w[g.extra(r)] : fabsf((float)(r - c));
}

// fills log_nu / log_mu / extra of one pixel (w already points at the pixel); false: masked out
__device__ __forceinline__ bool generic_load(const GenericSinkhorn& g, float* __restrict__ w, const float* __restrict__ gt,
                                             const float* __restrict__ hypo, const float* __restrict__ attn,
                                             const uint8_t* __restrict__ mask, size_t bp, size_t bdp, int HW) {
    if (mask[bp] == 0) return false;
    const float gv = gt[bp];
    int gi = 0;
    float best = 0.0f;
    for (int d = 0; d < g.D; ++d) {
        w[g.log_nu(d)] = logf(attn[bdp + (size_t)d * HW] + 1e-12f);
        const float dist = fabsf(hypo[bdp + (size_t)d * HW] - gv);
        if (d == 0 || dist < best) { best = dist; gi = d; }
    }
    if (g.continuous) {
        for (int c = 0; c < g.D; ++c) w[g.log_mu(c)] = logf(1e-12f);
        w[g.log_mu(g.D)] = logf(1.0f + 1e-12f);
        const float h0 = hypo[bdp], h1 = hypo[bdp + (size_t)HW], h2 = hypo[bdp + 2 * (size_t)HW];
        const float itv = 1.0f / h2 - 1.0f / h1;
        const float bin = (1.0f / gv - 1.0f / h0) / itv;
        for (int r = 0; r < g.D; ++r) w[g.extra(r)] = fabsf(bin - (float)r);
    } else {
        for (int c = 0; c < g.D; ++c) w[g.log_mu(c)] = logf((c == gi ? 1.0f : 0.0f) + 1e-12f);
    }
    return true;
}

// one sweep from (u_from) to (v_to, u_to); the slots may coincide (forward: everything in sweep 0)
__device__ __forceinline__ void generic_sweep(const GenericSinkhorn& g, float* __restrict__ w, int from, int to) {
    for (int c = 0; c < g.NC; ++c) {
        float mx = -INFINITY;
        for (int r = 0; r < g.D; ++r) mx = fmaxf(mx, generic_cost(g, w, r, c) / g.eps + w[g.u(from, r)]);
        float s = 0.0f;
        for (int r = 0; r < g.D; ++r) s += expf(generic_cost(g, w, r, c) / g.eps + w[g.u(from, r)] - mx);
        w[g.v(to, c)] = w[g.log_mu(c)] - (mx + logf(s));
    }
    for (int r = 0; r < g.D; ++r) {
        float mx = -INFINITY;
        for (int c = 0; c < g.NC; ++c) mx = fmaxf(mx, generic_cost(g, w, r, c) / g.eps + w[g.v(to, c)]);
        float s = 0.0f;
        for (int c = 0; c < g.NC; ++c) s += expf(generic_cost(g, w, r, c) / g.eps + w[g.v(to, c)] - mx);
        const float unew = w[g.log_nu(r)] - (mx + logf(s));
        w[g.u(to, r)] = unew;       // from == to: row r of u is not read again in this sweep (its LSE is over v only)
    }
}

__global__ void __launch_bounds__(kBlock)
sinkhorn_generic_fwd_kernel(const float* __restrict__ gt, const float* __restrict__ hypo, const float* __restrict__ attn,
                            const uint8_t* __restrict__ mask, float* __restrict__ pix_loss, float* __restrict__ tmap,
                            float* __restrict__ workspace, int HW, GenericSinkhorn g) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= HW) return;
    const size_t bp = (size_t)b * HW + p, bdp = (size_t)b * g.D * HW + p;
    float* __restrict__ w = workspace + bp;
    float* __restrict__ tm = tmap != nullptr ? tmap + bp * (size_t)(g.D * g.NC) : nullptr;
    if (!generic_load(g, w, gt, hypo, attn, mask, bp, bdp, HW)) {
        pix_loss[bp] = 0.0f;
        if (tm != nullptr)
            for (int i = 0; i < g.D * g.NC; ++i) tm[i] = __builtin_nanf("");
        return;
    }
    for (int r = 0; r < g.D; ++r) w[g.u(0, r)] = 0.0f;
    for (int c = 0; c < g.NC; ++c) w[g.v(0, c)] = 0.0f;
    for (int it = 0; it < g.iters; ++it) generic_sweep(g, w, 0, 0);
    float loss = 0.0f;
    for (int r = 0; r < g.D; ++r)
        for (int c = 0; c < g.NC; ++c) {
            const float cost = generic_cost(g, w, r, c);
            const float T = expf(cost / g.eps + w[g.u(0, r)] + w[g.v(0, c)]);
            loss += T * cost;
            if (tm != nullptr) tm[r * g.NC + c] = T;
        }
    pix_loss[bp] = loss;
}

// reverse sweep as in sinkhorn_bwd_kernel, the history u_k, v_k read back from the workspace; ubar / vbar / g_lognu reuse the
// slots of sweep 0's v (vbar), and two scratch rows appended behind the history (ubar, g_lognu)
__global__ void __launch_bounds__(kBlock)
sinkhorn_generic_bwd_kernel(const float* __restrict__ gt, const float* __restrict__ hypo, const float* __restrict__ attn,
                            const uint8_t* __restrict__ mask, const float* __restrict__ gpix, float* __restrict__ gattn,
                            float* __restrict__ workspace, int HW, GenericSinkhorn g) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= HW) return;
    const size_t bp = (size_t)b * HW + p, bdp = (size_t)b * g.D * HW + p;
    float* __restrict__ w = workspace + bp;
    if (!generic_load(g, w, gt, hypo, attn, mask, bp, bdp, HW)) {
        for (int d = 0; d < g.D; ++d) gattn[bdp + (size_t)d * HW] = 0.0f;
        return;
    }
    const int K = g.iters;
    for (int r = 0; r < g.D; ++r) w[g.u(0, r)] = 0.0f;
    for (int k = 1; k <= K; ++k) generic_sweep(g, w, k - 1, k);
    // three more rows behind the history: ubar [D], glognu [D], vbar [NC]
    const size_t ubar0 = g.u(K + 1, 0), gln0 = ubar0 + (size_t)g.D * g.stride, vbar0 = gln0 + (size_t)g.D * g.stride;
    for (int r = 0; r < g.D; ++r) { w[ubar0 + r * g.stride] = 0.0f; w[gln0 + r * g.stride] = 0.0f; }
    for (int c = 0; c < g.NC; ++c) w[vbar0 + c * g.stride] = 0.0f;
    for (int r = 0; r < g.D; ++r)
        for (int c = 0; c < g.NC; ++c) {
            const float cost = generic_cost(g, w, r, c);
            const float uK = K > 0 ? w[g.u(K, r)] : 0.0f, vK = K > 0 ? w[g.v(K, c)] : 0.0f;
            const float wt = expf(cost / g.eps + uK + vK) * cost;
            w[ubar0 + r * g.stride] += wt;
            w[vbar0 + c * g.stride] += wt;
        }
    for (int k = K; k >= 1; --k) {
        for (int r = 0; r < g.D; ++r) w[gln0 + r * g.stride] += w[ubar0 + r * g.stride];
        for (int c = 0; c < g.NC; ++c) {
            float acc = 0.0f;
            for (int r = 0; r < g.D; ++r)
                acc = fmaf(w[ubar0 + r * g.stride], expf(generic_cost(g, w, r, c) / g.eps + w[g.v(k, c)] + w[g.u(k, r)] - w[g.log_nu(r)]), acc);
            w[vbar0 + c * g.stride] -= acc;
        }
        for (int r = 0; r < g.D; ++r) {
            float acc = 0.0f;
            for (int c = 0; c < g.NC; ++c)
                acc = fmaf(w[vbar0 + c * g.stride], expf(generic_cost(g, w, r, c) / g.eps + w[g.u(k - 1, r)] + w[g.v(k, c)] - w[g.log_mu(c)]), acc);
            w[ubar0 + r * g.stride] = -acc;
        }
        for (int c = 0; c < g.NC; ++c) w[vbar0 + c * g.stride] = 0.0f;
    }
    const float gp = gpix[bp];
    for (int d = 0; d < g.D; ++d) gattn[bdp + (size_t)d * HW] = gp * w[gln0 + d * g.stride] / (attn[bdp + (size_t)d * HW] + 1e-12f);
}

constexpr int kGenericSinkhornMaxD = 64;

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

// Generic route: any D <= 64 (continuous: D >= 3), any iters >= 0.  workspace: mvster_sinkhorn_generic_workspace() floats.
extern "C" long long mvster_sinkhorn_generic_workspace(int B, int D, int H, int W, int iters, int continuous, int backward) {
    if (B <= 0 || D <= 0 || D > mvster::kGenericSinkhornMaxD || H <= 0 || W <= 0 || iters < 0) return -1;
    const long long NC = D + (continuous ? 1 : 0), px = (long long)B * H * W;
    const long long sweeps = backward ? iters + 1 : 1;
    return px * (2 * D + NC + sweeps * (D + NC) + (backward ? 2 * D + NC : 0));
}

static int check_generic(const void* a, const void* b, const void* c, const void* d, const void* e, const void* ws, long long have,
                         int B, int D, int H, int W, int iters, float eps, int continuous, int backward) {
    MVSTER_REQUIRE_PTR(a); MVSTER_REQUIRE_PTR(b); MVSTER_REQUIRE_PTR(c); MVSTER_REQUIRE_PTR(d); MVSTER_REQUIRE_PTR(e); MVSTER_REQUIRE_PTR(ws);
    MVSTER_REQUIRE_SHAPE(B > 0 && B <= 65535 && D > 0 && H > 0 && W > 0 && (long long)H * W <= mvster::kMaxPlane);
    MVSTER_REQUIRE_SHAPE(iters >= 0 && eps != 0.0f && (!continuous || D >= 3));
    if (D > mvster::kGenericSinkhornMaxD) return MVSTER_ERR_UNSUPPORTED;
    MVSTER_REQUIRE_SHAPE(have >= mvster_sinkhorn_generic_workspace(B, D, H, W, iters, continuous, backward));
    return MVSTER_OK;
}

extern "C" int mvster_sinkhorn_generic_fwd(const float* gt_depth, const float* hypo, const float* attn, const uint8_t* mask,
                                           float* pix_loss, float* tmap, float* workspace, long long workspace_floats, int B, int D,
                                           int H, int W, int iters, float eps, int continuous, mvster_stream_t stream) {
    using namespace mvster;
    if (int rc = check_generic(gt_depth, hypo, attn, mask, pix_loss, workspace, workspace_floats, B, D, H, W, iters, eps, continuous, 0))
        return rc;
    const int HW = H * W;
    GenericSinkhorn g{D, D + (continuous ? 1 : 0), continuous ? 1 : 0, iters, eps, (size_t)B * HW};
    hipLaunchKernelGGL(sinkhorn_generic_fwd_kernel, dim3(grid_for(HW, kBlock), B), dim3(kBlock), 0, as_stream(stream), gt_depth, hypo,
                       attn, mask, pix_loss, tmap, workspace, HW, g);
    return launch_status();
}

extern "C" int mvster_sinkhorn_generic_bwd(const float* gt_depth, const float* hypo, const float* attn, const uint8_t* mask,
                                           const float* gpix, float* gattn, float* workspace, long long workspace_floats, int B,
                                           int D, int H, int W, int iters, float eps, int continuous, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(gpix);
    if (int rc = check_generic(gt_depth, hypo, attn, mask, gattn, workspace, workspace_floats, B, D, H, W, iters, eps, continuous, 1))
        return rc;
    const int HW = H * W;
    GenericSinkhorn g{D, D + (continuous ? 1 : 0), continuous ? 1 : 0, iters, eps, (size_t)B * HW};
    hipLaunchKernelGGL(sinkhorn_generic_bwd_kernel, dim3(grid_for(HW, kBlock), B), dim3(kBlock), 0, as_stream(stream), gt_depth, hypo,
                       attn, mask, gpix, gattn, workspace, HW, g);
    return launch_status();
}
