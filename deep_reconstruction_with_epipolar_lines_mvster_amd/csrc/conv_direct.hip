// F2 / F3  direct convolution for the small-channel, full-resolution layers of FPN4 / reg2d / the mono head.
// replaces: nn.Conv2d / the (1,3,3) nn.Conv3d slices with stride 1, "same" padding, kernel 1x1 or 3x3 and few output
// channels -- FPN4 conv0.0 (3->8), conv0.1 (8->8), out3 (64->16), out4 (64->8), reg2d conv0 (G->8) and prob (8->1), the mono
// head's c->1 layers (models/mvs4net_utils.py:436-440,452-455,860-870,890,914) -- forward, input gradient and weight
// gradient.
//
// Why.  These layers move hundreds of MB per call for little arithmetic, and MIOpen's Winograd / implicit-GEMM kernels
// are tiled for many output channels: measured on MI355X at DTU resolution (tools/probes/conv_probe.py), 8->8 3x3 at
// 1152x1600 runs at 8.6 TFLOP/s (0.25 ms for 118 MB of activations), 64->8 at 22 TFLOP/s (0.77 ms forward, 1.49 ms
// backward, five times per step), the 8->1 1x1 at 0.49 ms for 265 MB -- 4-10x off the memory floor -- while the
// many-channel layers (96->32, 192->64) reach 70-90 TFLOP/s and stay on MIOpen.
//
// Forward / input gradient: one lane per output pixel, all COUT accumulators in registers, the K*K taps of one input
// channel read as coalesced row segments of its plane (neighbouring lanes and the next taps hit L1), the weights of one
// (input channel, tap) -- wave-uniform -- fetched by scalar loads from a [CIN][K*K][COUT] copy.  The input gradient is the
// same kernel on dy with flipped, transposed weights.
// Weight gradient: dW[co,ci,tap] = sum_p dy[co,p] x[ci,p+tap].  A workgroup owns a 64x16 pixel tile: dy of the tile (all
// COUT channels) sits in LDS, the input channels go by one at a time with their haloed tile; thread (co, row of taps)
// walks the pixels of the tile with 3 running sums; partial sums per tile are reduced by a second tiny kernel
// (no atomics, bitwise reproducible).
#include "conv_epilogue.h"
#include "mvster_device.h"
#include "mvster_host.h"

namespace mvster {

struct ConvDims {
    int N, CIN, H, W;
    int out_nhwc;            // forward: y is [N,H,W,COUT]
    ConvEpilogue ep;         // inference: BatchNorm (running statistics) + ReLU + skip connection on the accumulator
};

// PX horizontally adjacent output pixels per lane.  PX = 2 (COUT >= 16): the K x (K+1) input window is loaded once and the two
// accumulator sets pair up into packed FMAs -- the many-accumulator cases (the 8->64 input gradient of the 64->8 layer,
// 64->16, 48->16) are bound by FMA issue: 0.57 -> 0.37, 0.26 -> 0.21, 0.62 -> 0.43 ms.  PX = 1 (COUT <= 8): those layers are
// bound by memory and want the lanes (64->8 forward 0.34 ms with one pixel per lane, 0.46 with two).
template <int COUT, int K, int PX, bool EP>
__global__ void __launch_bounds__(kBlock)
conv_direct_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
                       float* __restrict__ y, ConvDims dm) {
    constexpr int P = K / 2, KK = K * K, WIN = K + PX - 1;
    const int n = blockIdx.y, HW = dm.H * dm.W;
    const int per_row = (dm.W + PX - 1) / PX;
    const int q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= dm.H * per_row) return;
    const int yy = q / per_row, xx = PX * (q - yy * per_row);
    float acc[COUT][PX];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float b0 = bias != nullptr ? bias[co] : 0.0f;
#pragma unroll
        for (int j = 0; j < PX; ++j) acc[co][j] = b0;
    }
    // window offsets and validity do not depend on the channel: rows yy-P..yy+P, columns xx-P..xx+P+PX-1
    int off[K * WIN];
    bool ok[K * WIN];
#pragma unroll
    for (int t = 0; t < K * WIN; ++t) {
        const int iy = yy + t / WIN - P, ix = xx + t % WIN - P;
        ok[t] = iy >= 0 && iy < dm.H && ix >= 0 && ix < dm.W;
        off[t] = ok[t] ? iy * dm.W + ix : yy * dm.W + xx;
    }
    const float* __restrict__ xb = x + (size_t)n * dm.CIN * HW;
    for (int ci = 0; ci < dm.CIN; ++ci) {
        const float* __restrict__ plane = xb + (size_t)ci * HW;
        const float* __restrict__ w = wt + (size_t)ci * KK * COUT;                 // wave-uniform: scalar loads
        float v[K * WIN];
#pragma unroll
        for (int t = 0; t < K * WIN; ++t) v[t] = ok[t] ? plane[off[t]] : 0.0f;
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    const float wv = w[(ky * K + kx) * COUT + co];
#pragma unroll
                    for (int j = 0; j < PX; ++j) acc[co][j] = fmaf(v[ky * WIN + kx + j], wv, acc[co][j]);
                }
    }
    if constexpr (EP) {                      // inference: BatchNorm (running statistics) + ReLU on the way out; wave-uniform
#pragma unroll
        for (int co = 0; co < COUT; ++co) {  // coefficients: scalar loads
            const NormCoef k{dm.ep.norm[co], dm.ep.norm[dm.ep.Cpad + co], dm.ep.norm[2 * dm.ep.Cpad + co]};
#pragma unroll
            for (int j = 0; j < PX; ++j) acc[co][j] = normalise(k, acc[co][j], dm.ep.relu != 0);
        }
    }
    if constexpr (COUT % 4 == 0) {
        if (dm.out_nhwc) {                   // channels-last: the lane's COUT sums of a pixel are one contiguous run
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                if (xx + j >= dm.W) continue;
                const size_t idx = ((size_t)n * HW + (size_t)yy * dm.W + xx + j) * COUT;
#pragma unroll
                for (int c4 = 0; c4 < COUT / 4; ++c4) {
                    float4 v = make_float4(acc[4 * c4][j], acc[4 * c4 + 1][j], acc[4 * c4 + 2][j], acc[4 * c4 + 3][j]);
                    if constexpr (EP) {
                        if (dm.ep.residual != nullptr) {
                            const float4 rs = reinterpret_cast<const float4*>(dm.ep.residual + idx)[c4];
                            v.x += rs.x; v.y += rs.y; v.z += rs.z; v.w += rs.w;
                        }
                    }
                    reinterpret_cast<float4*>(y + idx)[c4] = v;
                }
            }
            return;
        }
    }
    const size_t base = (size_t)n * COUT * HW + (size_t)yy * dm.W + xx;
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int j = 0; j < PX; ++j)
            if (xx + j < dm.W) {
                float v = acc[co][j];
                if constexpr (EP) {
                    if (dm.ep.residual != nullptr) v += dm.ep.residual[base + (size_t)co * HW + j];
                }
                y[base + (size_t)co * HW + j] = v;
            }
}

// ---- weight gradient ------------------------------------------------------------------------------------------------
#ifndef MVSTER_WG_TILE_H
#define MVSTER_WG_TILE_H 16
#endif
constexpr int kWgTileW = 64, kWgTileH = MVSTER_WG_TILE_H;   // pixels of a workgroup's tile (64 x 8 tiles: no faster, and the scratch no longer fits the input tiles for COUT = 16)
constexpr int kDyRow = kWgTileW + 1;                 // LDS strides chosen so that the lanes of a wave -- same column, different
constexpr int kDyChan = kWgTileH * kDyRow + (8 - (kWgTileH * kDyRow) % 64 + 64) % 64;   // (channel, row) -- fall on different banks: row stride = 1, channel stride = 8 (mod 64)

struct WgradDims {
    int N, CIN, H, W, tiles_x, tiles_y;
    int rg_n;                // a (channel, ky) task is split over rg_n row groups of the tile
    int cib;                 // input channels staged per pass
};

// partial[tile][ci][co][tap] (+ [co] bias sums) for this workgroup's tile.  Per pass `cib` input channels are staged with
// their haloed tiles; thread (channel of the pass, ky, row group) walks its rows of the tile with a sliding window of the
// input row and keeps COUT x K running sums -- one new input value and COUT dy values (LDS broadcasts) feed COUT*K FMAs.
// The row groups' sums are added through LDS (the scratch aliases the input tiles, dead by then).
// (First version: one (co, ky) pair per thread, 2 LDS reads per 3 FMAs and one or two channels per pass: 1.34 ms for the
// 64->8 layer at DTU resolution against 1.22 ms for MIOpen's implicit GEMM.)
template <int COUT, int K>
__global__ void __launch_bounds__(kBlock)
conv_direct_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial, WgradDims dm) {
    constexpr int P = K / 2, KK = K * K;
    constexpr int XW = kWgTileW + 2 * P + (K == 1 ? 1 : 0), XH = kWgTileH + 2 * P;      // row stride: rows on different banks
    constexpr int XC = kWgTileW + 2 * P;                                                   // columns actually loaded
    extern __shared__ float lds[];                    // dy tile [COUT][kDyChan]; x tiles [cib][XH][XW] / reduction scratch
    float* __restrict__ dyl = lds;
    float* __restrict__ xl = dyl + (size_t)COUT * kDyChan;
    float* __restrict__ red = xl;                     // [threads][COUT*K] after the walk
    const int n = blockIdx.y, HW = dm.H * dm.W;
    const int ty = blockIdx.x / dm.tiles_x, tx = blockIdx.x - ty * dm.tiles_x;
    const int x0 = tx * kWgTileW, y0 = ty * kWgTileH;
    // dy tile: eight loads in flight per thread (a plain copy loop waits for every load before the next)
    for (int e0 = 0; e0 < COUT * kWgTileH * kWgTileW; e0 += 8 * kBlock) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * kBlock + threadIdx.x;
            const int co = e / (kWgTileH * kWgTileW), r = (e / kWgTileW) % kWgTileH, c = e % kWgTileW;
            const int gy = y0 + r, gx = x0 + c;
            v[u] = e < COUT * kWgTileH * kWgTileW && gy < dm.H && gx < dm.W
                       ? dy[((size_t)n * COUT + co) * HW + (size_t)gy * dm.W + gx] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * kBlock + threadIdx.x;
            const int co = e / (kWgTileH * kWgTileW), r = (e / kWgTileW) % kWgTileH, c = e % kWgTileW;
            if (e < COUT * kWgTileH * kWgTileW) dyl[co * kDyChan + r * kDyRow + c] = v[u];
        }
    }
    const int slots = dm.cib * K;                     // (channel of the pass, ky)
    const int slot = threadIdx.x / dm.rg_n, rg = threadIdx.x - slot * dm.rg_n;
    const bool working = slot < slots;
    const int cl = working ? slot / K : 0, ky = working ? slot - cl * K : 0;
    const int row = dm.CIN * COUT * KK + COUT;                                  // + the tile's bias-gradient sums
    float* __restrict__ out = partial + ((size_t)(n * dm.tiles_y + ty) * dm.tiles_x + tx) * row;
    // The haloed tiles of the NEXT pass's channels are fetched into registers while this pass walks: the global latency
    // hides behind the arithmetic (with the copy inside the pass, its ~37 dependent loads per thread cost more than the walk).
    constexpr int kMaxStage = (8 * XH * XC + kBlock - 1) / kBlock;
    float stage[kMaxStage];
    const int stage_n = (dm.cib * XH * XC + kBlock - 1) / kBlock;
    auto fetch = [&](int ci0) {
#pragma unroll
        for (int u = 0; u < kMaxStage; ++u) {
            const int e = threadIdx.x + u * kBlock;
            float v = 0.0f;
            if (u < stage_n && e < dm.cib * XH * XC) {
                const int k = e / (XH * XC), rem = e - k * (XH * XC);
                const int r = rem / XC, c = rem - r * XC;
                const int gy = y0 + r - P, gx = x0 + c - P, ci = ci0 + k;
                if (ci < dm.CIN && gy >= 0 && gy < dm.H && gx >= 0 && gx < dm.W)
                    v = x[((size_t)n * dm.CIN + ci) * HW + (size_t)gy * dm.W + gx];
            }
            stage[u] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < kMaxStage; ++u) {
            const int e = threadIdx.x + u * kBlock;
            if (u < stage_n && e < dm.cib * XH * XC) {
                const int k = e / (XH * XC), rem = e - k * (XH * XC);
                const int r = rem / XC, c = rem - r * XC;
                xl[(k * XH + r) * XW + c] = stage[u];
            }
        }
    };
    fetch(0);
    for (int ci0 = 0; ci0 < dm.CIN; ci0 += dm.cib) {
        __syncthreads();                              // dy tile loaded / previous pass's sums read
        commit();
        __syncthreads();
        if (ci0 + dm.cib < dm.CIN) fetch(ci0 + dm.cib);
        float acc[COUT][K], accb[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            accb[co] = 0.0f;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc[co][kx] = 0.0f;
        }
        if (working) {
            for (int r = rg; r < kWgTileH; r += dm.rg_n) {
                const float* __restrict__ xr = xl + ((cl * XH) + r + ky) * XW;
                const float* __restrict__ dr = dyl + r * kDyRow;
                float win[K];                                                   // sliding window of the input row
#pragma unroll
                for (int kx = 0; kx + 1 < K; ++kx) win[kx + 1] = xr[kx];
#pragma unroll 4
                for (int c = 0; c < kWgTileW; ++c) {
#pragma unroll
                    for (int kx = 0; kx + 1 < K; ++kx) win[kx] = win[kx + 1];
                    win[K - 1] = xr[c + K - 1];
#pragma unroll
                    for (int co = 0; co < COUT; ++co) {
                        const float g = dr[co * kDyChan + c];
                        accb[co] += g;
#pragma unroll
                        for (int kx = 0; kx < K; ++kx) acc[co][kx] = fmaf(g, win[kx], acc[co][kx]);
                    }
                }
            }
        }
        __syncthreads();                              // everyone is done with the input tiles: reuse them as scratch
        if (working) {
#pragma unroll
            for (int co = 0; co < COUT; ++co)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) red[(size_t)threadIdx.x * (COUT * K) + co * K + kx] = acc[co][kx];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < dm.cib * COUT * KK; e += kBlock) {       // e = (k * COUT + co) * KK + ky * K + kx
            const int kx = e % K, kyy = (e / K) % K, co = (e / KK) % COUT, k = e / (KK * COUT);
            if (ci0 + k >= dm.CIN) continue;
            const int base = (k * K + kyy) * dm.rg_n;                          // first thread of that (channel, ky)
            float tot = 0.0f;
            for (int sh = 0; sh < dm.rg_n; ++sh) tot += red[(size_t)(base + sh) * (COUT * K) + co * K + kx];
            out[(size_t)(ci0 + k) * COUT * KK + co * KK + kyy * K + kx] = tot;
        }
        if (ci0 == 0) {                               // bias gradient: the (channel 0, ky 0) threads saw every dy of the tile
            __syncthreads();
            if (working && cl == 0 && ky == 0) {
#pragma unroll
                for (int co = 0; co < COUT; ++co) red[rg * COUT + co] = accb[co];
            }
            __syncthreads();
            for (int co = threadIdx.x; co < COUT; co += kBlock) {
                float tot = 0.0f;
                for (int sh = 0; sh < dm.rg_n; ++sh) tot += red[sh * COUT + co];
                out[(size_t)dm.CIN * COUT * KK + co] = tot;
            }
        }
    }
}

// Tree reduction of the per-tile rows: out[r][e] = sum of in[r * kFan + 0 .. kFan)[e]; applied until one row is left.
constexpr int kFan = 32;

__global__ void __launch_bounds__(kBlock)
conv_direct_rows_reduce_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int width) {
    const int e = blockIdx.x * kBlock + threadIdx.x, r = blockIdx.y;
    if (e >= width) return;
    const int lo = r * kFan, hi = min(rows, lo + kFan);
    float acc = 0.0f;
    for (int t = lo; t < hi; ++t) acc += in[(size_t)t * width + e];
    out[(size_t)r * width + e] = acc;
}

// final row [ci][co][tap] (+ bias sums) -> dweight [co][ci][tap], dbias
__global__ void __launch_bounds__(kBlock)
conv_direct_wgrad_finish_kernel(const float* __restrict__ row, float* __restrict__ dweight, float* __restrict__ dbias, int CIN,
                                int COUT, int KK) {
    const int e = blockIdx.x * kBlock + threadIdx.x, total = CIN * COUT * KK;
    if (e < total) {
        const int tap = e % KK, co = (e / KK) % COUT, ci = e / (KK * COUT);
        dweight[((size_t)co * CIN + ci) * KK + tap] = row[e];
    } else if (e < total + COUT && dbias != nullptr) {
        dbias[e - total] = row[e];
    }
}

static bool conv_cout_supported(int c) { return c == 1 || c == 3 || c == 4 || c == 8 || c == 16 || c == 32 || c == 64; }

}  // namespace mvster

extern "C" int mvster_conv_direct_supported(int CIN, int COUT, int K) {
    using namespace mvster;
    return (K == 1 || K == 3) && CIN > 0 && conv_cout_supported(COUT) ? 1 : 0;
}

#define MVSTER_CONV_CASE(CO)                                                                                                \
    case CO: {                                                                                                              \
        constexpr int PX = CO >= 16 ? 2 : 1;                                                                                \
        const dim3 grid(grid_for((long long)H * ((W + PX - 1) / PX), kBlock), N);                                           \
        if constexpr (CO <= 16) {            /* the epilogue's coefficients sit in scalar registers: few-channel layers only */ \
            if (norm != nullptr) {                                                                                          \
                if (K == 3) hipLaunchKernelGGL((conv_direct_fwd_kernel<CO, 3, PX, true>), grid, dim3(kBlock), 0, st, x, wt, bias, y, dm); \
                else hipLaunchKernelGGL((conv_direct_fwd_kernel<CO, 1, PX, true>), grid, dim3(kBlock), 0, st, x, wt, bias, y, dm); \
                break;                                                                                                      \
            }                                                                                                               \
        }                                                                                                                   \
        if (K == 3) hipLaunchKernelGGL((conv_direct_fwd_kernel<CO, 3, PX, false>), grid, dim3(kBlock), 0, st, x, wt, bias, y, dm); \
        else hipLaunchKernelGGL((conv_direct_fwd_kernel<CO, 1, PX, false>), grid, dim3(kBlock), 0, st, x, wt, bias, y, dm);      \
        break;                                                                                                              \
    }

extern "C" int mvster_conv_direct_fwd(const float* x, const float* wt, const float* bias, float* y, int N, int CIN, int COUT,
                                      int H, int W, int K, mvster_stream_t stream) {
    return mvster_conv_direct_fwd_layout(x, wt, bias, y, N, CIN, COUT, H, W, K, MVSTER_LAYOUT_NCHW, stream);
}

extern "C" int mvster_conv_direct_fwd_layout(const float* x, const float* wt, const float* bias, float* y, int N, int CIN,
                                             int COUT, int H, int W, int K, int out_layout, mvster_stream_t stream) {
    return mvster_conv_direct_fwd_ep(x, wt, bias, y, N, CIN, COUT, H, W, K, out_layout, nullptr, nullptr, 0, stream);
}

extern "C" int mvster_conv_direct_fwd_ep(const float* x, const float* wt, const float* bias, float* y, int N, int CIN, int COUT,
                                         int H, int W, int K, int out_layout, const float* norm, const float* residual, int relu,
                                         mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(x); MVSTER_REQUIRE_PTR(wt); MVSTER_REQUIRE_PTR(y);
    MVSTER_REQUIRE_SHAPE(N > 0 && N <= 65535 && CIN > 0 && COUT > 0 && H > 0 && W > 0 && (long long)H * W <= kMaxPlane);
    MVSTER_REQUIRE_SHAPE(out_layout == MVSTER_LAYOUT_NCHW || out_layout == MVSTER_LAYOUT_NHWC);
    if (!mvster_conv_direct_supported(CIN, COUT, K)) return MVSTER_ERR_UNSUPPORTED;
    if (out_layout == MVSTER_LAYOUT_NHWC && COUT % 4 != 0) return MVSTER_ERR_UNSUPPORTED;
    // the epilogue exists for the few-channel layers the policy routes here (COUT <= 16); without `norm` neither ReLU nor
    // the residual is applied by the plain kernels
    if ((norm != nullptr && COUT > 16) || (norm == nullptr && (residual != nullptr || relu))) return MVSTER_ERR_UNSUPPORTED;
    const ConvDims dm{N, CIN, H, W, out_layout == MVSTER_LAYOUT_NHWC ? 1 : 0,
                      ConvEpilogue{norm, residual, relu ? 1 : 0, conv_norm_pad(COUT)}};
    const hipStream_t st = as_stream(stream);
    switch (COUT) {
        MVSTER_CONV_CASE(1) MVSTER_CONV_CASE(3) MVSTER_CONV_CASE(4) MVSTER_CONV_CASE(8)
        MVSTER_CONV_CASE(16) MVSTER_CONV_CASE(32) MVSTER_CONV_CASE(64)
        default: return MVSTER_ERR_UNSUPPORTED;
    }
    return launch_status();
}

extern "C" long long mvster_conv_direct_wgrad_workspace(int N, int CIN, int COUT, int H, int W, int K) {
    using namespace mvster;
    if (N <= 0 || CIN <= 0 || COUT <= 0 || H <= 0 || W <= 0 || !(K == 1 || K == 3)) return 0;
    const long long tiles = (long long)N * ((H + kWgTileH - 1) / kWgTileH) * ((W + kWgTileW - 1) / kWgTileW);
    const long long width = (long long)CIN * COUT * K * K + COUT;
    return (tiles + (tiles + mvster::kFan - 1) / mvster::kFan + 1) * width;      // tile rows + the reduction tree's ping-pong rows
}

extern "C" int mvster_conv_direct_wgrad(const float* x, const float* dy, float* dweight, float* dbias, float* workspace,
                                        long long workspace_floats, int N, int CIN, int COUT, int H, int W, int K,
                                        mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(x); MVSTER_REQUIRE_PTR(dy); MVSTER_REQUIRE_PTR(dweight); MVSTER_REQUIRE_PTR(workspace);
    MVSTER_REQUIRE_SHAPE(N > 0 && N <= 65535 && CIN > 0 && COUT > 0 && H > 0 && W > 0 && (long long)H * W <= kMaxPlane);
    if (!(K == 1 || K == 3)) return MVSTER_ERR_UNSUPPORTED;
    const long long need = mvster_conv_direct_wgrad_workspace(N, CIN, COUT, H, W, K);
    MVSTER_REQUIRE_SHAPE(workspace_floats >= need);
    const int P = K / 2;
    if (!(COUT == 1 || COUT == 8 || COUT == 16)) return MVSTER_ERR_UNSUPPORTED;
    // threads = cib channels x K tap rows x rg_n row groups; scratch [threads][COUT*K] must fit where the input tiles were
    const int xtile = (kWgTileH + 2 * P) * (kWgTileW + 2 * P + (K == 1 ? 1 : 0));
    int cib = 1;
    while (cib * 2 <= 8 && cib * 2 <= CIN) cib *= 2;
    int rg_n = kWgTileH;
    while (rg_n > 1 && cib * K * rg_n > kBlock) rg_n /= 2;
    while (cib > 1 && (size_t)cib * K * rg_n * COUT * K > (size_t)cib * xtile) cib /= 2;     // scratch aliasing the tiles
    if ((size_t)cib * K * rg_n * COUT * K > (size_t)cib * xtile) return MVSTER_ERR_UNSUPPORTED;
    const size_t lds = sizeof(float) * ((size_t)COUT * kDyChan + (size_t)cib * xtile);
    if (lds > 150 * 1024) return MVSTER_ERR_UNSUPPORTED;
    WgradDims dm{N, CIN, H, W, (W + kWgTileW - 1) / kWgTileW, (H + kWgTileH - 1) / kWgTileH, rg_n, cib};
    const hipStream_t st = as_stream(stream);
    const dim3 grid(dm.tiles_x * dm.tiles_y, N);
#define MVSTER_WGRAD_LAUNCH(CO, KK_)                                                                                      \
    {                                                                                                                      \
        if (lds > 64 * 1024 &&                                                                                             \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_direct_wgrad_kernel<CO, KK_>),                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {                      \
            (void)hipGetLastError();                                                                                       \
            return MVSTER_ERR_LAUNCH;                                                                                      \
        }                                                                                                                  \
        hipLaunchKernelGGL((conv_direct_wgrad_kernel<CO, KK_>), grid, dim3(kBlock), lds, st, x, dy, workspace, dm);        \
    }
    if (COUT == 1 && K == 1) MVSTER_WGRAD_LAUNCH(1, 1)
    else if (COUT == 1) MVSTER_WGRAD_LAUNCH(1, 3)
    else if (COUT == 8 && K == 1) MVSTER_WGRAD_LAUNCH(8, 1)
    else if (COUT == 8) MVSTER_WGRAD_LAUNCH(8, 3)
    else if (K == 1) MVSTER_WGRAD_LAUNCH(16, 1)
    else MVSTER_WGRAD_LAUNCH(16, 3)
#undef MVSTER_WGRAD_LAUNCH
    if (int rc = launch_status()) return rc;
    const int width = CIN * COUT * K * K + COUT;
    // reduction tree, ping-pong between the tile rows (dead after the first level) and the area behind them
    int rows = dm.tiles_x * dm.tiles_y * N;
    float* const area_a = workspace;
    float* const area_b = workspace + (size_t)rows * width;
    float* in = area_a;
    while (rows > 1) {
        const int out_rows = (rows + kFan - 1) / kFan;
        float* const out = in == area_a ? area_b : area_a;
        hipLaunchKernelGGL(conv_direct_rows_reduce_kernel, dim3(grid_for(width, kBlock), out_rows), dim3(kBlock), 0, st, in, out,
                           rows, width);
        if (int rc = launch_status()) return rc;
        in = out;
        rows = out_rows;
    }
    hipLaunchKernelGGL(conv_direct_wgrad_finish_kernel, dim3(grid_for(width, kBlock)), dim3(kBlock), 0, st, in, dweight, dbias, CIN,
                       COUT, K * K);
    return launch_status();
}
