// F3  top-down merge of the feature pyramid, one fused operator each way:
//        y = bilinear_x2(feat, align_corners=True) + conv1x1(skip; weight, bias)
// replaces: `F.interpolate(intra_feat, scale_factor=2, mode="bilinear", align_corners=True) + self.innerK(convK)` in
// FPN4.forward (models/mvs4net_utils.py:490-505), i.e. ATen's upsample kernel, a MIOpen/rocBLAS 1x1 convolution, an
// element-wise add -- and, backwards, ATen's atomic upsample backward, two convolution-gradient kernels with their
// layout transposes and a bias reduction.  At DTU resolution the merged map has 64 channels at FULL resolution
// (472 MB per view): every one of those operators is a pass over it.  Here the forward writes it once and the backward
// reads its gradient three times (skip gradient, weight/bias gradient, transposed upsample), all streaming.
//
// Layout: NCHW planes, one lane per pixel, so every access to a channel plane is a coalesced row segment.
#include "mvster_device.h"
#include "mvster_host.h"

namespace mvster {

constexpr int kMaxTop = 64;       // output channels (8 * base_channels)

// PyTorch's align_corners=True source index: src = dst * (in - 1) / (out - 1) in float; i0 = (int)src,
// i1 = i0 + (i0 < in - 1), lambda1 = src - i0, lambda0 = 1 - lambda1   (aten/src/ATen/native/UpSample.h)
struct Lerp {
    int i0, i1;
    float l0, l1;
};

__device__ __forceinline__ Lerp source_index(int dst, float scale, int in) {
    const float src = scale * (float)dst;
    Lerp r;
    r.i0 = (int)src;
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.0f - r.l1;
    return r;
}

struct MergeDims {
    int N, Ct, Cs, h, w, H, W;    // feat [N,Ct,h,w], skip [N,Cs,H,W], y [N,Ct,H,W] with H = 2h, W = 2w
    float sy, sx;                  // (h - 1) / (H - 1), (w - 1) / (W - 1)
};

template <int CS>
__global__ void __launch_bounds__(kBlock)
fpn_merge_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ skip, const float* __restrict__ weight,
                     const float* __restrict__ bias, float* __restrict__ y, MergeDims dm) {
    __shared__ float wl[kMaxTop * CS], bl[kMaxTop];
    for (int i = threadIdx.x; i < dm.Ct * CS; i += kBlock) wl[i] = weight[i];
    for (int i = threadIdx.x; i < dm.Ct; i += kBlock) bl[i] = bias != nullptr ? bias[i] : 0.0f;
    __syncthreads();
    const int n = blockIdx.y;
    const int HW = dm.H * dm.W, hw = dm.h * dm.w;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= HW) return;
    const int yy = p / dm.W, xx = p - yy * dm.W;
    const Lerp ly = source_index(yy, dm.sy, dm.h), lx = source_index(xx, dm.sx, dm.w);
    const int o00 = ly.i0 * dm.w + lx.i0, o01 = ly.i0 * dm.w + lx.i1, o10 = ly.i1 * dm.w + lx.i0, o11 = ly.i1 * dm.w + lx.i1;
    float sk[CS];
#pragma unroll
    for (int ci = 0; ci < CS; ++ci) sk[ci] = skip[((size_t)n * CS + ci) * HW + p];
    const float* __restrict__ fb = feat + (size_t)n * dm.Ct * hw;
    float* __restrict__ yb = y + (size_t)n * dm.Ct * HW + p;
    for (int c = 0; c < dm.Ct; ++c) {
        const float* __restrict__ f = fb + (size_t)c * hw;
        const float up = ly.l0 * (lx.l0 * f[o00] + lx.l1 * f[o01]) + ly.l1 * (lx.l0 * f[o10] + lx.l1 * f[o11]);
        float lat = bl[c];
#pragma unroll
        for (int ci = 0; ci < CS; ++ci) lat = fmaf(wl[c * CS + ci], sk[ci], lat);
        yb[(size_t)c * HW] = up + lat;
    }
}

// dskip[ci] = sum_c weight[c,ci] * dy[c]   (the 1x1 convolution's input gradient)
template <int CS>
__global__ void __launch_bounds__(kBlock)
fpn_merge_bwd_skip_kernel(const float* __restrict__ dy, const float* __restrict__ weight, float* __restrict__ dskip,
                          MergeDims dm) {
    __shared__ float wl[kMaxTop * CS];
    for (int i = threadIdx.x; i < dm.Ct * CS; i += kBlock) wl[i] = weight[i];
    __syncthreads();
    const int n = blockIdx.y;
    const int HW = dm.H * dm.W;
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= HW) return;
    float acc[CS];
#pragma unroll
    for (int ci = 0; ci < CS; ++ci) acc[ci] = 0.0f;
    const float* __restrict__ d = dy + (size_t)n * dm.Ct * HW + p;
    for (int c = 0; c < dm.Ct; ++c) {
        const float g = d[(size_t)c * HW];
#pragma unroll
        for (int ci = 0; ci < CS; ++ci) acc[ci] = fmaf(wl[c * CS + ci], g, acc[ci]);
    }
#pragma unroll
    for (int ci = 0; ci < CS; ++ci) dskip[((size_t)n * CS + ci) * HW + p] = acc[ci];
}

// dweight[c,ci] = sum_{n,p} dy[n,c,p] * skip[n,ci,p],  dbias[c] = sum_{n,p} dy[n,c,p].
// Persistent workgroups; wave v of a workgroup owns 128/CS of the Ct channels per pass and keeps 128 (+ bias) running sums per lane
// over all the pixels the workgroup visits; one butterfly reduction and one atomic per sum at the very end.
template <int CS>
constexpr int kWChan = 128 / CS;      // 128 running products per lane: 16 / 8 / 4 channels for 8 / 16 / 32 skip channels

template <int CS>
__global__ void __launch_bounds__(kBlock)
fpn_merge_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ skip, float* __restrict__ dweight,
                            float* __restrict__ dbias, MergeDims dm) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int HW = dm.H * dm.W;
    const long long total = (long long)dm.N * HW;
    constexpr int WC = kWChan<CS>;
    for (int c0 = wave * WC; c0 < dm.Ct; c0 += (kBlock / kWave) * WC) {
        float acc[WC][CS], accb[WC];
#pragma unroll
        for (int k = 0; k < WC; ++k) {
            accb[k] = 0.0f;
#pragma unroll
            for (int ci = 0; ci < CS; ++ci) acc[k][ci] = 0.0f;
        }
        for (long long q = (long long)blockIdx.x * kWave + lane; q < total; q += (long long)gridDim.x * kWave) {
            const int n = (int)(q / HW), p = (int)(q - (long long)n * HW);
            float sk[CS];
#pragma unroll
            for (int ci = 0; ci < CS; ++ci) sk[ci] = skip[((size_t)n * CS + ci) * HW + p];
            const float* __restrict__ d = dy + ((size_t)n * dm.Ct + c0) * HW + p;
#pragma unroll
            for (int k = 0; k < WC; ++k) {
                const float g = c0 + k < dm.Ct ? d[(size_t)k * HW] : 0.0f;
                accb[k] += g;
#pragma unroll
                for (int ci = 0; ci < CS; ++ci) acc[k][ci] = fmaf(g, sk[ci], acc[k][ci]);
            }
        }
#pragma unroll
        for (int k = 0; k < WC; ++k) {
            const float sb = group_sum<kWave>(accb[k]);
            if (lane == 0 && c0 + k < dm.Ct && dbias != nullptr) atomicAdd(dbias + c0 + k, sb);
#pragma unroll
            for (int ci = 0; ci < CS; ++ci) {
                const float sw = group_sum<kWave>(acc[k][ci]);
                if (lane == 0 && c0 + k < dm.Ct) atomicAdd(dweight + (size_t)(c0 + k) * CS + ci, sw);
            }
        }
    }
}

// dfeat = transpose of the bilinear x2 upsample applied to dy, as a GATHER (ATen scatters with atomics): a low-resolution
// pixel (i,j) collects from the <= 6 x 6 high-resolution pixels whose source footprint can contain it, with exactly the
// forward's weights.  A workgroup owns a 32 x 8 tile of low-resolution pixels and walks kUpPlanes planes of it: per plane
// it stages the 68 x 20 high-resolution pixels the tile can touch in LDS with coalesced row loads (every dy element is read
// ~1.2 times; gathering straight from memory -- lanes two pixels apart, six rows each -- ran at ~1 TB/s), the next plane's
// window already in flight in registers while this one is reduced; the 6 + 6 interpolation weights are per lane, not per plane.
constexpr int kUpTw = 32, kUpTh = 8;                                   // low-resolution tile
constexpr int kUpW = 2 * kUpTw + 4, kUpH = 2 * kUpTh + 4;              // staged high-resolution window (2i-2 .. 2i+3)
constexpr int kUpPlanes = 8;                                           // planes per workgroup
constexpr int kUpLoads = (kUpH * kUpW + kBlock - 1) / kBlock;          // window elements per lane

__global__ void __launch_bounds__(kBlock)
fpn_upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dfeat, MergeDims dm, int planes) {
    __shared__ float win[kUpH][kUpW + 1];
    const int hw = dm.h * dm.w, HW = dm.H * dm.W;
    const int tiles_x = (dm.w + kUpTw - 1) / kUpTw;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int i0 = ty * kUpTh, j0 = tx * kUpTw;
    const int Y0 = 2 * i0 - 2, X0 = 2 * j0 - 2;
    int woff[kUpLoads], loff[kUpLoads];                            // the lane's window elements: offset in the plane (-1: outside), slot in LDS
#pragma unroll
    for (int k = 0; k < kUpLoads; ++k) {
        const int e = threadIdx.x + kBlock * k;
        const int r = e / kUpW, c = e - r * kUpW;
        const int yy = Y0 + r, xx = X0 + c;
        loff[k] = e < kUpH * kUpW ? r * (kUpW + 1) + c : -1;
        woff[k] = e < kUpH * kUpW && yy >= 0 && yy < dm.H && xx >= 0 && xx < dm.W ? yy * dm.W + xx : -1;
    }
    const int li = threadIdx.x / kUpTw, lj = threadIdx.x - li * kUpTw;
    const int i = i0 + li, j = j0 + lj;
    float wy[6], wx[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int yy = 2 * i - 2 + k, xx = 2 * j - 2 + k;
        wy[k] = 0.0f;
        wx[k] = 0.0f;
        if (yy >= 0 && yy < dm.H) {
            const Lerp l = source_index(yy, dm.sy, dm.h);
            wy[k] = (l.i0 == i ? l.l0 : 0.0f) + (l.i1 == i ? l.l1 : 0.0f);
        }
        if (xx >= 0 && xx < dm.W) {
            const Lerp l = source_index(xx, dm.sx, dm.w);
            wx[k] = (l.i0 == j ? l.l0 : 0.0f) + (l.i1 == j ? l.l1 : 0.0f);
        }
    }
    const int p0 = blockIdx.y * kUpPlanes, p1 = min(p0 + kUpPlanes, planes);
    float v[kUpLoads];
    auto fetch = [&](int plane) {
        const float* __restrict__ d = dy + (size_t)plane * HW;
#pragma unroll
        for (int k = 0; k < kUpLoads; ++k) v[k] = woff[k] >= 0 ? d[woff[k]] : 0.0f;
    };
    fetch(p0);
    for (int plane = p0; plane < p1; ++plane) {
        if (plane > p0) __syncthreads();                           // the previous plane's window has been consumed
#pragma unroll
        for (int k = 0; k < kUpLoads; ++k)
            if (loff[k] >= 0) (&win[0][0])[loff[k]] = v[k];
        __syncthreads();
        if (plane + 1 < p1) fetch(plane + 1);
        if (i < dm.h && j < dm.w) {
            float acc = 0.0f;
#pragma unroll
            for (int ky = 0; ky < 6; ++ky) {
                float r = 0.0f;
#pragma unroll
                for (int kx = 0; kx < 6; ++kx) r = fmaf(wx[kx], win[2 * li + ky][2 * lj + kx], r);
                acc = fmaf(wy[ky], r, acc);
            }
            dfeat[(size_t)plane * hw + (size_t)i * dm.w + j] = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Top pyramid level without its 64-channel full-resolution map.  The reference computes
//     intra = bilinear_x2(feat) + inner3(conv0)          [N,64,H,W]   (472 MB per view at DTU resolution)
//     out4  = conv3x3(intra, W4)                          [N, 8,H,W]
// (models/mvs4net_utils.py:500-505) and nothing else reads `intra`.  Both steps are linear, so the channel contraction of
// the 3x3 layer can be done BEFORE the upsampling, at a quarter of the pixels:
//     G[(co,tap)] = sum_c W4[co][c][tap] * (feat[c] + bias[c])      low resolution, 72 channels   (a 1x1 layer, csrc/conv_mfma.hip)
//     out4[co](Y,X) = sum_tap [ (Y,X)+tap inside the image ] * bilinear_x2(G[(co,tap)])((Y,X)+tap)   <- this kernel
//                   + conv3x3(conv0, W4 o W_inner3)(Y,X)                                              (an 8 -> 8 layer, `base`)
// Same function, different association: the 64-channel full-resolution tensor and the five passes over it (merge forward,
// three merge-backward passes, the 64 -> 8 layer's three operations) disappear, the flops drop 4x.
// Forward: a workgroup owns a 32 x 8 output tile and walks the eight output channels: the <= 8 x 20 low-resolution window of
// a channel's nine tap planes goes through LDS (double-buffered, the next channel's window in flight in registers) and every
// lane interpolates its pixel's nine shifted samples from there; the 36 offsets and weights are per lane, not per channel.
constexpr int kHeadTaps = 9, kHeadCo = 8, kHeadPlanes = kHeadTaps * kHeadCo;
constexpr int kHeadTw = 32, kHeadTh = 8;
constexpr int kHeadLw = 20, kHeadLh = 8;       // low-resolution window of a tile plus its one-pixel ring: <= 19 x 7 (scale < 1/2)

struct HeadDims {
    int N, h, w, H, W;
    float sy, sx;
    int out_nhwc;            // forward: out is [N,H,W,8]
};

constexpr int kHeadWin = kHeadLh * kHeadLw;
constexpr int kHeadLoads = (kHeadTaps * kHeadWin + kBlock - 1) / kBlock;   // window elements of one output channel's nine planes per lane

__global__ void __launch_bounds__(kBlock)
fpn_head_gather_fwd_kernel(const float* __restrict__ g, const float* __restrict__ base, float* __restrict__ out, HeadDims dm) {
    __shared__ float win[2][kHeadTaps][kHeadWin];                  // one output channel's nine tap planes, double-buffered
    const int n = blockIdx.y;
    const int tiles_x = (dm.W + kHeadTw - 1) / kHeadTw;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int Y0 = ty * kHeadTh, X0 = tx * kHeadTw;
    const int i_lo = source_index(max(Y0 - 1, 0), dm.sy, dm.h).i0, j_lo = source_index(max(X0 - 1, 0), dm.sx, dm.w).i0;
    const int hw = dm.h * dm.w, HW = dm.H * dm.W;
    const float* __restrict__ gn = g + (size_t)n * kHeadPlanes * hw;
    // this lane's window elements: the same positions for every output channel (its planes are 9 * hw further on)
    int goff[kHeadLoads];
#pragma unroll
    for (int k = 0; k < kHeadLoads; ++k) {
        const int e = threadIdx.x + kBlock * k;
        const int tap = e / kHeadWin, rem = e - tap * kHeadWin;
        const int r = rem / kHeadLw, c = rem - r * kHeadLw;
        const int i = i_lo + r, j = j_lo + c;
        goff[k] = e < kHeadTaps * kHeadWin && i < dm.h && j < dm.w ? tap * hw + i * dm.w + j : -1;
    }
    const int ly = threadIdx.x / kHeadTw, lx = threadIdx.x - ly * kHeadTw;
    const int Y = min(Y0 + ly, dm.H - 1), X = min(X0 + lx, dm.W - 1);
    const bool live = Y0 + ly < dm.H && X0 + lx < dm.W;
    int off[3][3][4];
    float wgt[3][3][4];                                            // zero weights for the taps that leave the image
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = Y + ky - 1;
        const bool oky = yy >= 0 && yy < dm.H;
        Lerp py = source_index(oky ? yy : Y, dm.sy, dm.h);
        py.i0 -= i_lo; py.i1 -= i_lo;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = X + kx - 1;
            const bool ok = oky && xx >= 0 && xx < dm.W;
            Lerp px = source_index(xx >= 0 && xx < dm.W ? xx : X, dm.sx, dm.w);
            px.i0 -= j_lo; px.i1 -= j_lo;
            const int t = (ky * 3 + kx) * kHeadWin;
            off[ky][kx][0] = t + py.i0 * kHeadLw + px.i0; off[ky][kx][1] = t + py.i0 * kHeadLw + px.i1;
            off[ky][kx][2] = t + py.i1 * kHeadLw + px.i0; off[ky][kx][3] = t + py.i1 * kHeadLw + px.i1;
            wgt[ky][kx][0] = ok ? py.l0 * px.l0 : 0.0f; wgt[ky][kx][1] = ok ? py.l0 * px.l1 : 0.0f;
            wgt[ky][kx][2] = ok ? py.l1 * px.l0 : 0.0f; wgt[ky][kx][3] = ok ? py.l1 * px.l1 : 0.0f;
        }
    }
    float v[kHeadLoads];
    auto fetch = [&](int co) {
        const float* __restrict__ p = gn + (size_t)co * kHeadTaps * hw;
#pragma unroll
        for (int k = 0; k < kHeadLoads; ++k) v[k] = goff[k] >= 0 ? p[goff[k]] : 0.0f;
    };
    fetch(0);
    const size_t o = (size_t)n * kHeadCo * HW + (size_t)Y * dm.W + X;
    float res[kHeadCo];                                            // channels-last output: the pixel's eight sums, stored together
#pragma unroll
    for (int co = 0; co < kHeadCo; ++co) {
        float* __restrict__ w = &win[co & 1][0][0];
#pragma unroll
        for (int k = 0; k < kHeadLoads; ++k) {
            const int e = threadIdx.x + kBlock * k;
            if (e < kHeadTaps * kHeadWin) w[e] = v[k];
        }
        __syncthreads();                                           // (the other buffer was last read two channels ago)
        if (co + 1 < kHeadCo) fetch(co + 1);
        float acc = live ? base[o + (size_t)co * HW] : 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                acc += (wgt[ky][kx][0] * w[off[ky][kx][0]] + wgt[ky][kx][1] * w[off[ky][kx][1]]) +
                       (wgt[ky][kx][2] * w[off[ky][kx][2]] + wgt[ky][kx][3] * w[off[ky][kx][3]]);
        res[co] = acc;
        if (live && !dm.out_nhwc) out[o + (size_t)co * HW] = acc;
    }
    if (live && dm.out_nhwc) {
        float4* __restrict__ op = reinterpret_cast<float4*>(out + ((size_t)n * HW + (size_t)Y * dm.W + X) * kHeadCo);
        op[0] = make_float4(res[0], res[1], res[2], res[3]);
        op[1] = make_float4(res[4], res[5], res[6], res[7]);
    }
}

// Backward of the gather w.r.t. G: dG[(co,tap)] = transposed upsample of dy[co] shifted by the tap.  Same tiling as
// fpn_upsample_bwd_kernel (32 x 8 low-resolution pixels, the high-resolution window they can touch in LDS, one ring wider
// for the shifts); a lane keeps its 8 x 8 window values in registers and forms the nine taps' sums from them.
constexpr int kHeadBw = 2 * kUpTw + 6, kHeadBh = 2 * kUpTh + 6;

__global__ void __launch_bounds__(kBlock)
fpn_head_gather_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dg, HeadDims dm) {
    __shared__ float win[kHeadBh][kHeadBw + 1];
    const int n = blockIdx.y / kHeadCo, co = blockIdx.y - n * kHeadCo;
    const int hw = dm.h * dm.w, HW = dm.H * dm.W;
    const int tiles_x = (dm.w + kUpTw - 1) / kUpTw;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int i0 = ty * kUpTh, j0 = tx * kUpTw;
    const int Y0 = 2 * i0 - 3, X0 = 2 * j0 - 3;                    // window origin: positions 2i-2 .. 2i+3, sources one further out
    const float* __restrict__ d = dy + ((size_t)n * kHeadCo + co) * HW;
    for (int e = threadIdx.x; e < kHeadBh * kHeadBw; e += kBlock) {
        const int r = e / kHeadBw, c = e - r * kHeadBw;
        const int yy = Y0 + r, xx = X0 + c;
        win[r][c] = yy >= 0 && yy < dm.H && xx >= 0 && xx < dm.W ? d[(size_t)yy * dm.W + xx] : 0.0f;
    }
    __syncthreads();
    const int li = threadIdx.x / kUpTw, lj = threadIdx.x - li * kUpTw;
    const int i = i0 + li, j = j0 + lj;
    if (i >= dm.h || j >= dm.w) return;
    float wy[6], wx[6];                                            // weight of position 2i-2+k (2j-2+k) on this low-resolution pixel
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int yy = 2 * i - 2 + k, xx = 2 * j - 2 + k;
        wy[k] = 0.0f;
        wx[k] = 0.0f;
        if (yy >= 0 && yy < dm.H) {
            const Lerp l = source_index(yy, dm.sy, dm.h);
            wy[k] = (l.i0 == i ? l.l0 : 0.0f) + (l.i1 == i ? l.l1 : 0.0f);
        }
        if (xx >= 0 && xx < dm.W) {
            const Lerp l = source_index(xx, dm.sx, dm.w);
            wx[k] = (l.i0 == j ? l.l0 : 0.0f) + (l.i1 == j ? l.l1 : 0.0f);
        }
    }
    float v[8][8];                                                 // dy at rows 2i-3 .. 2i+4, columns 2j-3 .. 2j+4
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) v[a][b] = win[2 * li + a][2 * lj + b];
    float* __restrict__ o = dg + ((size_t)n * kHeadPlanes + co * kHeadTaps) * hw + (size_t)i * dm.w + j;
#pragma unroll
    for (int ty3 = 0; ty3 < 3; ++ty3)
#pragma unroll
        for (int tx3 = 0; tx3 < 3; ++tx3) {
            // the sample at position p = (Y,X) + tap came from output pixel p - tap: rows 2i-2+ky - (ty3-1) = window row ky + 2 - ty3
            float acc = 0.0f;
#pragma unroll
            for (int ky = 0; ky < 6; ++ky) {
                float r = 0.0f;
#pragma unroll
                for (int kx = 0; kx < 6; ++kx) r = fmaf(wx[kx], v[ky + 2 - ty3][kx + 2 - tx3], r);
                acc = fmaf(wy[ky], r, acc);
            }
            o[(size_t)(ty3 * 3 + tx3) * hw] = acc;
        }
}

static int check_merge(int N, int Ct, int Cs, int h, int w, MergeDims* dm) {
    MVSTER_REQUIRE_SHAPE(N > 0 && N <= 65535 && Ct > 0 && Cs > 0 && h > 1 && w > 1);
    MVSTER_REQUIRE_SHAPE((long long)4 * h * w <= kMaxPlane && (long long)N * Ct <= 65535);
    if (Ct > kMaxTop || !(Cs == 8 || Cs == 16 || Cs == 32)) return MVSTER_ERR_UNSUPPORTED;
    *dm = MergeDims{N, Ct, Cs, h, w, 2 * h, 2 * w, (float)(h - 1) / (float)(2 * h - 1), (float)(w - 1) / (float)(2 * w - 1)};
    return MVSTER_OK;
}

}  // namespace mvster

#define MVSTER_BY_SKIP(KERNEL, GRID, ...)                                                                \
    switch (Cs) {                                                                                        \
        case 8: hipLaunchKernelGGL((KERNEL<8>), GRID, dim3(kBlock), 0, st, __VA_ARGS__); break;          \
        case 16: hipLaunchKernelGGL((KERNEL<16>), GRID, dim3(kBlock), 0, st, __VA_ARGS__); break;        \
        default: hipLaunchKernelGGL((KERNEL<32>), GRID, dim3(kBlock), 0, st, __VA_ARGS__); break;        \
    }

extern "C" int mvster_fpn_merge_fwd(const float* feat, const float* skip, const float* weight, const float* bias, float* y,
                                    int N, int Ct, int Cs, int h, int w, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(feat); MVSTER_REQUIRE_PTR(skip); MVSTER_REQUIRE_PTR(weight); MVSTER_REQUIRE_PTR(y);
    MergeDims dm;
    if (int rc = check_merge(N, Ct, Cs, h, w, &dm)) return rc;
    const hipStream_t st = as_stream(stream);
    const dim3 grid(grid_for((long long)dm.H * dm.W, kBlock), N);
    MVSTER_BY_SKIP(fpn_merge_fwd_kernel, grid, feat, skip, weight, bias, y, dm)
    return launch_status();
}

extern "C" int mvster_fpn_merge_bwd(const float* dy, const float* skip, const float* weight, float* dfeat, float* dskip,
                                    float* dweight, float* dbias, int N, int Ct, int Cs, int h, int w,
                                    mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(dy); MVSTER_REQUIRE_PTR(skip); MVSTER_REQUIRE_PTR(weight);
    MVSTER_REQUIRE_PTR(dfeat); MVSTER_REQUIRE_PTR(dskip); MVSTER_REQUIRE_PTR(dweight);
    MergeDims dm;
    if (int rc = check_merge(N, Ct, Cs, h, w, &dm)) return rc;
    const hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(dweight, 0, sizeof(float) * (size_t)Ct * Cs, st) != hipSuccess ||
        (dbias != nullptr && hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)Ct, st) != hipSuccess)) {
        (void)hipGetLastError();
        return MVSTER_ERR_LAUNCH;
    }
    const dim3 grid(grid_for((long long)dm.H * dm.W, kBlock), N);
    MVSTER_BY_SKIP(fpn_merge_bwd_skip_kernel, grid, dy, weight, dskip, dm)
    if (int rc = launch_status()) return rc;
    // few, long-lived workgroups: every wave ends with 128 (+ bias) atomics onto the same Ct x Cs sums, and 1800 workgroups
    // doing that took ~1 ms per call whatever the level (same-address contention); two workgroups per CU take < 0.3 ms
    // (measured: 1800 workgroups 1.0-1.1 ms per call, 512 workgroups 0.36-0.49 ms); and at least 32 pixels per lane so
    // that the epilogue (128 butterflies + 128 one-lane atomics per wave and pass) does not dominate the small levels
    const long long want = ((long long)N * dm.H * dm.W + 32 * kWave - 1) / (32 * kWave);
    const dim3 wgrid((unsigned)(want < 1 ? 1 : want < 512 ? want : 512));
    MVSTER_BY_SKIP(fpn_merge_bwd_weight_kernel, wgrid, dy, skip, dweight, dbias, dm)
    if (int rc = launch_status()) return rc;
    static_assert(kUpTw * kUpTh == kBlock, "one lane per low-resolution pixel of the tile");
    hipLaunchKernelGGL(fpn_upsample_bwd_kernel,
                       dim3(((w + kUpTw - 1) / kUpTw) * ((h + kUpTh - 1) / kUpTh), (N * Ct + kUpPlanes - 1) / kUpPlanes),
                       dim3(kBlock), 0, st, dy, dfeat, dm, N * Ct);
    return launch_status();
}

extern "C" int mvster_fpn_head_gather_fwd(const float* g, const float* base, float* out, int N, int h, int w,
                                          mvster_stream_t stream) {
    return mvster_fpn_head_gather_fwd_layout(g, base, out, N, h, w, MVSTER_LAYOUT_NCHW, stream);
}

extern "C" int mvster_fpn_head_gather_fwd_layout(const float* g, const float* base, float* out, int N, int h, int w,
                                                 int out_layout, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(g); MVSTER_REQUIRE_PTR(base); MVSTER_REQUIRE_PTR(out);
    MVSTER_REQUIRE_SHAPE(N > 0 && N <= 65535 && h > 1 && w > 1 && (long long)4 * h * w <= kMaxPlane);
    MVSTER_REQUIRE_SHAPE(out_layout == MVSTER_LAYOUT_NCHW || (out_layout == MVSTER_LAYOUT_NHWC && out != base));
    const HeadDims dm{N, h, w, 2 * h, 2 * w, (float)(h - 1) / (float)(2 * h - 1), (float)(w - 1) / (float)(2 * w - 1),
                      out_layout == MVSTER_LAYOUT_NHWC ? 1 : 0};
    const dim3 grid(((dm.W + kHeadTw - 1) / kHeadTw) * ((dm.H + kHeadTh - 1) / kHeadTh), N);
    hipLaunchKernelGGL(fpn_head_gather_fwd_kernel, grid, dim3(kBlock), 0, as_stream(stream), g, base, out, dm);
    return launch_status();
}

extern "C" int mvster_fpn_head_gather_bwd(const float* dy, float* dg, int N, int h, int w, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(dy); MVSTER_REQUIRE_PTR(dg);
    MVSTER_REQUIRE_SHAPE(N > 0 && (long long)N * kHeadCo <= 65535 && h > 1 && w > 1 && (long long)4 * h * w <= kMaxPlane);
    const HeadDims dm{N, h, w, 2 * h, 2 * w, (float)(h - 1) / (float)(2 * h - 1), (float)(w - 1) / (float)(2 * w - 1), 0};
    const dim3 grid(((w + kUpTw - 1) / kUpTw) * ((h + kUpTh - 1) / kUpTh), N * kHeadCo);
    hipLaunchKernelGGL(fpn_head_gather_bwd_kernel, grid, dim3(kBlock), 0, as_stream(stream), dy, dg, dm);
    return launch_status();
}
