// F2 / F3  many-channel convolutions of FPN4 / reg2d as implicit GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32:
// f32 in, f32 accumulate -- bit-for-bit a k-ordered fmaf chain, so no precision is given up against the reference's fp32).
// replaces: nn.Conv2d / the (1,3,3) and (3,3,3) nn.Conv3d layers (as conv2d over depth slices) with 16-64 output channels
// -- FPN4 conv1.x/conv2.x/conv3.x, the 5x5 stride-2 level heads, out1/out2/out3 (models/mvs4net_utils.py:441-455,
// 466-470), reg2d conv1..conv6 (models/mvs4net_utils.py:891-899) -- forward, input gradient (stride 1: the same kernel
// on dy with flipped, transposed weights) and weight gradient.  NCHW fp32 in and out, "same"-style padding K/2.
//
// Why.  MIOpen runs these shapes (8-64 channels, 10^5-10^6 pixels) at 15-45 TFLOP/s forward and pays two layout transposes
// around its NHWC implicit-GEMM weight-gradient kernels (tools/probes/conv_probe.py); the fp32 MFMA peak is 157 TFLOP/s.
//
// Forward, one workgroup = an 8 x 32 tile of output pixels x up to 64 output channels:
//   D[co 32][pixel 32] += A[co][ci pair] * B[ci pair][pixel]     per tap (ky,kx) and pair of input channels
//   A = weights, re-packed once per call to [ci][tap][co] and staged per CB input channels in LDS;
//   B = the haloed input tile of those CB channels in LDS, plane-major exactly as NCHW delivers it: lane = pixel reads
//       consecutive words (stride-2 layers: every other word, the second half-wave = the next channel sits on the odd banks).
//   wave w owns tile rows 2w, 2w+1: 2 x NB accumulators of 32x32.  Stores: for each accumulator register a half-wave
//   writes 32 consecutive pixels of one output-channel plane.
// Weight gradient, dW[co][ci][ky][kx] = sum_p dy[co][p] x[ci][p*S + (ky,kx) - P]:
//   D[co 32][(kx,ci) 32] += A[co][pixel pair] * B[pixel pair][(kx,ci)]   per ky; the GEMM's K runs over the pixels.
//   Both tiles stay plane-major in LDS; a plane stride of (32/CIB) * odd words makes the 32 (kx,ci) columns of a lane group
//   fall on 32 different banks.  Waves split the tile's rows, workgroups are persistent over tiles, partial sums are folded
//   through LDS per workgroup and across workgroups by a second small kernel (no atomics: bitwise reproducible).
#include "conv_epilogue.h"
#include "mvster_device.h"
#include "mvster_host.h"

namespace mvster {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef MVSTER_MF_CB_S1         // input channels staged per chunk by the stride-1 forward kernels.  Measured (TFLOP/s forward,
#define MVSTER_MF_CB_S1 8       // 32->32 / 64->32 / 96->32 / 1x1 64->72): 4: 85 / 99 / 102 / 36, 8: 83 / 97 / 100 / 39, 16: 70 / 88 / 91 / 39
#endif
#ifndef MVSTER_MF_FWD_WAVES     // > 0: ask the compiler for at least this many waves per SIMD in the forward kernels
#define MVSTER_MF_FWD_WAVES 0   // (3 instead of the natural 2 at 64 output channels: within 2 % either way)
#endif
constexpr int kMfTileW = 32;          // output pixels per MFMA column block
constexpr int kMfRowsFwd = 8;         // output rows per forward tile: two per wave

struct MfDims {
    int N, Cin, Cout, H, W, Ho, Wo;   // x [N,Cin,H,W] -> y [N,Cout,Ho,Wo]
    int CinPad, CoutPad;              // packed weights [CinPad][K*K][CoutPad]
    int tiles_x, tiles_y;
    int depth, Creal;                 // depth > 0: the Cin = 3 * Creal input channels are VIRTUAL, see input_plane()
    int out_nhwc;                     // forward kernels: y is [N,Ho,Wo,Cout] (the layout the cost-volume kernels gather from)
    ConvEpilogue ep;                  // forward / transposed kernels: inference BatchNorm + ReLU + skip connection on the accumulator
};

// (3,3,3) layers over depth slices stacked into the batch (sample n = b * depth + d): input channel cv = dz * Creal + c of
// sample n is channel c of sample n + dz - 1, zero where d + dz - 1 leaves [0, depth) -- the three depth-shifted copies the
// reference's Conv3d sees, read in place instead of from a concatenated tensor.  -> plane pointer, or nullptr for "zero".
__device__ __forceinline__ const float* input_plane(const float* __restrict__ x, int n, int cv, int Cin, int depth, int Creal,
                                                     int HW) {
    if (cv >= Cin) return nullptr;
    if (depth <= 0) return x + ((size_t)n * Cin + cv) * HW;
    const int dz = cv / Creal, c = cv - dz * Creal;
    const int dd = n % depth + dz - 1;
    return dd >= 0 && dd < depth ? x + ((size_t)(n + dz - 1) * Creal + c) * HW : nullptr;
}

__host__ __device__ constexpr int round_to_residue(int v, int mod, int res) { return v + ((res - v % mod) + mod) % mod; }

// wp[ci][tap][co], zero-padded.  mode 0: w is [Cout][Cin][KK] (a layer's own weights, forward).  mode 1: w is
// [Cin][Cout][KK] and the taps are mirrored (the input gradient of a stride-1 layer is the convolution of dy with these).
// mode 2: w is [Cin][Cout][KK], taps as they are (transposed convolution).
// mode 3: input gradient of a depth-stacked (3,3,3) layer: w is [Cin/3][3*Cout][KK] (the layer's folded weights, Cin/3 = its
//         output channels, Cout = its real input channels); virtual input channel cv = dz' * (Cin/3) + co reads dy of slice
//         n + dz' - 1, which saw this slice under depth tap 2 - dz': wp[cv][tap][c] = w[co][(2 - dz') * Cout + c][KK-1-tap].
__global__ void __launch_bounds__(kBlock)
conv_mfma_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int KK, int CinPad, int CoutPad,
                      int mode) {
    const int e = blockIdx.x * kBlock + threadIdx.x;
    if (e >= CinPad * KK * CoutPad) return;
    const int co = e % CoutPad, tap = (e / CoutPad) % KK, ci = e / (CoutPad * KK);
    float v = 0.0f;
    if (ci < Cin && co < Cout) {
        if (mode == 3) {
            const int cl = Cin / 3, dz = ci / cl, col = ci - dz * cl;
            v = w[((size_t)col * 3 * Cout + (2 - dz) * Cout + co) * KK + (KK - 1 - tap)];
        } else {
            v = mode == 0 ? w[((size_t)co * Cin + ci) * KK + tap] : w[((size_t)ci * Cout + co) * KK + (mode == 1 ? KK - 1 - tap : tap)];
        }
    }
    wp[e] = v;
}

// ------------------------------------------------------------------------------------------------------------ forward
template <int KS, int S, int NB, int CB>
struct FwdTile {
    static constexpr int KK = KS * KS, P = KS / 2;
    static constexpr int IN_H = (kMfRowsFwd - 1) * S + KS, IN_W = (kMfTileW - 1) * S + KS, PLANE = IN_H * IN_W;
    // the two half-waves of an MFMA operand read consecutive channel planes: 32 banks apart (stride 1) / odd banks (stride 2)
    static constexpr int PS = S == 1 ? round_to_residue(PLANE, 64, 32) : (PLANE | 1);
    static constexpr int COW = NB * 32;
    static constexpr int WPS = round_to_residue(KK * COW, 64, 32);
    // Two LDS buffers (the next chunk is written while this one feeds the matrix cores, one barrier per chunk) where two
    // workgroups still fit a CU with them.  Measured without (tools/probes/variant.sh, MVSTER_PROBE_MF_*): kernel time =
    // MFMA time (at 99 % of the pipe's rate) + the time of everything else -- the co-resident workgroups run in lockstep,
    // so nothing fills the pipe while a workgroup stages.
#ifndef MVSTER_MF_DOUBLE_BUFFER
#define MVSTER_MF_DOUBLE_BUFFER 1
#endif
    static constexpr bool DB = MVSTER_MF_DOUBLE_BUFFER && 2 * CB * (PS + WPS) * 4 <= 80 * 1024;
    static constexpr int NBUF = DB ? 2 : 1;
};

template <int KS, int S, int NB, int CB, bool EP>
__global__ void __launch_bounds__(kBlock)
#if MVSTER_MF_FWD_WAVES > 0
__attribute__((amdgpu_waves_per_eu(MVSTER_MF_FWD_WAVES)))
#endif
conv_mfma_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                     float* __restrict__ y, MfDims d) {
    using T = FwdTile<KS, S, NB, CB>;
    constexpr int KK = T::KK, P = T::P, IN_W = T::IN_W, PLANE = T::PLANE, PS = T::PS, COW = T::COW, WPS = T::WPS;
    constexpr int PPW = CB / 4;                                    // input planes of a chunk per wave
    constexpr int NI = (PLANE + kWave - 1) / kWave;                // elements of a plane per lane
    constexpr int ROW4 = COW / 4, W4 = CB * KK * ROW4;             // packed weights of a chunk as float4s
    constexpr int NW = (W4 + kBlock - 1) / kBlock;
    static_assert(CB % 4 == 0, "planes are dealt to the four waves");
    constexpr bool DB = T::DB;
    __shared__ float il_all[T::NBUF * CB * PS];
    __shared__ __attribute__((aligned(16))) float wl_all[T::NBUF * CB * WPS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    int t = blockIdx.x;
    const int tx = t % d.tiles_x;
    t /= d.tiles_x;
    const int ty = t % d.tiles_y, n = t / d.tiles_y;
    const int co0 = blockIdx.y * COW;
    const int gy0 = ty * kMfRowsFwd * S - P, gx0 = tx * kMfTileW * S - P;
    const int HW = d.H * d.W;

    // this lane's elements of an input plane: the same window positions for every plane of every chunk
    int goff[NI];                                                  // offset inside the plane, -1: outside the image / the window
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int idx = lane + kWave * k;
        const int r = idx / IN_W, col = idx - r * IN_W;
        const int gy = gy0 + r, gx = gx0 + col;
        goff[k] = idx < PLANE && (unsigned)gy < (unsigned)d.H && (unsigned)gx < (unsigned)d.W ? gy * d.W + gx : -1;
    }

    f32x16 acc[2][NB];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][nb][i] = 0.0f;

    float vin[PPW][NI];
    float4 vw[NW];
    auto fetch = [&](int ci0) {                                    // global -> registers: in flight while the MFMAs of the previous chunk run
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp) {
            const float* __restrict__ plane = input_plane(x, n, ci0 + wave * PPW + pp, d.Cin, d.depth, d.Creal, HW);
#pragma unroll
            for (int k = 0; k < NI; ++k) vin[pp][k] = plane != nullptr && goff[k] >= 0 ? plane[goff[k]] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int q = tid + kBlock * k;
            const int row = q / ROW4, within = q - row * ROW4;     // row = c * KK + tap
            vw[k] = q < W4 ? *reinterpret_cast<const float4*>(wp + ((size_t)ci0 * KK + row) * d.CoutPad + co0 + 4 * within)
                           : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    };
    auto stage = [&](int buf) {                                    // registers -> LDS
        float* __restrict__ il = il_all + buf * (CB * PS);
        float* __restrict__ wl = wl_all + buf * (CB * WPS);
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp)
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int idx = lane + kWave * k;
                if (idx < PLANE) il[(wave * PPW + pp) * PS + idx] = vin[pp][k];
            }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int q = tid + kBlock * k;
            const int row = q / ROW4, within = q - row * ROW4;
            if (q < W4) *reinterpret_cast<float4*>(&wl[(row / KK) * WPS + (row % KK) * COW + 4 * within]) = vw[k];
        }
    };

    // One iteration = one chunk of CB input channels.  Double-buffered: chunk k feeds the matrix cores from buffer k & 1
    // while chunk k+1 (fetched during chunk k-1) is written to the other buffer after the first MFMAs have been issued,
    // and chunk k+2 leaves for the registers -- one barrier per chunk, the pipe never waits for a staging phase.
    constexpr int UNITS = KK * (CB / 2);                           // (tap, channel pair) steps of a chunk
    constexpr int SPLIT = UNITS >= 8 ? 4 : UNITS / 2;              // the next chunk is staged after this many of them
    fetch(0);
    if constexpr (DB) {
        stage(0);
        __syncthreads();
        if (CB < d.CinPad) fetch(CB);
    }
    int buf = 0;
    for (int ci0 = 0; ci0 < d.CinPad; ci0 += CB) {
        if constexpr (!DB) {
            if (ci0 > 0) __syncthreads();                          // the previous chunk has been consumed
            stage(0);
            __syncthreads();
            if (ci0 + CB < d.CinPad) fetch(ci0 + CB);
        }
        const float* __restrict__ il = il_all + buf * (CB * PS);
        const float* __restrict__ wl = wl_all + buf * (CB * WPS);
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int tap = u / (CB / 2), c = 2 * (u % (CB / 2));
            const int ky = tap / KS, kx = tap - ky * KS;
            if constexpr (DB) {
                if (u == SPLIT && ci0 + CB < d.CinPad) {
                    stage(buf ^ 1);
                    if (ci0 + 2 * CB < d.CinPad) fetch(ci0 + 2 * CB);
                }
            }
            {
                float a[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) a[nb] = wl[(c + h) * WPS + tap * COW + nb * 32 + p];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const float b = il[(c + h) * PS + ((2 * wave + r) * S + ky) * IN_W + p * S + kx];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
#if defined(MVSTER_PROBE_MF_NO_MFMA)         // timing probes (tools/probes/variant.sh); results are wrong when defined
                        acc[r][nb][0] += a[nb] * b;
#elif defined(MVSTER_PROBE_MF_NO_LDS_READS)
                        acc[r][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)tap, (float)c, acc[r][nb], 0, 0, 0);
#else
                        acc[r][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb], b, acc[r][nb], 0, 0, 0);
#endif
                    }
                }
            }
        }
        if constexpr (DB) {
            __syncthreads();                                       // buffer `buf` is free, the other one is complete
            buf ^= 1;
        }
    }
    // D: column (pixel) = lane & 31, row (output channel) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5): registers
    // 4j .. 4j+3 hold four consecutive channels.  EP: the layer's BatchNorm / ReLU / skip connection on the way out.
    const int ox = tx * kMfTileW + p;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = co0 + nb * 32 + 8 * j + 4 * h;
            float add[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (bias != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) add[e] = co + e < d.Cout ? bias[co + e] : 0.0f;
            }
            float4 ka = {}, kb = {}, km = {};
            if constexpr (EP) {                                    // (padded to a multiple of 32 channels: always in bounds)
                ka = *reinterpret_cast<const float4*>(d.ep.norm + co);
                kb = *reinterpret_cast<const float4*>(d.ep.norm + d.ep.Cpad + co);
                km = *reinterpret_cast<const float4*>(d.ep.norm + 2 * d.ep.Cpad + co);
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int oy = ty * kMfRowsFwd + 2 * wave + r;
                if (oy >= d.Ho || ox >= d.Wo) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[r][nb][4 * j + e] + add[e];
                if constexpr (EP) {
                    v[0] = normalise(NormCoef{ka.x, kb.x, km.x}, v[0], d.ep.relu != 0);
                    v[1] = normalise(NormCoef{ka.y, kb.y, km.y}, v[1], d.ep.relu != 0);
                    v[2] = normalise(NormCoef{ka.z, kb.z, km.z}, v[2], d.ep.relu != 0);
                    v[3] = normalise(NormCoef{ka.w, kb.w, km.w}, v[3], d.ep.relu != 0);
                }
                if (d.out_nhwc) {                                  // channels-last (Cout % 4 == 0, checked by the host)
                    if (co >= d.Cout) continue;
                    const size_t idx = (((size_t)n * d.Ho + oy) * d.Wo + ox) * d.Cout + co;
                    float4 o = make_float4(v[0], v[1], v[2], v[3]);
                    if constexpr (EP) {
                        if (d.ep.residual != nullptr) {
                            const float4 rs = *reinterpret_cast<const float4*>(d.ep.residual + idx);
                            o.x += rs.x; o.y += rs.y; o.z += rs.z; o.w += rs.w;
                        }
                    }
                    *reinterpret_cast<float4*>(y + idx) = o;
                    continue;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (co + e >= d.Cout) continue;
                    const size_t idx = (((size_t)n * d.Cout + co + e) * d.Ho + oy) * d.Wo + ox;
                    float o = v[e];
                    if constexpr (EP) {
                        if (d.ep.residual != nullptr) o += d.ep.residual[idx];
                    }
                    y[idx] = o;
                }
            }
        }
}

// ------------------------------------------------------------------------------------- transposed convolution, stride 2
// y[co][oy][ox] = sum over ci and the taps with (oy + P - ky), (ox + P - kx) even of w[ci][co][ky][kx] * x[ci][(oy+P-ky)/2][(ox+P-kx)/2]
// -- reg2d's ConvTranspose3d (1,3,3) up-convolutions (models/mvs4net_utils.py:901-912) and the input gradient of every
// stride-2 layer.  As a gather: an output pixel of row parity r and column parity q receives only the taps of matching
// parity, and along a row the pixels of one column parity read CONSECUTIVE input columns -- so one MFMA column block is 32
// output pixels of one parity (a 64-wide output tile holds both), and no multiplication by an inserted zero is ever issued.
// Workgroup = 8 output rows x 64 output columns; wave w owns rows 2w (even), 2w+1 (odd) x both column parities.
constexpr int kMfTrRows = 8, kMfTrInH = kMfTrRows / 2 + 2, kMfTrInW = kMfTileW + 2;

template <int KS, int NB, int CB>
struct TrTile {
    static constexpr int KK = KS * KS, P = KS / 2;
    static constexpr int PLANE = kMfTrInH * kMfTrInW;              // input rows oy0/2 - 1 .. oy0/2 + 4, columns ox0/2 - 1 .. ox0/2 + 32
    static constexpr int PS = round_to_residue(PLANE, 64, 32);
    static constexpr int COW = NB * 32;
    static constexpr int WPS = round_to_residue(KK * COW, 64, 32);
};

template <int KS, int NB, int CB, bool EP>
__global__ void __launch_bounds__(kBlock)
conv_mfma_transposed_kernel(const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ y, MfDims d) {
    using T = TrTile<KS, NB, CB>;
    constexpr int KK = T::KK, P = T::P, IN_W = kMfTrInW, PLANE = T::PLANE, PS = T::PS, COW = T::COW, WPS = T::WPS;
    constexpr int PPW = CB / 4, NI = (PLANE + kWave - 1) / kWave;
    constexpr int ROW4 = COW / 4, W4 = CB * KK * ROW4, NW = (W4 + kBlock - 1) / kBlock;
    __shared__ float il[CB * PS];
    __shared__ __attribute__((aligned(16))) float wl[CB * WPS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    int t = blockIdx.x;
    const int tx = t % d.tiles_x;
    t /= d.tiles_x;
    const int ty = t % d.tiles_y, n = t / d.tiles_y;
    const int co0 = blockIdx.y * COW;
    const int oy0 = ty * kMfTrRows, ox0 = tx * 2 * kMfTileW;       // even
    const int gy0 = oy0 / 2 - 1, gx0 = ox0 / 2 - 1;
    const int HW = d.H * d.W;
    const float* __restrict__ xn = x + (size_t)n * d.Cin * HW;

    int goff[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int idx = lane + kWave * k;
        const int r = idx / IN_W, col = idx - r * IN_W;
        const int gy = gy0 + r, gx = gx0 + col;
        goff[k] = idx < PLANE && (unsigned)gy < (unsigned)d.H && (unsigned)gx < (unsigned)d.W ? gy * d.W + gx : -1;
    }
    f32x16 acc[2][2][NB];                                          // [row parity][column parity][co block]
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[r][q][nb][i] = 0.0f;

    float vin[PPW][NI];
    float4 vw[NW];
    auto fetch = [&](int ci0) {
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp) {
            const int ci = ci0 + wave * PPW + pp;
            const float* __restrict__ plane = xn + (size_t)ci * HW;
#pragma unroll
            for (int k = 0; k < NI; ++k) vin[pp][k] = ci < d.Cin && goff[k] >= 0 ? plane[goff[k]] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int q = tid + kBlock * k;
            const int row = q / ROW4, within = q - row * ROW4;
            vw[k] = q < W4 ? *reinterpret_cast<const float4*>(wp + ((size_t)ci0 * KK + row) * d.CoutPad + co0 + 4 * within)
                           : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp)
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int idx = lane + kWave * k;
                if (idx < PLANE) il[(wave * PPW + pp) * PS + idx] = vin[pp][k];
            }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int q = tid + kBlock * k;
            const int row = q / ROW4, within = q - row * ROW4;
            if (q < W4) *reinterpret_cast<float4*>(&wl[(row / KK) * WPS + (row % KK) * COW + 4 * within]) = vw[k];
        }
    };

    fetch(0);
    for (int ci0 = 0; ci0 < d.CinPad; ci0 += CB) {
        if (ci0 > 0) __syncthreads();
        stage();
        __syncthreads();
        if (ci0 + CB < d.CinPad) fetch(ci0 + CB);
#pragma unroll
        for (int c = 0; c < CB; c += 2) {
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int r = (ky + P) & 1;                        // the row parity this tap row feeds (oy0 is even)
                const int lrow = wave + (r + P - ky) / 2 + 1;      // input row of output row 2*wave + r, inside the tile
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const int q = (kx + P) & 1;
                    const int lcol = (q + P - kx) / 2 + 1;
                    const float b = il[(c + h) * PS + lrow * IN_W + p + lcol];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const float a = wl[(c + h) * WPS + (ky * KS + kx) * COW + nb * 32 + p];
                        acc[r][q][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r][q][nb], 0, 0, 0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = co0 + nb * 32 + 8 * j + 4 * h;          // registers 4j .. 4j+3: four consecutive channels
            float4 ka = {}, kb = {}, km = {};
            if constexpr (EP) {
                ka = *reinterpret_cast<const float4*>(d.ep.norm + co);
                kb = *reinterpret_cast<const float4*>(d.ep.norm + d.ep.Cpad + co);
                km = *reinterpret_cast<const float4*>(d.ep.norm + 2 * d.ep.Cpad + co);
            }
            const float fa[4] = {ka.x, ka.y, ka.z, ka.w}, fb[4] = {kb.x, kb.y, kb.z, kb.w}, fm[4] = {km.x, km.y, km.z, km.w};
            // a lane holds the horizontally adjacent outputs 2p and 2p + 1 (q = 0, 1): one 8-byte store (and skip-connection
            // load) per pair where the row is even-sized, instead of two half-used passes over every cache line
            const bool pairs = (d.Wo & 1) == 0;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int oy = oy0 + 2 * wave + r;
                const int ox = ox0 + 2 * p;
                if (oy >= d.Ho || ox >= d.Wo) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (co + e >= d.Cout) continue;
                    const size_t idx = (((size_t)n * d.Cout + co + e) * d.Ho + oy) * d.Wo + ox;
                    float v0 = acc[r][0][nb][4 * j + e], v1 = acc[r][1][nb][4 * j + e];
                    if constexpr (EP) {
                        v0 = normalise(NormCoef{fa[e], fb[e], fm[e]}, v0, d.ep.relu != 0);
                        v1 = normalise(NormCoef{fa[e], fb[e], fm[e]}, v1, d.ep.relu != 0);
                    }
                    if (pairs) {                                   // ox even, Wo even: idx is even, ox + 1 < Wo
                        if constexpr (EP) {
                            if (d.ep.residual != nullptr) {
                                const float2 rs = *reinterpret_cast<const float2*>(d.ep.residual + idx);
                                v0 += rs.x;
                                v1 += rs.y;
                            }
                        }
                        *reinterpret_cast<float2*>(y + idx) = make_float2(v0, v1);
                    } else {
                        if constexpr (EP) {
                            if (d.ep.residual != nullptr) {
                                v0 += d.ep.residual[idx];
                                if (ox + 1 < d.Wo) v1 += d.ep.residual[idx + 1];
                            }
                        }
                        y[idx] = v0;
                        if (ox + 1 < d.Wo) y[idx + 1] = v1;
                    }
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------- weight gradient
struct MfWgDims {
    int N, Cin, Cout, H, W, Ho, Wo;
    int tiles_x, tiles_y, ntiles;
    int ci_blocks;                    // blockIdx.y = co_block * ci_blocks + ci_block
    int depth, Creal;                 // depth > 0: virtual input channels, see input_plane()
};

template <int KS, int S, int CIB, int TH>
struct WgTile {
    static constexpr int P = KS / 2, RPW = TH / 4;                 // tile rows per wave
    static constexpr int IN_H = (TH - 1) * S + KS, IN_W = (kMfTileW - 1) * S + KS, PLANE = IN_H * IN_W;
    static constexpr int Q = 32 / CIB;                             // (kx, ci) columns of one MFMA span Q taps
    static constexpr int PSX = (((PLANE + Q - 1) / Q) | 1) * Q;    // (32/CIB) * odd: the 32 columns fall on 32 banks
    static constexpr int NBN = (KS * CIB + 31) / 32;               // column blocks per ky
    static constexpr int DPS = TH * kMfTileW + 1;                  // dy plane stride, odd
    static constexpr int XL = CIB * PSX + 64, DL = 32 * DPS;       // + slack: columns past the last tap read on (never stored)
    static constexpr int E = KS * NBN * 1024;                      // partial sums per workgroup, MFMA-native order
    static constexpr int LDS = (XL + DL) > E ? (XL + DL) : E;
};

template <int KS, int S, int CIB, int TH>
__global__ void __launch_bounds__(kBlock)
conv_mfma_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partials, MfWgDims d) {
    using T = WgTile<KS, S, CIB, TH>;
    constexpr int P = T::P, RPW = T::RPW, IN_W = T::IN_W, PLANE = T::PLANE, PSX = T::PSX, NBN = T::NBN, DPS = T::DPS, E = T::E;
    constexpr int PPW = CIB / 4;                                   // x planes per wave
    constexpr int NI = (PLANE + kWave - 1) / kWave;                // elements of an x plane per lane
    constexpr int DPL = TH * kMfTileW, ND = DPL / kWave;           // dy plane of the tile, elements per lane; 8 planes per wave
    static_assert(DPL % kWave == 0, "dy rows are dealt to whole waves");
    extern __shared__ float lds[];                                 // T::LDS floats: x tile + dy tile, later the partial sums
    float* __restrict__ xl = lds;
    float* __restrict__ dyl = lds + T::XL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    const int cob = blockIdx.y / d.ci_blocks, cib = blockIdx.y - cob * d.ci_blocks;
    const int co0 = cob * 32, ci0 = cib * CIB;
    const int HW = d.H * d.W, HWo = d.Ho * d.Wo;

    int boff[NBN];                                                 // lane's column (kx, ci) of block j inside the x tile
#pragma unroll
    for (int j = 0; j < NBN; ++j) {
        const int f = j * 32 + p;
        boff[j] = (f % CIB) * PSX + f / CIB;
    }
    f32x16 acc[KS][NBN];
#pragma unroll
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int j = 0; j < NBN; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ky][j][i] = 0.0f;

    float vx[PPW][NI], vd[8][ND];
    auto fetch = [&](int t) {                                      // global -> registers: in flight while the previous tile's MFMAs run
        const int tx = t % d.tiles_x;
        t /= d.tiles_x;
        const int ty = t % d.tiles_y, n = t / d.tiles_y;
        const int oy0 = ty * TH, ox0 = tx * kMfTileW;
        const int gy0 = oy0 * S - P, gx0 = ox0 * S - P;
        int goff[NI];
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int idx = lane + kWave * k;
            const int r = idx / IN_W, col = idx - r * IN_W;
            const int gy = gy0 + r, gx = gx0 + col;
            goff[k] = idx < PLANE && (unsigned)gy < (unsigned)d.H && (unsigned)gx < (unsigned)d.W ? gy * d.W + gx : -1;
        }
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp) {
            const float* __restrict__ plane = input_plane(x, n, ci0 + wave * PPW + pp, d.Cin, d.depth, d.Creal, HW);
#pragma unroll
            for (int k = 0; k < NI; ++k) vx[pp][k] = plane != nullptr && goff[k] >= 0 ? plane[goff[k]] : 0.0f;
        }
        int doff[ND];
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int idx = lane + kWave * k;
            const int r = idx / kMfTileW, col = idx - r * kMfTileW;
            doff[k] = oy0 + r < d.Ho && ox0 + col < d.Wo ? (oy0 + r) * d.Wo + ox0 + col : -1;
        }
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            const int co = co0 + wave * 8 + pp;
            const float* __restrict__ plane = dy + ((size_t)n * d.Cout + co) * HWo;
#pragma unroll
            for (int k = 0; k < ND; ++k) vd[pp][k] = co < d.Cout && doff[k] >= 0 ? plane[doff[k]] : 0.0f;
        }
    };
    auto stage = [&]() {                                           // registers -> LDS
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp)
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int idx = lane + kWave * k;
                if (idx < PLANE) xl[(wave * PPW + pp) * PSX + idx] = vx[pp][k];
            }
#pragma unroll
        for (int pp = 0; pp < 8; ++pp)
#pragma unroll
            for (int k = 0; k < ND; ++k) dyl[(wave * 8 + pp) * DPS + lane + kWave * k] = vd[pp][k];
    };

    if ((int)blockIdx.x < d.ntiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < d.ntiles; t += gridDim.x) {
        if (t != (int)blockIdx.x) __syncthreads();                 // the previous tile has been consumed
        stage();
        __syncthreads();
        if (t + (int)gridDim.x < d.ntiles) fetch(t + gridDim.x);
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave * RPW + rr;
#pragma unroll 4
            for (int ox = 0; ox < kMfTileW; ox += 2) {
                const float a = dyl[p * DPS + r * kMfTileW + ox + h];
#pragma unroll
                for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                    for (int j = 0; j < NBN; ++j) {
                        const float b = xl[boff[j] + (r * S + ky) * IN_W + (ox + h) * S];
                        acc[ky][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[ky][j], 0, 0, 0);
                    }
            }
        }
    }
    // fold the four waves' sums through LDS (wave after wave: plain reads and writes), then one coalesced store
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int j = 0; j < NBN; ++j)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float* __restrict__ slot = &lds[((ky * NBN + j) * 16 + i) * 64 + lane];
                        *slot = w == 0 ? acc[ky][j][i] : *slot + acc[ky][j][i];
                    }
        }
        __syncthreads();
    }
    float* __restrict__ out = partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * E;
    for (int e = tid; e < E; e += kBlock) out[e] = lds[e];
}

// Weight gradient of the 3x3 stride-1 layers with <= 8 output channels (FPN4 conv0.1, out4; reg2d conv0).  With the output
// channels on an MFMA axis, 8 of 32 rows would do work.  Here the roles are swapped and the three tap rows ride along:
//   D[(kx,ci) 32][(ky,co) 32] += A[(kx,ci)][pixel pair] * B[pixel pair][(ky,co)],   B = dy[co] shifted up by ky rows
// (dW[co][ci][ky][kx] = sum_p' dy[co][p' - ky rows] x[ci][p' + kx], p' running over the INPUT tile's pixels), so 24 of the 32
// columns are live and one pass over the input tile's rows feeds all nine taps.  Tile = 6 output rows (8 input rows: two per
// wave); the dy tile carries 3 zero rows above and 2 below for the shifted reads that fall outside it.
constexpr int kPkRows = 6, kPkDyRows = kPkRows + 5;

template <int CIB>
struct WgPackTile {
    static constexpr int IN_H = kPkRows + 2, IN_W = kMfTileW + 2, PLANE = IN_H * IN_W;
    static constexpr int Q = 32 / CIB;
    static constexpr int PSX = (((PLANE + Q - 1) / Q) | 1) * Q;
    static constexpr int NBN = (3 * CIB + 31) / 32;
    static constexpr int DPS = kPkDyRows * kMfTileW + 1;           // odd
    static constexpr int XL = CIB * PSX + 64, DL = 8 * DPS;
    static constexpr int E = NBN * 1024;
    static constexpr int LDS = (XL + DL) > E ? (XL + DL) : E;
};

template <int CIB>
__global__ void __launch_bounds__(kBlock)
conv_mfma_wgrad_packed_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partials, MfWgDims d) {
    using T = WgPackTile<CIB>;
    constexpr int IN_H = T::IN_H, IN_W = T::IN_W, PLANE = T::PLANE, PSX = T::PSX, NBN = T::NBN, DPS = T::DPS, E = T::E;
    constexpr int PPW = CIB / 4, NI = (PLANE + kWave - 1) / kWave;
    constexpr int DPL = kPkRows * kMfTileW, ND = DPL / kWave;      // dy plane of the tile: 192 pixels, 3 per lane; 2 planes per wave
    static_assert(IN_H % 4 == 0 && DPL % kWave == 0, "input rows and dy pixels are dealt evenly");
    extern __shared__ float lds[];
    float* __restrict__ xl = lds;
    float* __restrict__ dyl = lds + T::XL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    const int ci0 = blockIdx.y * CIB;
    const int HW = d.H * d.W;

    int aoff[NBN];                                                 // lane's row (kx, ci) of block j inside the x tile
#pragma unroll
    for (int j = 0; j < NBN; ++j) {
        const int f = j * 32 + p;
        aoff[j] = (f % CIB) * PSX + f / CIB;
    }
    const int boff = (p & 7) * DPS + (3 - (p >> 3)) * kMfTileW;    // lane's column (ky, co): dy plane co, shifted up by ky rows
    f32x16 acc[NBN];
#pragma unroll
    for (int j = 0; j < NBN; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.0f;
    for (int e = tid; e < T::DL; e += kBlock) dyl[e] = 0.0f;       // the padding rows stay zero for the whole kernel

    float vx[PPW][NI], vd[2][ND];
    auto fetch = [&](int t) {
        const int tx = t % d.tiles_x;
        t /= d.tiles_x;
        const int ty = t % d.tiles_y, n = t / d.tiles_y;
        const int oy0 = ty * kPkRows, ox0 = tx * kMfTileW;
        int goff[NI];
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int idx = lane + kWave * k;
            const int r = idx / IN_W, col = idx - r * IN_W;
            const int gy = oy0 - 1 + r, gx = ox0 - 1 + col;
            goff[k] = idx < PLANE && (unsigned)gy < (unsigned)d.H && (unsigned)gx < (unsigned)d.W ? gy * d.W + gx : -1;
        }
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp) {
            const int ci = ci0 + wave * PPW + pp;
            const float* __restrict__ plane = x + ((size_t)n * d.Cin + ci) * HW;
#pragma unroll
            for (int k = 0; k < NI; ++k) vx[pp][k] = ci < d.Cin && goff[k] >= 0 ? plane[goff[k]] : 0.0f;
        }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int co = wave * 2 + pp;
            const float* __restrict__ plane = dy + ((size_t)n * d.Cout + co) * HW;
#pragma unroll
            for (int k = 0; k < ND; ++k) {
                const int idx = lane + kWave * k;
                const int r = idx / kMfTileW, col = idx - r * kMfTileW;
                vd[pp][k] = co < d.Cout && oy0 + r < d.H && ox0 + col < d.W ? plane[(oy0 + r) * d.W + ox0 + col] : 0.0f;
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp)
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int idx = lane + kWave * k;
                if (idx < PLANE) xl[(wave * PPW + pp) * PSX + idx] = vx[pp][k];
            }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int k = 0; k < ND; ++k) dyl[(wave * 2 + pp) * DPS + 3 * kMfTileW + lane + kWave * k] = vd[pp][k];
    };

    if ((int)blockIdx.x < d.ntiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < d.ntiles; t += gridDim.x) {
        __syncthreads();                                           // the previous tile has been consumed (first: dyl is zeroed)
        stage();
        __syncthreads();
        if (t + (int)gridDim.x < d.ntiles) fetch(t + gridDim.x);
#pragma unroll
        for (int rr = 0; rr < IN_H / 4; ++rr) {
            const int r = wave * (IN_H / 4) + rr;                  // input-tile row; output row r - ky
#pragma unroll 4
            for (int ox = 0; ox < kMfTileW; ox += 2) {
                // input column of output pixel ox under tap kx is ox + kx: the tap offset sits in aoff
                const float b = dyl[boff + r * kMfTileW + ox + h];
#pragma unroll
                for (int j = 0; j < NBN; ++j) {
                    const float a = xl[aoff[j] + r * IN_W + ox + h];
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < NBN; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float* __restrict__ slot = &lds[(j * 16 + i) * 64 + lane];
                    *slot = w == 0 ? acc[j][i] : *slot + acc[j][i];
                }
        }
        __syncthreads();
    }
    float* __restrict__ out = partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * E;
    for (int e = tid; e < E; e += kBlock) out[e] = lds[e];
}

// Weight gradient of the 1x1 layers: dW[co][ci] = sum_p dy[co][p] x[ci][p] -- a plain GEMM whose K runs over the pixels.  The
// tap-row kernel above gives a 1x1 layer ONE MFMA per pair of LDS tiles, and every (co, ci) block group re-reads both tensors
// (FPN4's head contraction 64 -> 72 at half resolution: 6 groups, 3.5 GB of loads for 1.25 GB of tensors, 0.85 ms).  Here a
// workgroup holds up to 96 x 64 outputs (3 x 2 accumulators per wave) and reads its 4 x 32 pixels of both tensors once.
constexpr int kW1Rows = 4, kW1Px = kW1Rows * kMfTileW;             // pixels per tile: one row per wave
constexpr int kW1Ps = kW1Px + 1;                                   // plane stride, odd: the 32 channels of a lane group on 32 banks

template <int COB, int CIBK>
__global__ void __launch_bounds__(kBlock)
conv_mfma_wgrad_1x1_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partials, MfWgDims d) {
    constexpr int CO = COB * 32, CI = CIBK * 32, E = COB * CIBK * 1024;
    constexpr int PD = CO / 4, PX = CI / 4, NP = kW1Px / kWave;    // planes per wave, elements of a plane per lane
    extern __shared__ float lds[];                                 // max((CO + CI) * kW1Ps, E) floats
    float* __restrict__ dyl = lds;
    float* __restrict__ xl = lds + CO * kW1Ps;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, p = lane & 31;
    const int cog = blockIdx.y / d.ci_blocks, cig = blockIdx.y - cog * d.ci_blocks;
    const int co0 = cog * CO, ci0 = cig * CI;
    const int HW = d.H * d.W;
    f32x16 acc[COB][CIBK];
#pragma unroll
    for (int a = 0; a < COB; ++a)
#pragma unroll
        for (int b = 0; b < CIBK; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
    float vd[PD][NP], vx[PX][NP];
    auto fetch = [&](int t) {
        const int tx = t % d.tiles_x;
        t /= d.tiles_x;
        const int ty = t % d.tiles_y, n = t / d.tiles_y;
        int off[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int idx = lane + kWave * k;
            const int gy = ty * kW1Rows + idx / kMfTileW, gx = tx * kMfTileW + idx % kMfTileW;
            off[k] = gy < d.H && gx < d.W ? gy * d.W + gx : -1;
        }
#pragma unroll
        for (int pp = 0; pp < PD; ++pp) {
            const int co = co0 + wave * PD + pp;
            const float* __restrict__ plane = dy + ((size_t)n * d.Cout + co) * HW;
#pragma unroll
            for (int k = 0; k < NP; ++k) vd[pp][k] = co < d.Cout && off[k] >= 0 ? plane[off[k]] : 0.0f;
        }
#pragma unroll
        for (int pp = 0; pp < PX; ++pp) {
            const int ci = ci0 + wave * PX + pp;
            const float* __restrict__ plane = x + ((size_t)n * d.Cin + ci) * HW;
#pragma unroll
            for (int k = 0; k < NP; ++k) vx[pp][k] = ci < d.Cin && off[k] >= 0 ? plane[off[k]] : 0.0f;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int pp = 0; pp < PD; ++pp)
#pragma unroll
            for (int k = 0; k < NP; ++k) dyl[(wave * PD + pp) * kW1Ps + lane + kWave * k] = vd[pp][k];
#pragma unroll
        for (int pp = 0; pp < PX; ++pp)
#pragma unroll
            for (int k = 0; k < NP; ++k) xl[(wave * PX + pp) * kW1Ps + lane + kWave * k] = vx[pp][k];
    };
    if ((int)blockIdx.x < d.ntiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < d.ntiles; t += gridDim.x) {
        if (t != (int)blockIdx.x) __syncthreads();
        stage();
        __syncthreads();
        if (t + (int)gridDim.x < d.ntiles) fetch(t + gridDim.x);
#pragma unroll 4
        for (int ox = 0; ox < kMfTileW; ox += 2) {
            const int q = wave * kMfTileW + ox + h;                // this wave's row of the tile, pixel pair (ox, ox + 1)
            float a[COB], b[CIBK];
#pragma unroll
            for (int i = 0; i < COB; ++i) a[i] = dyl[(i * 32 + p) * kW1Ps + q];
#pragma unroll
            for (int j = 0; j < CIBK; ++j) b[j] = xl[(j * 32 + p) * kW1Ps + q];
#pragma unroll
            for (int i = 0; i < COB; ++i)
#pragma unroll
                for (int j = 0; j < CIBK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < COB; ++i)
#pragma unroll
                for (int j = 0; j < CIBK; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float* __restrict__ slot = &lds[((i * CIBK + j) * 16 + r) * 64 + lane];
                        *slot = w == 0 ? acc[i][j][r] : *slot + acc[i][j][r];
                    }
        }
        __syncthreads();
    }
    float* __restrict__ out = partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * E;
    for (int e = tid; e < E; e += kBlock) out[e] = lds[e];
}

// finish for the 1x1 kernel: e = ((i * CIBK + j) * 16 + reg) * 64 + lane -> co = cog*CO + i*32 + row(reg, lane), ci = cig*CI + j*32 + (lane & 31)
__global__ void __launch_bounds__(kBlock)
conv_mfma_wgrad_1x1_finish_kernel(const float* __restrict__ partials, float* __restrict__ dweight, int nparts, int E, int COB,
                                  int CIBK, int Cin, int Cout, int ci_blocks) {
    __shared__ float red[kBlock];
    const int grp = blockIdx.y;
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
    float s = 0.0f;
    if (e < E) {
        const float* __restrict__ src = partials + (size_t)grp * nparts * E + e;
        float part[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        int q = slice;
        for (; q + 28 < nparts; q += 32)
#pragma unroll
            for (int u = 0; u < 8; ++u) part[u] += src[(size_t)(q + 4 * u) * E];
        for (; q < nparts; q += 4) part[0] += src[(size_t)q * E];
        s = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (slice != 0 || e >= E) return;
    s = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
    const int lane = e & 63, reg = (e >> 6) & 15, ij = e >> 10;
    const int i = ij / CIBK, j = ij - i * CIBK;
    const int cog = grp / ci_blocks, cig = grp - cog * ci_blocks;
    const int co = cog * COB * 32 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    const int ci = cig * CIBK * 32 + j * 32 + (lane & 31);
    if (co < Cout && ci < Cin) dweight[(size_t)co * Cin + ci] = s;
}

// dW[co][ci][ky][kx] = sum over the workgroups' partial sums; E elements per (co block, ci block) in MFMA-native order:
// e = ((ky * NBN + j) * 16 + reg) * 64 + lane.  A workgroup = 64 elements x 4 slices of the partials.
__global__ void __launch_bounds__(kBlock)
conv_mfma_wgrad_finish_kernel(const float* __restrict__ partials, float* __restrict__ dweight, int nparts, int E, int NBN, int KS,
                              int CIB, int Cin, int Cout, int ci_blocks, int packed) {
    __shared__ float red[kBlock];
    const int grp = blockIdx.y;
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
    float s = 0.0f;
    if (e < E) {
        // eight loads in flight per lane (one dependent load per iteration made this kernel ~25 us whatever the layer);
        // the order of the additions is fixed, so the result stays bitwise reproducible
        const float* __restrict__ src = partials + (size_t)grp * nparts * E + e;
        float part[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        int q = slice;
        for (; q + 28 < nparts; q += 32)
#pragma unroll
            for (int u = 0; u < 8; ++u) part[u] += src[(size_t)(q + 4 * u) * E];
        for (; q < nparts; q += 4) part[0] += src[(size_t)q * E];
        s = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (slice != 0 || e >= E) return;
    s = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
    const int lane = e & 63, reg = (e >> 6) & 15, kj = e >> 10;
    if (packed) {                                                  // rows = (kx, ci) of block kj, columns = (ky, co): see the packed kernel
        const int co = lane & 7, ky = (lane & 31) >> 3;
        const int f = kj * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const int kx = f / CIB, ci = grp * CIB + f % CIB;
        if (ky < KS && kx < KS && ci < Cin && co < Cout) dweight[(((size_t)co * Cin + ci) * KS + ky) * KS + kx] = s;
        return;
    }
    const int j = kj % NBN, ky = kj / NBN;
    const int cob = grp / ci_blocks, cib = grp - cob * ci_blocks;
    const int co = cob * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    const int f = j * 32 + (lane & 31);
    const int kx = f / CIB, ci = cib * CIB + f % CIB;
    if (kx < KS && ci < Cin && co < Cout) dweight[(((size_t)co * Cin + ci) * KS + ky) * KS + kx] = s;
}

static bool mfma_conv_shape_ok(int K, int stride) {
    return (stride == 1 && (K == 1 || K == 3)) || (stride == 2 && (K == 3 || K == 5));
}

static int wgrad_cib(int Cin, int K) {
    const int cib = Cin <= 8 ? 8 : Cin <= 16 ? 16 : 32;
    return K == 5 && cib == 32 ? 16 : cib;                         // 5 x 5 columns of 32 channels: 400 accumulator registers
}

// persistent workgroups over all (co, ci) block groups together.  The 32-channel-block kernels hold > 256 registers: one
// workgroup per CU is resident, a second round would only add partial sums and a tail (32->32: 0.18 -> 0.16 ms with 256);
// the others fit twice (8->8: 0.43 ms with 512, 0.50 with 256)
constexpr int kWgGridOnePerCu = 256, kWgGridTwoPerCu = 512;

struct WgPlan {
    int cib, th, ci_blocks, co_blocks, groups, nbn, E, grid_x, tiles_x, tiles_y, ntiles;
    bool packed;                      // <= 8 output channels, 3x3 stride 1: conv_mfma_wgrad_packed_kernel
    int cob1, cibk1;                  // K = 1: 32-channel blocks per workgroup of conv_mfma_wgrad_1x1_kernel (0: not that kernel)
};

static WgPlan wgrad_plan(int N, int Cin, int Cout, int Ho, int Wo, int K, int stride) {
    WgPlan pl;
    pl.cob1 = pl.cibk1 = 0;
    if (K == 1 && stride == 1) {                                   // whole-GEMM kernel: up to 96 x 64 outputs per workgroup
        const int cob_all = (Cout + 31) / 32, cib_all = (Cin + 31) / 32;
        pl.cob1 = cob_all >= 3 ? 3 : cob_all;
        pl.cibk1 = cib_all >= 2 ? 2 : 1;
        pl.cib = 32; pl.packed = false; pl.th = kW1Rows; pl.nbn = 1;
        pl.co_blocks = (cob_all + pl.cob1 - 1) / pl.cob1;
        pl.ci_blocks = (cib_all + pl.cibk1 - 1) / pl.cibk1;
        pl.groups = pl.co_blocks * pl.ci_blocks;
        pl.E = pl.cob1 * pl.cibk1 * 1024;
        pl.tiles_x = (Wo + kMfTileW - 1) / kMfTileW;
        pl.tiles_y = (Ho + kW1Rows - 1) / kW1Rows;
        pl.ntiles = N * pl.tiles_x * pl.tiles_y;
        int g1 = kWgGridOnePerCu / pl.groups;
        if (g1 < 1) g1 = 1;
        pl.grid_x = pl.ntiles < g1 ? pl.ntiles : g1;
        return pl;
    }
    pl.cib = wgrad_cib(Cin, K);
    pl.packed = K == 3 && stride == 1 && Cout <= 8;
    pl.th = pl.packed ? kPkRows : stride == 1 ? 8 : 4;
    pl.ci_blocks = (Cin + pl.cib - 1) / pl.cib;
    pl.co_blocks = (Cout + 31) / 32;
    pl.groups = pl.ci_blocks * pl.co_blocks;
    pl.nbn = (K * pl.cib + 31) / 32;
    pl.E = (pl.packed ? 1 : K) * pl.nbn * 1024;
    pl.tiles_x = (Wo + kMfTileW - 1) / kMfTileW;
    pl.tiles_y = (Ho + pl.th - 1) / pl.th;
    pl.ntiles = N * pl.tiles_x * pl.tiles_y;
    int gx = (pl.cib == 32 && !pl.packed ? kWgGridOnePerCu : kWgGridTwoPerCu) / pl.groups;
    if (gx < 1) gx = 1;
    pl.grid_x = pl.ntiles < gx ? pl.ntiles : gx;
    return pl;
}

}  // namespace mvster

extern "C" int mvster_conv_mfma_supported(int Cin, int Cout, int K, int stride) {
    using namespace mvster;
    return Cin > 0 && Cout > 0 && Cin <= 1024 && Cout <= 1024 && mfma_conv_shape_ok(K, stride) ? 1 : 0;
}

extern "C" long long mvster_conv_mfma_workspace(int N, int Cin, int Cout, int H, int W, int K, int stride, int op) {
    using namespace mvster;
    if (N <= 0 || H <= 0 || W <= 0 || !mvster_conv_mfma_supported(Cin, Cout, K, stride)) return 0;
    if (op == MVSTER_CONV_OP_WGRAD) {
        const int P = K / 2, Ho = (H + 2 * P - K) / stride + 1, Wo = (W + 2 * P - K) / stride + 1;
        const WgPlan pl = wgrad_plan(N, Cin, Cout, Ho, Wo, K, stride);
        return (long long)pl.groups * pl.grid_x * pl.E;
    }
    // packed weights [CinPad][K*K][CoutPad]: the forward kernels stage 8 (stride 1) / 4 (stride 2) input channels per step,
    // the transposed kernel 8 (K = 3) / 4 (K = 5)
    const int cb = op == MVSTER_CONV_OP_TRANSPOSED ? (K == 3 ? 8 : 4) : (stride == 1 ? MVSTER_MF_CB_S1 : 4);
    return (long long)((Cin + cb - 1) / cb * cb) * K * K * ((Cout + 31) / 32 * 32);
}

#define MVSTER_MFMA_FWD(KS_, S_, NB_, CB_)                                                                                    \
    do {                                                                                                                         \
        if (d.ep.norm != nullptr)                                                                                             \
            hipLaunchKernelGGL((conv_mfma_fwd_kernel<KS_, S_, NB_, CB_, true>), grid, dim3(kBlock), 0, st, x, wp, bias, y, d);  \
        else                                                                                                                  \
            hipLaunchKernelGGL((conv_mfma_fwd_kernel<KS_, S_, NB_, CB_, false>), grid, dim3(kBlock), 0, st, x, wp, bias, y, d); \
    } while (0)

namespace mvster {
// 64 or 32 output channels per workgroup?  The kernels are bound by the matrix pipe, and a launch is a few "rounds" of
// resident workgroups: what counts is how many MFMA-carrying waves a SIMD has to work through, round by round.  With 64
// channels a workgroup stages its input tile once for twice the work (2 workgroups fit a CU); with 32 there are twice as
// many workgroups of half the length (3 fit): better whenever the grid does not fill the chip several times over -- the
// low-resolution layers of reg2d / FPN4 (tools/probes/conv_calls.py: 64 -> 64 at 18 x 25 pixels is 24 workgroups of 92 us).
#ifndef MVSTER_MF_PICK_BLOCKS
#define MVSTER_MF_PICK_BLOCKS 1
#endif
static bool two_blocks_pay(long long tiles, int CoutPad) {
    if (CoutPad % 64 != 0) return false;
    if (!MVSTER_MF_PICK_BLOCKS) return true;
    auto simd_work = [](long long wgs, int per_cu, double work) {       // sum over rounds of (waves per SIMD x work per wave)
        const long long slots = 256LL * per_cu, full = wgs / slots, rest = wgs % slots;
        return work * ((double)full * per_cu + (double)((rest + 255) / 256));
    };
    const long long wg2 = tiles * (CoutPad / 64), wg1 = tiles * (CoutPad / 32);
    // (1.3: with 32 channels per workgroup the input tile is staged twice and every MFMA needs 1.5 instead of 1 LDS reads --
    // measured, a 32-channel workgroup takes 0.65 of a 64-channel one, not 0.5: 64 -> 64 at 18 x 25 and 36 x 50 pixels
    // 133 -> 87 us, while the 630-tile layers (64 -> 64 at 144 x 200 x 5 views) gain nothing and stay on 64)
    return simd_work(wg2, 2, 2.0) <= 1.3 * simd_work(wg1, 3, 1.0);
}

struct EpilogueArgs {         // host side of ConvEpilogue
    const float* norm;
    const float* residual;
    int relu;
    int prepacked;            // `weight` is the array mvster_conv_mfma_pack wrote: no pack launch, no workspace
};
static ConvEpilogue make_epilogue(const EpilogueArgs* e, int Cout) {
    return e != nullptr ? ConvEpilogue{e->norm, e->residual, e->relu ? 1 : 0, conv_norm_pad(Cout)}
                        : ConvEpilogue{nullptr, nullptr, 0, conv_norm_pad(Cout)};
}
}  // namespace mvster

// pack_mode: see conv_mfma_pack_kernel; depth > 0: Cin = 3 * Creal virtual input channels over depth-shifted samples
static int conv_mfma_fwd_impl(const float* x, const float* weight, const float* bias, float* y, float* workspace,
                              long long workspace_floats, int N, int Cin, int Cout, int H, int W, int K, int stride,
                              int pack_mode, int depth, int Creal, int out_nhwc, mvster_stream_t stream,
                              const mvster::EpilogueArgs* ep = nullptr) {
    using namespace mvster;
    const bool prepacked = ep != nullptr && ep->prepacked;
    MVSTER_REQUIRE_PTR(x); MVSTER_REQUIRE_PTR(weight); MVSTER_REQUIRE_PTR(y);
    if (!prepacked) MVSTER_REQUIRE_PTR(workspace);
    if (out_nhwc && Cout % 4 != 0) return MVSTER_ERR_UNSUPPORTED;  // 16-byte channel groups
    MVSTER_REQUIRE_SHAPE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && (long long)H * W <= kMaxPlane);
    if (!mvster_conv_mfma_supported(Cin, Cout, K, stride)) return MVSTER_ERR_UNSUPPORTED;
    MVSTER_REQUIRE_SHAPE(pack_mode == 0 || stride == 1);
    MVSTER_REQUIRE_SHAPE(prepacked || workspace_floats >= mvster_conv_mfma_workspace(N, Cin, Cout, H, W, K, stride, MVSTER_CONV_OP_FWD));
    const int P = K / 2, cb = stride == 1 ? MVSTER_MF_CB_S1 : 4;
    MfDims d;
    d.N = N; d.Cin = Cin; d.Cout = Cout; d.H = H; d.W = W; d.depth = depth; d.Creal = Creal; d.out_nhwc = out_nhwc ? 1 : 0;
    d.ep = make_epilogue(ep, Cout);
    d.Ho = (H + 2 * P - K) / stride + 1;
    d.Wo = (W + 2 * P - K) / stride + 1;
    d.CinPad = (Cin + cb - 1) / cb * cb;
    d.CoutPad = (Cout + 31) / 32 * 32;
    d.tiles_x = (d.Wo + kMfTileW - 1) / kMfTileW;
    d.tiles_y = (d.Ho + kMfRowsFwd - 1) / kMfRowsFwd;
    const long long tiles = (long long)N * d.tiles_x * d.tiles_y;
    MVSTER_REQUIRE_SHAPE(tiles < (1LL << 31));
    const hipStream_t st = as_stream(stream);
    const int packed = d.CinPad * K * K * d.CoutPad;
    const float* wp = prepacked ? weight : workspace;
    if (!prepacked)
        hipLaunchKernelGGL(conv_mfma_pack_kernel, dim3(grid_for(packed, kBlock)), dim3(kBlock), 0, st, weight, workspace, Cin, Cout,
                           K * K, d.CinPad, d.CoutPad, pack_mode);
    const bool two = two_blocks_pay(tiles, d.CoutPad);             // 64 output channels per workgroup, or 32
    if (K == 1 && d.CoutPad == 96) {                               // 1x1, 65-96 output channels (FPN4's head contraction 64 -> 72):
        const dim3 grid((unsigned)tiles, 1);                       // all three blocks from one pass over the input tile
        MVSTER_MFMA_FWD(1, 1, 3, MVSTER_MF_CB_S1);
        return launch_status();
    }
    const dim3 grid((unsigned)tiles, d.CoutPad / (two ? 64 : 32));
    if (K == 1) { if (two) MVSTER_MFMA_FWD(1, 1, 2, MVSTER_MF_CB_S1); else MVSTER_MFMA_FWD(1, 1, 1, MVSTER_MF_CB_S1); }
    else if (K == 3 && stride == 1) { if (two) MVSTER_MFMA_FWD(3, 1, 2, MVSTER_MF_CB_S1); else MVSTER_MFMA_FWD(3, 1, 1, MVSTER_MF_CB_S1); }
    else if (K == 3) { if (two) MVSTER_MFMA_FWD(3, 2, 2, 4); else MVSTER_MFMA_FWD(3, 2, 1, 4); }
    else { if (two) MVSTER_MFMA_FWD(5, 2, 2, 4); else MVSTER_MFMA_FWD(5, 2, 1, 4); }
    return launch_status();
}

extern "C" int mvster_conv_mfma_fwd(const float* x, const float* weight, const float* bias, float* y, float* workspace,
                                    long long workspace_floats, int N, int Cin, int Cout, int H, int W, int K, int stride,
                                    int flags, mvster_stream_t stream) {
    MVSTER_REQUIRE_SHAPE((flags & ~(MVSTER_CONV_FLAG_FLIP | MVSTER_CONV_FLAG_OUT_NHWC)) == 0);
    return conv_mfma_fwd_impl(x, weight, bias, y, workspace, workspace_floats, N, Cin, Cout, H, W, K, stride,
                              flags & MVSTER_CONV_FLAG_FLIP ? 1 : 0, 0, 0, flags & MVSTER_CONV_FLAG_OUT_NHWC, stream);
}

extern "C" int mvster_conv_mfma_depth3_fwd(const float* x, const float* weight, const float* bias, float* y, float* workspace,
                                           long long workspace_floats, int N, int C, int Cout, int H, int W, int K, int depth,
                                           int input_gradient, mvster_stream_t stream) {
    MVSTER_REQUIRE_SHAPE(depth > 0 && N > 0 && N % depth == 0 && C > 0 && Cout > 0 && (K == 1 || K == 3));
    // forward: 3*C virtual channels of x -> Cout.  Input gradient: x := dy [N,Cout,..], 3*Cout virtual channels -> C.
    if (input_gradient)
        return conv_mfma_fwd_impl(x, weight, nullptr, y, workspace, workspace_floats, N, 3 * Cout, C, H, W, K, 1, 3, depth, Cout, 0, stream);
    return conv_mfma_fwd_impl(x, weight, bias, y, workspace, workspace_floats, N, 3 * C, Cout, H, W, K, 1, 0, depth, C, 0, stream);
}

#define MVSTER_MFMA_TR(KS_, NB_, CB_)                                                                                       \
    do {                                                                                                                       \
        if (d.ep.norm != nullptr)                                                                                           \
            hipLaunchKernelGGL((conv_mfma_transposed_kernel<KS_, NB_, CB_, true>), grid, dim3(kBlock), 0, st, x, wp, y, d);   \
        else                                                                                                                \
            hipLaunchKernelGGL((conv_mfma_transposed_kernel<KS_, NB_, CB_, false>), grid, dim3(kBlock), 0, st, x, wp, y, d);  \
    } while (0)

static int conv_mfma_transposed_impl(const float* x, const float* weight, float* y, float* workspace,
                                     long long workspace_floats, int N, int Cin, int Cout, int H, int W, int Ho, int Wo,
                                     int K, mvster_stream_t stream, const mvster::EpilogueArgs* ep) {
    using namespace mvster;
    const bool prepacked = ep != nullptr && ep->prepacked;
    MVSTER_REQUIRE_PTR(x); MVSTER_REQUIRE_PTR(weight); MVSTER_REQUIRE_PTR(y);
    if (!prepacked) MVSTER_REQUIRE_PTR(workspace);
    MVSTER_REQUIRE_SHAPE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && (long long)Ho * Wo <= kMaxPlane);
    if (!(K == 3 || K == 5) || Cin > 1024 || Cout > 1024) return MVSTER_ERR_UNSUPPORTED;
    const int P = K / 2;
    // the stride-2 convolution that maps [Ho,Wo] back onto [H,W]: (Ho + 2P - K) / 2 + 1 = H
    MVSTER_REQUIRE_SHAPE((Ho + 2 * P - K) / 2 + 1 == H && (Wo + 2 * P - K) / 2 + 1 == W);
    const int cb = K == 3 ? 8 : 4;
    MfDims d;
    d.N = N; d.Cin = Cin; d.Cout = Cout; d.H = H; d.W = W; d.Ho = Ho; d.Wo = Wo; d.depth = 0; d.Creal = 0; d.out_nhwc = 0;
    d.ep = make_epilogue(ep, Cout);
    d.CinPad = (Cin + cb - 1) / cb * cb;
    d.CoutPad = (Cout + 31) / 32 * 32;
    const long long packed = (long long)d.CinPad * K * K * d.CoutPad;
    MVSTER_REQUIRE_SHAPE(prepacked || workspace_floats >= packed);
    d.tiles_x = (Wo + 2 * kMfTileW - 1) / (2 * kMfTileW);
    d.tiles_y = (Ho + kMfTrRows - 1) / kMfTrRows;
    const long long tiles = (long long)N * d.tiles_x * d.tiles_y;
    MVSTER_REQUIRE_SHAPE(tiles < (1LL << 31));
    const hipStream_t st = as_stream(stream);
    const float* wp = prepacked ? weight : workspace;
    if (!prepacked)
        hipLaunchKernelGGL(conv_mfma_pack_kernel, dim3(grid_for(packed, kBlock)), dim3(kBlock), 0, st, weight, workspace, Cin, Cout,
                           K * K, d.CinPad, d.CoutPad, 2);
    const bool two = two_blocks_pay(tiles, d.CoutPad);
    const dim3 grid((unsigned)tiles, d.CoutPad / (two ? 64 : 32));
    if (K == 3) { if (two) MVSTER_MFMA_TR(3, 2, 8); else MVSTER_MFMA_TR(3, 1, 8); }
    else { if (two) MVSTER_MFMA_TR(5, 2, 4); else MVSTER_MFMA_TR(5, 1, 4); }
    return launch_status();
}

extern "C" int mvster_conv_mfma_transposed(const float* x, const float* weight, float* y, float* workspace,
                                           long long workspace_floats, int N, int Cin, int Cout, int H, int W, int Ho, int Wo,
                                           int K, mvster_stream_t stream) {
    return conv_mfma_transposed_impl(x, weight, y, workspace, workspace_floats, N, Cin, Cout, H, W, Ho, Wo, K, stream, nullptr);
}

// ---- inference entry points: BatchNorm (running statistics) + ReLU + skip connection in the epilogue, weights packed once ----
extern "C" long long mvster_conv_mfma_packed_floats(int Cin, int Cout, int K, int stride, int op) {
    return mvster_conv_mfma_workspace(1, Cin, Cout, 1, 1, K, stride, op == MVSTER_CONV_OP_TRANSPOSED ? op : MVSTER_CONV_OP_FWD);
}

extern "C" int mvster_conv_mfma_pack(const float* weight, float* packed, long long packed_floats, int Cin, int Cout, int K,
                                     int stride, int op, int mode, mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(weight); MVSTER_REQUIRE_PTR(packed);
    MVSTER_REQUIRE_SHAPE(Cin > 0 && Cout > 0 && mode >= 0 && mode <= 3 && (op == MVSTER_CONV_OP_FWD || op == MVSTER_CONV_OP_TRANSPOSED));
    if (op == MVSTER_CONV_OP_FWD ? !mvster_conv_mfma_supported(Cin, Cout, K, stride) : !(K == 3 || K == 5)) return MVSTER_ERR_UNSUPPORTED;
    const long long need = mvster_conv_mfma_packed_floats(Cin, Cout, K, stride, op);
    MVSTER_REQUIRE_SHAPE(packed_floats >= need && need < (1LL << 31));
    const int cb = op == MVSTER_CONV_OP_TRANSPOSED ? (K == 3 ? 8 : 4) : (stride == 1 ? MVSTER_MF_CB_S1 : 4);
    const int CinPad = (Cin + cb - 1) / cb * cb, CoutPad = (Cout + 31) / 32 * 32;
    hipLaunchKernelGGL(conv_mfma_pack_kernel, dim3(grid_for(need, kBlock)), dim3(kBlock), 0, as_stream(stream), weight, packed, Cin,
                       Cout, K * K, CinPad, CoutPad, mode);
    return launch_status();
}

extern "C" int mvster_conv_mfma_fwd_ep(const float* x, const float* packed, const float* bias, float* y, int N, int Cin, int Cout,
                                       int H, int W, int K, int stride, int flags, int depth, const float* norm,
                                       const float* residual, int relu, mvster_stream_t stream) {
    MVSTER_REQUIRE_SHAPE((flags & ~MVSTER_CONV_FLAG_OUT_NHWC) == 0 && depth >= 0);
    MVSTER_REQUIRE_PTR(norm);
    const mvster::EpilogueArgs ep{norm, residual, relu, 1};
    if (depth > 0) {          // a (3,k,k) layer over depth slices stacked into the batch: Cin = 3 * C virtual channels (depth3_fwd)
        MVSTER_REQUIRE_SHAPE(N % depth == 0 && Cin % 3 == 0 && (K == 1 || K == 3) && stride == 1);
        return conv_mfma_fwd_impl(x, packed, bias, y, nullptr, 0, N, Cin, Cout, H, W, K, 1, 0, depth, Cin / 3,
                                  flags & MVSTER_CONV_FLAG_OUT_NHWC, stream, &ep);
    }
    return conv_mfma_fwd_impl(x, packed, bias, y, nullptr, 0, N, Cin, Cout, H, W, K, stride, 0, 0, 0,
                              flags & MVSTER_CONV_FLAG_OUT_NHWC, stream, &ep);
}

extern "C" int mvster_conv_mfma_transposed_ep(const float* x, const float* packed, float* y, int N, int Cin, int Cout, int H,
                                              int W, int Ho, int Wo, int K, const float* norm, const float* residual, int relu,
                                              mvster_stream_t stream) {
    MVSTER_REQUIRE_PTR(norm);
    const mvster::EpilogueArgs ep{norm, residual, relu, 1};
    return conv_mfma_transposed_impl(x, packed, y, nullptr, 0, N, Cin, Cout, H, W, Ho, Wo, K, stream, &ep);
}

#define MVSTER_MFMA_WG(KS_, S_, CIB_, TH_)                                                                                   \
    {                                                                                                                        \
        constexpr int bytes = (int)sizeof(float) * WgTile<KS_, S_, CIB_, TH_>::LDS;                                          \
        static_assert(bytes <= 160 * 1024, "tile does not fit the LDS");                                                    \
        if (bytes > 64 * 1024 &&                                                                                             \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_wgrad_kernel<KS_, S_, CIB_, TH_>),                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {                          \
            (void)hipGetLastError();                                                                                         \
            return MVSTER_ERR_LAUNCH;                                                                                        \
        }                                                                                                                    \
        hipLaunchKernelGGL((conv_mfma_wgrad_kernel<KS_, S_, CIB_, TH_>), grid, dim3(kBlock), bytes, st, x, dy, workspace, d); \
    }
#define MVSTER_MFMA_WG_PACKED(CIB_)                                                                                      \
    {                                                                                                                    \
        constexpr int bytes = (int)sizeof(float) * WgPackTile<CIB_>::LDS;                                                \
        if (bytes > 64 * 1024 &&                                                                                         \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_wgrad_packed_kernel<CIB_>),                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {                      \
            (void)hipGetLastError();                                                                                     \
            return MVSTER_ERR_LAUNCH;                                                                                    \
        }                                                                                                                \
        hipLaunchKernelGGL((conv_mfma_wgrad_packed_kernel<CIB_>), grid, dim3(kBlock), bytes, st, x, dy, workspace, d);   \
    }
#define MVSTER_MFMA_WG_BY_CIB(KS_, S_, TH_)                 \
    switch (pl.cib) {                                       \
        case 8: MVSTER_MFMA_WG(KS_, S_, 8, TH_) break;      \
        case 16: MVSTER_MFMA_WG(KS_, S_, 16, TH_) break;    \
        default: MVSTER_MFMA_WG(KS_, S_, 32, TH_) break;    \
    }

static int conv_mfma_wgrad_impl(const float* x, const float* dy, float* dweight, float* workspace, long long workspace_floats,
                                int N, int Cin, int Cout, int H, int W, int K, int stride, int depth, int Creal,
                                mvster_stream_t stream) {
    using namespace mvster;
    MVSTER_REQUIRE_PTR(x); MVSTER_REQUIRE_PTR(dy); MVSTER_REQUIRE_PTR(dweight); MVSTER_REQUIRE_PTR(workspace);
    MVSTER_REQUIRE_SHAPE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && (long long)H * W <= kMaxPlane);
    if (!mvster_conv_mfma_supported(Cin, Cout, K, stride)) return MVSTER_ERR_UNSUPPORTED;
    MVSTER_REQUIRE_SHAPE(workspace_floats >= mvster_conv_mfma_workspace(N, Cin, Cout, H, W, K, stride, MVSTER_CONV_OP_WGRAD));
    const int P = K / 2;
    MfWgDims d;
    d.N = N; d.Cin = Cin; d.Cout = Cout; d.H = H; d.W = W; d.depth = depth; d.Creal = Creal;
    d.Ho = (H + 2 * P - K) / stride + 1;
    d.Wo = (W + 2 * P - K) / stride + 1;
    const WgPlan pl = wgrad_plan(N, Cin, Cout, d.Ho, d.Wo, K, stride);
    d.tiles_x = pl.tiles_x; d.tiles_y = pl.tiles_y; d.ntiles = pl.ntiles; d.ci_blocks = pl.ci_blocks;
    const hipStream_t st = as_stream(stream);
    const dim3 grid(pl.grid_x, pl.groups);
    if (pl.cob1 > 0) {
#define MVSTER_MFMA_WG_1X1(COB_, CIBK_)                                                                                    \
    {                                                                                                                        \
        constexpr int fl = (COB_ + CIBK_) * 32 * kW1Ps > COB_ * CIBK_ * 1024 ? (COB_ + CIBK_) * 32 * kW1Ps : COB_ * CIBK_ * 1024; \
        constexpr int bytes = (int)sizeof(float) * fl;                                                                      \
        if (bytes > 64 * 1024 &&                                                                                             \
            hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_wgrad_1x1_kernel<COB_, CIBK_>),                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {                          \
            (void)hipGetLastError();                                                                                         \
            return MVSTER_ERR_LAUNCH;                                                                                        \
        }                                                                                                                    \
        hipLaunchKernelGGL((conv_mfma_wgrad_1x1_kernel<COB_, CIBK_>), grid, dim3(kBlock), bytes, st, x, dy, workspace, d);   \
    }
        if (pl.cob1 == 3 && pl.cibk1 == 2) MVSTER_MFMA_WG_1X1(3, 2)
        else if (pl.cob1 == 3) MVSTER_MFMA_WG_1X1(3, 1)
        else if (pl.cob1 == 2 && pl.cibk1 == 2) MVSTER_MFMA_WG_1X1(2, 2)
        else if (pl.cob1 == 2) MVSTER_MFMA_WG_1X1(2, 1)
        else if (pl.cibk1 == 2) MVSTER_MFMA_WG_1X1(1, 2)
        else MVSTER_MFMA_WG_1X1(1, 1)
#undef MVSTER_MFMA_WG_1X1
        if (int rc = launch_status()) return rc;
        hipLaunchKernelGGL(conv_mfma_wgrad_1x1_finish_kernel, dim3(grid_for(pl.E, 64), pl.groups), dim3(kBlock), 0, st, workspace,
                           dweight, pl.grid_x, pl.E, pl.cob1, pl.cibk1, Cin, Cout, pl.ci_blocks);
        return launch_status();
    }
    if (pl.packed) {
        switch (pl.cib) {
            case 8: MVSTER_MFMA_WG_PACKED(8) break;
            case 16: MVSTER_MFMA_WG_PACKED(16) break;
            default: MVSTER_MFMA_WG_PACKED(32) break;
        }
    }
    else if (K == 3 && stride == 1) { MVSTER_MFMA_WG_BY_CIB(3, 1, 8) }
    else if (K == 3) { MVSTER_MFMA_WG_BY_CIB(3, 2, 4) }
    else {
        if (pl.cib == 8) MVSTER_MFMA_WG(5, 2, 8, 4) else MVSTER_MFMA_WG(5, 2, 16, 4)
    }
    if (int rc = launch_status()) return rc;
    hipLaunchKernelGGL(conv_mfma_wgrad_finish_kernel, dim3(grid_for(pl.E, 64), pl.groups), dim3(kBlock), 0, st, workspace, dweight,
                       pl.grid_x, pl.E, pl.nbn, K, pl.cib, Cin, Cout, pl.ci_blocks, pl.packed ? 1 : 0);
    return launch_status();
}

extern "C" int mvster_conv_mfma_wgrad(const float* x, const float* dy, float* dweight, float* workspace,
                                      long long workspace_floats, int N, int Cin, int Cout, int H, int W, int K, int stride,
                                      mvster_stream_t stream) {
    return conv_mfma_wgrad_impl(x, dy, dweight, workspace, workspace_floats, N, Cin, Cout, H, W, K, stride, 0, 0, stream);
}

extern "C" int mvster_conv_mfma_depth3_wgrad(const float* x, const float* dy, float* dweight, float* workspace,
                                             long long workspace_floats, int N, int C, int Cout, int H, int W, int K, int depth,
                                             mvster_stream_t stream) {
    MVSTER_REQUIRE_SHAPE(depth > 0 && N > 0 && N % depth == 0 && C > 0 && Cout > 0);
    if (K != 3 || Cout <= 8) return MVSTER_ERR_UNSUPPORTED;       // the 1x1 and packed-tap kernels read real planes only
    return conv_mfma_wgrad_impl(x, dy, dweight, workspace, workspace_floats, N, 3 * C, Cout, H, W, K, 1, depth, C, stream);
}
