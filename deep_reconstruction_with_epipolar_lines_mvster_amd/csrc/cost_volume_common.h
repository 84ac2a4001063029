// Shared declarations of the fused cost-volume kernels (cost_volume_fwd.hip / cost_volume_bwd.hip).
#pragma once

#include <math.h>

#include "mvster_device.h"
#include "mvster_host.h"

namespace mvster {

struct SrcPtrs {
    const float* p[MVSTER_MAX_SRC_VIEWS];
};
struct GradPtrs {
    float* p[MVSTER_MAX_SRC_VIEWS];
};

struct VolumeDims {
    int H, W, Hs, Ws, S, ntiles;
    float temp;     // attn_temp
    float sqrt_c;   // (float)sqrt(C), the reference divides the view weight by math.sqrt(C)
    float inv_temp; // 1 / attn_temp when attn_temp is a power of two (x / temp == x * inv_temp bit for bit), else 0
    float inv_sqrt_c; // (float)(1 / sqrt(C)): x / sqrt_c as div_const(x, sqrt_c, inv_sqrt_c) (exact product when C is a power of 4)
    float inv_tc;   // 1 / (attn_temp * sqrt_c): backward of the scaled softmax
    int grad_planar; // backward only: write gref / gsrcs as [B,C,H,W] planes (MVSTER_GRAD_PLANAR) although features are read [B,H,W,C]
};

// x / attn_temp as the reference computes it; a power-of-two temperature (the scripts' attn_temp 2) divides exactly by
// multiplication -- one instruction instead of the ~10 of an IEEE division, in kernels that are VALU-bound.
#ifdef __HIPCC__
__device__ __forceinline__ float div_temp(float x, const VolumeDims& dm) {
    return dm.inv_temp != 0.0f ? x * dm.inv_temp : div_fast(x, dm.temp);
}
// the view weight: softmax / sqrt(C)  (models/mvs4net_utils.py:1083)
__device__ __forceinline__ float div_sqrt_c(float x, const VolumeDims& dm) { return div_const(x, dm.sqrt_c, dm.inv_sqrt_c); }
#endif

inline VolumeDims make_dims(int H, int W, int Hs, int Ws, int S, int C, float attn_temp) {
    int e = 0;
    const float m = frexpf(fabsf(attn_temp), &e);
    const bool pow2 = m == 0.5f && e > -100 && e < 100;
    return VolumeDims{H, W, Hs, Ws, S, 0, attn_temp, (float)sqrt((double)C), pow2 ? 1.0f / attn_temp : 0.0f,
                      (float)(1.0 / sqrt((double)C)), 1.0f / (attn_temp * (float)sqrt((double)C)), 0};
}

#ifdef __HIPCC__
// ---- gather of one sample from a channels-last map (forward and coefficient kernels) ------------------------------------
// One sample's gather + group correlation for NCH channels held by this lane (channels ch0 .. ch0+NCH-1 of the tap vectors
// at byte offsets o[4] from srcb); ref2 = the lane's reference channels, pre-scaled by 1/CPG, as pairs.
// cor[g] = mean over the group's channels of warped * ref  (models/mvs4net_utils.py:1066-1069); optionally returns the
// interpolated ("warped") channels for the backward.
template <int NCH, int CPG, bool KEEP>
__device__ __forceinline__ void gather_correlate(const float* __restrict__ srcb, const unsigned (&o)[4], const float (&w)[4],
                                                 const f32x2 (&ref2)[NCH / 2], float (&cor)[NCH / CPG],
                                                 f32x2 (&warped)[KEEP ? NCH / 2 : 1]) {
    constexpr int N4 = NCH / 4;
    float4 t0[N4], t1[N4], t2[N4], t3[N4];
#pragma unroll
    for (int i = 0; i < N4; ++i) {
        t0[i] = load16(srcb, o[0] + 16 * i);
        t1[i] = load16(srcb, o[1] + 16 * i);
        t2[i] = load16(srcb, o[2] + 16 * i);
        t3[i] = load16(srcb, o[3] + 16 * i);
    }
    const f32x2 w0 = {w[0], w[0]}, w1 = {w[1], w[1]}, w2 = {w[2], w[2]}, w3 = {w[3], w[3]};
    f32x2 v[NCH / 2];
#pragma unroll
    for (int i = 0; i < N4; ++i) {
        const f32x2 a0 = {t0[i].x, t0[i].y}, a1 = {t0[i].z, t0[i].w};
        const f32x2 b0 = {t1[i].x, t1[i].y}, b1 = {t1[i].z, t1[i].w};
        const f32x2 c0 = {t2[i].x, t2[i].y}, c1 = {t2[i].z, t2[i].w};
        const f32x2 d0 = {t3[i].x, t3[i].y}, d1 = {t3[i].z, t3[i].w};
        v[2 * i] = __builtin_elementwise_fma(d0, w3, __builtin_elementwise_fma(c0, w2, __builtin_elementwise_fma(b0, w1, a0 * w0)));
        v[2 * i + 1] = __builtin_elementwise_fma(d1, w3, __builtin_elementwise_fma(c1, w2, __builtin_elementwise_fma(b1, w1, a1 * w0)));
    }
    if constexpr (KEEP) {
#pragma unroll
        for (int i = 0; i < NCH / 2; ++i) warped[i] = v[i];
    }
    if constexpr (CPG == 1) {
#pragma unroll
        for (int i = 0; i < NCH / 2; ++i) {
            const f32x2 pr = v[i] * ref2[i];
            cor[2 * i] = pr.x;
            cor[2 * i + 1] = pr.y;
        }
    } else {
        constexpr int PPG = CPG / 2;                  // channel pairs per group
#pragma unroll
        for (int g = 0; g < NCH / CPG; ++g) {
            f32x2 acc = v[g * PPG] * ref2[g * PPG];
#pragma unroll
            for (int k = 1; k < PPG; ++k) acc = __builtin_elementwise_fma(v[g * PPG + k], ref2[g * PPG + k], acc);
            cor[g] = acc.x + acc.y;
        }
    }
}

// byte offsets of the four taps inside one [Hs,Ws,C] map and the zero-padded bilinear weights
template <int C>
__device__ __forceinline__ void tap_addresses(const Landing& l, int Hs, int Ws, unsigned (&o)[4], float (&w)[4]) {
    const float fe = 1.0f - l.fx, fs = 1.0f - l.fy;
    const float wx0 = (l.x0in | l.bad) ? fe : 0.0f, wx1 = (l.x1in | l.bad) ? l.fx : 0.0f;
    const float wy0 = (l.y0in | l.bad) ? fs : 0.0f, wy1 = (l.y1in | l.bad) ? l.fy : 0.0f;
    w[0] = wy0 * wx0;
    w[1] = wy0 * wx1;
    w[2] = wy1 * wx0;
    w[3] = wy1 * wx1;
    const int x0 = clamp_i32(l.ix, 0, Ws - 1) * (C * 4), x1 = clamp_i32(l.ix + 1, 0, Ws - 1) * (C * 4);
    const int y0 = clamp_i32(l.iy, 0, Hs - 1), y1 = clamp_i32(l.iy + 1, 0, Hs - 1);
    const int row = Ws * (C * 4);                      // < 2^24: checked on the host (nhwc_addressable)
    o[0] = (unsigned)(__mul24(y0, row) + x0);
    o[1] = (unsigned)(__mul24(y0, row) + x1);
    o[2] = (unsigned)(__mul24(y1, row) + x0);
    o[3] = (unsigned)(__mul24(y1, row) + x1);
}

#endif

// ---- which reference pixels a workgroup owns (channels-last forward and coefficient kernels) ---------------------------
// A workgroup's PXB pixels form a 16-wide block (16 x PXB/16); blocks are numbered DOWN a band of kBandRows block rows
// first, then to the right.  With the XCD-contiguous tile mapping the ~256 workgroups in flight on an XCD then cover a
// compact region (16 block columns x a 64-row band at D = 4) whose source footprint -- the region widened by the depth
// scatter -- fits the XCD's 4 MB L2, and the front moves sideways through the source maps: a source line is fetched once
// per band instead of once per few reference rows (row strips: 3.7 GB fetched for 1.33 GB at DTU stage 4 on noise).
// Planar tensors ([B,D,H,W]) are still touched in 64-byte row segments per wave.
constexpr int kBandRows = 16;

template <int PXB>
struct PixelBlocks {
    static constexpr int BW = PXB >= 16 ? 16 : PXB, BH = PXB / BW;
    static inline int tiles(int H, int W) {
        const int tx = (W + BW - 1) / BW, ty = (H + BH - 1) / BH;
        return tx * (((ty + kBandRows - 1) / kBandRows) * kBandRows);
    }
};

#ifdef __HIPCC__
struct OwnedPixel {
    int p, x, y;       // clamped into the map
    bool live;         // false: padding lane outside the map
};

template <int PXB>
__device__ __forceinline__ OwnedPixel owned_pixel(int tile, int pl, int H, int W) {
    using PB = PixelBlocks<PXB>;
    const int tx = (W + PB::BW - 1) / PB::BW;
    const int per_band = tx * kBandRows;
    const int band = tile / per_band, rem = tile - band * per_band;
    const int bx = rem / kBandRows, by = band * kBandRows + (rem - bx * kBandRows);
    const int xr = bx * PB::BW + pl % PB::BW, yr = by * PB::BH + pl / PB::BW;
    OwnedPixel o;
    o.live = xr < W && yr < H;
    o.x = min(xr, W - 1);
    o.y = min(yr, H - 1);
    o.p = o.y * W + o.x;
    return o;
}
#endif

// (channels per group, groups) pairs instantiated: C in {8,16,32,64} x G in {4,8}
#define MVSTER_FOR_SHAPES(X) \
    X(2, 4) X(4, 4) X(8, 4) X(16, 4) X(1, 8) X(2, 8) X(4, 8) X(8, 8)

inline bool shape_instantiated(int C, int G, int D) {
    if (G <= 0 || C % G != 0) return false;
    const int cpg = C / G;
    const bool cg = (G == 4 && (cpg == 2 || cpg == 4 || cpg == 8 || cpg == 16)) ||
                    (G == 8 && (cpg == 1 || cpg == 2 || cpg == 4 || cpg == 8));
    return cg && (D == 4 || D == 8 || D == 16);
}

// Lane mapping.  FORWARD: always D lanes per pixel (DPL = 1): the kernel is gather-latency bound (SQ_WAIT_ANY 75 % of
// wave cycles with one lane per pixel at 132 VGPRs / 3 waves per SIMD); D lanes per pixel need 65 VGPRs, run 7 waves
// per SIMD and measured 1.55x faster at DTU stages 3-4.  BACKWARD: one lane per pixel (DPL = D) whenever D == 4,
// C <= 16 and there are enough pixels to fill the chip, because only then can a lane merge the hypotheses of its
// pixel before the atomic flush (stage 4: 19.0 ms with D lanes per pixel vs 7.8 ms); otherwise D lanes per pixel.
constexpr long long kMinPixelWaves = 8192;
constexpr int kMaxWideChannels = 16;

inline bool pick_pixel_per_lane(int B, int HW, int C, int D) {
    return D == 4 && C <= kMaxWideChannels && ((long long)B * HW) / kWave >= kMinPixelWaves;
}

// Tiled scatter of the deferred backward (cost_volume_scatter.hip): used for the stages with D = 8 hypotheses spread
// along the whole epipolar segment (no per-pixel merging possible).  Measured against the direct kernels (B x S views):
// DTU stage 2 (C = 32) 6.72 -> 3.31 ms, stage 1 (C = 64) 2.17 -> 1.47 ms; hi-res config stage 3 (768 x 1024, C = 16,
// 1 x 9 views) 8.76 -> 6.94 ms, stage 2 (384 x 512, C = 32, 1 x 9) 4.83 -> 2.93 ms (tools/probes/hires_bwd_probe.py).
inline bool tile_scatter_applies(int C, int G, int D, int layout) {
    return layout == MVSTER_LAYOUT_NHWC && D == 8 && (C == 16 || C == 32 || C == 64) && shape_instantiated(C, G, D);
}

// ---- binned (atomics-free) backward, cost_volume_binned.hip ---------------------------------------------------------
// The source map of every (batch, view) is cut into tiles of 2^lw x 2^lh pixels; one workgroup of the owner kernel sums
// everything that lands in a tile.  A (pixel, hypothesis) sample is listed ONCE, under the tile of its upper-left tap and
// one of four categories: interior, bottom row (its lower taps fall into the tile below), right column, corner.  The
// owner of a tile reads its own four lists plus the right-column / bottom-row / corner lists of its left / upper /
// upper-left neighbours -- so no sample is ever duplicated in memory and the item lists hold exactly one entry per
// (sample, view) (round 2 listed a sample under every tile its footprint touched: up to four entries, a 4x worst-case
// reservation and three extra rounds of wave-level grouping in the kernels that count and fill).
struct BinGrid {
    int lw, lh;               // log2 of the tile width / height in source pixels
    int tiles_x, tiles_y;     // tiles per source map
};
constexpr int kBinCats = 4;   // bin = tile * 4 + category; category = 2 * (right column) + (bottom row)

// tile area x C x 4 B of LDS for the tile image: 4 / 8 / 16 / 16 KB per workgroup
template <int C> struct BinTile;
#ifndef MVSTER_C8_LW          // tuning hooks (tools/probes/owner_sweep.sh builds variants); the defaults are the measured best
#define MVSTER_C8_LW 4
#endif
#ifndef MVSTER_C8_LH
#define MVSTER_C8_LH 3
#endif
#ifndef MVSTER_C8_KPT          // swept on MI355X in round 2, DTU stage 4, B = 4, noise / scene, ms per backward:
#define MVSTER_C8_KPT 2        //   16x8 tile: KPT 1 4.96/4.30, 2 4.45/3.73, 3 4.57/3.93, 4 4.91/4.01, 6 5.76/4.84
#endif                         //   8x8: KPT 2 4.62/3.86, 3 4.85/3.85, 4 5.04/4.11;  32x8: KPT 2 5.45/4.51, 3 7.03/6.02
//                             // round 3 (one item per sample; owner kernel alone, ms): 16x8 1.94; 8x4 (one wave per tile)
//                             //   KPT 2 / 3 / 4 2.25 / 2.10 / 2.14; 8x8 2.28; 16x4 2.36 -- smaller tiles also cost the
//                             //   count / fill passes more distinct bins per workgroup (tools/probes/ab_owner.sh)
#ifndef MVSTER_OWNER_KPT
#define MVSTER_OWNER_KPT 8
#endif
template <> struct BinTile<8> { static constexpr int kLw = MVSTER_C8_LW, kLh = MVSTER_C8_LH; };     // 16 x 8
template <> struct BinTile<16> { static constexpr int kLw = 4, kLh = 3; };    // 16 x 8
template <> struct BinTile<32> { static constexpr int kLw = 4, kLh = 3; };    // 16 x 8
template <> struct BinTile<64> { static constexpr int kLw = 3, kLh = 3; };    //  8 x 8

inline BinGrid bin_grid_for(int C, int Hs, int Ws) {
    BinGrid g{};
    g.lw = C == 8 ? BinTile<8>::kLw : C == 16 ? BinTile<16>::kLw : C == 32 ? BinTile<32>::kLw : BinTile<64>::kLw;
    g.lh = C == 8 ? BinTile<8>::kLh : C == 16 ? BinTile<16>::kLh : C == 32 ? BinTile<32>::kLh : BinTile<64>::kLh;
    g.tiles_x = (Ws + (1 << g.lw) - 1) >> g.lw;
    g.tiles_y = (Hs + (1 << g.lh) - 1) >> g.lh;
    return g;
}

// An item names its sample: reference pixel (x, y) and hypothesis d packed as y << 18 | x << 4 | d (no integer division
// by the run-time image width when it is unpacked).
constexpr int kItemXBits = 14, kItemDBits = 4;
constexpr int kMaxItemDim = 1 << kItemXBits;       // H, W <= 16384, D <= 16

// Workspace of the binned route, in 4-byte words: per-hypothesis group coefficients [B,S,HW,D,G] (float), then the
// counting sort's arrays (int): counts[nbins], partials[nbins / 1024 rounded up] (scan), offsets[nbins + 1], cursor[nbins],
// items[B*S*HW*D] -- exactly one item per (sample, view).
struct BinnedLayout {
    long long coef, counts, partials, offsets, cursor, items, total;     // word offsets; total = words needed
    long long nbins, max_items;
};

constexpr long long kMaxBinnedItems = (1LL << 31) - 1;           // 32-bit item positions
constexpr long long kMaxBins = 1LL << 24;                        // two-level scan with 1024-wide blocks

// Measured on MI355X against the atomic routes in round 2 (B = 4, S = 4, noise / smooth-scene hypotheses, ms per backward):
//   DTU stage 1 (C 64, D 8) 0.96 / 0.96 vs 1.50 / 1.50      stage 2 (C 32, D 8) 2.13 / 1.84 vs 3.35 / 2.47
//   stage 3 (C 16, D 4) 2.47 / 1.98 vs 3.02 / 3.43          stage 4 (C 8, D 4)  6.31 / 5.15 vs 7.07 / 7.62
// so the binned route is the recommendation wherever it exists: mvster_cost_volume_bwd_workspace() asks for its scratch.
constexpr int kMinBinnedChannels = 8;

inline bool binned_layout(int S, int B, int C, int G, int D, int H, int W, int Hs, int Ws, BinnedLayout* out) {
    if (!(C == 8 || C == 16 || C == 32 || C == 64) || !shape_instantiated(C, G, D)) return false;
    if (S <= 0 || B <= 0 || H <= 0 || W <= 0 || Hs <= 0 || Ws <= 0) return false;
    if (H > kMaxItemDim || W > kMaxItemDim || D > (1 << kItemDBits)) return false;
    const BinGrid g = bin_grid_for(C, Hs, Ws);
    BinnedLayout l{};
    l.nbins = (long long)B * S * g.tiles_x * g.tiles_y * kBinCats;
    l.max_items = (long long)B * S * H * W * D;
    if (l.max_items > kMaxBinnedItems || l.nbins > kMaxBins) return false;
    auto align4 = [](long long v) { return (v + 3) & ~3LL; };
    l.coef = 0;
    l.counts = align4((long long)B * S * H * W * D * G);
    l.partials = align4(l.counts + l.nbins);
    l.offsets = align4(l.partials + (l.nbins + 1023) / 1024);
    l.cursor = align4(l.offsets + l.nbins + 1);
    l.items = align4(l.cursor + l.nbins);
    l.total = align4(l.items + l.max_items);
    if (out != nullptr) *out = l;
    return true;
}

// fwd_counts: the bin counts the forward pass already took (mvster_cost_volume_fwd_counted), or nullptr: count here.
// items_filled: the coefficient kernel has already claimed the slots and written the item lists (BinFill below).
int launch_binned_scatter(const float* ref, const float* pose, const float* hypo, const GradPtrs& gsrcs, int B, int C, int G,
                          int D, const VolumeDims& dm, float* workspace, const BinnedLayout& lay, bool items_filled,
                          hipStream_t st);
int launch_bin_scan(const int* counts, float* workspace, const BinnedLayout& lay, hipStream_t st);

#ifdef __HIPCC__
// ---- the counting sort's two passes, riding on kernels that project every sample anyway ------------------------------
// The forward kernel counts (training mode) and the backward's coefficient kernel fills; both see bitwise the same
// landing positions.  Lanes of a wave are aggregated by RUNS of equal bins (neighbouring lanes are the hypotheses of one
// pixel and then the next pixel of the row: long runs on smooth depth, one per pixel on noise): the head of a run adds the
// run's length -- about 20 instructions per sample without a loop, where round 2's grouping by distinct tile looped over
// readlanes four times per sample (~150).
struct BinCount {            // forward: counters[((b*S + s) * tiles_per_view + tile) * 4 + cat] += 1 per sample
    int* counters;           // nullptr: do not count
    BinGrid bg;
    int tiles_per_view;
};
struct BinFill {             // backward coefficient kernel: claim slots from cursor[] (starts at the bin offsets), write items
    int* cursor;             // nullptr: do not fill
    int* items;
    BinGrid bg;
    int tiles_per_view;
    int max_items;           // capacity of items[]: a stale or foreign counts buffer cannot make the fill write past it
};

__device__ __forceinline__ int pack_item(int x, int y, int d) { return (y << (kItemXBits + kItemDBits)) | (x << kItemDBits) | d; }
__device__ __forceinline__ void unpack_item(int item, int& x, int& y, int& d) {
    d = item & ((1 << kItemDBits) - 1);
    x = (item >> kItemDBits) & (kMaxItemDim - 1);
    y = (int)((unsigned)item >> (kItemXBits + kItemDBits));
}

// Bin (within one (batch, view)) of a sample; listed = at least one tap inside the map and the position finite.
// entry: where the bin lives in a workgroup's direct-mapped LDS table (BinAgg below).
constexpr int kAggViews = 4;          // views whose tables are live together
constexpr int kAggEntries = 128;      // 8 tile columns x 4 tile rows x 4 categories (2 views x 512 entries measured slower)
__device__ __forceinline__ int sample_bin(const Landing& l, const BinGrid& bg, bool& listed, int& entry) {
    listed = (l.x0in | l.x1in) & (l.y0in | l.y1in) & !l.bad;
    const int xc = max(l.ix, 0), yc = max(l.iy, 0);            // ix, iy >= -1 for a listed sample
    const int tw1 = (1 << bg.lw) - 1, th1 = (1 << bg.lh) - 1;
    const int tx = xc >> bg.lw, ty = yc >> bg.lh;
    const int cat = ((xc & tw1) == tw1 ? 2 : 0) | ((yc & th1) == th1 ? 1 : 0);
    entry = (tx & 7) | ((ty & 3) << 3) | (cat << 5);
    return (__mul24(ty, bg.tiles_x) + tx) * kBinCats + cat;
}
__device__ __forceinline__ int sample_bin(const Landing& l, const BinGrid& bg, bool& listed) {
    int entry;
    return sample_bin(l, bg, listed, entry);
}

// bits of the lanes at or below this lane, as two 32-bit halves (computed once per thread)
struct LaneMasks {
    unsigned le_lo, le_hi;
};
__device__ __forceinline__ LaneMasks lane_masks(int lane) {
    LaneMasks m;
    m.le_lo = lane >= 31 ? 0xFFFFFFFFu : ((2u << lane) - 1u);
    m.le_hi = lane < 32 ? 0u : (lane == 63 ? 0xFFFFFFFFu : ((2u << (lane - 32)) - 1u));
    return m;
}

// Runs of equal keys over the lanes of a wave.  head: first lane of its run; len: lanes in the run (valid in the head);
// head_lane: the run's first lane (every lane).
struct Run {
    bool head;
    int len, head_lane;
};
__device__ __forceinline__ Run wave_runs(int key, int lane, const LaneMasks& lm) {
    const int prev = dpp_i<kDppWaveShr1>(key);
    Run r;
    r.head = (lane == 0) | (key != prev);
    const unsigned long long hm = __ballot(r.head);
    const unsigned lo = (unsigned)hm, hi = (unsigned)(hm >> 32);
    const unsigned alo = lo & ~lm.le_lo, ahi = hi & ~lm.le_hi;              // heads above this lane
    const int next = alo != 0u ? __builtin_ctz(alo) : (ahi != 0u ? 32 + __builtin_ctz(ahi) : kWave);
    r.len = next - lane;
    const unsigned blo = lo & lm.le_lo, bhi = hi & lm.le_hi;                // heads at or below this lane (lane 0 always is)
    r.head_lane = bhi != 0u ? 63 - __builtin_clz(bhi) : 31 - __builtin_clz(blo | 1u);
    return r;
}

// A sample may be shared by 2^HSLOG adjacent lanes (the channel-split kernels): all of them pass the same bin, a run then
// holds len >> HSLOG samples and only a sample's first lane writes its item.
// forward side: one non-returning int atomic per run of equal bins
template <int HSLOG = 0>
__device__ __forceinline__ void count_sample(const BinCount& bc, int* __restrict__ view_counters, const Landing& l, bool live,
                                             int lane, const LaneMasks& lm) {
    bool listed;
    const int bin = sample_bin(l, bc.bg, listed);
    listed &= live;
    const Run r = wave_runs(listed ? bin : -1, lane, lm);
    if (r.head && listed) atomicAdd(view_counters + bin, r.len >> HSLOG);
}

// fill side: the head of a run claims the run's slots with one returning atomic; the slots are consumed later (the value
// needs a round trip to the memory side), see consume_claim()
struct Claim {
    int base;                // head lanes: first slot of the run
    int head_lane;
    bool listed;
};
template <int HSLOG = 0>
__device__ __forceinline__ Claim claim_slots(const BinFill& bf, int* __restrict__ view_cursor, const Landing& l, bool live,
                                             int lane, const LaneMasks& lm) {
    bool listed;
    const int bin = sample_bin(l, bf.bg, listed);
    listed &= live;
    const Run r = wave_runs(listed ? bin : -1, lane, lm);
    Claim c;
    c.base = (r.head && listed) ? atomicAdd(view_cursor + bin, r.len >> HSLOG) : 0;
    c.head_lane = r.head_lane;
    c.listed = listed;
    return c;
}
template <int HSLOG = 0>
__device__ __forceinline__ void consume_claim(const BinFill& bf, const Claim& c, int lane, int item) {
    const int slot = __shfl(c.base, c.head_lane, kWave) + ((lane - c.head_lane) >> HSLOG);
    const bool first = (lane & ((1 << HSLOG) - 1)) == 0;
    if (c.listed && first && (unsigned)slot < (unsigned)bf.max_items) bf.items[slot] = item;
}

// ---- workgroup-level aggregation of the two passes ------------------------------------------------------------------------
// Device-scope integer atomics execute at the memory side at about the rate of the float ones (~20 G requests/s): with one
// request per run of equal bins per wave, the count and fill passes sent 22 M requests each at DTU stage 4 on noise and
// cost the kernels carrying them 0.15-0.3 ms.  The heads of the runs therefore first meet in a direct-mapped LDS table per
// (workgroup, view) -- entry = low bits of the tile coordinates and the category; a workgroup's samples of one view touch
// a handful of neighbouring tiles -- and ONE lane per occupied entry talks to memory.  Two heads whose bins collide on an
// entry (tiles 8 columns or 4 rows apart in one workgroup: far-spread hypotheses) fall back to their own global atomic.
struct BinAgg {
    int keys[kAggViews][kAggEntries];
    int cnts[kAggViews][kAggEntries];
    int bases[kAggViews][kAggEntries];       // fill pass only
};

__device__ __forceinline__ void agg_reset(BinAgg& t) {
    for (int e = threadIdx.x; e < kAggViews * kAggEntries; e += blockDim.x) {
        (&t.keys[0][0])[e] = -1;
        (&t.cnts[0][0])[e] = 0;
    }
    lds_barrier();
}

// count pass, one view: heads add their run to the workgroup's table (slot v)
template <int HSLOG = 0>
__device__ __forceinline__ void agg_count(BinAgg& t, int v, const BinCount& bc, int* __restrict__ view_counters, const Landing& l,
                                          bool live, int lane, const LaneMasks& lm) {
    bool listed;
    int entry;
    const int bin = sample_bin(l, bc.bg, listed, entry);
    listed &= live;
    const Run r = wave_runs(listed ? bin : -1, lane, lm);
    if (r.head && listed) {
        const int old = atomicCAS(&t.keys[v][entry], -1, bin);
        if (old == -1 || old == bin) atomicAdd(&t.cnts[v][entry], r.len >> HSLOG);
        else atomicAdd(view_counters + bin, r.len >> HSLOG);
    }
}

// after the last view of a batch (views s0 .. s0 + nv - 1 of batch item b): one global atomic per occupied entry
__device__ __forceinline__ void agg_flush_counts(BinAgg& t, int* __restrict__ counters, int bins_per_view, int b, int S, int s0) {
    lds_barrier();
    for (int e = threadIdx.x; e < kAggViews * kAggEntries; e += blockDim.x) {
        const int c = (&t.cnts[0][0])[e];
        if (c > 0) {
            atomicAdd(counters + ((size_t)b * S + s0 + e / kAggEntries) * bins_per_view + (&t.keys[0][0])[e], c);
            (&t.keys[0][0])[e] = -1;
            (&t.cnts[0][0])[e] = 0;
        }
    }
    lds_barrier();
}

// fill pass, one view: returns this lane's place as (entry << 16 | rank inside the workgroup's samples of that bin), or -1
// when the sample is not listed (or was written already: table collision, handled on the spot).
template <int HSLOG = 0>
__device__ __forceinline__ int agg_claim(BinAgg& t, int v, const BinFill& bf, int* __restrict__ view_cursor, const Landing& l,
                                         bool live, int lane, const LaneMasks& lm, int item) {
    bool listed;
    int entry;
    const int bin = sample_bin(l, bf.bg, listed, entry);
    listed &= live;
    const Run r = wave_runs(listed ? bin : -1, lane, lm);
    int place = -1;            // heads: entry << 16 | first rank of the run;  -2: collided
    if (r.head && listed) {
        const int old = atomicCAS(&t.keys[v][entry], -1, bin);
        place = (old == -1 || old == bin) ? ((entry << 16) | atomicAdd(&t.cnts[v][entry], r.len >> HSLOG)) : -2;
    }
    if (__ballot(place == -2) != 0) {                                    // rare, wave-uniform
        Claim c;
        c.base = place == -2 ? atomicAdd(view_cursor + bin, r.len >> HSLOG) : 0;
        c.head_lane = r.head_lane;
        c.listed = listed && __shfl(place, r.head_lane, kWave) == -2;
        consume_claim<HSLOG>(bf, c, lane, item);
    }
    const int head_place = __shfl(place, r.head_lane, kWave);
    return (listed && head_place >= 0) ? head_place + ((lane - r.head_lane) >> HSLOG) : -1;
}

// after the last view of a batch: one returning global atomic per occupied entry claims the workgroup's slots of that bin
__device__ __forceinline__ void agg_flush_claims(BinAgg& t, int* __restrict__ cursor, int bins_per_view, int b, int S, int s0) {
    lds_barrier();
    for (int e = threadIdx.x; e < kAggViews * kAggEntries; e += blockDim.x) {
        const int c = (&t.cnts[0][0])[e];
        if (c > 0) {
            (&t.bases[0][0])[e] = atomicAdd(cursor + ((size_t)b * S + s0 + e / kAggEntries) * bins_per_view + (&t.keys[0][0])[e], c);
            (&t.keys[0][0])[e] = -1;
            (&t.cnts[0][0])[e] = 0;
        }
    }
    lds_barrier();
}
template <int HSLOG = 0>
__device__ __forceinline__ void agg_write_item(const BinAgg& t, int v, const BinFill& bf, int place, int lane, int item) {
    const bool first = (lane & ((1 << HSLOG) - 1)) == 0;
    if (place >= 0 && first) {
        const int slot = t.bases[v][place >> 16] + (place & 0xFFFF);
        if ((unsigned)slot < (unsigned)bf.max_items) bf.items[slot] = item;
    }
}
#endif

// Which kernel family the backward entry point runs for a shape (mvster_cost_volume_bwd_route in mvster_hip.h).
inline int pick_bwd_route(int S, int B, int C, int G, int D, int H, int W, int Hs, int Ws, int layout,
                          long long workspace_floats) {
    if (S <= 0 || B <= 0 || H <= 0 || W <= 0 || !shape_instantiated(C, G, D)) return MVSTER_BWD_ROUTE_NONE;
    if (layout != MVSTER_LAYOUT_NHWC) return MVSTER_BWD_ROUTE_PLANAR;
    BinnedLayout lay;
    if (binned_layout(S, B, C, G, D, H, W, Hs, Ws, &lay) && workspace_floats >= lay.total) return MVSTER_BWD_ROUTE_BINNED;
    if (tile_scatter_applies(C, G, D, layout) && workspace_floats >= (long long)B * S * H * W * D * G)
        return MVSTER_BWD_ROUTE_TILE_SCATTER;        // a caller that only offers the round-1 workspace size
    if (pick_pixel_per_lane(B, H * W, C, D)) return MVSTER_BWD_ROUTE_SPLIT_WINDOW;
    return MVSTER_BWD_ROUTE_DENSE;
}

int launch_tile_scatter(const float* dcor, const float* ref, const float* pose, const float* hypo, const GradPtrs& gsrcs,
                        int B, int C, int G, int D, const VolumeDims& dm, hipStream_t st);

inline int check_volume(int S, int B, int C, int G, int D, int H, int W, int Hs, int Ws, float temp, int layout) {
    MVSTER_REQUIRE_SHAPE(layout == MVSTER_LAYOUT_NCHW || layout == MVSTER_LAYOUT_NHWC);
    MVSTER_REQUIRE_SHAPE(B > 0 && C > 0 && G > 0 && D > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0 && S > 0);
    MVSTER_REQUIRE_SHAPE(B <= 65535 && C % G == 0 && temp != 0.0f);
    MVSTER_REQUIRE_SHAPE((long long)H * W <= kMaxPlane && (long long)Hs * Ws <= kMaxPlane);
    if (S > MVSTER_MAX_SRC_VIEWS) return MVSTER_ERR_UNSUPPORTED;
    if (!shape_instantiated(C, G, D)) return MVSTER_ERR_UNSUPPORTED;
    if (layout == MVSTER_LAYOUT_NHWC) {
        // the channels-last kernels address a tap by a 32-bit byte offset inside one [Hs,Ws,C] map (24-bit row pitch) and a
        // volume element by a 32-bit offset inside one batch item
        MVSTER_REQUIRE_SHAPE((long long)Hs * Ws * C * 4 < (1LL << 32) && (long long)Ws * C * 4 < (1LL << 23) && Hs < (1 << 23));
        MVSTER_REQUIRE_SHAPE((long long)G * D * H * W * 4 < (1LL << 32));
    }
    return MVSTER_OK;
}

}  // namespace mvster
