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
    int grad_planar; // backward only: write gref / gsrcs as [B,C,H,W] planes (MVSTER_GRAD_PLANAR) although features are read [B,H,W,C]
};

// x / attn_temp as the reference computes it; a power-of-two temperature (the scripts' attn_temp 2) divides exactly by
// multiplication -- one instruction instead of the ~10 of an IEEE division, in kernels that are VALU-bound.
#ifdef __HIPCC__
__device__ __forceinline__ float div_temp(float x, const VolumeDims& dm) {
    return dm.inv_temp != 0.0f ? x * dm.inv_temp : x / dm.temp;
}
#endif

inline VolumeDims make_dims(int H, int W, int Hs, int Ws, int S, int C, float attn_temp) {
    int e = 0;
    const float m = frexpf(fabsf(attn_temp), &e);
    const bool pow2 = m == 0.5f && e > -100 && e < 100;
    return VolumeDims{H, W, Hs, Ws, S, 0, attn_temp, (float)sqrt((double)C), pow2 ? 1.0f / attn_temp : 0.0f, 0};
}

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
// The source map of every (batch, view) is cut into tiles of 2^lw x 2^lh pixels; a tile is one bin of the counting sort
// and one workgroup of the accumulate kernel, whose LDS holds the tile as C doubles per pixel.
struct BinGrid {
    int lw, lh;               // log2 of the tile width / height in source pixels
    int tiles_x, tiles_y;     // tiles per source map
};

// tile area x C x 4 B of LDS for the tile image: 4 / 8 / 16 / 16 KB per workgroup
template <int C> struct BinTile;
#ifndef MVSTER_C8_LW          // tuning hooks (tools/probes/owner_sweep.sh builds variants); the defaults are the measured best
#define MVSTER_C8_LW 4
#endif
#ifndef MVSTER_C8_KPT          // swept on MI355X, DTU stage 4, B = 4, noise / scene, ms per backward:
#define MVSTER_C8_KPT 2        //   16x8 tile: KPT 1 4.96/4.30, 2 4.45/3.73, 3 4.57/3.93, 4 4.91/4.01, 6 5.76/4.84
#endif                         //   8x8: KPT 2 4.62/3.86, 3 4.85/3.85, 4 5.04/4.11;  32x8: KPT 2 5.45/4.51, 3 7.03/6.02
#ifndef MVSTER_OWNER_KPT
#define MVSTER_OWNER_KPT 8
#endif
template <> struct BinTile<8> { static constexpr int kLw = MVSTER_C8_LW, kLh = 3; };     // 16 x 8
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

// Workspace of the binned route, in 4-byte words: per-hypothesis group coefficients [B,S,HW,D,G] (float), then the
// counting sort's arrays (int): counts[nbins], partials[nbins / 1024 rounded up] (scan), offsets[nbins + 1], cursor[nbins],
// items[4 * B*S*HW*D] -- an item (pixel, hypothesis) of one view is listed in every tile its 2x2 footprint touches, at
// most four.
struct BinnedLayout {
    long long coef, counts, partials, offsets, cursor, items, total;     // word offsets; total = words needed
    long long nbins, max_items;
};

constexpr long long kMaxBinnedItems = (1LL << 31) - 1;           // 32-bit item positions
constexpr long long kMaxBins = 1LL << 22;                        // single-workgroup scan

// Measured on MI355X against the atomic routes (B = 4, S = 4, noise / smooth-scene hypotheses, ms per backward):
//   DTU stage 1 (C 64, D 8) 0.96 / 0.96 vs 1.50 / 1.50      stage 2 (C 32, D 8) 2.13 / 1.84 vs 3.35 / 2.47
//   stage 3 (C 16, D 4) 2.47 / 1.98 vs 3.02 / 3.43          stage 4 (C 8, D 4)  6.31 / 5.15 vs 7.07 / 7.62
// so the binned route is the recommendation wherever it exists: mvster_cost_volume_bwd_workspace() asks for its scratch.
constexpr int kMinBinnedChannels = 8;

inline bool binned_layout(int S, int B, int C, int G, int D, int H, int W, int Hs, int Ws, BinnedLayout* out) {
    if (!(C == 8 || C == 16 || C == 32 || C == 64) || !shape_instantiated(C, G, D)) return false;
    if (S <= 0 || B <= 0 || H <= 0 || W <= 0 || Hs <= 0 || Ws <= 0) return false;
    const BinGrid g = bin_grid_for(C, Hs, Ws);
    BinnedLayout l{};
    l.nbins = (long long)B * S * g.tiles_x * g.tiles_y;
    l.max_items = 4LL * B * S * H * W * D;
    if (l.max_items > kMaxBinnedItems || l.nbins > kMaxBins || (long long)H * W * D > kMaxBinnedItems) return false;
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

// fwd_counts: the tile counts the forward pass already took (mvster_cost_volume_fwd_counted), or nullptr: count here.
// items_filled: the coefficient kernel has already claimed the slots and written the item lists (BinFill below).
int launch_binned_scatter(const float* ref, const float* pose, const float* hypo, const GradPtrs& gsrcs, int B, int C, int G,
                          int D, const VolumeDims& dm, float* workspace, const BinnedLayout& lay, bool items_filled,
                          hipStream_t st);
int launch_bin_scan(const int* counts, float* workspace, const BinnedLayout& lay, hipStream_t st);

#ifdef __HIPCC__
// ---- the counting sort's two passes, riding on kernels that project every sample anyway ------------------------------
// Projecting a sample costs ~100 VALU instructions and these kernels are VALU-bound (4 IEEE divisions per projection),
// so a separate count and a separate fill pass cost 0.6 + 0.85 ms at DTU stage 4.  The forward kernel therefore counts
// (training mode) and the backward's coefficient kernel fills; both see bitwise the same taps.
struct BinCount {            // forward: counters[(b*S + s) * tiles_per_view + tile] += 1 per touched tile
    int* counters;           // nullptr: do not count
    BinGrid bg;
    int tiles_per_view;
};
struct BinFill {             // backward coefficient kernel: claim slots from cursor[] (starts at the tile offsets), write items
    int* cursor;             // nullptr: do not fill
    int* items;
    BinGrid bg;
    int tiles_per_view;
};

// Tiles (within one (batch, view) plane) that receive a non-zero tap of one sample: id[k] is the tile of tap k and
// emit[k] tells whether tap k is the first tap with a non-zero weight in that tile.  At most four tiles.
__device__ __forceinline__ void footprint_tiles(const Taps& t, const BinGrid& bg, int (&id)[4], bool (&emit)[4]) {
    bool nz[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        nz[k] = masked_weight(t, k) != 0.0f;                       // in-map taps only: coordinates are >= 0 there
        id[k] = ((t.iy + (k >> 1)) >> bg.lh) * bg.tiles_x + ((t.ix + (k & 1)) >> bg.lw);
    }
    emit[0] = nz[0];
    emit[1] = nz[1] && !(nz[0] && id[0] == id[1]);
    emit[2] = nz[2] && !(nz[0] && id[0] == id[2]) && !(nz[1] && id[1] == id[2]);
    emit[3] = nz[3] && !(nz[0] && id[0] == id[3]) && !(nz[1] && id[1] == id[3]) && !(nz[2] && id[2] == id[3]);
}

// Groups the lanes of a wave that emit the same tile: leader = lowest such lane, rank = position among them, count = size
// of the group (leader = -1 for lanes that emit nothing).  The loop runs once per distinct tile in the wave (1-4 at the
// fine stages); its trip count is wave-uniform.
__device__ __forceinline__ void wave_group(int tile, bool emit, int lane, int& leader, int& rank, int& count) {
    leader = -1;
    rank = 0;
    count = 0;
    unsigned long long pend = __ballot(emit);
    while (pend) {
        const int l = __ffsll((long long)pend) - 1;
        const int key = __builtin_amdgcn_readlane(tile, l);
        const unsigned long long same = __ballot(emit && tile == key);
        if (emit && tile == key) {
            leader = l;
            rank = __popcll(same & ((1ull << lane) - 1ull));
            count = __popcll(same);
        }
        pend &= ~same;
    }
}

// forward side: one non-returning int atomic per distinct tile of the wave
__device__ __forceinline__ void count_sample(const BinCount& bc, int* __restrict__ view_counters, const Taps& t, bool live,
                                             int lane) {
    int id[4];
    bool emit[4];
    footprint_tiles(t, bc.bg, id, emit);
    // taps 1-3 open a new tile only when the footprint straddles a tile edge: one ballot skips their three rounds
    const bool straddles = __ballot(live && (emit[1] | emit[2] | emit[3])) != 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k > 0 && !straddles) break;
        int leader, rank, count;
        wave_group(id[k], live && emit[k], lane, leader, rank, count);
        if (lane == leader) atomicAdd(view_counters + id[k], count);
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
    return MVSTER_OK;
}

}  // namespace mvster
