// Atomics-free scatter of the source-feature gradient: second half of the "binned" backward route.
//
// replaces: the autograd backward of F.grid_sample w.r.t. the source feature inside homo_warping
// (models/mvs4net_utils.py:59), summed over hypotheses and reference pixels, for the channels-last layout.
//
// Why.  Global float atomics execute at the memory side at ~1.3 TB/s of added bytes (~20 G 64-byte requests/s,
// MI355X_MICROARCH.md "Global float atomics"); every scatter route built on them is pinned to that rate however the taps
// are merged first (cost_volume_bwd.hip, cost_volume_scatter.hip).  Here the destination gets OWNERS instead: the source
// map of every (batch, view) is cut into tiles; every (pixel, hypothesis) sample is listed in the tiles its 2x2 footprint
// touches (a counting sort keyed by tile: count -> scan -> fill); one workgroup per tile then sorts its list once more in
// LDS by the source pixel of the sample's upper-left tap, and every lane -- the owner of one such pixel -- walks its own
// samples and sums their four taps in registers.  The four tap images are folded into the tile (four conflict-free
// passes over an LDS image) and the tile is written once with plain 16-byte stores.  No global float atomic, no memset of
// the gradient maps, a cost that does not depend on how coherent the depth maps are.
//
// Kernels (all launched by launch_binned_scatter on the caller's stream):
//   bin_items_kernel<false, D>  a workgroup = an 8x8 / 8x4 / 4x4 block of reference pixels x D hypotheses: project, count
//                               the tiles touched per view in an LDS hash table, one global int atomic per distinct tile
//   bin_scan_partials_kernel, bin_scan_apply_kernel   exclusive scan of the tile counts
//   bin_items_kernel<true, D>   same projection; ranks inside the workgroup from the LDS table, one RETURNING global
//                               atomic per distinct tile claims the slots, items written
//   tile_owner_kernel<C,G,D>    one workgroup per tile as described above (C = 8: tile_owner_c8_kernel<G,D>)
// The per-(view, pixel, hypothesis) group coefficients come from cost_volume_bwd_nhwc_kernel<..., DEFER = true>
// (cost_volume_bwd.hip): tap gradient = coef[group(c)] * ref[c] * bilinear weight.
#include <type_traits>

#include "cost_volume_common.h"

namespace mvster {

// ------------------------------------------------------------------------------------------------ counting sort by bin
struct BinArgs {
    const float* pose;      // [B,S,12]
    const float* hypo;      // [B,D,HW]
    VolumeDims dm;
    BinGrid bg;
    int tiles_per_view, max_items;
};

// Stand-alone count (FILL = false) / fill (FILL = true) pass for a backward whose forward did not count: the same
// projection and the same run aggregation as the fused passes (cost_volume_common.h), one lane per (pixel, hypothesis),
// all views in turn.  FILL: `counters` is the cursor array (starts at the bin offsets).
template <bool FILL, int D>
__global__ void __launch_bounds__(kBlock)
bin_items_kernel(BinArgs a, int* __restrict__ counters, int* __restrict__ items) {
    constexpr int PXB = kBlock / D;
    const VolumeDims& dm = a.dm;
    const int tile = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    if (tile >= dm.ntiles) return;
    const int HW = dm.H * dm.W;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int d = threadIdx.x % D, pl = threadIdx.x / D;
    const OwnedPixel own = owned_pixel<PXB>(tile, pl, dm.H, dm.W);
    const float hyp = a.hypo[((size_t)b * D + d) * HW + own.p];
    const LaneMasks lm = lane_masks(lane);
    const int item = pack_item(own.x, own.y, d);
    BinCount bc{counters, a.bg, a.tiles_per_view};
    BinFill bf{counters, items, a.bg, a.tiles_per_view, a.max_items};
    for (int s = 0; s < dm.S; ++s) {
        const Pose q = load_pose(a.pose + ((size_t)b * dm.S + s) * 12);
        const Landing l = land(q, rotate_pixel(q, (float)own.x, (float)own.y), hyp, dm.Hs, dm.Ws);
        int* __restrict__ view = counters + ((size_t)b * dm.S + s) * a.tiles_per_view * kBinCats;
        if constexpr (FILL) {
            const Claim c = claim_slots<0>(bf, view, l, own.live, lane, lm);
            consume_claim<0>(bf, c, lane, item);
        } else {
            count_sample<0>(bc, view, l, own.live, lane, lm);
        }
    }
}

// exclusive scan of counts[n] in two launches: per-block sums, then every block adds the sums of the blocks before it
constexpr int kScanBlock = 1024;

__device__ __forceinline__ int block_sum(int v, int* wave_tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int m = 1; m < kWave; m <<= 1) v += __shfl_xor(v, m, kWave);
    if (lane == 0) wave_tot[wave] = v;
    __syncthreads();
    int tot = 0;
    for (int w = 0; w < kScanBlock / kWave; ++w) tot += wave_tot[w];
    __syncthreads();
    return tot;
}

__global__ void __launch_bounds__(kScanBlock)
bin_scan_partials_kernel(const int* __restrict__ counts, int* __restrict__ partials, int n) {
    __shared__ int wave_tot[kScanBlock / kWave];
    const int i = blockIdx.x * kScanBlock + threadIdx.x;
    const int tot = block_sum(i < n ? counts[i] : 0, wave_tot);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// offsets[i] = sum of counts[0..i), offsets[n] = total; cursor[i] = offsets[i] (consumed by the fill pass)
__global__ void __launch_bounds__(kScanBlock)
bin_scan_apply_kernel(const int* __restrict__ counts, const int* __restrict__ partials, int* __restrict__ offsets,
                      int* __restrict__ cursor, int n) {
    __shared__ int wave_tot[kScanBlock / kWave];
    int before = 0;
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += kScanBlock) before += partials[j];
    const int base = block_sum(before, wave_tot);
    const int i = blockIdx.x * kScanBlock + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mine = i < n ? counts[i] : 0;
    int incl = mine;
#pragma unroll
    for (int m = 1; m < kWave; m <<= 1) {
        const int up = __shfl_up(incl, m, kWave);
        if (lane >= m) incl += up;
    }
    if (lane == kWave - 1) wave_tot[wave] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
    const int excl = base + wbase + incl - mine;
    if (i < n) {
        offsets[i] = excl;
        cursor[i] = excl;
    }
    if (i == n - 1) offsets[n] = excl + mine;
}

// ------------------------------------------------------------------------------------------------ one workgroup per tile
struct OwnerArgs {
    const float* dcor;      // [B,S,HW,D,G]
    const float* ref;       // [B,HW,C]
    const float* pose;
    const float* hypo;
    const int* offsets;     // [nbins + 1], nbins = ntiles * 4
    const int* items;
    GradPtrs gsrcs;
    VolumeDims dm;
    BinGrid bg;
    int tiles_per_view, ntiles;          // ntiles = B * S * tiles_per_view workgroups
};

template <int C>
struct OwnerTile {
    static constexpr int TW = 1 << BinTile<C>::kLw, TH = 1 << BinTile<C>::kLh;
    static constexpr int BW = TW + 1, BH = TH + 1;                 // source pixels that can hold the upper-left tap of a sample
    static constexpr int NCELL = BW * BH;                          // touching the tile: one column / row of halo
    static constexpr int NT = ((NCELL + kWave - 1) / kWave) * kWave;   // one lane per such pixel: 192 / 192 / 192 / 128
    static constexpr int KPT = C == 8 ? MVSTER_C8_KPT : MVSTER_OWNER_KPT;                     // samples a lane sorts per chunk
    static constexpr int CHUNK = NT * KPT;                         // 384 (48 B each in LDS) / 1536 / 1536 / 1024 samples (16 B each)
};

// The samples a tile's owner has to look at: its own four lists (interior, bottom row, right column, corner -- one
// contiguous stretch of the item array, which the right-column and corner lists of the LEFT neighbour directly precede),
// the bottom-row and corner lists of the tile above and the corner list of the upper-left tile.  Position i of the
// concatenation is item i + shift(i).
struct OwnerLists {
    int n;                  // samples in all
    int c0, c1, c2;         // cumulative ends of the first three stretches
    int s0, d1, d2, d3;     // shift of the first stretch, shift increments of the following ones
};
__device__ __forceinline__ OwnerLists owner_lists(const int* __restrict__ offsets, int tile, int tx, int ty, int tiles_x) {
    const int t4 = tile * kBinCats, u4 = t4 - tiles_x * kBinCats;
    const int a_beg = offsets[tx > 0 ? t4 - 2 : t4], a_end = offsets[t4 + 4];
    int b_beg = 0, b_end = 0, c_beg = 0, c_end = 0, d_beg = 0, d_end = 0;
    if (ty > 0) {
        c_beg = offsets[u4 + 1];            // above: bottom row
        c_end = offsets[u4 + 2];
        d_beg = offsets[u4 + 3];            // above: corner
        d_end = offsets[u4 + 4];
        if (tx > 0) {
            b_beg = offsets[u4 - 1];        // upper left: corner
            b_end = offsets[u4];
        }
    }
    OwnerLists o;
    o.c0 = a_end - a_beg;
    o.c1 = o.c0 + (b_end - b_beg);
    o.c2 = o.c1 + (c_end - c_beg);
    o.n = o.c2 + (d_end - d_beg);
    const int s1 = b_beg - o.c0, s2 = c_beg - o.c1, s3 = d_beg - o.c2;
    o.s0 = a_beg;
    o.d1 = s1 - a_beg;
    o.d2 = s2 - s1;
    o.d3 = s3 - s2;
    return o;
}
// (written with arithmetic on the comparison results: a chain of selects on four values becomes a lookup table in scratch)
__device__ __forceinline__ int owner_item(const int* __restrict__ items, const OwnerLists& o, int i) {
    const int shift = o.s0 + (i >= o.c0 ? o.d1 : 0) + (i >= o.c1 ? o.d2 : 0) + (i >= o.c2 ? o.d3 : 0);
    return items[i + shift];
}

struct __attribute__((aligned(16))) SampleRec {
    int id;                 // packed item: reference pixel and hypothesis
    int mask;               // in-map bits of the four taps
    float fx, fy;           // fractional sample position
};

template <int C, int G, int D>
__global__ void __launch_bounds__(OwnerTile<C>::NT)
tile_owner_kernel(OwnerArgs a) {
    using T = OwnerTile<C>;
    constexpr int TW = T::TW, TH = T::TH, BW = T::BW, NCELL = T::NCELL, NT = T::NT, KPT = T::KPT, CHUNK = T::CHUNK;
    constexpr int LW = BinTile<C>::kLw, LH = BinTile<C>::kLh;
    constexpr int CPG = C / G;
    // Channels per walk and samples in flight per owner, swept in round 3 (tools/probes/ab_owner.sh; us per launch at DTU
    // stages 3 / 2 / 1, B = 4, noise):   CH 8, PF 2 (round 2)  975 / 790 / 309      CH 8, PF 1  982 / 793 / 349      CH 8, PF 4  1468 / 924 / 313
    //                                    CH 16, PF 1           920 / 704 / 283      CH 16, PF 2  1324 / 788 / 267
    // A walk step is one dependent round trip to L2 / HBM whatever it fetches, so 16 channels per step halve the steps of the
    // 16-channel stage (one pass over the lists instead of two) -- as long as the registers still allow three waves per SIMD.
    // What the walk really loses is balance: with every owner walking the MEAN list length it takes 626 / 441 us (the longest
    // list of a wave is about twice the mean on noise).  Tried against that and dropped: lists capped at ~1.25 x mean with the
    // excess dealt to all lanes and everything accumulated by LDS double-precision atomics (ds_add_f64; the tile image
    // channel-major so that an instruction's lanes fall on different banks): 923 / 812 / 286 -- the excess pass is its own
    // dependent chain (binary search, record, gathers, atomics) in front of every wave's walk, and tighter caps made it worse
    // (mean + 1: 1077 / 1004 / 294); the C = 8 kernel's LDS-staged records at C = 16 (80-byte records, 192 or 384 per chunk):
    // 931 / 1199 us against 924 -- three waves per SIMD at best, and twice the LDS traffic per sample.
#ifndef MVSTER_OWNER_CH
#define MVSTER_OWNER_CH 16
#endif
    constexpr int CH = MVSTER_OWNER_CH < C ? MVSTER_OWNER_CH : C;  // channels per walk: 4 taps x CH accumulators per lane
    constexpr int GPC = CH / CPG > 0 ? CH / CPG : 1;               // coefficient groups a channel chunk spans
#ifndef MVSTER_OWNER_PF
#define MVSTER_OWNER_PF (C == 64 ? 2 : 1)
#endif
    constexpr int PF = MVSTER_OWNER_PF;                            // samples in flight per owner
    static_assert(CPG <= CH ? CH % CPG == 0 : CPG % CH == 0, "a channel chunk holds whole groups or lies inside one");
    __shared__ SampleRec sorted[CHUNK];
    __shared__ int cnt[NT], start[NT];
    __shared__ __attribute__((aligned(16))) float out[TH * TW * C];

    const int bin = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    if (bin >= a.ntiles) return;
    const VolumeDims& dm = a.dm;
    const int HW = dm.H * dm.W, HWs = dm.Hs * dm.Ws;
    const int tv = bin % a.tiles_per_view, bs = bin / a.tiles_per_view;
    const int s = bs % dm.S, b = bs / dm.S;
    const int ty = tv / a.bg.tiles_x, tx = tv - ty * a.bg.tiles_x;
    const int x0 = tx << LW, y0 = ty << LH;
    const int tid = threadIdx.x;
    // this lane owns the samples whose upper-left tap is source pixel (x0 - 1 + cbx, y0 - 1 + cby)
    const int cby = tid / BW, cbx = tid - cby * BW;
    const bool owner = tid < NCELL;

    for (int i = tid; i < TH * TW * C; i += NT) out[i] = 0.0f;

    const OwnerLists lists = owner_lists(a.offsets, bin, tx, ty, a.bg.tiles_x);
    const int n = lists.n;
    const Pose q = load_pose(a.pose + ((size_t)b * dm.S + s) * 12);
    const float* __restrict__ hypo = a.hypo + (size_t)b * D * HW;
    const float* __restrict__ dcor = a.dcor + ((size_t)b * dm.S + s) * HW * D * G;
    const float* __restrict__ ref = a.ref + (size_t)b * HW * C;

    for (int c0 = 0; c0 < n; c0 += CHUNK) {
        const int m = min(CHUNK, n - c0);
        cnt[tid] = 0;
        lds_barrier();                                           // also: the previous chunk's walk is over, `out` is zeroed
        // ---- sort the chunk by owning pixel: count ...
        SampleRec rec[KPT];
        int cell[KPT], slot[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int i = tid + j * NT;
            cell[j] = -1;
            if (i < m) {
                const int id = owner_item(a.items, lists, c0 + i);
                int x, y, d;
                unpack_item(id, x, y, d);
                const int p = __mul24(y, dm.W) + x;
                const Landing l = land(q, rotate_pixel(q, (float)x, (float)y), hypo[(size_t)d * HW + p], dm.Hs, dm.Ws);
                const int cx = l.ix - x0 + 1, cy = l.iy - y0 + 1;
                rec[j].id = p * D + d;
                rec[j].mask = landing_mask(l);
                rec[j].fx = l.fx;
                rec[j].fy = l.fy;
                if ((unsigned)cx < (unsigned)BW && (unsigned)cy < (unsigned)T::BH) {     // always, for a sample listed here
                    cell[j] = cy * BW + cx;
                    slot[j] = atomicAdd(&cnt[cell[j]], 1);
                }
            }
        }
        lds_barrier();
        // ... scan (first wave; NT / 64 consecutive counters per lane) ...
        if (tid < kWave) {
            constexpr int R = NT / kWave;
            int v[R], tot = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                v[r] = cnt[tid * R + r];
                tot += v[r];
            }
            int incl = tot;
#pragma unroll
            for (int mm = 1; mm < kWave; mm <<= 1) {
                const int up = __shfl_up(incl, mm, kWave);
                if (tid >= mm) incl += up;
            }
            int run = incl - tot;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                start[tid * R + r] = run;
                run += v[r];
            }
        }
        lds_barrier();
        // ... place
#pragma unroll
        for (int j = 0; j < KPT; ++j)
            if (cell[j] >= 0) sorted[start[cell[j]] + slot[j]] = rec[j];
        lds_barrier();

        // ---- every owner walks its samples, CH channels at a time ---------------------------------------------------
#ifdef MVSTER_PROBE_OWNER_NO_WALK            // timing probes (tools/probes/ab_owner.sh); results are wrong when defined
        const int mine = 0, first = 0;
#elif defined(MVSTER_PROBE_OWNER_EVEN_WALK)
        const int mine = owner ? (m + NCELL - 1) / NCELL : 0, first = 0;        // every owner walks the mean count
#else
        const int mine = owner ? cnt[tid] : 0, first = owner ? start[tid] : 0;
#endif
        for (int ch = 0; ch < C; ch += CH) {
            float acc[4][CH];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < CH; ++c) acc[k][c] = 0.0f;
            // The samples of one owner are walked PF at a time with all their loads issued together: the walk is bound by
            // the latency of these gathers (coefficients and reference features come from HBM / L2), not by arithmetic.
            for (int k = 0; k < mine; k += PF) {
                SampleRec r[PF];
                float4 rq[PF][CH / 4];
                float cf[PF][GPC];
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    r[u] = sorted[first + min(k + u, mine - 1)];
                    if (k + u >= mine) r[u].mask = 0;              // past the end: a repeat of the last sample, weightless
                }
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const int pj = r[u].id / D;
                    const float4* __restrict__ rp = reinterpret_cast<const float4*>(ref + (size_t)pj * C + ch);
#pragma unroll
                    for (int i = 0; i < CH / 4; ++i) rq[u][i] = rp[i];
                    const float* __restrict__ cp = dcor + (size_t)r[u].id * G + ch / CPG;
                    if constexpr (GPC == 8) {
                        const float4 a0 = reinterpret_cast<const float4*>(cp)[0], a1 = reinterpret_cast<const float4*>(cp)[1];
                        cf[u][0] = a0.x; cf[u][1] = a0.y; cf[u][2] = a0.z; cf[u][3] = a0.w;
                        cf[u][4] = a1.x; cf[u][5] = a1.y; cf[u][6] = a1.z; cf[u][7] = a1.w;
                    } else if constexpr (GPC == 4) {
                        const float4 a0 = reinterpret_cast<const float4*>(cp)[0];
                        cf[u][0] = a0.x; cf[u][1] = a0.y; cf[u][2] = a0.z; cf[u][3] = a0.w;
                    } else if constexpr (GPC == 2) {
                        const float2 a0 = reinterpret_cast<const float2*>(cp)[0];
                        cf[u][0] = a0.x; cf[u][1] = a0.y;
                    } else {
                        cf[u][0] = cp[0];
                    }
                }
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    float rv[CH];
#pragma unroll
                    for (int i = 0; i < CH / 4; ++i) {
                        rv[4 * i] = rq[u][i].x; rv[4 * i + 1] = rq[u][i].y; rv[4 * i + 2] = rq[u][i].z; rv[4 * i + 3] = rq[u][i].w;
                    }
                    float w[4];
                    weights_from_fraction(r[u].fx, r[u].fy, r[u].mask, w);
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const float dwarp = cf[u][CPG <= CH ? c / CPG : 0] * rv[c];
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) acc[kk][c] = fmaf(dwarp, w[kk], acc[kk][c]);
                    }
                }
            }
            // ---- fold the four tap images into the tile: within a pass every lane targets its own pixel ----------------
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int lx = cbx - 1 + (kk & 1), ly = cby - 1 + (kk >> 1);
                if (owner && (unsigned)lx < (unsigned)TW && (unsigned)ly < (unsigned)TH) {
                    float4* __restrict__ o = reinterpret_cast<float4*>(&out[(ly * TW + lx) * C + ch]);
#pragma unroll
                    for (int i = 0; i < CH / 4; ++i) {
                        float4 u = o[i];
                        u.x += acc[kk][4 * i]; u.y += acc[kk][4 * i + 1]; u.z += acc[kk][4 * i + 2]; u.w += acc[kk][4 * i + 3];
                        o[i] = u;
                    }
                }
                lds_barrier();
            }
        }
    }
    lds_barrier();

    // ---- the tile's only trip to memory: plain 16-byte stores, every pixel of the tile (zeros included) -------------
    float* __restrict__ gsrc = a.gsrcs.p[s] + (size_t)b * HWs * C;
    if (dm.grad_planar) {                    // [B,C,Hs,Ws] planes: rows of TW pixels per (channel, tile row)
        for (int e = tid; e < TH * TW * C; e += NT) {
            const int cellp = e % (TH * TW), chn = e / (TH * TW);
            const int lx = cellp & (TW - 1), ly = cellp >> LW;
            const int gx = x0 + lx, gy = y0 + ly;
            if (gx < dm.Ws && gy < dm.Hs) gsrc[(size_t)chn * HWs + (size_t)gy * dm.Ws + gx] = out[cellp * C + chn];
        }
        return;
    }
    for (int e = tid; e < TH * TW * C / 4; e += NT) {
        const int cellp = (4 * e) / C, chn = (4 * e) % C;
        const int lx = cellp & (TW - 1), ly = cellp >> LW;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx < dm.Ws && gy < dm.Hs)
            *reinterpret_cast<float4*>(gsrc + ((size_t)gy * dm.Ws + gx) * C + chn) = *reinterpret_cast<const float4*>(&out[cellp * C + chn]);
    }
}

// ---- C = 8: the same ownership scheme with everything a sample contributes staged in LDS ---------------------------------
// At eight channels the stage has the most samples (DTU stage 4: 121 M per backward) and the least arithmetic per
// sample; walking them with one gather per step (kernel above) is bound by the latency of those gathers (3.9 ms).
// Here the sorting pass -- one lane per SAMPLE, all lanes busy, neighbouring lanes = the hypotheses of one pixel, so
// their coefficient and reference loads share lines -- also fetches the coefficients and the reference feature and
// leaves the finished per-channel gradient coef[group(c)] * ref[c] and the four bilinear weights in the sample's LDS record
// (48 bytes).  The owners' walk then touches LDS only: three 16-byte reads and 16 packed FMAs per sample.
struct __attribute__((aligned(16))) StagedRec {
    float w[4];             // bilinear weights, 0 for a tap outside the map
    float dw[8];            // coef[group(c)] * ref[c]
};

#ifndef MVSTER_C8_OWNER_WAVES
#define MVSTER_C8_OWNER_WAVES 1
#endif
template <int G, int D>
__global__ void __launch_bounds__(OwnerTile<8>::NT, MVSTER_C8_OWNER_WAVES)
tile_owner_c8_kernel(OwnerArgs a) {
    constexpr int C = 8;
    using T = OwnerTile<C>;
    constexpr int TW = T::TW, TH = T::TH, BW = T::BW, NCELL = T::NCELL, NT = T::NT, KPT = T::KPT, CHUNK = T::CHUNK;
    constexpr int LW = BinTile<C>::kLw, LH = BinTile<C>::kLh;
    constexpr int CPG = C / G;
    __shared__ StagedRec sorted[CHUNK];
    __shared__ int cnt[NT], start[NT];
    __shared__ __attribute__((aligned(16))) float out[TH * TW * C];

    const int bin = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    if (bin >= a.ntiles) return;
    const VolumeDims& dm = a.dm;
    const int HW = dm.H * dm.W, HWs = dm.Hs * dm.Ws;
    const int tv = bin % a.tiles_per_view, bs = bin / a.tiles_per_view;
    const int s = bs % dm.S, b = bs / dm.S;
    const int ty = tv / a.bg.tiles_x, tx = tv - ty * a.bg.tiles_x;
    const int x0 = tx << LW, y0 = ty << LH;
    const int tid = threadIdx.x;
    const int cby = tid / BW, cbx = tid - cby * BW;
    const bool owner = tid < NCELL;

    for (int i = tid; i < TH * TW * C; i += NT) out[i] = 0.0f;

    const OwnerLists lists = owner_lists(a.offsets, bin, tx, ty, a.bg.tiles_x);
    const int n = lists.n;
    const Pose q = load_pose(a.pose + ((size_t)b * dm.S + s) * 12);
    const float* __restrict__ hypo = a.hypo + (size_t)b * D * HW;
    const float* __restrict__ dcor = a.dcor + ((size_t)b * dm.S + s) * HW * D * G;
    const float* __restrict__ ref = a.ref + (size_t)b * HW * C;

    f32x2 acc[4][C / 2];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < C / 2; ++c) acc[k][c] = f32x2{0.0f, 0.0f};

    // the items of the NEXT chunk are requested while this chunk's data is in flight: one dependent round trip per
    // chunk instead of two
    int id_next[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        const int i = tid + j * NT;
        id_next[j] = i < n ? owner_item(a.items, lists, i) : -1;
    }
    for (int c0 = 0; c0 < n; c0 += CHUNK) {
        cnt[tid] = 0;
        lds_barrier();
        // ---- fetch: everything that depends on the items, all in flight together ---------------------------------------
        int id[KPT], px[KPT], py[KPT];
        float hyp[KPT];
        float4 cf0[KPT], cf1[KPT], r0[KPT], r1[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            id[j] = id_next[j];
            int d;
            unpack_item(max(id[j], 0), px[j], py[j], d);
            const int p = __mul24(py[j], dm.W) + px[j];
            hyp[j] = hypo[(size_t)d * HW + p];
            const float4* __restrict__ cp = reinterpret_cast<const float4*>(dcor + ((size_t)p * D + d) * G);
            cf0[j] = cp[0];
            if constexpr (G == 8) cf1[j] = cp[1];
            const float4* __restrict__ rp = reinterpret_cast<const float4*>(ref + (size_t)p * C);
            r0[j] = rp[0];
            r1[j] = rp[1];
        }
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            const int i = c0 + CHUNK + tid + j * NT;
            id_next[j] = i < n ? owner_item(a.items, lists, i) : -1;
        }
        // ---- project, count per owner ----------------------------------------------------------------------------------
        StagedRec rec[KPT];
        int cell[KPT], slot[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            cell[j] = -1;
            if (id[j] >= 0) {
#ifdef MVSTER_PROBE_OWNER_NO_PROJECT
                Landing l{};
                l.ix = x0 + ((px[j] + id[j]) & (TW - 1)); l.iy = y0 + ((py[j] + (id[j] >> 4)) & (TH - 1)); l.fx = hyp[j] * 1e-3f; l.fy = 0.5f;
                l.x0in = l.x1in = l.y0in = l.y1in = true; l.bad = false;
#else
                const Landing l = land(q, rotate_pixel(q, (float)px[j], (float)py[j]), hyp[j], dm.Hs, dm.Ws);
#endif
                const int cx = l.ix - x0 + 1, cy = l.iy - y0 + 1;
                weights_from_fraction(l.fx, l.fy, landing_mask(l), rec[j].w);
                const float rv[C] = {r0[j].x, r0[j].y, r0[j].z, r0[j].w, r1[j].x, r1[j].y, r1[j].z, r1[j].w};
                const float cf[8] = {cf0[j].x, cf0[j].y, cf0[j].z, cf0[j].w, G == 8 ? cf1[j].x : 0.0f, G == 8 ? cf1[j].y : 0.0f,
                                     G == 8 ? cf1[j].z : 0.0f, G == 8 ? cf1[j].w : 0.0f};
#pragma unroll
                for (int c = 0; c < C; ++c) rec[j].dw[c] = cf[c / CPG] * rv[c];
                if ((unsigned)cx < (unsigned)BW && (unsigned)cy < (unsigned)T::BH) {
                    cell[j] = cy * BW + cx;
                    slot[j] = atomicAdd(&cnt[cell[j]], 1);
                }
            }
        }
        lds_barrier();
        if (tid < kWave) {
            constexpr int R = NT / kWave;
            int v[R], tot = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                v[r] = cnt[tid * R + r];
                tot += v[r];
            }
            int incl = tot;
#pragma unroll
            for (int mm = 1; mm < kWave; mm <<= 1) {
                const int up = __shfl_up(incl, mm, kWave);
                if (tid >= mm) incl += up;
            }
            int run = incl - tot;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                start[tid * R + r] = run;
                run += v[r];
            }
        }
        lds_barrier();
#pragma unroll
        for (int j = 0; j < KPT; ++j)
            if (cell[j] >= 0) sorted[start[cell[j]] + slot[j]] = rec[j];
        lds_barrier();

        // ---- walk: LDS only ------------------------------------------------------------------------------------------------
#ifdef MVSTER_PROBE_OWNER_NO_WALK            // timing probes (tools/probes/ab_owner.sh); results are wrong when defined
        const int mine = 0, first = 0;
#elif defined(MVSTER_PROBE_OWNER_EVEN_WALK)
        const int mine = owner ? (min(CHUNK, n - c0) + NCELL - 1) / NCELL : 0, first = 0;          // every owner walks the mean count
#else
        const int mine = owner ? cnt[tid] : 0, first = owner ? start[tid] : 0;
#endif
        for (int k = 0; k < mine; ++k) {
            const float4* __restrict__ rp = reinterpret_cast<const float4*>(&sorted[first + k]);
            const float4 w = rp[0], d0 = rp[1], d1 = rp[2];
            const f32x2 dv[4] = {{d0.x, d0.y}, {d0.z, d0.w}, {d1.x, d1.y}, {d1.z, d1.w}};
            const float wk[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f32x2 ww = {wk[kk], wk[kk]};
#pragma unroll
                for (int c = 0; c < C / 2; ++c) acc[kk][c] = __builtin_elementwise_fma(dv[c], ww, acc[kk][c]);
            }
        }
        lds_barrier();                                           // `sorted`, `cnt`, `start` are rewritten by the next chunk
    }

    // ---- fold the four tap images into the tile: within a pass every lane targets its own pixel ----------------------------
    lds_barrier();                                               // an empty tile gets here without any barrier since `out` was zeroed
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int lx = cbx - 1 + (kk & 1), ly = cby - 1 + (kk >> 1);
        if (owner && (unsigned)lx < (unsigned)TW && (unsigned)ly < (unsigned)TH) {
            float4* __restrict__ o = reinterpret_cast<float4*>(&out[(ly * TW + lx) * C]);
            float4 u = o[0], v = o[1];
            u.x += acc[kk][0].x; u.y += acc[kk][0].y; u.z += acc[kk][1].x; u.w += acc[kk][1].y;
            v.x += acc[kk][2].x; v.y += acc[kk][2].y; v.z += acc[kk][3].x; v.w += acc[kk][3].y;
            o[0] = u;
            o[1] = v;
        }
        lds_barrier();
    }
    float* __restrict__ gsrc = a.gsrcs.p[s] + (size_t)b * HWs * C;
    if (dm.grad_planar) {                    // [B,C,Hs,Ws] planes: rows of TW pixels per (channel, tile row)
        for (int e = tid; e < TH * TW * C; e += NT) {
            const int cellp = e % (TH * TW), chn = e / (TH * TW);
            const int lx = cellp & (TW - 1), ly = cellp >> LW;
            const int gx = x0 + lx, gy = y0 + ly;
            if (gx < dm.Ws && gy < dm.Hs) gsrc[(size_t)chn * HWs + (size_t)gy * dm.Ws + gx] = out[cellp * C + chn];
        }
        return;
    }
    for (int e = tid; e < TH * TW * C / 4; e += NT) {
        const int cellp = (4 * e) / C, chn = (4 * e) % C;
        const int lx = cellp & (TW - 1), ly = cellp >> LW;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx < dm.Ws && gy < dm.Hs)
            *reinterpret_cast<float4*>(gsrc + ((size_t)gy * dm.Ws + gx) * C + chn) = *reinterpret_cast<const float4*>(&out[cellp * C + chn]);
    }
}

template <int C, int G>
static int launch_owner(const OwnerArgs& a, int D, hipStream_t st) {
    const dim3 grid((unsigned)(((a.ntiles + 7) / 8) * 8)), block(OwnerTile<C>::NT);
    if constexpr (C == 8) {
        if (D == 4) hipLaunchKernelGGL((tile_owner_c8_kernel<G, 4>), grid, block, 0, st, a);
        else if (D == 8) hipLaunchKernelGGL((tile_owner_c8_kernel<G, 8>), grid, block, 0, st, a);
        else if (D == 16) hipLaunchKernelGGL((tile_owner_c8_kernel<G, 16>), grid, block, 0, st, a);
        else return MVSTER_ERR_UNSUPPORTED;
        return launch_status();
    }
    if (D == 4) hipLaunchKernelGGL((tile_owner_kernel<C, G, 4>), grid, block, 0, st, a);
    else if (D == 8) hipLaunchKernelGGL((tile_owner_kernel<C, G, 8>), grid, block, 0, st, a);
    else if (D == 16) hipLaunchKernelGGL((tile_owner_kernel<C, G, 16>), grid, block, 0, st, a);
    else return MVSTER_ERR_UNSUPPORTED;
    return launch_status();
}

template <bool FILL>
static int launch_bin_items(const BinArgs& ba, int B, int D, int* counters, int* items, hipStream_t st) {
    BinArgs a = ba;
    auto launch = [&](auto dtag) {
        constexpr int DD = decltype(dtag)::value;
        a.dm.ntiles = PixelBlocks<kBlock / DD>::tiles(a.dm.H, a.dm.W);
        const dim3 grid((unsigned)(((a.dm.ntiles + 7) / 8) * 8), (unsigned)B);
        hipLaunchKernelGGL((bin_items_kernel<FILL, DD>), grid, dim3(kBlock), 0, st, a, counters, items);
    };
    if (D == 4) launch(std::integral_constant<int, 4>{});
    else if (D == 8) launch(std::integral_constant<int, 8>{});
    else if (D == 16) launch(std::integral_constant<int, 16>{});
    else return MVSTER_ERR_UNSUPPORTED;
    return launch_status();
}

int launch_bin_scan(const int* counts, float* workspace, const BinnedLayout& lay, hipStream_t st) {
    int* const words = reinterpret_cast<int*>(workspace);
    const int nbins = (int)lay.nbins, nblocks = grid_for(nbins, kScanBlock);
    hipLaunchKernelGGL(bin_scan_partials_kernel, dim3(nblocks), dim3(kScanBlock), 0, st, counts, words + lay.partials, nbins);
    if (int rc = launch_status()) return rc;
    hipLaunchKernelGGL(bin_scan_apply_kernel, dim3(nblocks), dim3(kScanBlock), 0, st, counts, words + lay.partials,
                       words + lay.offsets, words + lay.cursor, nbins);
    return launch_status();
}

int launch_binned_scatter(const float* ref, const float* pose, const float* hypo, const GradPtrs& gsrcs, int B, int C, int G,
                          int D, const VolumeDims& dm, float* workspace, const BinnedLayout& lay, bool items_filled,
                          hipStream_t st) {
    const BinGrid bg = bin_grid_for(C, dm.Hs, dm.Ws);
    int* const words = reinterpret_cast<int*>(workspace);
    int* const counts = words + lay.counts;
    int* const offsets = words + lay.offsets;
    int* const cursor = words + lay.cursor;
    int* const items = words + lay.items;
    const int nbins = (int)lay.nbins;
    if (!items_filled) {                   // stand-alone counting sort: the forward pass did not count for us
        if (hipMemsetAsync(counts, 0, sizeof(int) * (size_t)nbins, st) != hipSuccess) {
            (void)hipGetLastError();
            return MVSTER_ERR_LAUNCH;
        }
        BinArgs ba{pose, hypo, dm, bg, bg.tiles_x * bg.tiles_y, (int)lay.max_items};
        if (int rc = launch_bin_items<false>(ba, B, D, counts, nullptr, st)) return rc;
        if (int rc = launch_bin_scan(counts, workspace, lay, st)) return rc;
        if (int rc = launch_bin_items<true>(ba, B, D, cursor, items, st)) return rc;
    }
    OwnerArgs oa{workspace + lay.coef, ref, pose, hypo, offsets, items, gsrcs, dm, bg, bg.tiles_x * bg.tiles_y, nbins / kBinCats};
#define X(C_, G_) \
    if (C == C_ && G == G_) return launch_owner<C_, G_>(oa, D, st);
    X(8, 4) X(16, 4) X(32, 4) X(64, 4) X(8, 8) X(16, 8) X(32, 8) X(64, 8)
#undef X
    return MVSTER_ERR_UNSUPPORTED;
}

}  // namespace mvster
