// Tiled scatter of the source-feature gradient (second half of the deferred backward, see cost_volume_bwd.hip DEFER).
//
// replaces: the autograd backward of F.grid_sample w.r.t. the source feature inside homo_warping
// (models/mvs4net_utils.py:59), summed over hypotheses, for the channels-last layout.
//
// Why a second kernel.  Float atomics on gfx950 execute at the memory side in 64-byte requests at ~20 G requests/s
// chip-wide, whatever the payload, and LDS float atomics (ds_add_f32) run at one lane per ~3 clocks per CU
// (tools/probes/lds_atomic_probe.hip) - so the only way below the request count of "one request per tap line" is to
// merge contributions to the same source line BEFORE they leave the CU, with plain LDS read-modify-writes.  The
// hypotheses of an 8x8 block of reference pixels land in a thin band along the epipolar line: 2048 taps hit only
// 200-400 distinct source pixels at the coarse stages.  Each wave therefore owns an 8x8 pixel tile and a private
// direct-mapped write-combining cache in LDS (NSLOT entries of 16 channels = one 64-byte line of one source pixel):
// a tap adds into its entry, a tap that maps to an entry held by another pixel first evicts that entry with ONE dense
// atomic wave-instruction part (16 lanes = 64 contiguous bytes), and the cache is flushed once per (tile, view).
// No LDS atomics: the four 16-lane groups of a wave work on four different taps per step, a step whose groups collide
// on an entry is serialised (wave-uniform branch), and waves never share an entry.
//
// Work split: grid = (16x8-pixel tiles, B, S * C/16); a workgroup's 2 waves take the two 8x8 halves; per tile row, lane l
// first projects (pixel l/8, hypothesis l%8), then the wave walks the 64 records in 16 steps of four: 16-lane group g
// takes pixel 2g + parity, lane c of the group channel 16*chunk + c.
// The taps are recomputed from (pose, hypothesis) exactly as in the forward pass; the per-(pixel, hypothesis)
// coefficients come from `dcor` [B,S,HW,D,G] written by cost_volume_bwd_nhwc_kernel<..., DEFER = true>.
#include "cost_volume_common.h"

namespace mvster {

constexpr int kTile = 8;            // reference pixels per wave tile edge
constexpr int kChunk = 16;          // channels per cache entry: one 64-byte line
// entry distance of vertically adjacent source pixels; swept 13 / 17 / 21 / 29 / 37 / 45 on the bench inputs: stage-2
// backward 3.21 / 3.24 / 3.36 / 3.28 / 3.35 / 3.83 ms, stage 1 1.50 / 1.58 / 1.65 / 1.49 / 1.63 / 2.11 ms (flat below ~30)
constexpr int kRowStep = 29;
constexpr int kPad = kRowStep + 3;  // entries e .. e+kRowStep+1 of one hypothesis never wrap: the arrays are padded instead
constexpr int kScatterBlock = 128;  // two waves = a 16 x 8 pixel tile per workgroup (LDS, not threads, limits residency)

struct ScatterArgs {
    const float* dcor;
    const float* ref;
    const float* pose;
    const float* hypo;
    GradPtrs gsrcs;
    VolumeDims dm;
    int C, G, D, tiles_x;
};

// Orders the LDS accesses of one wave across its lanes: a wavefront-scope release / acquire pair around a wave barrier
// (s_waitcnt lgkmcnt(0) + barrier).  Round 1 relied on the LDS returning one wave's operations in issue order and used a
// compiler barrier only; this kernel is a fallback route now (the binned route is the default), so it pays the few
// per cent for the documented ordering instead.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The four taps of one hypothesis are the source pixels (ix,iy) (ix+1,iy) (ix,iy+1) (ix+1,iy+1).  With the entry index
// e = (ix + kRowStep iy) mod NB they occupy the four DIFFERENT entries e, e+1, e+29, e+30 (no wrap: kPad spare entries), so a
// group deposits its four taps as one batch with all LDS reads in flight together.  Two groups can touch the same entry
// only when their base entries differ by 0, 1, 28, 29 or 30.
__device__ __forceinline__ bool entries_overlap(int ea, int eb) {
    const int diff = abs(ea - eb);
    return (diff <= 1) | ((diff >= kRowStep - 1) & (diff <= kRowStep + 1));
}

struct __attribute__((aligned(16))) TapRecord {     // one (pixel, hypothesis), written by one lane, read by a 16-lane group
    int entry;          // base cache entry
    int off;            // iy * Ws + ix (element index of tap 0; taps 1..3 at +1, +Ws, +Ws+1 when their weight is non-zero)
    int pad0, pad1;
    float w[4];         // bilinear weights, 0 for taps outside the source map or a dead pixel
};

template <int NSLOT, int D>
__global__ void __launch_bounds__(kScatterBlock)
tile_scatter_kernel(ScatterArgs a) {
    static_assert(D == 8 || D == 4, "a pass = 64 / D tile pixels x D hypotheses = one (pixel, hypothesis) per lane");
    constexpr int kRowsPerPass = 8 / D;               // tile rows projected at once (1 or 2)
    constexpr int NB = NSLOT - kPad + 1;              // base entries 0 .. NB-1, taps reach NB-1+30 < NSLOT
    __shared__ int keys[kScatterBlock / kWave][NSLOT];
    __shared__ __attribute__((aligned(16))) float vals[kScatterBlock / kWave][NSLOT * kChunk];
    __shared__ TapRecord recs[kScatterBlock / kWave][kWave];
    const VolumeDims& dm = a.dm;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 4, cl = lane & 15;
    const int nchunk = a.C / kChunk;
    const int s = blockIdx.z / nchunk, chunk = blockIdx.z - s * nchunk;
    const int b = blockIdx.y;
    const int ty = blockIdx.x / a.tiles_x, tx = blockIdx.x - ty * a.tiles_x;
    const int x0 = tx * (2 * kTile) + wave * kTile, y0 = ty * kTile;
    if (x0 >= dm.W || y0 >= dm.H) return;                 // wave-uniform; waves never synchronise with each other
    const int HW = dm.H * dm.W, HWs = dm.Hs * dm.Ws;
    const int C = a.C, G = a.G, cpg = C / G;
    const int c = chunk * kChunk + cl;
    const Pose q = load_pose(a.pose + ((size_t)b * dm.S + s) * 12);
    float* __restrict__ gsrc = a.gsrcs.p[s] + (size_t)b * HWs * C + c;      // + pixel * C
    const float* __restrict__ dcor = a.dcor + ((size_t)b * dm.S + s) * HW * D * G + c / cpg;
    const float* __restrict__ hypo = a.hypo + (size_t)b * D * HW;
    int* __restrict__ wkeys = keys[wave];
    float* __restrict__ wvals = vals[wave];
    TapRecord* __restrict__ wrecs = recs[wave];

    for (int i = lane; i < NSLOT; i += kWave) wkeys[i] = -1;

    // phase-1 role of this lane: tile row (within the pass) r1, pixel column px1, hypothesis d1
    const int d1 = lane % D, px1 = (lane / D) & 7, r1 = lane / (8 * D);
    const int xr1 = x0 + px1;
    const bool live_x1 = xr1 < dm.W;
    const int x1 = live_x1 ? xr1 : dm.W - 1;
    // phase-2 role: pixel columns 2 grp and 2 grp + 1 (the four groups of a step are two pixels apart, so that at a
    // source/reference scale near 1 their 2x2 footprints do not touch), channel c
    const int xa = min(x0 + 2 * grp, dm.W - 1), xb = min(x0 + 2 * grp + 1, dm.W - 1);

    const int rows = min(kTile, dm.H - y0);
    for (int rp = 0; rp < rows; rp += kRowsPerPass) {
        // ---- phase 1: one projection per lane -----------------------------------------------------------------------
        const int yr1 = y0 + rp + r1;
        const bool live1 = live_x1 & (yr1 < dm.H);
        const int y1 = min(yr1, dm.H - 1);
        const int p1 = y1 * dm.W + x1;
        const Taps t = project_taps(q, rotate_pixel(q, (float)x1, (float)y1), hypo[(size_t)d1 * HW + p1], dm.Hs, dm.Ws);
        TapRecord rec;
        const int e = t.ix + kRowStep * t.iy;                        // ix, iy >= -2
        rec.entry = (e + 128 * NB) % NB;
        rec.off = t.iy * dm.Ws + t.ix;
        rec.pad0 = rec.pad1 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) rec.w[k] = live1 ? masked_weight(t, k) : 0.0f;
        const bool any1 = (rec.w[0] != 0.0f) | (rec.w[1] != 0.0f) | (rec.w[2] != 0.0f) | (rec.w[3] != 0.0f);
        wave_sync();                                                 // the previous pass's phase 2 is done with recs
        wrecs[lane] = rec;
        // does this (pixel, hypothesis) share a cache entry with the same hypothesis of another pixel of its step
        // (same row, pixels of equal parity)?
        const int mine = any1 ? rec.entry : -100000 - 1000 * lane;
        bool overlap = false;
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            const int pxo = (((px1 >> 1) + j) & 3) * 2 + (px1 & 1);
            const int other = __shfl(mine, (r1 * 8 + pxo) * D + d1, kWave);
            overlap |= entries_overlap(mine, other);
        }
        // bit = lane of the record; a phase-2 step is serialised when any of its four records has its bit set
        const unsigned long long clash_bits = __ballot(overlap);
        wave_sync();

        // ---- phase 2: 16 steps, four (pixel, hypothesis) records per step -------------------------------------------
        // step t: hypothesis t % D, pixel parity (t / D) & 1, row t / (2 D) of the pass; group g: pixel 2 g + parity
        float rv[kRowsPerPass][2];
#pragma unroll
        for (int rr = 0; rr < kRowsPerPass; ++rr) {
            const int yy = min(y0 + rp + rr, dm.H - 1);
            rv[rr][0] = a.ref[((size_t)b * HW + yy * dm.W + xa) * C + c];
            rv[rr][1] = a.ref[((size_t)b * HW + yy * dm.W + xb) * C + c];
        }
        float dc[16];
#pragma unroll
        for (int tstep = 0; tstep < 16; ++tstep) {
            const int d = tstep % D, par = (tstep / D) & 1, rr = tstep / (2 * D);
            const int yy = min(y0 + rp + rr, dm.H - 1);
            const int pp = yy * dm.W + (par ? xb : xa);
            dc[tstep] = dcor[((size_t)pp * D + d) * G] * rv[rr][par];
        }
#pragma unroll
        for (int tstep = 0; tstep < 16; ++tstep) {
            const int d = tstep % D, par = (tstep / D) & 1, rr = tstep / (2 * D);
            const TapRecord rc = wrecs[(rr * 8 + 2 * grp + par) * D + d];           // broadcast within the group
            const float base = dc[tstep];
            constexpr unsigned long long kQuad = 1ull | (1ull << (2 * D)) | (1ull << (4 * D)) | (1ull << (6 * D));
            const unsigned long long step_bits = kQuad << ((rr * 8 + par) * D + d);
            const bool clash = (clash_bits & step_bits) != 0;                        // wave-uniform
            const bool any = (rc.w[0] != 0.0f) | (rc.w[1] != 0.0f) | (rc.w[2] != 0.0f) | (rc.w[3] != 0.0f);
            auto deposit4 = [&]() {
                const int e0 = rc.entry, e2 = rc.entry + kRowStep;
                const int k0 = wkeys[e0], k1 = wkeys[e0 + 1], k2 = wkeys[e2], k3 = wkeys[e2 + 1];
                const float c0 = wvals[e0 * kChunk + cl], c1 = wvals[(e0 + 1) * kChunk + cl];
                const float c2 = wvals[e2 * kChunk + cl], c3 = wvals[(e2 + 1) * kChunk + cl];
                const int key[4] = {k0, k1, k2, k3};
                const float cur[4] = {c0, c1, c2, c3};
                const int off[4] = {rc.off, rc.off + 1, rc.off + dm.Ws, rc.off + dm.Ws + 1};
                float nv[4];
                int nk[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool act = rc.w[k] != 0.0f;
                    const bool hit = key[k] == off[k];
                    const float v = base * rc.w[k];
                    if (act & !hit & (key[k] >= 0) & (cur[k] != 0.0f)) atomicAdd(gsrc + (size_t)key[k] * C, cur[k]);
                    // an inactive tap of an active group rewrites its entry unchanged (the entry is reserved
                    // for this group in this step either way)
                    nv[k] = !act ? cur[k] : hit ? cur[k] + v : v;
                    nk[k] = (act & !hit) ? off[k] : key[k];
                }
                wvals[e0 * kChunk + cl] = nv[0];
                wvals[(e0 + 1) * kChunk + cl] = nv[1];
                wvals[e2 * kChunk + cl] = nv[2];
                wvals[(e2 + 1) * kChunk + cl] = nv[3];
                if (cl == 0) {
                    wkeys[e0] = nk[0];
                    wkeys[e0 + 1] = nk[1];
                    wkeys[e2] = nk[2];
                    wkeys[e2 + 1] = nk[3];
                }
            };
            if (!clash) {
                if (any) deposit4();
            } else {
                for (int g = 0; g < 4; ++g) {
                    if (any & (grp == g)) deposit4();
                    wave_sync();
                }
            }
            wave_sync();
        }
    }

    // final flush: four entries per wave-instruction, 64 contiguous bytes each
    wave_sync();
    for (int i = 0; i < NSLOT / 4; ++i) {
        const int slot = 4 * i + grp;
        const int key = wkeys[slot];
        if (key >= 0) {
            const float v = wvals[slot * kChunk + cl];
            if (v != 0.0f) atomicAdd(gsrc + (size_t)key * C, v);
        }
    }
}

int launch_tile_scatter(const float* dcor, const float* ref, const float* pose, const float* hypo, const GradPtrs& gsrcs,
                        int B, int C, int G, int D, const VolumeDims& dm, hipStream_t st) {
    if (C % kChunk != 0 || dm.S * (C / kChunk) > 65535) return MVSTER_ERR_UNSUPPORTED;
    ScatterArgs a{dcor, ref, pose, hypo, gsrcs, dm, C, G, D, grid_for(dm.W, 2 * kTile)};
    const int tiles_y = grid_for(dm.H, kTile);
    dim3 grid(a.tiles_x * tiles_y, B, dm.S * (C / kChunk));
    // The kernel also handles D = 4 (two tile rows per pass).  Measured on the stage-3 shape (C = 16, B = 4): coefficient
    // kernel 1.12 ms + scatter 2.21 ms = 3.39 ms against 3.30 ms for the two-lanes-per-pixel window kernel - the scatter
    // is instruction-bound (0.28 us per four records) before the saved requests pay off, so it is not dispatched.
    if (D != 8) return MVSTER_ERR_UNSUPPORTED;
    // 128 entries per wave (10.5 KB): measured best on MI355X against 96/160/192/256 - more entries merge more taps but
    // leave fewer waves per CU to hide the LDS round trips (stage 2, B=4: 3.54 / 3.32 / 3.80 / 3.77 / 4.03 ms)
    hipLaunchKernelGGL((tile_scatter_kernel<128, 8>), grid, dim3(kScatterBlock), 0, st, a);
    return launch_status();
}

}  // namespace mvster
