// Device-side helpers shared by the gfx950 kernels of libmvster_hip.so.
//
// Round 3: written for instruction count (the round-2 stage-4 forward issued ~415 wave-instructions per (sample, view)):
//   * in-bounds tests on the converted integers (unsigned compares), zero-padding folded into the separable weights;
//   * reciprocal + residual correction instead of IEEE division sequences where the quotient is well conditioned
//     (soft-max normalisation, num / den); DPP cross-lane reductions (no ds_bpermute) inside aligned lane groups;
//   * packed fp32 arithmetic for the channel pairs.
// ONE thing is deliberately NOT cheapened: the SAMPLE POSITION.  It follows the reference's CPU operator sequence
// rounding for rounding -- separately rounded ray * depth + trans, IEEE divisions by z, and F.grid_sample's [-1,1]
// normalise / un-normalise round trip (models/mvs4net_utils.py:42-52,59).  The fp32 reference's own positions are ~1e-4 px
// from the float64 result at W = 1600, and on high-frequency features that much position noise moves the fused volume by
// ~1e-3 and flips near-tie arg-max picks: a kernel that is merely "as accurate as" the reference differs from it by that
// much (measured: 1.2e-3 on the volume at DTU stage 4), one that rounds like it agrees to ~1e-6 and reproduces its picks.
// The ~60 extra VALU instructions per sample are free: the gather kernels are bound by the L1 / texture-address path
// (SQ_WAIT_INST_ANY 60 % of wave cycles, VALU pipe < 50 % busy; profiles/README.md), not by VALU issue.
// The library is compiled with -ffp-contract=off; every fused multiply-add below is written explicitly.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvster {

constexpr int kWave = 64;    // CDNA wavefront width
constexpr int kBlock = 256;  // threads per workgroup used by every kernel here (4 waves)

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8 shares an XCD, each with its own L2).
// Remap so that every XCD walks one contiguous band of pixel tiles: the two reference rows that
// touch the same source row are then served by the same L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_contiguous_tile(int block, int nblocks) {
    const int per_xcd = (nblocks + 7) >> 3;
    return (block & 7) * per_xcd + (block >> 3);
}

// ---- arithmetic primitives -------------------------------------------------------------------------------------------
__device__ __forceinline__ float rcp_approx(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32, 1 ulp

// a / b: reciprocal, product, one residual correction -- the correctly rounded quotient except in rare last-bit cases
// (4 instructions against ~11 for the IEEE sequence).  b must be a normal number: a denormal or infinite divisor gives
// NaN / 0 where IEEE division would not; no divisor on the path can be one (den >= 1e-8, softmax sums >= 1, and a
// projected z that is exactly 0 is replaced by 1e-9 as in the reference).
__device__ __forceinline__ float div_fast(float a, float b) {
    const float r = rcp_approx(b);
    const float q = a * r;
    const float e = fmaf(-q, b, a);
    return fmaf(e, r, q);
}

// a / c for a constant c with rc = 1/c precomputed: 3 instructions, same rounding quality as div_fast
__device__ __forceinline__ float div_const(float a, float c, float rc) {
    const float q = a * rc;
    const float e = fmaf(-q, c, a);
    return fmaf(e, rc, q);
}

// exp(x) for x <= 0 (soft-max arguments after the maximum is subtracted): 2^(x log2 e) with the product carried in two
// floats, v_exp_f32 on the high part and a first-order correction for the low part; ~1.5 ulp, 6 instructions.
// Results below the normal range flush to 0 (the reference produces denormals there: < 1.2e-38 absolute).
__device__ __forceinline__ float exp_nonpos(float x) {
    constexpr float kLog2eHi = 1.44269502162933349609375f;
    constexpr float kLog2eLo = 1.92596299112661746e-8f;
    constexpr float kLn2 = 0.693147180559945309f;
    const float t = x * kLog2eHi;
    float r = fmaf(x, kLog2eHi, -t);
    r = fmaf(x, kLog2eLo, r);
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, r * kLn2, e);
}

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() fences every address space: it drains the
// wave's global loads (s_waitcnt vmcnt(0)) before the barrier, so a prefetch issued ahead of a barrier is over when the
// barrier returns and nothing overlaps.  The LDS-only fence waits for the wave's LDS operations alone (lgkmcnt); global
// loads, stores and atomics stay in flight across it.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- cross-lane: DPP ---------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
constexpr int kDppXor1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141;   // lane i <-> 7 - i within 8 lanes: the "other quad" once quads hold equal values
constexpr int kDppMirror = 0x140;       // lane i <-> 15 - i within 16 lanes: the "other half" once halves hold equal values
constexpr int kDppWaveShr1 = 0x138;     // lane i reads lane i - 1 (lane 0: bound_ctrl zero)

// All-reductions over the LANES consecutive lanes (LANES a power of two <= 64) that share one pixel.  Up to 16 lanes stay
// inside a DPP row: quad permutes, then the mirrored halves (valid because every lane of a quad / half already holds the
// same partial result).
template <int LANES>
__device__ __forceinline__ float group_max(float v) {
    if constexpr (LANES >= 2) v = fmaxf(v, dpp<kDppXor1>(v));
    if constexpr (LANES >= 4) v = fmaxf(v, dpp<kDppXor2>(v));
    if constexpr (LANES >= 8) v = fmaxf(v, dpp<kDppHalfMirror>(v));
    if constexpr (LANES >= 16) v = fmaxf(v, dpp<kDppMirror>(v));
#pragma unroll
    for (int m = 16; m < LANES; m <<= 1) v = fmaxf(v, __shfl_xor(v, m, kWave));
    return v;
}

template <int LANES>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (LANES >= 2) v += dpp<kDppXor1>(v);
    if constexpr (LANES >= 4) v += dpp<kDppXor2>(v);
    if constexpr (LANES >= 8) v += dpp<kDppHalfMirror>(v);
    if constexpr (LANES >= 16) v += dpp<kDppMirror>(v);
#pragma unroll
    for (int m = 16; m < LANES; m <<= 1) v += __shfl_xor(v, m, kWave);
    return v;
}

// The same reductions over N lanes that are STRIDE apart (lane bits log2(STRIDE) .. log2(STRIDE*N)-1): used when a pixel's
// lanes are laid out [hypothesis][channel part] and only one of the two indices is to be reduced.
template <int N, int STRIDE>
__device__ __forceinline__ float strided_max(float v) {
#pragma unroll
    for (int m = STRIDE; m < N * STRIDE; m <<= 1) {
        if (m == 1) v = fmaxf(v, dpp<kDppXor1>(v));
        else if (m == 2) v = fmaxf(v, dpp<kDppXor2>(v));
        else v = fmaxf(v, __shfl_xor(v, m, kWave));
    }
    return v;
}

template <int N, int STRIDE>
__device__ __forceinline__ float strided_sum(float v) {
#pragma unroll
    for (int m = STRIDE; m < N * STRIDE; m <<= 1) {
        if (m == 1) v += dpp<kDppXor1>(v);
        else if (m == 2) v += dpp<kDppXor2>(v);
        else v += __shfl_xor(v, m, kWave);
    }
    return v;
}

// ---- geometry ------------------------------------------------------------------------------------------------------------
// Relative pose of one (batch, source view): rot row-major then trans.  Wave-uniform -> SGPRs.
struct Pose {
    float r00, r01, r02, r10, r11, r12, r20, r21, r22, t0, t1, t2;
};

__device__ __forceinline__ Pose load_pose(const float* __restrict__ p) {
    Pose q;
    q.r00 = p[0]; q.r01 = p[1]; q.r02 = p[2];
    q.r10 = p[3]; q.r11 = p[4]; q.r12 = p[5];
    q.r20 = p[6]; q.r21 = p[7]; q.r22 = p[8];
    q.t0 = p[9]; q.t1 = p[10]; q.t2 = p[11];
    return q;
}

// rot @ [x, y, 1]  (models/mvs4net_utils.py:42, a 3-term dot product per row)
struct Ray {
    float x, y, z;
};

__device__ __forceinline__ Ray rotate_pixel(const Pose& q, float x, float y) {
    Ray r;
    r.x = fmaf(q.r01, y, q.r00 * x) + q.r02;
    r.y = fmaf(q.r11, y, q.r10 * x) + q.r12;
    r.z = fmaf(q.r21, y, q.r20 * x) + q.r22;
    return r;
}

__device__ __forceinline__ int cvt_sat_i32(float v) {     // v_cvt_i32_f32: saturates, NaN -> 0 (the C++ cast is undefined there)
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

__device__ __forceinline__ int clamp_i32(int v, int lo, int hi) {     // v_med3_i32
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}

// Where a sample lands in the source map: floor of the position, fraction, which of the two columns / rows of its 2x2
// footprint are inside the map.  A non-finite position (NaN, or Inf - Inf in the fraction) is flagged `bad`: the forward
// pass keeps its NaN weights so that NaN propagates like in the reference; its backward scatters nothing, as grid_sample's.
struct Landing {
    int ix, iy;                  // floor of the sample position (saturated to the int range); tap (dy,dx) = (iy+dy, ix+dx)
    float fx, fy;                // fractional position: w00 = (1-fy)(1-fx), w01 = (1-fy) fx, w10 = fy (1-fx), w11 = fy fx
    bool x0in, x1in, y0in, y1in; // column ix / ix+1, row iy / iy+1 inside the map
    bool bad;                    // position not finite
};

//   models/mvs4net_utils.py:43-52  p = ray*depth + trans (separately rounded), z==0 -> 1e-9,
//                                  uv = p.xy / z, grid = uv / ((size-1)/2) - 1
//   F.grid_sample (:59), CPU kernel: pix = (grid + 1) * ((size-1)/2); w = pix - floor(pix);
//                                  taps outside [0,size) contribute zero.
// Position arithmetic: the reference's operations in the reference's order, IEEE divisions included (see the file header).
__device__ __forceinline__ Landing land(const Pose& q, const Ray& r, float depth, int Hs, int Ws) {
    const float half_w = (float)((Ws - 1) / 2.0), half_h = (float)((Hs - 1) / 2.0);      // wave-uniform
    const float px = r.x * depth + q.t0;
    const float py = r.y * depth + q.t1;
    float pz = r.z * depth + q.t2;
    pz = pz == 0.0f ? 1e-9f : pz;
    const float gx = (px / pz) / half_w - 1.0f;
    const float gy = (py / pz) / half_h - 1.0f;
    const float u = (gx + 1.0f) * half_w;
    const float v = (gy + 1.0f) * half_h;
    const float xw = floorf(u), yn = floorf(v);
    Landing l;
    l.fx = u - xw;
    l.fy = v - yn;
    l.ix = cvt_sat_i32(xw);
    l.iy = cvt_sat_i32(yn);
    l.x0in = (unsigned)l.ix < (unsigned)Ws;
    l.x1in = (unsigned)(l.ix + 1) < (unsigned)Ws;
    l.y0in = (unsigned)l.iy < (unsigned)Hs;
    l.y1in = (unsigned)(l.iy + 1) < (unsigned)Hs;
    l.bad = (l.fx != l.fx) | (l.fy != l.fy);
    return l;
}

// in-map bits of the four taps (bit 2*dy+dx); 0 for a non-finite position
__device__ __forceinline__ int landing_mask(const Landing& l) {
    const bool ok = !l.bad;
    return (int)(l.x0in & l.y0in & ok) | ((int)(l.x1in & l.y0in & ok) << 1) | ((int)(l.x0in & l.y1in & ok) << 2) |
           ((int)(l.x1in & l.y1in & ok) << 3);
}

// Four bilinear taps of one sample: element offsets inside one channel plane of the source map and
// the weight of each tap with the zero-padding mask folded in (weight 0 for a tap outside the map;
// a NaN weight is kept so that NaN coordinates still produce NaN like the reference does).
struct Taps {
    int o00, o01, o10, o11;      // (y0,x0) (y0,x1) (y1,x0) (y1,x1); always inside the plane
    float w00, w01, w10, w11;
    int ix, iy;                  // floor of the sample position, limited to [-2, size+1]
    int mask;                    // bit (2*dy+dx) set when that tap lies inside the source map (0 for non-finite positions)
    float fx, fy;
};

__device__ __forceinline__ Taps taps_of(const Landing& l, int Hs, int Ws) {
    Taps t;
    const float fe = 1.0f - l.fx, fs = 1.0f - l.fy;
    // separable zero padding: a column / row outside the map zeroes its factor (NaN factors are kept)
    const float wx0 = (l.x0in | l.bad) ? fe : 0.0f, wx1 = (l.x1in | l.bad) ? l.fx : 0.0f;
    const float wy0 = (l.y0in | l.bad) ? fs : 0.0f, wy1 = (l.y1in | l.bad) ? l.fy : 0.0f;
    t.w00 = wy0 * wx0;
    t.w01 = wy0 * wx1;
    t.w10 = wy1 * wx0;
    t.w11 = wy1 * wx1;
    const int x0 = clamp_i32(l.ix, 0, Ws - 1), x1 = clamp_i32(l.ix + 1, 0, Ws - 1);
    const int y0 = clamp_i32(l.iy, 0, Hs - 1), y1 = clamp_i32(l.iy + 1, 0, Hs - 1);
    t.o00 = __mul24(y0, Ws) + x0;
    t.o01 = __mul24(y0, Ws) + x1;
    t.o10 = __mul24(y1, Ws) + x0;
    t.o11 = __mul24(y1, Ws) + x1;
    t.ix = clamp_i32(l.ix, -2, Ws + 1);       // limited like this for the kernels that index small tables with it
    t.iy = clamp_i32(l.iy, -2, Hs + 1);
    t.mask = landing_mask(l);
    t.fx = l.fx;
    t.fy = l.fy;
    return t;
}

// Projection of (pixel ray, depth) into the source view and the bilinear footprint there.
__device__ __forceinline__ Taps project_taps(const Pose& q, const Ray& r, float depth, int Hs, int Ws) {
    return taps_of(land(q, r, depth, Hs, Ws), Hs, Ws);
}

// The four bilinear weights from the fractional position, with the products formed exactly as taps_of() forms them;
// taps outside the source map (mask bit clear) get weight 0 -- the backward weights, see masked_weight().
__device__ __forceinline__ void weights_from_fraction(float fx, float fy, int mask, float (&w)[4]) {
    const float fe = 1.0f - fx, fs = 1.0f - fy;
    w[0] = (mask & 1) ? fs * fe : 0.0f;
    w[1] = (mask & 2) ? fs * fx : 0.0f;
    w[2] = (mask & 4) ? fy * fe : 0.0f;
    w[3] = (mask & 8) ? fy * fx : 0.0f;
}

// Backward weight of tap k: zero unless the tap is inside the map and the position finite.
__device__ __forceinline__ float masked_weight(const Taps& t, int k) {
    const float w = k == 0 ? t.w00 : k == 1 ? t.w01 : k == 2 ? t.w10 : t.w11;
    return (t.mask >> k) & 1 ? w : 0.0f;
}

// Bilinear sample of one channel plane.
__device__ __forceinline__ float sample(const float* __restrict__ plane, const Taps& t) {
    float acc = plane[t.o00] * t.w00;
    acc = fmaf(plane[t.o01], t.w01, acc);
    acc = fmaf(plane[t.o10], t.w10, acc);
    acc = fmaf(plane[t.o11], t.w11, acc);
    return acc;
}

// Scatter-add of one channel's gradient onto the four taps (skips masked taps).
__device__ __forceinline__ void scatter(float* __restrict__ plane, const Taps& t, float g) {
    const float w0 = masked_weight(t, 0), w1 = masked_weight(t, 1), w2 = masked_weight(t, 2), w3 = masked_weight(t, 3);
    if (w0 != 0.0f) atomicAdd(plane + t.o00, g * w0);
    if (w1 != 0.0f) atomicAdd(plane + t.o01, g * w1);
    if (w2 != 0.0f) atomicAdd(plane + t.o10, g * w2);
    if (w3 != 0.0f) atomicAdd(plane + t.o11, g * w3);
}

// 16-byte load at a 32-bit byte offset from a wave-uniform base: compiles to global_load_dwordx4 v, v_off, s[base]
// (no 64-bit address arithmetic per lane).
__device__ __forceinline__ float4 load16(const float* __restrict__ base, unsigned byte_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}

}  // namespace mvster
