// Device-side helpers shared by the gfx950 kernels of libmvster_hip.so.
//
// Arithmetic follows the reference's CPU PyTorch operator sequence step by step (file:line refer to
// the upstream tree) so that fp32 results agree to rounding noise; the library is compiled with
// -ffp-contract=off and every fused multiply-add below is written explicitly.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvster {

constexpr int kWave = 64;    // CDNA wavefront width
constexpr int kBlock = 256;  // threads per workgroup used by every kernel here (4 waves)

// Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8 shares an XCD, each with its own L2).
// Remap so that every XCD walks one contiguous band of pixel tiles: the two reference rows that
// touch the same source row are then served by the same L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_contiguous_tile(int block, int nblocks) {
    const int per_xcd = (nblocks + 7) >> 3;
    return (block & 7) * per_xcd + (block >> 3);
}

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

// Four bilinear taps of one sample: element offsets inside one channel plane of the source map and
// the weight of each tap with the zero-padding mask folded in (weight 0 for a tap outside the map;
// a NaN weight is kept so that NaN coordinates still produce NaN like the reference does).
struct Taps {
    int o00, o01, o10, o11;      // (y0,x0) (y0,x1) (y1,x0) (y1,x1); always inside the plane
    float w00, w01, w10, w11;
    int ix, iy;                  // floor of the sample position (limited to [-2, size+1]); tap (dy,dx) = (iy+dy, ix+dx)
    int mask;                    // bit (2*dy+dx) set when that tap lies inside the source map
    float fx, fy;                // fractional sample position: w00 = (1-fy)(1-fx), w01 = (1-fy) fx, w10 = fy (1-fx), w11 = fy fx
};

// The four bilinear weights from the fractional position, with the products formed exactly as project_taps() forms them;
// taps outside the source map (mask bit clear) get weight 0 -- the backward weights, see masked_weight().
__device__ __forceinline__ void weights_from_fraction(float fx, float fy, int mask, float (&w)[4]) {
    const float fe = 1.0f - fx, fs = 1.0f - fy;
    w[0] = (mask & 1) ? fs * fe : 0.0f;
    w[1] = (mask & 2) ? fs * fx : 0.0f;
    w[2] = (mask & 4) ? fy * fe : 0.0f;
    w[3] = (mask & 8) ? fy * fx : 0.0f;
}

// Backward weight of tap k: zero unless the tap is inside the map.  (A NaN sample position keeps NaN weights in
// the forward pass so that NaN propagates like in the reference; its backward scatters nothing, as grid_sample's.)
__device__ __forceinline__ float masked_weight(const Taps& t, int k) {
    const float w = k == 0 ? t.w00 : k == 1 ? t.w01 : k == 2 ? t.w10 : t.w11;
    return (t.mask >> k) & 1 ? w : 0.0f;
}

// Projection of (pixel ray, depth) into the source view and the bilinear footprint there.
//   models/mvs4net_utils.py:43-52  p = ray*depth + trans (separately rounded), z==0 -> 1e-9,
//                                  uv = p.xy / z, grid = uv / ((size-1)/2) - 1
//   F.grid_sample (:59), CPU kernel: pix = (grid + 1) * ((size-1)/2); w = pix - floor(pix); e = 1 - w;
//                                  taps outside [0,size) contribute zero.
__device__ __forceinline__ Taps project_taps(const Pose& q, const Ray& r, float depth, int Hs, int Ws,
                                             float half_w, float half_h) {
    float px = r.x * depth + q.t0;
    float py = r.y * depth + q.t1;
    float pz = r.z * depth + q.t2;
    if (pz == 0.0f) pz = 1e-9f;
    const float u = px / pz;
    const float v = py / pz;
    const float gx = u / half_w - 1.0f;
    const float gy = v / half_h - 1.0f;
    const float ix = (gx + 1.0f) * half_w;
    const float iy = (gy + 1.0f) * half_h;

    const float xw = floorf(ix), yn = floorf(iy);
    const float fw = ix - xw, fe = 1.0f - fw;
    const float fn = iy - yn, fs = 1.0f - fn;
    const float wmax = (float)(Ws - 1), hmax = (float)(Hs - 1);
    // comparisons on the integer-valued floats (no float->int overflow); NaN compares false
    const bool x0in = (xw >= 0.0f) & (xw <= wmax);
    const bool x1in = (xw >= -1.0f) & (xw <= wmax - 1.0f);
    const bool y0in = (yn >= 0.0f) & (yn <= hmax);
    const bool y1in = (yn >= -1.0f) & (yn <= hmax - 1.0f);
    const int x0 = (int)fminf(fmaxf(xw, 0.0f), wmax);
    const int x1 = (int)fminf(fmaxf(xw + 1.0f, 0.0f), wmax);
    const int y0 = (int)fminf(fmaxf(yn, 0.0f), hmax);
    const int y1 = (int)fminf(fmaxf(yn + 1.0f, 0.0f), hmax);

    Taps t;
    t.o00 = y0 * Ws + x0;
    t.o01 = y0 * Ws + x1;
    t.o10 = y1 * Ws + x0;
    t.o11 = y1 * Ws + x1;
    const float nw = fs * fe, ne = fs * fw, sw = fn * fe, se = fn * fw;
    t.w00 = ((x0in & y0in) | (nw != nw)) ? nw : 0.0f;
    t.w01 = ((x1in & y0in) | (ne != ne)) ? ne : 0.0f;
    t.w10 = ((x0in & y1in) | (sw != sw)) ? sw : 0.0f;
    t.w11 = ((x1in & y1in) | (se != se)) ? se : 0.0f;
    t.ix = (int)fminf(fmaxf(xw, -2.0f), wmax + 1.0f);
    t.iy = (int)fminf(fmaxf(yn, -2.0f), hmax + 1.0f);
    t.mask = (int)(x0in & y0in) | ((int)(x1in & y0in) << 1) | ((int)(x0in & y1in) << 2) | ((int)(x1in & y1in) << 3);
    t.fx = fw;
    t.fy = fn;
    return t;
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

// Reductions over the LANES consecutive lanes (LANES a power of two <= 64) that share one pixel.
template <int LANES>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int m = 1; m < LANES; m <<= 1) v = fmaxf(v, __shfl_xor(v, m, kWave));
    return v;
}

template <int LANES>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int m = 1; m < LANES; m <<= 1) v += __shfl_xor(v, m, kWave);
    return v;
}

// The same reductions over N lanes that are STRIDE apart (lane bits log2(STRIDE) .. log2(STRIDE*N)-1): used when a pixel's
// lanes are laid out [hypothesis][channel part] and only one of the two indices is to be reduced.
template <int N, int STRIDE>
__device__ __forceinline__ float strided_max(float v) {
#pragma unroll
    for (int m = STRIDE; m < N * STRIDE; m <<= 1) v = fmaxf(v, __shfl_xor(v, m, kWave));
    return v;
}

template <int N, int STRIDE>
__device__ __forceinline__ float strided_sum(float v) {
#pragma unroll
    for (int m = STRIDE; m < N * STRIDE; m <<= 1) v += __shfl_xor(v, m, kWave);
    return v;
}

}  // namespace mvster
