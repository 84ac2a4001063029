// Probe: what bounds a bilinear gather of 32-byte taps (channels-last, C = 8) from an L2-resident map on gfx950 -- the
// number of wave-level load instructions, the bytes they return, or the distinct cache lines one instruction touches?
//
//   A  lane = sample:            8 x global_load_dwordx4 per lane (4 taps x 2 halves), 64 samples per wave
//   B  4 lanes = one sample:     lane (kx, kc) loads channel half kc of tap column kx in both rows: 2 loads per lane,
//                                16 samples per wave; adjacent lanes read adjacent 16-byte pieces of one 64-byte run
//   C  2 lanes = one sample:     lane kc loads its channel half of all four taps: 4 loads per lane
//   D  as A but every lane quad shares one position (one fourth of the distinct addresses)
// Every variant moves the same 128 bytes per sample into registers and reduces them to one float per sample.
// Positions: pixel + smooth disparity + per-pixel random offset of up to +-R px + d * step px (hypothesis d of 4).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/gather_probe.hip -o tools/probes/build/gather_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int H = 1184, W = 1600, C = 8, D = 4;

__device__ __forceinline__ float4 ld(const float* base, unsigned off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off);
}
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

__device__ __forceinline__ void position(const float* __restrict__ jitter, int p, int d, float step, float& u, float& v) {
    const int y = p / W, x = p - y * W;
    const float j = jitter[p];
    u = (float)x - 40.0f + j + d * step;
    v = (float)y + 3.0f - 0.33f * (j + d * step);
}

__device__ __forceinline__ void taps(float u, float v, unsigned (&o)[4], float (&w)[4]) {
    const float xf = floorf(u), yf = floorf(v);
    const float fx = u - xf, fy = v - yf;
    const int x0 = min(max((int)xf, 0), W - 2), y0 = min(max((int)yf, 0), H - 2);
    w[0] = (1 - fy) * (1 - fx); w[1] = (1 - fy) * fx; w[2] = fy * (1 - fx); w[3] = fy * fx;
    o[0] = (unsigned)((y0 * W + x0) * C * 4);
    o[1] = o[0] + C * 4;
    o[2] = o[0] + W * C * 4;
    o[3] = o[2] + C * 4;
}

template <bool SHARE>
__global__ void __launch_bounds__(256) variant_a(const float* __restrict__ map, const float* __restrict__ jitter, float* __restrict__ out,
                                                 int npix, float step) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int p = min(t / D, npix - 1), d = SHARE ? 0 : t % D;
    float u, v;
    position(jitter, p, d, step, u, v);
    unsigned o[4];
    float w[4];
    taps(u, v, o, w);
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 a = ld(map, o[k]), b = ld(map, o[k] + 16);
        acc += w[k] * (a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w);
    }
    if (t < npix * D) out[t] = acc;
}

__global__ void __launch_bounds__(256) variant_b(const float* __restrict__ map, const float* __restrict__ jitter, float* __restrict__ out,
                                                 int npix, float step) {
    // 16 lanes per pixel: [d][kx][kc]
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int kc = t & 1, kx = (t >> 1) & 1, d = (t >> 2) & 3, p = min(t >> 4, npix - 1);
    float u, v;
    position(jitter, p, d, step, u, v);
    unsigned o[4];
    float w[4];
    taps(u, v, o, w);
    const float4 a = ld(map, o[kx] + 16 * kc), b = ld(map, o[2 + kx] + 16 * kc);
    float acc = w[kx] * (a.x + a.y + a.z + a.w) + w[2 + kx] * (b.x + b.y + b.z + b.w);
    acc += dpp<0xB1>(acc);
    acc += dpp<0x4E>(acc);
    if ((t & 3) == 0 && (t >> 4) < npix) out[(t >> 4) * D + d] = acc;
}

__global__ void __launch_bounds__(256) variant_c(const float* __restrict__ map, const float* __restrict__ jitter, float* __restrict__ out,
                                                 int npix, float step) {
    // 8 lanes per pixel: [d][kc]
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int kc = t & 1, d = (t >> 1) & 3, p = min(t >> 3, npix - 1);
    float u, v;
    position(jitter, p, d, step, u, v);
    unsigned o[4];
    float w[4];
    taps(u, v, o, w);
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float4 a = ld(map, o[k] + 16 * kc);
        acc += w[k] * (a.x + a.y + a.z + a.w);
    }
    acc += dpp<0xB1>(acc);
    if ((t & 1) == 0 && (t >> 3) < npix) out[(t >> 3) * D + d] = acc;
}

int main(int argc, char** argv) {
    const int npix = H * W;
    const int views = 16;                      // 4 batch items x 4 views: 16 passes over different maps
    float *maps, *jit, *out;
    CHECK(hipMalloc(&maps, (size_t)views * npix * C * 4));
    CHECK(hipMalloc(&jit, (size_t)npix * 4));
    CHECK(hipMalloc(&out, (size_t)npix * D * 4));
    CHECK(hipMemset(maps, 0, (size_t)views * npix * C * 4));
    std::vector<float> hj(npix);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const float steps[] = {0.5f, 2.0f};
    const float radii[] = {0.0f, 28.0f};
    for (float R : radii)
        for (float step : steps) {
            srand(1);
            for (int i = 0; i < npix; ++i) hj[i] = R * (2.0f * rand() / (float)RAND_MAX - 1.0f);
            CHECK(hipMemcpy(jit, hj.data(), (size_t)npix * 4, hipMemcpyHostToDevice));
            for (int var = 0; var < 4; ++var) {
                const int lanes_per_px = var == 0 || var == 3 ? D : var == 1 ? 16 : 8;
                const int blocks = (int)(((long long)npix * lanes_per_px + 255) / 256);
                float ms = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CHECK(hipEventRecord(e0));
                    for (int s = 0; s < views; ++s) {
                        const float* m = maps + (size_t)s * npix * C;
                        if (var == 0) hipLaunchKernelGGL(variant_a<false>, dim3(blocks), dim3(256), 0, 0, m, jit, out, npix, step);
                        if (var == 1) hipLaunchKernelGGL(variant_b, dim3(blocks), dim3(256), 0, 0, m, jit, out, npix, step);
                        if (var == 2) hipLaunchKernelGGL(variant_c, dim3(blocks), dim3(256), 0, 0, m, jit, out, npix, step);
                        if (var == 3) hipLaunchKernelGGL(variant_a<true>, dim3(blocks), dim3(256), 0, 0, m, jit, out, npix, step);
                    }
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                }
                const double samples = (double)views * npix * D;
                printf("jitter +-%2.0f px, step %.1f px  variant %c: %7.3f ms for %d maps  (%.1f G samples/s, %.1f TB/s into registers)\n", R, step,
                       "ABCD"[var], ms, views, samples / ms / 1e6, samples * 128 / ms / 1e9);
                fflush(stdout);
            }
        }
    return 0;
}
