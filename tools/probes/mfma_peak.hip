// Probe: what does the fp32 matrix pipe sustain on this box?  v_mfma_f32_32x32x2_f32 in a register-only loop, 1..4 waves per
// SIMD, 2 or 4 independent accumulators per wave.  Nominal peak: 256 CUs x 4 SIMDs x 2.4 GHz x 64 flop/cycle = 157 TFLOP/s.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_peak.hip -o tools/probes/build/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) spin(float* out, int iters, float a0) {
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k)
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.0f;
    float a = a0 + threadIdx.x, b = a0 * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.0f;
    for (int k = 0; k < NACC; ++k)
        for (int i = 0; i < 16; ++i) s += acc[k][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 4096 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    for (int nacc = 2; nacc <= 4; nacc += 2)
        for (int wg_per_cu = 1; wg_per_cu <= 4; ++wg_per_cu) {
            const int blocks = 256 * wg_per_cu;
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (nacc == 2) hipLaunchKernelGGL(spin<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
                else hipLaunchKernelGGL(spin<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double flop = (double)blocks * 4 * iters * 8 * nacc * (32.0 * 32 * 2 * 2);
            printf("%d accumulators, %d waves per SIMD: %.3f ms, %.1f TFLOP/s (%.0f %% of 157.3)\n", nacc, wg_per_cu, ms, flop / ms / 1e9,
                   100.0 * flop / ms / 1e9 / 157.3);
        }
    return 0;
}
