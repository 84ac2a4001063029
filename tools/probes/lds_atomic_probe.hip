// Probe: throughput of LDS float atomics (ds_add_f32) on gfx950 under different address patterns and occupancies.
// Build: hipcc -O3 --offload-arch=gfx950 lds_atomic_probe.hip -o build/lds_atomic_probe   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int kTile = 8192;       // floats (32 KB)

// MODE 0: lane-linear addresses (conflict free)   1: stride 8 floats   2: per-lane LCG random within the tile
// MODE 3: random 32-byte groups: 8 consecutive lanes add 8 consecutive floats at a random 8-float-aligned offset
// ATOMIC 1: ds_add_f32   0: plain read-modify-write (racy, only as a speed reference)
template <int MODE, int ATOMIC>
__global__ void __launch_bounds__(1024) lds_add_kernel(float* __restrict__ out, int iters) {
    __shared__ float tile[kTile];
    for (int i = threadIdx.x; i < kTile; i += blockDim.x) tile[i] = 0.0f;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    if (MODE == 3) s = (threadIdx.x / 8) * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int it = 0; it < iters; ++it) {
        int a;
        if (MODE == 0) a = (threadIdx.x + it * 64) & (kTile - 1);
        else if (MODE == 1) a = (threadIdx.x * 8 + it) & (kTile - 1);
        else if (MODE == 2) { s = s * 1664525u + 1013904223u; a = (s >> 8) & (kTile - 1); }
        else { s = s * 1664525u + 1013904223u; a = (((s >> 8) & (kTile / 8 - 1)) * 8) + (threadIdx.x & 7); }
        if (ATOMIC) atomicAdd(&tile[a], 1.0f);
        else tile[a] += 1.0f;
    }
    __syncthreads();
    float acc = 0.0f;
    for (int i = threadIdx.x; i < kTile; i += blockDim.x) acc += tile[i];
    if (acc == -1.0f) out[0] = acc;          // keep the loop alive
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float tot = 0.0f;
        for (int i = 0; i < kTile; ++i) tot += tile[i];
        out[1] = tot;                        // ATOMIC: must equal blockDim.x * iters
    }
}

template <int MODE, int ATOMIC>
int run(const char* name, float* out, int block, int grid, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((lds_add_kernel<MODE, ATOMIC>), dim3(grid), dim3(block), 0, 0, out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((lds_add_kernel<MODE, ATOMIC>), dim3(grid), dim3(block), 0, 0, out, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.0f, host[2];
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost));
    const double adds = (double)grid * block * iters;
    printf("%-28s block %4d grid %5d: %8.3f ms  %8.2f G lane-adds/s  (%.2f lanes/clk/CU at 2.4 GHz, 256 CUs)  sum %s\n", name,
           block, grid, ms, adds / ms * 1e-6, adds / (ms * 1e-3) / 256.0 / 2.4e9,
           ATOMIC ? (host[1] == (float)block * iters ? "exact" : "WRONG") : "n/a");
    fflush(stdout);
    return 0;
}

int main() {
    float* out;
    CHECK(hipMalloc(&out, 64));
    const int iters = 4096;
    for (int block : {256, 512, 1024}) {
        const int grid = 256 * (2048 / block);
        if (run<0, 1>("atomic linear", out, block, grid, iters)) return 1;
        if (run<1, 1>("atomic stride-8", out, block, grid, iters)) return 1;
        if (run<2, 1>("atomic random", out, block, grid, iters)) return 1;
        if (run<3, 1>("atomic random 8-float groups", out, block, grid, iters)) return 1;
        if (run<0, 0>("plain rmw linear", out, block, grid, iters)) return 1;
        if (run<2, 0>("plain rmw random", out, block, grid, iters)) return 1;
    }
    CHECK(hipFree(out));
    return 0;
}
