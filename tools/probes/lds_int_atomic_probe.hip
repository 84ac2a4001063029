// Probe: throughput of LDS atomics by operand type on gfx950 (ds_add_u32 / ds_add_u64 / ds_add_f32 / ds_add_f64 and the
// returning ds_add_rtn_u32), linear and random addresses.  Question behind it: can a workgroup accumulate scattered
// contributions in an LDS tile with INTEGER (fixed-point) atomics where ds_add_f32 is too slow (lds_atomic_probe.hip)?
// Build: hipcc -O3 --offload-arch=gfx950 lds_int_atomic_probe.hip -o build/lds_int_atomic_probe   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int kBytes = 32768;

template <typename T> struct One { static __device__ T v() { return (T)1; } };

// MODE 0: lane-linear   2: per-lane random   3: random groups of 8 consecutive elements (8 lanes each)
template <typename T, int MODE, bool RTN>
__global__ void __launch_bounds__(1024) lds_add_kernel(T* __restrict__ out, int iters) {
    constexpr int kElems = kBytes / sizeof(T);
    __shared__ T tile[kElems];
    for (int i = threadIdx.x; i < kElems; i += blockDim.x) tile[i] = (T)0;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    if (MODE == 3) s = (threadIdx.x / 8) * 2654435761u + blockIdx.x * 40503u + 12345u;
    T keep = (T)0;
    for (int it = 0; it < iters; ++it) {
        int a;
        if (MODE == 0) a = (threadIdx.x + it * 64) & (kElems - 1);
        else if (MODE == 2) { s = s * 1664525u + 1013904223u; a = (s >> 8) & (kElems - 1); }
        else { s = s * 1664525u + 1013904223u; a = (((s >> 8) & (kElems / 8 - 1)) * 8) + (threadIdx.x & 7); }
        if (RTN) keep += atomicAdd(&tile[a], One<T>::v());
        else atomicAdd(&tile[a], One<T>::v());
    }
    __syncthreads();
    T acc = keep == (T)12345 ? (T)1 : (T)0;
    for (int i = threadIdx.x; i < kElems; i += blockDim.x) acc += tile[i];
    if (acc == (T)-1) out[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double tot = 0.0;
        for (int i = 0; i < kElems; ++i) tot += (double)tile[i];
        out[1] = (T)tot;
    }
}

template <typename T, int MODE, bool RTN>
int run(const char* name, void* outv, int block, int grid, int iters) {
    T* out = (T*)outv;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((lds_add_kernel<T, MODE, RTN>), dim3(grid), dim3(block), 0, 0, out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((lds_add_kernel<T, MODE, RTN>), dim3(grid), dim3(block), 0, 0, out, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    T host[2];
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost));
    const double adds = (double)grid * block * iters;
    printf("%-34s block %4d grid %5d: %8.3f ms  %9.2f G lane-adds/s  (%6.2f lanes/clk/CU at 2.4 GHz, 256 CUs)  sum %s\n", name,
           block, grid, ms, adds / ms * 1e-6, adds / (ms * 1e-3) / 256.0 / 2.4e9,
           (double)host[1] == (double)block * iters ? "exact" : "WRONG");
    fflush(stdout);
    return 0;
}

int main() {
    void* out;
    CHECK(hipMalloc(&out, 64));
    const int iters = 2048;
    for (int block : {256, 1024}) {
        const int grid = 256 * (2048 / block);
#define ALL(T, nm)                                                                      \
        if (run<T, 0, false>(nm " linear", out, block, grid, iters)) return 1;          \
        if (run<T, 2, false>(nm " random", out, block, grid, iters)) return 1;          \
        if (run<T, 3, false>(nm " random 8-groups", out, block, grid, iters)) return 1;
        ALL(unsigned, "u32")
        ALL(unsigned long long, "u64")
        ALL(float, "f32")
        ALL(double, "f64")
        if (run<unsigned, 2, true>("u32 rtn random", out, block, grid, iters)) return 1;
        if (run<unsigned long long, 2, true>("u64 rtn random", out, block, grid, iters)) return 1;
    }
    CHECK(hipFree(out));
    return 0;
}
