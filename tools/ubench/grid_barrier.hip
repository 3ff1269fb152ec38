// Micro-benchmark (MI355X): what does the grid barrier of the persistent per-detection launch cost?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I augmentedautoencoder_amd/csrc -o grid_barrier tools/ubench/grid_barrier.hip && ./grid_barrier
// One resident block per CU (96 KB of LDS each); P barriers of csrc/device_intrinsics.h (two-level counters + per-group gates,
// arrive / wait halves) in one launch.  Variants: bare; with a 16-byte coherent (sc1) store per thread before every arrival
// (what a phase's epilogue leaves in flight); with 16 KB of HBM loads per wave requested between arrive and wait (the prefetch
// slot) -- per barrier = (t(P) - t(0)) / P.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "device_intrinsics.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256, 1) void k_barriers(aae::GridBarrier gb, int phases, float* sink, const float* stream, unsigned stream_bytes) {
    extern __shared__ unsigned char smem[];
    const unsigned G = gridDim.x, blk = blockIdx.x;
    const aae::buffer_rsrc sb = aae::make_buffer(sink, G * 256u * 16u);
    const aae::buffer_rsrc rb = aae::make_buffer(stream, stream_bytes);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = 1; p <= phases; ++p) {
        if (MODE >= 1) aae::coherent_store4(sb, (blk * 256u + threadIdx.x) * 16u, acc);
        aae::grid_barrier_arrive(gb, G, blk, (unsigned)p);
        f32x4 e[16];
        if (MODE >= 2) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                e[u] = aae::buffer_load4(rb, (unsigned)(((((unsigned)p * G + blk) * 4u + (threadIdx.x >> 6)) * 16u + u) * 1024u + (threadIdx.x & 63) * 16u) % (stream_bytes - 16u));
        }
        aae::grid_barrier_wait(gb, G, blk, (unsigned)p);
        if (MODE >= 2) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += e[u];
        }
    }
    if (MODE >= 2 && acc.x == 12345.f) sink[0] = acc.y;
    (void)smem;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long* words; float* sink; float* stream;
    const unsigned stream_bytes = 256u << 20;
    CHECK(hipMalloc(&words, aae::kGridBarrierWords * 8));
    CHECK(hipMalloc(&sink, 1024 * 256 * 16));
    CHECK(hipMalloc(&stream, stream_bytes));
    CHECK(hipMemset(stream, 0, stream_bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    unsigned nonce = 100;
    const int smem = 96 * 1024;
    CHECK(hipFuncSetAttribute((const void*)k_barriers<0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    CHECK(hipFuncSetAttribute((const void*)k_barriers<1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    CHECK(hipFuncSetAttribute((const void*)k_barriers<2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    const char* names[] = {"bare", "sc1 store before every arrival", "sc1 store + 16 KB per wave requested between arrive and wait"};
    for (int mode = 0; mode < 3; ++mode)
        for (int grid : {cus / 4, cus / 2, cus}) {
            float us[2] = {0.f, 0.f};
            const int phases_of[2] = {0, 20};
            for (int v = 0; v < 2; ++v) {
                const int reps = 30;
                for (int r = -3; r < reps; ++r) {
                    if (r == 0) CHECK(hipEventRecord(e0, 0));
                    CHECK(hipMemsetAsync(words, 0, aae::kGridBarrierWords * 8, 0));
                    aae::GridBarrier gb{words, ++nonce};
                    if (mode == 0) hipLaunchKernelGGL(k_barriers<0>, dim3(grid), dim3(256), smem, 0, gb, phases_of[v], sink, stream, stream_bytes);
                    else if (mode == 1) hipLaunchKernelGGL(k_barriers<1>, dim3(grid), dim3(256), smem, 0, gb, phases_of[v], sink, stream, stream_bytes);
                    else hipLaunchKernelGGL(k_barriers<2>, dim3(grid), dim3(256), smem, 0, gb, phases_of[v], sink, stream, stream_bytes);
                }
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms = 0.f;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                us[v] = ms / reps * 1e3f;
            }
            printf("{\"what\": \"grid_barrier\", \"variant\": \"%s\", \"blocks\": %d, \"cus\": %d, \"us_launch_no_barrier\": %.2f, \"us_launch_20_barriers\": %.2f, \"us_per_barrier\": %.3f}\n",
                   names[mode], grid, cus, us[0], us[1], (us[1] - us[0]) / 20.f);
        }
    return 0;
}
