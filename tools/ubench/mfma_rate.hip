// Micro-benchmark (MI355X): fp32 MFMA rate of short kernels.  One wave per SIMD on every CU runs N back-to-back
// v_mfma_f32_32x32x2_f32 on four independent accumulators; launched back-to-back (steady) and with host-side idle
// gaps in between (the per-detection regime), to see what clock a 10-us kernel actually gets.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_mfma(float* out, int n, long long* cycles) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    const float x = (float)threadIdx.x, y = 1.0f;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int n : {16, 64, 256, 1024, 4096}) {
        for (int gap_us : {0, 200, 2000}) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_mfma, dim3(256), dim3(256), 0, 0, out, n, cyc);
            hipDeviceSynchronize();
            double tot = 0; const int reps = 20;
            for (int i = 0; i < reps; ++i) {
                if (gap_us) usleep(gap_us);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k_mfma, dim3(256), dim3(256), 0, 0, out, n, cyc);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms;
            }
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double us = tot / reps * 1e3, flop = 256.0 * 4 * 4 * n * 4096.0;
            printf("{\"what\": \"mfma_rate\", \"mfma_per_wave\": %d, \"idle_gap_us\": %d, \"us_per_launch\": %.2f, \"tflops\": %.1f, \"counter_ticks\": %lld, \"ideal_us_at_2.4GHz\": %.2f}\n",
                   4 * n, gap_us, us, flop / us / 1e6, c, 4.0 * n * 64 / 2400.0);
        }
    }
    return 0;
}
