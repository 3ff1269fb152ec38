// Micro-benchmark (MI355X): what the kernel-argument fetch costs a short kernel, and whether gfx950's kernarg preload
// (SGPRs filled by the dispatcher: scalar kernel parameters only, -mllvm -amdgpu-kernarg-preload-count=N) removes it.
// Two kernels do the same trivial work from 16 integer arguments: one takes them in a struct by value (what the library's
// kernels do), one as 16 scalar parameters (eligible for preload).  Back-to-back launch period and the in-kernel time from
// wave start to "arguments available" (s_memtime before / after the first use).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 -o kernarg_preload kernarg_preload.hip && ./kernarg_preload
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Args { int v[16]; int* out; long long* ticks; };
__global__ void k_struct(const Args a) {
    const long long t0 = __builtin_readcyclecounter();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a.v[i] * (i + 1);
    a.out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) a.ticks[0] = t1 - t0;
}
__global__ void k_scalar(int v0, int v1, int v2, int v3, int v4, int v5, int v6, int v7, int v8, int v9, int v10, int v11, int v12, int v13,
                         int v14, int v15, int* out, long long* ticks) {
    const long long t0 = __builtin_readcyclecounter();
    const int s = v0 + 2 * v1 + 3 * v2 + 4 * v3 + 5 * v4 + 6 * v5 + 7 * v6 + 8 * v7 + 9 * v8 + 10 * v9 + 11 * v10 + 12 * v11 + 13 * v12 + 14 * v13 +
                  15 * v14 + 16 * v15;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
int main() {
    int* out; long long* ticks;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&ticks, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    Args a; for (int i = 0; i < 16; ++i) a.v[i] = i; a.out = out; a.ticks = ticks;
    for (int variant = 0; variant < 2; ++variant)
        for (int rep = 0; rep < 3; ++rep) {
            const int n = 2000;
            auto launch = [&](int j) {
                a.v[0] = j;                                   // fresh argument values every launch
                if (variant == 0) hipLaunchKernelGGL(k_struct, dim3(256), dim3(256), 0, 0, a);
                else hipLaunchKernelGGL(k_scalar, dim3(256), dim3(256), 0, 0, j, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, out, ticks);
            };
            for (int j = 0; j < 100; ++j) launch(j);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            for (int j = 0; j < n; ++j) launch(j);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
            printf("{\"what\": \"kernarg\", \"args\": \"%s\", \"rep\": %d, \"launch_period_us\": %.3f, \"ticks_first_use_to_store\": %lld}\n",
                   variant ? "16 scalars (preloaded)" : "struct by value", rep, ms / n * 1e3, t);
        }
    return 0;
}
