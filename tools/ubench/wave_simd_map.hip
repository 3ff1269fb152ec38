// Which waves of a 512-thread block share a SIMD?  (MI355X; prints HW_ID fields per wave of block 0.)
//   hipcc --offload-arch=gfx950 -O3 -o wave_simd_map tools/ubench/wave_simd_map.hip && ./wave_simd_map
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(512) void k(unsigned* out) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID, all 32 bits
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
}

int main() {
    unsigned* d; unsigned h[64];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(8), dim3(512), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int b = 0; b < 2; ++b)
        for (int w = 0; w < 8; ++w) {
            const unsigned v = h[b * 8 + w];
            printf("{\"what\": \"wave_simd_map\", \"block\": %d, \"wave\": %d, \"wave_id\": %u, \"simd_id\": %u, \"pipe_id\": %u, \"cu_id\": %u, \"sh_id\": %u, \"se_id\": %u}\n",
                   b, w, v & 15u, (v >> 4) & 3u, (v >> 6) & 3u, (v >> 8) & 15u, (v >> 12) & 1u, (v >> 13) & 7u);
        }
    return 0;
}
