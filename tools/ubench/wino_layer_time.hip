// Timing harness (MI355X) for the one-launch Winograd layer kernel of the library (csrc/kernels/conv_winograd_f32.h) on the three layer
// shapes of the default encoder at B = 256, random operands (no correctness check: tests/ do that) -- for quick scheduling experiments.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I augmentedautoencoder_amd/csrc -o wino_layer_time tools/ubench/wino_layer_time.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "device_intrinsics.h"
#include "kernels/conv_winograd_f32.h"

#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

int main() {
    const int B = 256;
    struct Shape { const char* name; int H, Cin, Cout, geom; } shapes[3] = {{"conv2", 64, 128, 256, 0}, {"conv3", 32, 256, 512, 0}, {"conv4", 16, 512, 512, 1}};
    CHECK(hipFuncSetAttribute((const void*)aae::conv_wino_layer_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_layer_smem_bytes<0>()));
    CHECK(hipFuncSetAttribute((const void*)aae::conv_wino_layer_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_layer_smem_bytes<1>()));
    CHECK(hipFuncSetAttribute((const void*)aae::conv_wino_layer_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_layer_smem_bytes<0>()));
    CHECK(hipFuncSetAttribute((const void*)aae::conv_wino_layer_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_layer_smem_bytes<1>()));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wide = 0; wide < 2; ++wide) {
    double total = 0;
    for (const Shape& s : shapes) {
        const int Ho = s.H / 2;
        const size_t nx = (size_t)B * s.H * s.H * s.Cin, nu = (size_t)16 * s.Cin * s.Cout, no = (size_t)B * Ho * Ho * s.Cout;
        std::vector<float> hx(nx), hu(nu);
        for (size_t i = 0; i < nx; ++i) hx[i] = (float)(rand() % 1000) * 1e-3f;
        for (size_t i = 0; i < nu; ++i) hu[i] = (float)(rand() % 1000 - 500) * 1e-4f;
        float *dx, *du, *dout, *dbias;
        CHECK(hipMalloc(&dx, nx * 4)); CHECK(hipMalloc(&du, nu * 4)); CHECK(hipMalloc(&dout, no * 4)); CHECK(hipMalloc(&dbias, s.Cout * 4));
        CHECK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(du, hu.data(), nu * 4, hipMemcpyHostToDevice));
        CHECK(hipMemset(dbias, 0, s.Cout * 4));
        aae::ConvWinoLayerArgs p;
        p.c.x = dx; p.c.U = nullptr; p.c.bias = dbias; p.c.bn_scale = nullptr; p.c.bn_shift = nullptr; p.c.out = dout;
        p.c.B = B; p.c.H = p.c.W = s.H; p.c.Cin = s.Cin; p.c.Cout = s.Cout; p.c.Ho = p.c.Wo = Ho; p.c.eh = p.c.ew = 0; p.c.mode = 0; p.c.relu = 1;
        p.c.blocks_x = p.c.blocks_y = s.geom == 0 ? Ho / 16 : 1;
        for (int i = 0; i < 4; ++i) p.U4[i] = du;          // (every phase reads a prefix of the 16-point array)
        const unsigned grid = (unsigned)(s.Cout / 64) * (s.geom == 0 ? (unsigned)(p.c.blocks_x * p.c.blocks_y * B) : (unsigned)(B / 4));
        auto launch = [&]() {
            if (wide) {
                if (s.geom == 0) hipLaunchKernelGGL((aae::conv_wino_layer_kernel<0, true>), dim3(grid), dim3(256), aae::wino_layer_smem_bytes<0>(), 0, p);
                else hipLaunchKernelGGL((aae::conv_wino_layer_kernel<1, true>), dim3(grid), dim3(256), aae::wino_layer_smem_bytes<1>(), 0, p);
            } else {
                if (s.geom == 0) hipLaunchKernelGGL((aae::conv_wino_layer_kernel<0, false>), dim3(grid), dim3(512), aae::wino_layer_smem_bytes<0>(), 0, p);
                else hipLaunchKernelGGL((aae::conv_wino_layer_kernel<1, false>), dim3(grid), dim3(512), aae::wino_layer_smem_bytes<1>(), 0, p);
            }
        };
        for (int w = 0; w < 3; ++w) launch();
        CHECK(hipDeviceSynchronize());
        const int reps = 10;
        CHECK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const double executed = 2.0 * B * (Ho / 2) * (Ho / 2) * 49.0 * s.Cin * s.Cout;
        printf("{\"wide\": %d, \"layer\": \"%s\", \"ms\": %.4f, \"mfma_tflops\": %.1f, \"mfma_frac_of_157\": %.3f, \"tf_equivalent\": %.1f}\n", wide, s.name, ms, executed / (ms * 1e-3) / 1e12,
               executed / (ms * 1e-3) / 1e12 / 157.3, executed * 100.0 / 49.0 / (ms * 1e-3) / 1e12);
        total += ms;
        CHECK(hipFree(dx)); CHECK(hipFree(du)); CHECK(hipFree(dout)); CHECK(hipFree(dbias));
    }
    printf("{\"wide\": %d, \"conv2_to_conv4_ms\": %.4f}\n", wide, total);
    }
    return 0;
}
