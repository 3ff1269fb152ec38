// Timing harness (MI355X) for the one-launch Winograd layer kernel of the library (csrc/kernels/conv_winograd_f32.h) on the three layer
// shapes of the default encoder at B = 256 (no correctness check: tests/ do that) -- for quick scheduling experiments.
//   * operands as the bench has them: activations non-negative with ~45 % zeros (post-ReLU), each phase its own weight array of its own
//     size (so that the weight traffic is the library's);
//   * every block mapping xcd_cols = 0 (plain), 1, 2, 4, 8 where the layer allows it (conv_winograd_f32.h: wino_block);
//   * built with -DAAE_WINO_STAMPS: one launch per layer with in-kernel shader-clock stamps per wave at every phase boundary, reduced here
//     to "cycles per block in: phase prologue (weights + first stage fill + barrier) | K loop | output transform | final store", the K
//     loop's MFMA-issue efficiency, and the spread over the waves (which XCD a block ran on is recorded too: the mapping's assumption).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I augmentedautoencoder_amd/csrc -o wino_layer_time tools/ubench/wino_layer_time.hip
//   hipcc ... -DAAE_WINO_STAMPS -o wino_layer_stamps tools/ubench/wino_layer_time.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "device_intrinsics.h"
#include "kernels/conv_winograd_f32.h"
#ifndef AAE_WINO_VAR
#define AAE_WINO_VAR 0          // (a tag for A/B builds of the header: printed, nothing else)
#endif

#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

static double frand(unsigned long long& st) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    return (double)(st >> 11) / 9007199254740992.0;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256;
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const int only_cols = argc > 3 ? atoi(argv[3]) : -1;         // >= 0: only the mappings with this xcd_cols (or the layer's largest valid one below it)
    struct Shape { const char* name; int H, Cin, Cout, geom; } shapes[3] = {{"conv2", 64, 128, 256, 0}, {"conv3", 32, 256, 512, 0}, {"conv4", 16, 512, 512, 1}};
    CHECK(hipFuncSetAttribute((const void*)aae::conv_wino_layer_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_layer_smem_bytes<0>()));
    CHECK(hipFuncSetAttribute((const void*)aae::conv_wino_layer_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::wino_layer_smem_bytes<1>()));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    unsigned long long seed = 0x9E3779B97F4A7C15ull;
    static const int points[4] = {9, 12, 12, 16};            // index 2 eh + ew
    for (const Shape& s : shapes) {
        const int Ho = s.H / 2;
        const size_t nx = (size_t)B * s.H * s.H * s.Cin, no = (size_t)B * Ho * Ho * s.Cout;
        std::vector<float> hx(nx);
        for (size_t i = 0; i < nx; ++i) { const double r = frand(seed); hx[i] = r < 0.45 ? 0.f : (float)(1.5 * (r - 0.45)); }
        float *dx, *dout, *dbias, *du[4];
        CHECK(hipMalloc(&dx, nx * 4)); CHECK(hipMalloc(&dout, no * 4)); CHECK(hipMalloc(&dbias, s.Cout * 4));
        CHECK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
        CHECK(hipMemset(dbias, 0, s.Cout * 4));
        for (int i = 0; i < 4; ++i) {
            const size_t nu = (size_t)points[i] * s.Cin * s.Cout;
            std::vector<float> hu(nu);
            for (size_t k = 0; k < nu; ++k) hu[k] = (float)((2.0 * frand(seed) - 1.0) * 0.03);
            CHECK(hipMalloc(&du[i], nu * 4));
            CHECK(hipMemcpy(du[i], hu.data(), nu * 4, hipMemcpyHostToDevice));
        }
        aae::ConvWinoLayerArgs p;
        memset(&p, 0, sizeof(p));
        p.c.x = dx; p.c.U = nullptr; p.c.bias = dbias; p.c.bn_scale = nullptr; p.c.bn_shift = nullptr; p.c.out = dout;
        p.c.B = B; p.c.H = p.c.W = s.H; p.c.Cin = s.Cin; p.c.Cout = s.Cout; p.c.Ho = p.c.Wo = Ho; p.c.eh = p.c.ew = 0; p.c.mode = 0; p.c.relu = 1;
        p.c.blocks_x = p.c.blocks_y = s.geom == 0 ? Ho / 16 : 1;
        p.c.regions = s.geom == 0 ? p.c.blocks_x * p.c.blocks_y * B : (B + 3) / 4;
        for (int i = 0; i < 4; ++i) p.U4[i] = du[i];
        const int nbn = s.Cout / 64;
        const double executed = 2.0 * B * (Ho / 2) * (Ho / 2) * 49.0 * s.Cin * s.Cout;
        static const int cols[5] = {0, 1, 2, 4, 8};
        for (int ci = 0; ci < 5; ++ci) {
            const int xc = cols[ci];
            if (xc > 0 && !aae::wino_xcd_cols_valid(nbn, xc)) continue;
            if (only_cols >= 0 && xc != (only_cols > nbn ? nbn : only_cols)) continue;
            p.c.xcd_cols = xc;
            const unsigned grid = aae::wino_grid_blocks(p.c.regions, nbn, xc);
#ifdef AAE_WINO_STAMPS
            const size_t nstamp = (size_t)grid * 8 * aae::kWinoStampSlots;
            long long* dst;
            CHECK(hipMalloc(&dst, nstamp * 8));
            CHECK(hipMemset(dst, 0, nstamp * 8));
            p.c.stamps = dst;
#endif
            auto launch = [&]() {
                if (s.geom == 0) hipLaunchKernelGGL((aae::conv_wino_layer_kernel<0, false>), dim3(grid), dim3(512), aae::wino_layer_smem_bytes<0>(), 0, p);
                else hipLaunchKernelGGL((aae::conv_wino_layer_kernel<1, false>), dim3(grid), dim3(512), aae::wino_layer_smem_bytes<1>(), 0, p);
            };
            // warm up until the clocks have settled (the host-side set-up above leaves the GPU idle for seconds: the first ~100 ms after it run slow)
            for (int w = 0; w < (ci == 0 || only_cols >= 0 ? 60 : 10); ++w) launch();
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            for (int r = 0; r < reps; ++r) launch();
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            ms /= reps;
            printf("{\"what\": \"wino_layer_time\", \"var\": %d, \"B\": %d, \"layer\": \"%s\", \"xcd_cols\": %d, \"grid\": %u, \"ms\": %.4f, \"mfma_tflops\": %.1f, \"mfma_frac_of_157\": %.3f, \"tf_equivalent\": %.1f",
                   AAE_WINO_VAR, B, s.name, xc, grid, ms, executed / (ms * 1e-3) / 1e12, executed / (ms * 1e-3) / 1e12 / 157.3, executed * 100.0 / 49.0 / (ms * 1e-3) / 1e12);
#ifdef AAE_WINO_STAMPS
            std::vector<long long> hs(nstamp);
            CHECK(hipMemcpy(hs.data(), dst, nstamp * 8, hipMemcpyDeviceToHost));
            // per block: earliest entry / latest exit over its 8 waves; per phase the wave-mean of every segment
            double seg[4][3] = {{0}}, store = 0, span = 0, kstage[7] = {0}, wave_kloop[8] = {0}, units[2][6] = {{0}}, units_w[2][8][6] = {{{0}}};
            long long nb = 0;
            int xcd_ok = 0, xcd_seen = 0;
            for (unsigned b = 0; b < grid; ++b) {
                const long long* blk = hs.data() + (size_t)b * 8 * aae::kWinoStampSlots;
                if (blk[16] == 0) continue;          // (a surplus block of a padded grid)
                ++nb;
                long long t_in = blk[0], t_out = blk[16];
                for (int w = 0; w < 8; ++w) {
                    const long long* st = blk + (size_t)w * aae::kWinoStampSlots;
                    t_in = std::min(t_in, st[0]);
                    t_out = std::max(t_out, st[16]);
                    for (int ph = 0; ph < 4; ++ph) {
                        seg[ph][0] += (double)(st[4 * ph + 1] - st[4 * ph + 0]) / 8;
                        seg[ph][1] += (double)(st[4 * ph + 2] - st[4 * ph + 1]) / 8;
                        seg[ph][2] += (double)(st[4 * ph + 3] - st[4 * ph + 2]) / 8;
                        wave_kloop[w] += (double)(st[4 * ph + 2] - st[4 * ph + 1]);
                    }
                    store += (double)(st[16] - st[15]) / 8;
                    for (int k = 0; k < 7; ++k)
                        if (st[21 + k] && st[20 + k]) kstage[k] += (double)(st[21 + k] - st[20 + k]) / 8;
                    // stages 2, 3 of phase 0: unit 0 .. 3 (start to start), last unit's end = the stamp before the barrier, the barrier
                    for (int q = 0; q < 2; ++q) {
                        const long long* u = st + 28 + 6 * q;
                        const long long next_start = st[20 + 2 + q + 1];          // start of the next stage
                        const double d[6] = {(double)(u[1] - u[0]), (double)(u[2] - u[1]), (double)(u[3] - u[2]), (double)(u[4] - u[3]), (double)(u[5] - u[4]), (double)(next_start - u[5])};
                        for (int k = 0; k < 6; ++k) { units[q][k] += d[k] / 8; units_w[q][w][k] += d[k]; }
                    }
                }
                span += (double)(t_out - t_in);
                ++xcd_seen;
                if ((int)blk[17] == (int)(b & 7)) ++xcd_ok;
            }
            const double inv = nb ? 1.0 / nb : 0;
            // MFMA issue cycles a SIMD needs for a block-phase K loop: its two waves' MFMAs x 64 cycles
            static const int pts[4] = {16, 12, 12, 9};          // phase order of the kernel: 3x3, 3x2, 2x3, 2x2
            printf(", \"blocks\": %lld, \"block_on_xcd_p_mod_8\": %.3f, \"cycles_per_block\": {\"span\": %.0f", nb, xcd_seen ? (double)xcd_ok / xcd_seen : 0.0, span * inv);
            double ksum = 0, psum = 0, osum = 0, ideal_sum = 0;
            for (int ph = 0; ph < 4; ++ph) {
                const double ideal = (double)pts[ph] * (s.Cin / 2) * 64.0;      // per SIMD: points x k-steps x (2 waves x half the points each = the points) x 64
                printf(", \"phase%d\": {\"prologue\": %.0f, \"k_loop\": %.0f, \"k_loop_mfma_cycles\": %.0f, \"k_loop_eff\": %.3f, \"output_transform\": %.0f}", ph, seg[ph][0] * inv,
                       seg[ph][1] * inv, ideal, ideal / (seg[ph][1] * inv), seg[ph][2] * inv);
                psum += seg[ph][0] * inv; ksum += seg[ph][1] * inv; osum += seg[ph][2] * inv; ideal_sum += ideal;
            }
            printf(", \"final_store\": %.0f, \"sum\": {\"prologues\": %.0f, \"k_loops\": %.0f, \"mfma_cycles\": %.0f, \"output_transforms\": %.0f}", store * inv, psum, ksum, ideal_sum, osum);
            printf(", \"phase0_stage_cycles\": [");
            for (int k = 0; k < 7; ++k) printf("%s%.0f", k ? ", " : "", kstage[k] * inv);
            printf("], \"phase0_stage2_units_then_barrier_wait_then_gap\": [");
            for (int k = 0; k < 6; ++k) printf("%s%.0f", k ? ", " : "", units[0][k] * inv);
            printf("], \"phase0_stage3_units_then_barrier_wait_then_gap\": [");
            for (int k = 0; k < 6; ++k) printf("%s%.0f", k ? ", " : "", units[1][k] * inv);
            printf("], \"stage2_by_wave\": [");
            for (int w = 0; w < 8; ++w) { printf("%s[", w ? ", " : ""); for (int k = 0; k < 6; ++k) printf("%s%.0f", k ? ", " : "", units_w[0][w][k] * inv); printf("]"); }
            printf("], \"k_loop_cycles_by_wave\": [");
            for (int w = 0; w < 8; ++w) printf("%s%.0f", w ? ", " : "", wave_kloop[w] * inv);
            printf("]}");
            CHECK(hipFree(dst));
#endif
            printf("}\n");
            fflush(stdout);
        }
        CHECK(hipFree(dx)); CHECK(hipFree(dout)); CHECK(hipFree(dbias));
        for (int i = 0; i < 4; ++i) CHECK(hipFree(du[i]));
    }
    return 0;
}
