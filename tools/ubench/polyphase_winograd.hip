// Micro-benchmark (MI355X): fewer multiplies in exact-ish fp32 -- is it worth building?  
//
// The 5 x 5 stride-2 'SAME' convolutions of /root/reference/auto_pose/ae/encoder.py:41-50 split into four stride-1 "phase"
// convolutions over the even / odd sub-images (taps {0,2,4} x {0,2,4} = 3 x 3, 3 x 2, 2 x 3, 2 x 2).  This program takes
// conv3's 3 x 3 phase alone -- sub-image [B, 18, 18, 256] (16 + halo), 256 -> 512 channels, outputs [B, 16, 16, 512], B = 256
// -- and evaluates it as Winograd F(2 x 2, 3 x 3): 16 element-wise products per 2 x 2 output tile instead of 36, i.e. 16
// GEMMs  [tiles x 256] x [256 x 512]  on the fp32 matrix cores (the product library's own 128 x 128 LDS-DMA implicit GEMM as a
// 1 x 1 convolution), with the input transform B^T d B and the output transform A^T m A (+ bias + ReLU) as their own launches:
//     nominal work of the phase (what the direct kernel multiplies): 2 * B*256 * 9*256 * 512 = 154.6 GFLOP
//     Winograd multiplies:                                            2 * 16 * B*64 * 256 * 512 =  68.7 GFLOP  (2.25 x fewer)
// Reported: time per stage, TF-EQUIVALENT = nominal work / total time (the direct conv3 kernel of the bench runs at 142 TF), and
// the largest error against a float64 direct evaluation (and the same for a direct fp32 fma chain, for scale).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I augmentedautoencoder_amd/csrc -o polyphase_winograd tools/ubench/polyphase_winograd.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "device_intrinsics.h"
#include "kernels/tile_f32.h"
#include "kernels/conv_igemm_f32.h"
#include "winograd_fused.h"

#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

constexpr int kB = 256, kS = 18, kO = 16, kC = 256, kN = 512, kTilesPerImage = 64, kT = kB * kTilesPerImage;

// V[p][tile][c] = (B^T d B)[p], d = the 4 x 4 patch of tile (b, ty, tx), channel c.  One thread per (tile, 4 channels).
__global__ __launch_bounds__(256) void input_transform_kernel(const float* __restrict__ s, float* __restrict__ V) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;            // (tile, c4)
    const int c4 = (int)(gid % (kC / 4));
    const long long tile = gid / (kC / 4);
    if (tile >= kT) return;
    const int b = (int)(tile / kTilesPerImage), t = (int)(tile % kTilesPerImage), ty = t / 8, tx = t % 8;
    f32x4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            d[i][j] = *reinterpret_cast<const f32x4*>(s + (((long long)b * kS + 2 * ty + i) * kS + 2 * tx + j) * kC + c4 * 4);
    f32x4 r[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                 // rows: B^T d
        r[0][j] = d[0][j] - d[2][j];
        r[1][j] = d[1][j] + d[2][j];
        r[2][j] = d[2][j] - d[1][j];
        r[3][j] = d[1][j] - d[3][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                 // columns: (B^T d) B
        const f32x4 v0 = r[i][0] - r[i][2], v1 = r[i][1] + r[i][2], v2 = r[i][2] - r[i][1], v3 = r[i][1] - r[i][3];
        *reinterpret_cast<f32x4*>(V + ((long long)(i * 4 + 0) * kT + tile) * kC + c4 * 4) = v0;
        *reinterpret_cast<f32x4*>(V + ((long long)(i * 4 + 1) * kT + tile) * kC + c4 * 4) = v1;
        *reinterpret_cast<f32x4*>(V + ((long long)(i * 4 + 2) * kT + tile) * kC + c4 * 4) = v2;
        *reinterpret_cast<f32x4*>(V + ((long long)(i * 4 + 3) * kT + tile) * kC + c4 * 4) = v3;
    }
}

// out[b][2 ty + dy][2 tx + dx][n] = relu((A^T m A)[dy][dx] + bias[n]), m = the 16 GEMM results of the tile.  One thread per (tile, 4 columns).
__global__ __launch_bounds__(256) void output_transform_kernel(const float* __restrict__ Mb, const float* __restrict__ bias, float* __restrict__ out) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int n4 = (int)(gid % (kN / 4));
    const long long tile = gid / (kN / 4);
    if (tile >= kT) return;
    const int b = (int)(tile / kTilesPerImage), t = (int)(tile % kTilesPerImage), ty = t / 8, tx = t % 8;
    f32x4 m[4][4];
#pragma unroll
    for (int p = 0; p < 16; ++p) m[p / 4][p % 4] = *reinterpret_cast<const f32x4*>(Mb + ((long long)p * kT + tile) * kN + n4 * 4);
    f32x4 q[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                 // A^T m
        q[0][j] = m[0][j] + m[1][j] + m[2][j];
        q[1][j] = m[1][j] - m[2][j] - m[3][j];
    }
    const f32x4 bs = *reinterpret_cast<const f32x4*>(bias + n4 * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        f32x4 y0 = q[i][0] + q[i][1] + q[i][2] + bs, y1 = q[i][1] - q[i][2] - q[i][3] + bs;
#pragma unroll
        for (int e = 0; e < 4; ++e) { y0[e] = fmaxf(y0[e], 0.f); y1[e] = fmaxf(y1[e], 0.f); }
        *reinterpret_cast<f32x4*>(out + (((long long)b * kO + 2 * ty + i) * kO + 2 * tx + 0) * kN + n4 * 4) = y0;
        *reinterpret_cast<f32x4*>(out + (((long long)b * kO + 2 * ty + i) * kO + 2 * tx + 1) * kN + n4 * 4) = y1;
    }
}

// [K][N] -> the library's packed layout [K/4][N][4] (aae_encoder_plan.h: pack_weights with one tap)
static std::vector<float> pack(const std::vector<float>& w) {
    std::vector<float> out((size_t)kC * kN);
    for (int k = 0; k < kC; ++k)
        for (int n = 0; n < kN; ++n) out[((size_t)(k >> 2) * kN + n) * 4 + (k & 3)] = w[(size_t)k * kN + n];
    return out;
}

static double frand(unsigned long long& st) {       // xorshift: deterministic inputs without <random>
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    return (double)(st >> 11) / 9007199254740992.0;
}

int main() {
    unsigned long long seed = 0x9E3779B97F4A7C15ull;
    // activations as a post-ReLU layer leaves them (non-negative, many zeros), weights glorot-uniform like synth.make_weights
    std::vector<float> hs((size_t)kB * kS * kS * kC), hg((size_t)9 * kC * kN), hbias(kN);
    for (auto& v : hs) { const double r = frand(seed); v = r < 0.45 ? 0.f : (float)(1.5 * (r - 0.45)); }
    const double lim = sqrt(6.0 / (25.0 * kC + 25.0 * kN));
    for (auto& v : hg) v = (float)((2.0 * frand(seed) - 1.0) * lim);
    for (auto& v : hbias) v = (float)(0.1 * frand(seed) - 0.05);
    // borders of the sub-image that come from the 'SAME' padding are zero
    for (int b = 0; b < kB; ++b)
        for (int i = 0; i < kS; ++i)
            for (int j = 0; j < kS; ++j)
                if (i == 0 || j == 0 || i == kS - 1 || j == kS - 1)
                    for (int c = 0; c < kC; ++c) hs[(((size_t)b * kS + i) * kS + j) * kC + c] = 0.f;
    // U = G g G^T in float64, rounded once to fp32; [16][K][N]
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<std::vector<float>> U(16, std::vector<float>((size_t)kC * kN));
    for (int k = 0; k < kC; ++k)
        for (int n = 0; n < kN; ++n) {
            double g[3][3], t[4][3];
            for (int a = 0; a < 3; ++a)
                for (int c = 0; c < 3; ++c) g[a][c] = hg[((size_t)(a * 3 + c) * kC + k) * kN + n];
            for (int i = 0; i < 4; ++i)
                for (int c = 0; c < 3; ++c) t[i][c] = G[i][0] * g[0][c] + G[i][1] * g[1][c] + G[i][2] * g[2][c];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) U[i * 4 + j][(size_t)k * kN + n] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
        }
    float *ds, *dV, *dM, *dout, *dbias, *dzero, *dU[16];
    CHECK(hipMalloc(&ds, hs.size() * 4));
    CHECK(hipMalloc(&dV, (size_t)16 * kT * kC * 4));
    CHECK(hipMalloc(&dM, (size_t)16 * kT * kN * 4));
    CHECK(hipMalloc(&dout, (size_t)kB * kO * kO * kN * 4));
    CHECK(hipMalloc(&dbias, kN * 4));
    CHECK(hipMalloc(&dzero, kN * 4));
    CHECK(hipMemset(dzero, 0, kN * 4));
    CHECK(hipMemcpy(ds, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dbias, hbias.data(), kN * 4, hipMemcpyHostToDevice));
    for (int p = 0; p < 16; ++p) {
        const std::vector<float> pk = pack(U[p]);
        CHECK(hipMalloc(&dU[p], pk.size() * 4));
        CHECK(hipMemcpy(dU[p], pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
    }
    using namespace aae;
    auto gemm_args = [&](int p) {
        ConvIgemmArgs a = ConvIgemmArgs();
        a.x = dV + (size_t)p * kT * kC; a.x_bytes = (unsigned)((size_t)kT * kC * 4);
        a.wp = dU[p]; a.wp_bytes = (unsigned)((size_t)kC * kN * 4);
        a.bias = dzero; a.bn_scale = nullptr; a.bn_shift = nullptr;
        a.out = dM + (size_t)p * kT * kN;
        a.H = a.W = a.Ho = a.Wo = 1; a.Cin = kC; a.Cout = a.CoutPad = kN; a.KS = 1; a.S = 1; a.pt = a.pl = 0;
        a.M = kT; a.slabs_total = kC / 32; a.slabs_per_split = a.slabs_total; a.num_mt = kT / 128; a.num_nt = kN / 128; a.splits = 1; a.relu = 0;
        return a;
    };
    (void)hipFuncSetAttribute((const void*)conv_igemm_f32_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)conv_igemm_f32_kernel<false, true, false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kConvIgemmSmem);
    auto run_once = [&](int variant) {
        hipLaunchKernelGGL(input_transform_kernel, dim3((unsigned)((long long)kT * (kC / 4) / 256)), dim3(256), 0, 0, ds, dV);
        for (int p = 0; p < 16; ++p) {
            const ConvIgemmArgs a = gemm_args(p);
            if (variant == 0) hipLaunchKernelGGL((conv_igemm_f32_kernel<false, true>), dim3(a.num_mt * a.num_nt), dim3(256), kConvIgemmSmem, 0, a);
            else hipLaunchKernelGGL((conv_igemm_f32_kernel<false, true, false, 1, true>), dim3(a.num_mt * a.num_nt), dim3(256), 2 * kSlabFloatsA * 4, 0, a);
        }
        hipLaunchKernelGGL(output_transform_kernel, dim3((unsigned)((long long)kT * (kN / 4) / 256)), dim3(256), 0, 0, dM, dbias, dout);
    };
    hipEvent_t e[4];
    for (auto& ev : e) CHECK(hipEventCreate(&ev));
    const double nominal = 2.0 * kB * 256.0 * 9.0 * kC * kN, wino = 2.0 * 16.0 * kT * kC * kN;
    for (int variant = 0; variant < 2; ++variant) {
        for (int w = 0; w < 3; ++w) run_once(variant);
        CHECK(hipDeviceSynchronize());
        const int reps = 10;
        double t_in = 0, t_gemm = 0, t_out = 0;
        for (int r = 0; r < reps; ++r) {
            CHECK(hipEventRecord(e[0], 0));
            hipLaunchKernelGGL(input_transform_kernel, dim3((unsigned)((long long)kT * (kC / 4) / 256)), dim3(256), 0, 0, ds, dV);
            CHECK(hipEventRecord(e[1], 0));
            for (int p = 0; p < 16; ++p) {
                const ConvIgemmArgs a = gemm_args(p);
                if (variant == 0) hipLaunchKernelGGL((conv_igemm_f32_kernel<false, true>), dim3(a.num_mt * a.num_nt), dim3(256), kConvIgemmSmem, 0, a);
                else hipLaunchKernelGGL((conv_igemm_f32_kernel<false, true, false, 1, true>), dim3(a.num_mt * a.num_nt), dim3(256), 2 * kSlabFloatsA * 4, 0, a);
            }
            CHECK(hipEventRecord(e[2], 0));
            hipLaunchKernelGGL(output_transform_kernel, dim3((unsigned)((long long)kT * (kN / 4) / 256)), dim3(256), 0, 0, dM, dbias, dout);
            CHECK(hipEventRecord(e[3], 0));
            CHECK(hipEventSynchronize(e[3]));
            float a = 0, b = 0, c = 0;
            CHECK(hipEventElapsedTime(&a, e[0], e[1])); CHECK(hipEventElapsedTime(&b, e[1], e[2])); CHECK(hipEventElapsedTime(&c, e[2], e[3]));
            t_in += a; t_gemm += b; t_out += c;
        }
        t_in /= reps; t_gemm /= reps; t_out /= reps;
        const double total = t_in + t_gemm + t_out;
        printf("{\"what\": \"winograd_f2x2_3x3_phase_of_conv3\", \"gemm_kernel\": \"%s\", \"B\": %d, \"input_transform_ms\": %.4f, \"gemms_16_ms\": %.4f, \"output_transform_ms\": %.4f, "
               "\"total_ms\": %.4f, \"gemm_tflops\": %.1f, \"tf_equivalent_of_the_direct_phase\": %.1f, \"vs_direct_kernel_at_142_tf\": %.3f, "
               "\"transform_bytes_MB\": %.0f, \"direct_phase_ms_at_142_tf\": %.4f}\n",
               variant == 0 ? "conv_igemm_f32 128x128 LDS-DMA (both operands through LDS)" : "conv_igemm_f32 128x128 LDS-DMA A, weights to registers", kB, t_in, t_gemm,
               t_out, total, wino / (t_gemm * 1e-3) / 1e12, nominal / (total * 1e-3) / 1e12, nominal / (total * 1e-3) / 1e12 / 142.0,
               ((double)hs.size() * 4 + 2.0 * 16 * kT * kC * 4 + 2.0 * 16 * kT * kN * 4 + (double)kB * kO * kO * kN * 4) / 1e6, nominal / 142e12 * 1e3);
    }
    // ---- the fused kernel (winograd_fused.h): one launch, transforms on MFMA fragments in registers
    std::vector<float> hUp((size_t)16 * kC * kN);
    for (int p = 0; p < 16; ++p)
        for (int k = 0; k < kC; ++k)
            for (int n = 0; n < kN; ++n) hUp[wf::packed_u_index(p, k, n, kC)] = U[p][(size_t)k * kN + n];
    float* dUp;
    CHECK(hipMalloc(&dUp, hUp.size() * 4));
    CHECK(hipMemcpy(dUp, hUp.data(), hUp.size() * 4, hipMemcpyHostToDevice));
    const wf::Args fa{ds, dUp, dbias, dout, kC, kN};
    CHECK(hipFuncSetAttribute((const void*)wf::wino_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, wf::kSmemBytes));
    CHECK(hipFuncSetAttribute((const void*)wf::wino_fused4x2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, wf::kSmemBytes4));
    CHECK(hipFuncSetAttribute((const void*)wf::wino_fused8_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, wf::kSmemBytes));
    CHECK(hipFuncSetAttribute((const void*)wf::wino_fused8_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, wf::kSmemBytes));
    CHECK(hipFuncSetAttribute((const void*)wf::wino_fused8_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, wf::kSmemBytes));
    CHECK(hipFuncSetAttribute((const void*)wf::wino_fused8_kernel<16 + 32 + 64>, hipFuncAttributeMaxDynamicSharedMemorySize, wf::kSmemBytes));
    CHECK(hipFuncSetAttribute((const void*)wf::wino_fused8_kernel<16 + 32 + 64 + 128>, hipFuncAttributeMaxDynamicSharedMemorySize, wf::kSmemBytes));
    CHECK(hipFuncSetAttribute((const void*)wf::wino_fused8_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, wf::kSmemBytes));
    CHECK(hipFuncSetAttribute((const void*)wf::wino_fused8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, wf::kSmemBytes));
    CHECK(hipFuncSetAttribute((const void*)wf::wino_fused8_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, wf::kSmemBytes));
    auto launch_fused = [&](int variant) {
        if (variant == 0) hipLaunchKernelGGL(wf::wino_fused_kernel, dim3(kB * (kN / 64)), dim3(256), wf::kSmemBytes, 0, fa);
        else if (variant == 1) hipLaunchKernelGGL(wf::wino_fused8_kernel<0>, dim3(kB * (kN / 64)), dim3(512), wf::kSmemBytes, 0, fa);
        else if (variant == 2) hipLaunchKernelGGL(wf::wino_fused4x2_kernel, dim3(kB * 2 * (kN / 64)), dim3(256), wf::kSmemBytes4, 0, fa);
        else if (variant == 3) hipLaunchKernelGGL(wf::wino_fused8_kernel<1>, dim3(kB * (kN / 64)), dim3(512), wf::kSmemBytes, 0, fa);
        else if (variant == 4) hipLaunchKernelGGL(wf::wino_fused8_kernel<2>, dim3(kB * (kN / 64)), dim3(512), wf::kSmemBytes, 0, fa);
        else if (variant == 5) hipLaunchKernelGGL(wf::wino_fused8_kernel<16>, dim3(kB * (kN / 64)), dim3(512), wf::kSmemBytes, 0, fa);
        else if (variant == 6) hipLaunchKernelGGL(wf::wino_fused8_kernel<32>, dim3(kB * (kN / 64)), dim3(512), wf::kSmemBytes, 0, fa);
        else if (variant == 7) hipLaunchKernelGGL(wf::wino_fused8_kernel<64>, dim3(kB * (kN / 64)), dim3(512), wf::kSmemBytes, 0, fa);
        else if (variant == 8) hipLaunchKernelGGL(wf::wino_fused8_kernel<16 + 32 + 64>, dim3(kB * (kN / 64)), dim3(512), wf::kSmemBytes, 0, fa);
        else hipLaunchKernelGGL(wf::wino_fused8_kernel<16 + 32 + 64 + 128>, dim3(kB * (kN / 64)), dim3(512), wf::kSmemBytes, 0, fa);
    };
    const int fused_variants = getenv("WF_ABLATE") ? 10 : 3;
    auto check_accuracy = [&](const char* what) -> int {
        std::vector<float> hout((size_t)2 * kO * kO * kN);
        CHECK(hipMemcpy(hout.data(), dout, hout.size() * 4, hipMemcpyDeviceToHost));
        double worst_w = 0, worst_d = 0, scale = 0;
        for (int b = 0; b < 2; ++b)
            for (int oy = 0; oy < kO; ++oy)
                for (int ox = 0; ox < kO; ++ox)
                    for (int n = 0; n < kN; n += 7) {                 // every 7th column: 74 of 512
                        double acc = hbias[n];
                        float acc32 = 0.f;
                        for (int a = 0; a < 3; ++a)
                            for (int c = 0; c < 3; ++c) {
                                const float* sp = &hs[(((size_t)b * kS + oy + a) * kS + ox + c) * kC];
                                const float* gp = &hg[(size_t)(a * 3 + c) * kC * kN + n];
                                for (int k = 0; k < kC; ++k) {
                                    acc += (double)sp[k] * (double)gp[(size_t)k * kN];
                                    acc32 = fmaf(sp[k], gp[(size_t)k * kN], acc32);
                                }
                            }
                        const double want = acc > 0 ? acc : 0, got = hout[(((size_t)b * kO + oy) * kO + ox) * kN + n];
                        const double d32 = (double)fmaxf(acc32 + hbias[n], 0.f);
                        worst_w = fmax(worst_w, fabs(got - want));
                        worst_d = fmax(worst_d, fabs(d32 - want));
                        scale = fmax(scale, fabs(want));
                    }
        printf("{\"what\": \"accuracy\", \"of\": \"%s\", \"outputs_checked\": %d, \"max_abs_output\": %.4f, \"winograd_max_abs_err\": %.3e, \"winograd_max_rel_to_scale\": %.3e, "
               "\"direct_fp32_fma_chain_max_abs_err\": %.3e, \"direct_max_rel_to_scale\": %.3e}\n",
               what, 2 * kO * kO * ((kN + 6) / 7), scale, worst_w, worst_w / scale, worst_d, worst_d / scale);
        return 0;
    };
    if (check_accuracy("unfused pipeline")) return 1;
    for (int variant = 0; variant < fused_variants; ++variant) {
        CHECK(hipMemset(dout, 0xFF, (size_t)kB * kO * kO * kN * 4));
        for (int w = 0; w < 3; ++w) launch_fused(variant);
        CHECK(hipDeviceSynchronize());
        const int reps = 20;
        CHECK(hipEventRecord(e[0], 0));
        for (int r = 0; r < reps; ++r) launch_fused(variant);
        CHECK(hipEventRecord(e[1], 0));
        CHECK(hipEventSynchronize(e[1]));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e[0], e[1]));
        ms /= reps;
        const char* name = variant == 0 ? "wino_fused_kernel (4 waves: 16 point accumulators per wave)"
                           : variant == 1 ? "wino_fused8_kernel (8 waves: the points of a tile split over two waves, two waves per SIMD)"
                           : variant == 2 ? "wino_fused4x2_kernel (4 waves = 32 tiles x 64 channels, two blocks per CU)"
                           : variant == 3 ? "ABLATION (results wrong): 8 waves, weight fragments of every group from one address"
                           : variant == 4 ? "ABLATION (results wrong): 8 waves, patch of every group from one LDS plane"
                           : variant == 5 ? "ABLATION (results wrong): 8 waves, no patch read / transform after the first group"
                           : variant == 6 ? "ABLATION (results wrong): 8 waves, no weight loads after the first group"
                           : variant == 7 ? "ABLATION (results wrong): 8 waves, no stage fill"
                           : variant == 8 ? "ABLATION (results wrong): 8 waves, MFMAs + barriers only" : "ABLATION (results wrong): 8 waves, MFMAs only";
        printf("{\"what\": \"winograd_f2x2_3x3_phase_of_conv3_FUSED\", \"kernel\": \"%s\", "
               "\"B\": %d, \"total_ms\": %.4f, \"mfma_tflops\": %.1f, \"mfma_frac_of_157\": %.3f, \"tf_equivalent_of_the_direct_phase\": %.1f, \"vs_direct_kernel_at_142_tf\": %.3f}\n",
               name, kB, ms, wino / (ms * 1e-3) / 1e12, wino / (ms * 1e-3) / 1e12 / 157.3, nominal / (ms * 1e-3) / 1e12, nominal / (ms * 1e-3) / 1e12 / 142.0);
        if (check_accuracy(name)) return 1;
    }
    return 0;
}
