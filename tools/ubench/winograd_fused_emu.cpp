// CPU check of tools/ubench/winograd_fused.h on the fiber emulator of tests/emu (index math, fragment maps, LDS layout):
//   cd tools/ubench && /opt/rocm/lib/llvm/bin/clang++ -O2 -std=c++17 -ffp-contract=off -I ../../tests/emu -o /tmp/wf_emu winograd_fused_emu.cpp ../../tests/emu/hip_emu.cpp && /tmp/wf_emu
#include "hip_emu.h"

#include <stdio.h>

#include <vector>

#include "winograd_fused.h"

static double frand(unsigned long long& st) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    return (double)(st >> 11) / 9007199254740992.0;
}

int main() {
    const int B = 2, C = 64, N = 128;
    unsigned long long seed = 0x9E3779B97F4A7C15ull;
    std::vector<float> hs((size_t)B * 18 * 18 * C), hg((size_t)9 * C * N), hb(N), hout((size_t)B * 16 * 16 * N, -1.f);
    for (auto& v : hs) { const double r = frand(seed); v = r < 0.45 ? 0.f : (float)(1.5 * (r - 0.45)); }
    for (auto& v : hg) v = (float)((2.0 * frand(seed) - 1.0) * 0.05);
    for (auto& v : hb) v = (float)(0.1 * frand(seed) - 0.05);
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::vector<float> U((size_t)16 * C * N);
    for (int k = 0; k < C; ++k)
        for (int n = 0; n < N; ++n) {
            double g[3][3], t[4][3];
            for (int a = 0; a < 3; ++a)
                for (int c = 0; c < 3; ++c) g[a][c] = hg[((size_t)(a * 3 + c) * C + k) * N + n];
            for (int i = 0; i < 4; ++i)
                for (int c = 0; c < 3; ++c) t[i][c] = G[i][0] * g[0][c] + G[i][1] * g[1][c] + G[i][2] * g[2][c];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j)
                    U[wf::packed_u_index(i * 4 + j, k, n, C)] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
        }
    wf::Args a{hs.data(), U.data(), hb.data(), hout.data(), C, N};
    int rc = 0;
    for (int variant = 0; variant < 3; ++variant) {
    std::fill(hout.begin(), hout.end(), -1.f);
    if (variant == 0) aae_emu::launch(dim3(B * (N / 64)), dim3(256), wf::kSmemBytes, [&]() { wf::wino_fused_kernel(a); });
    else if (variant == 1) aae_emu::launch(dim3(B * (N / 64)), dim3(512), wf::kSmemBytes, [&]() { wf::wino_fused8_kernel<0>(a); });
    else aae_emu::launch(dim3(B * 2 * (N / 64)), dim3(256), wf::kSmemBytes4, [&]() { wf::wino_fused4x2_kernel(a); });
    double worst = 0, scale = 0;
    for (int b = 0; b < B; ++b)
        for (int oy = 0; oy < 16; ++oy)
            for (int ox = 0; ox < 16; ++ox)
                for (int n = 0; n < N; ++n) {
                    double acc = hb[n];
                    for (int ka = 0; ka < 3; ++ka)
                        for (int kc = 0; kc < 3; ++kc)
                            for (int k = 0; k < C; ++k)
                                acc += (double)hs[(((size_t)b * 18 + oy + ka) * 18 + ox + kc) * C + k] * (double)hg[((size_t)(ka * 3 + kc) * C + k) * N + n];
                    const double want = acc > 0 ? acc : 0, got = hout[(((size_t)b * 16 + oy) * 16 + ox) * N + n];
                    worst = fmax(worst, fabs(got - want));
                    scale = fmax(scale, fabs(want));
                }
    printf("variant %d: max abs err %.3e (scale %.3f)\n", variant, worst, scale);
    if (!(worst < 1e-5 * fmax(scale, 1.0))) rc = 1;
    }
    return rc;
}
