// Micro-benchmark (MI355X): where does a tile of the query-resident bf16 scan (config 5: 368 928 x 128 bf16 rows, B = 256)
// spend its time?  The production kernel, built with parts of the per-tile work removed at compile time:
//   for a in 0 1 2 4 8 3 6 7 14 15; do
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -I augmentedautoencoder_amd/csrc -DAAE_SCAN_RESIDENT_ABLATE=$a \
//           -o scan_resident_ablate_$a tools/ubench/scan_resident_ablate.hip; done
// (tools/gpu_scan_ablate.sh builds and runs them).  1 = no arg-max fold, 2 = no LDS fragment reads, 4 = no MFMAs,
// 8 = no global loads / LDS writes.  Results of an ablated build are wrong by construction; only the time is looked at.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "device_intrinsics.h"
#include "kernels/tile_f32.h"
#include "kernels/multi_launch.h"
#include "kernels/codebook_scan_f32.h"
#include "kernels/codebook_scan_bf16.h"
#include "kernels/codebook_scan_resident.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void reset_prune(int* w, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = aae::kScoreKeyEmpty;
}

// prune != nullptr: the top-k bound words, reset by a small kernel in front of every launch (as the normalise kernel of the
// library does) -- its ~2-3 us are inside the figure
template <int K, int RH>
static int run(const char* what, int N, int B, const void* E, const void* qp, float* pval, int* pidx, float* cv, int* ci, int cus, int* prune = nullptr) {
    aae::ScanResidentArgs a;
    a.E = E; a.e_bytes = (unsigned)((size_t)N * 256);
    a.qp = qp; a.pval = pval; a.pidx = pidx;
    a.N = N; a.B = B; a.Bpad = (B + 31) / 32 * 32; a.Bstride = a.Bpad;
    const int QB = aae::scan_resident_queries<RH>();
    const int chunks = (a.Bpad + QB - 1) / QB;
    const int ntiles = (N + 127) / 128;
    const int blocks = cus / chunks > 0 ? cus / chunks : 1;
    a.tiles_per_block = (ntiles + blocks - 1) / blocks;
    const int gx = (ntiles + a.tiles_per_block - 1) / a.tiles_per_block;
    a.k = K; a.cand_v = cv; a.cand_i = ci; a.prune = prune;
#ifdef AAE_SCAN_COUNT
    int* dbg; CHECK(hipMalloc(&dbg, 64 * 4 + 16 * 8)); CHECK(hipMemset(dbg, 0, 64 * 4 + 16 * 8)); a.dbg = dbg;
#endif
    const int pw = aae::kPruneReplicas * a.Bpad * aae::kPruneGroups;
    CHECK(hipFuncSetAttribute((const void*)aae::scan_resident_kernel<true, K, RH>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::scan_resident_smem<true, RH>()));
    const int smem = aae::scan_resident_smem<true, RH>();
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int reps = 50;
    for (int w = 0; w < 5; ++w) {
        if (prune) hipLaunchKernelGGL(reset_prune, dim3((pw + 255) / 256), dim3(256), 0, 0, prune, pw);
        hipLaunchKernelGGL((aae::scan_resident_kernel<true, K, RH>), dim3(gx, chunks), dim3(aae::kScanResidentThreads), smem, 0, a);
    }
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) {
        if (prune) hipLaunchKernelGGL(reset_prune, dim3((pw + 255) / 256), dim3(256), 0, 0, prune, pw);
        hipLaunchKernelGGL((aae::scan_resident_kernel<true, K, RH>), dim3(gx, chunks), dim3(aae::kScanResidentThreads), smem, 0, a);
    }
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
#ifdef AAE_SCAN_COUNT
    { int h[64]; CHECK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost)); const double per = 55.0 * gx * chunks * 8;   // launches x waves
      printf("%s: per wave and step (accumulator tiles passed of %d, value slots inserted of %d):", what, RH == 1 ? 4 : 2, RH == 1 ? 64 : 32);
      for (int st = 0; st < 12; ++st) printf(" [%.2f %.2f]", h[2 * st] / per, h[2 * st + 1] / per); printf("\n");
      long long tk[16]; CHECK(hipMemcpy(tk, dbg + 64, sizeof(tk), hipMemcpyDeviceToHost));
      printf("   step durations of block 100 in the last launch (us):"); for (int st = 0; st < 12; ++st) printf(" %.2f", (tk[st + 1] - tk[st]) * 0.01); if (K > 0) printf(" | after the loop %.2f", (tk[13] - tk[12]) * 0.01); printf("\n"); }
#endif
    printf("{\"what\": \"scan_resident_ablate\", \"ablate\": %d, \"kernel\": \"%s\", \"N\": %d, \"B\": %d, \"grid\": [%d, %d], \"tiles_per_block\": %d, \"us_per_launch\": %.2f}\n",
           AAE_SCAN_RESIDENT_ABLATE, what, N, B, gx, chunks, a.tiles_per_block, ms * 1000.f / reps);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int N = 368928, B = 256;
    unsigned short* E; unsigned short* qp; float* pval; int* pidx; float* cv; int* ci;
    CHECK(hipMalloc(&E, (size_t)N * 256));
    CHECK(hipMalloc(&qp, (size_t)3 * 16 * 256 * 8 * 2));
    CHECK(hipMalloc(&pval, (size_t)1024 * 256 * 4));
    CHECK(hipMalloc(&pidx, (size_t)1024 * 256 * 4));
    CHECK(hipMalloc(&cv, (size_t)256 * 1024 * 8 * 4));
    CHECK(hipMalloc(&ci, (size_t)256 * 1024 * 8 * 4));
    // bf16 values of magnitude < 1 with varied signs and exponents (no NaN patterns)
    {
        const size_t n = (size_t)N * 128;
        unsigned short* h = (unsigned short*)malloc(n * 2);
        unsigned s = 12345u;
        for (size_t k = 0; k < n; ++k) { s = s * 1664525u + 1013904223u; h[k] = (unsigned short)(((s >> 16) & 0x80ffu) | 0x3d00u | ((s >> 9) & 0x0100u)); }
        CHECK(hipMemcpy(E, h, n * 2, hipMemcpyHostToDevice));
        for (size_t k = 0; k < (size_t)3 * 16 * 256 * 8; ++k) { s = s * 1664525u + 1013904223u; h[k] = (unsigned short)(((s >> 16) & 0x80ffu) | 0x3d00u); }
        CHECK(hipMemcpy(qp, h, (size_t)3 * 16 * 256 * 8 * 2, hipMemcpyHostToDevice));
        free(h);
    }
    if (run<0, 1>("bf16 arg-max, 256 queries per block", N, B, E, qp, pval, pidx, cv, ci, cus)) return 1;
    if (run<5, 1>("bf16 top-5, 256 queries per block", N, B, E, qp, pval, pidx, cv, ci, cus)) return 1;
    int* prune;
    CHECK(hipMalloc(&prune, (size_t)aae::kPruneReplicas * 256 * aae::kPruneGroups * 4));
    if (run<5, 1>("bf16 top-5, 256 queries per block, pruned", N, B, E, qp, pval, pidx, cv, ci, cus, prune)) return 1;
    if (run<5, 2>("bf16 top-5, 128 queries per block (B = 32), pruned", N, 32, E, qp, pval, pidx, cv, ci, cus, prune)) return 1;
    if (run<5, 2>("bf16 top-5, 128 queries per block (B = 32)", N, 32, E, qp, pval, pidx, cv, ci, cus)) return 1;
    if (run<0, 2>("bf16 arg-max, 128 queries per block", N, 128, E, qp, pval, pidx, cv, ci, cus)) return 1;
    return 0;
}
