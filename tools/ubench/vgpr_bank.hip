// Micro-benchmark (MI355X): does the VGPR BANK (register number mod 4) of an instruction's source operands change what it costs next to / inside a stream of
// fp32 MFMAs (v_mfma_f32_32x32x2_f32)?  Background: two builds of the Winograd layer kernel with the same instruction stream but another register assignment
// differ by 3.5 % in their K loops (profiles/r15/wino_persistent_blocks_ab.json).  Explicit registers through inline asm; one block of 4 or 8 waves per CU
// (1 or 2 waves per SIMD), shader-clock cycles per loop iteration.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/vgpr_bank tools/ubench/vgpr_bank.hip && ./build/vgpr_bank
// Result (profiles/r15/vgpr_bank.jsonl): NO effect -- 8 MFMAs 512.23 cycles whatever the banks of A and B, 16 packed adds + 8 MFMAs 600.26 whatever the banks of the
// adds' sources.  (The 2-waves-per-SIMD rows average an older wave that keeps the pipe and a younger one that waits for it: independent accumulators never yield.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define CLOBBERS "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", \
    "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", \
    "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", \
    "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", \
    "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", \
    "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", \
    "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", \
    "v159", "s20", "scc", "memory"

// eight independent accumulators v[32:47] ... v[144:159]; operands in v0 ... v31
#define MFMA8(A0, B0, A1, B1, A2, B2, A3, B3, A4, B4, A5, B5, A6, B6, A7, B7)                        \
    "v_mfma_f32_32x32x2_f32 v[32:47], " A0 ", " B0 ", v[32:47]\n"                                      \
    "v_mfma_f32_32x32x2_f32 v[48:63], " A1 ", " B1 ", v[48:63]\n"                                      \
    "v_mfma_f32_32x32x2_f32 v[64:79], " A2 ", " B2 ", v[64:79]\n"                                      \
    "v_mfma_f32_32x32x2_f32 v[80:95], " A3 ", " B3 ", v[80:95]\n"                                      \
    "v_mfma_f32_32x32x2_f32 v[96:111], " A4 ", " B4 ", v[96:111]\n"                                    \
    "v_mfma_f32_32x32x2_f32 v[112:127], " A5 ", " B5 ", v[112:127]\n"                                  \
    "v_mfma_f32_32x32x2_f32 v[128:143], " A6 ", " B6 ", v[128:143]\n"                                  \
    "v_mfma_f32_32x32x2_f32 v[144:159], " A7 ", " B7 ", v[144:159]\n"

// sixteen packed adds, sources (S1, S2) pairs; destinations v[16:31] in pairs
#define PK16(S1, S2)                                                                                    \
    "v_pk_add_f32 v[16:17], " S1 ", " S2 "\n v_pk_add_f32 v[18:19], " S1 ", " S2 "\n"                   \
    "v_pk_add_f32 v[20:21], " S1 ", " S2 "\n v_pk_add_f32 v[22:23], " S1 ", " S2 "\n"                   \
    "v_pk_add_f32 v[24:25], " S1 ", " S2 "\n v_pk_add_f32 v[26:27], " S1 ", " S2 "\n"                   \
    "v_pk_add_f32 v[28:29], " S1 ", " S2 "\n v_pk_add_f32 v[30:31], " S1 ", " S2 "\n"                   \
    "v_pk_add_f32 v[16:17], " S1 ", " S2 "\n v_pk_add_f32 v[18:19], " S1 ", " S2 "\n"                   \
    "v_pk_add_f32 v[20:21], " S1 ", " S2 "\n v_pk_add_f32 v[22:23], " S1 ", " S2 "\n"                   \
    "v_pk_add_f32 v[24:25], " S1 ", " S2 "\n v_pk_add_f32 v[26:27], " S1 ", " S2 "\n"                   \
    "v_pk_add_f32 v[28:29], " S1 ", " S2 "\n v_pk_add_f32 v[30:31], " S1 ", " S2 "\n"

#define LOOP(BODY) asm volatile("s_mov_b32 s20, %0\n 1:\n" BODY "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n" ::"s"(iters) : CLOBBERS)

template <int CASE>
__global__ __launch_bounds__(512) void k(int iters, long long* cyc) {
    const long long t0 = __builtin_readcyclecounter();
    if (CASE == 0) LOOP(MFMA8("v0", "v4", "v0", "v4", "v8", "v12", "v8", "v12", "v0", "v4", "v0", "v4", "v8", "v12", "v8", "v12"));          // A, B same bank (0, 0)
    if (CASE == 1) LOOP(MFMA8("v0", "v5", "v0", "v5", "v8", "v13", "v8", "v13", "v0", "v5", "v0", "v5", "v8", "v13", "v8", "v13"));          // A, B banks (0, 1)
    if (CASE == 2) LOOP(MFMA8("v0", "v6", "v0", "v6", "v8", "v14", "v8", "v14", "v0", "v6", "v0", "v6", "v8", "v14", "v8", "v14"));          // (0, 2)
    if (CASE == 3) LOOP(PK16("v[0:1]", "v[4:5]") MFMA8("v0", "v5", "v0", "v5", "v8", "v13", "v8", "v13", "v0", "v5", "v0", "v5", "v8", "v13", "v8", "v13"));   // pk sources same banks
    if (CASE == 4) LOOP(PK16("v[0:1]", "v[6:7]") MFMA8("v0", "v5", "v0", "v5", "v8", "v13", "v8", "v13", "v0", "v5", "v0", "v5", "v8", "v13", "v8", "v13"));   // pk sources banks (0,1) (2,3)
    if (CASE == 5) LOOP(PK16("v[0:1]", "v[2:3]") MFMA8("v0", "v5", "v0", "v5", "v8", "v13", "v8", "v13", "v0", "v5", "v0", "v5", "v8", "v13", "v8", "v13"));   // pk sources banks (0,1) (2,3), neighbours
    if (CASE == 6) LOOP(PK16("v[0:1]", "v[4:5]") MFMA8("v0", "v4", "v0", "v4", "v8", "v12", "v8", "v12", "v0", "v4", "v0", "v4", "v8", "v12", "v8", "v12"));   // both clash
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int CASE>
static int run(const char* what, int waves) {
    const int iters = 2000, blocks = 256;
    long long* d;
    CHECK(hipMalloc(&d, blocks * 8 * sizeof(long long)));
    CHECK(hipMemset(d, 0, blocks * 8 * sizeof(long long)));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<CASE>, dim3(blocks), dim3(64 * waves), 0, 0, iters, d);
    CHECK(hipDeviceSynchronize());
    std::vector<long long> h(blocks * 8);
    CHECK(hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double sum = 0;
    int n = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) { sum += (double)h[b * 8 + w]; ++n; }
    // (s_memtime / readcyclecounter ticks at 100 MHz on this chip: report ticks per iteration; ratios between cases are what matters)
    printf("{\"what\": \"vgpr_bank\", \"case\": \"%s\", \"waves_per_simd\": %d, \"ticks_per_iteration\": %.4f}\n", what, waves / 4, sum / n / iters);
    CHECK(hipFree(d));
    return 0;
}

int main() {
    for (int waves : {4, 8}) {
        if (waves == 4) {
            run<0>("8 mfma, A and B in the same bank", 4); run<1>("8 mfma, A and B in banks 0 / 1", 4); run<2>("8 mfma, A and B in banks 0 / 2", 4);
            run<3>("16 pk_add (sources same banks) + 8 mfma", 4); run<4>("16 pk_add (sources banks 01 / 23) + 8 mfma", 4); run<5>("16 pk_add (sources banks 01 / 23 neighbouring pair) + 8 mfma", 4);
            run<6>("16 pk_add + 8 mfma, both with bank clashes", 4);
        } else {
            run<0>("8 mfma, A and B in the same bank", 8); run<1>("8 mfma, A and B in banks 0 / 1", 8); run<2>("8 mfma, A and B in banks 0 / 2", 8);
            run<3>("16 pk_add (sources same banks) + 8 mfma", 8); run<4>("16 pk_add (sources banks 01 / 23) + 8 mfma", 8); run<5>("16 pk_add (sources banks 01 / 23 neighbouring pair) + 8 mfma", 8);
            run<6>("16 pk_add + 8 mfma, both with bank clashes", 8);
        }
    }
    return 0;
}
