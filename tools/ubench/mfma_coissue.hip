// Micro-benchmark (MI355X): what does an instruction cost NEXT TO a stream of fp32 MFMAs (v_mfma_f32_32x32x2_f32, 64 cycles each on a
// SIMD's matrix pipe) when one or two waves share the SIMD?  The Winograd layer kernel (csrc/kernels/conv_winograd_f32.h) runs ~16 MFMAs
// per wave and "unit" with 16 packed fp32 adds, 8 ds_read_b128 and 4 buffer loads in between and reaches 0.74-0.79 of the MFMA rate in
// its K loop (in-kernel stamps, profiles/r15): this program prices each ingredient alone.
//   One block per CU (LDS-limited), 4 or 8 waves; every wave runs ITER iterations of [16 MFMAs on 8 accumulators + the fillers, spread
//   evenly or in one cluster]; reported: shader cycles per MFMA-slot of a SIMD (64 = the pipe never idles).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_coissue tools/ubench/mfma_coissue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

// fillers per iteration: NPK v_pk_add_f32, NPL plain v_add_f32, NDS ds_read_b128, NVM buffer/global 16-B loads (L2-resident), NSA s_nop-free
// scalar ALU ops; CLUSTER: all fillers behind MFMA 7 instead of spread over the 16 gaps; BAR: an s_barrier per iteration.
template <int NW, int NPK, int NPL, int NDS, int NVM, bool CLUSTER, bool BAR, int PRIO, int LK>
__global__ __launch_bounds__(NW * 64) void k(const float* __restrict__ g, float* __restrict__ out, long long* __restrict__ cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* lds = reinterpret_cast<f32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += NW * 64) lds[i] = f32x4{(float)i, 1.f, 2.f, 3.f};
    __syncthreads();
    if (PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    f32x16 acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f32x2 p0 = {1.f + lane, 2.f}, p1 = {0.5f, 0.25f}, p2 = {3.f, 4.f}, p3 = {5.f, 6.f};
    float s0 = 1.f + lane, s1 = 0.5f, s2 = 2.f, s3 = 3.f;
    f32x4 d[8], gl[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) gl[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4* gp = reinterpret_cast<const f32x4*>(g) + (wave * 64 + lane);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, 64 * 512 * 16 + 8 * 64 * 16, 0x00020000);
    const int lane_off = (wave * 64 + lane) * 16;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto do_load = [&](int it, int vm) {
        const int soff = ((it * 4 + vm) & 63) * 8192;
        if (LK == 0) gl[vm & 3] = gp[((it * 4 + vm) & 63) * 512];
        else if (LK == 1) gl[vm & 3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, soff, 0));
        else if (LK == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 65536 + wave * 4096 + (vm & 3) * 1024), 16, lane_off, soff, 0, 0);
        else {
            const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_off, soff, 0), b = __builtin_amdgcn_raw_buffer_load_b64(rs, lane_off + 8, soff, 0);
            gl[vm & 3] = __builtin_bit_cast(f32x4, u32x4{a[0], a[1], b[0], b[1]});
        }
    };
    float av = 1.f + lane * 1e-3f, bv = 0.5f;
    constexpr int kFill = NPK + NPL + NDS + NVM;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        int pk = 0, pl = 0, ds = 0, vm = 0, done = 0;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m & 7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // how many fillers go behind this MFMA
            const int upto = CLUSTER ? (m == 7 ? kFill : (m < 7 ? 0 : kFill)) : (kFill * (m + 1)) / 16;
#pragma unroll
            for (; done < upto; ++done) {
                // round-robin over the filler kinds that still have work
                if (pk < NPK && (pk * kFill <= done * NPK)) {
                    if (pk & 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p2) : "v"(p2), "v"(p1));
                    else asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p3) : "v"(p3), "v"(p0));
                    ++pk;
                } else if (pl < NPL && (pl * kFill <= done * NPL)) {
                    if (pl & 1) asm volatile("v_add_f32 %0, %1, %2" : "=v"(s2) : "v"(s2), "v"(s1));
                    else asm volatile("v_add_f32 %0, %1, %2" : "=v"(s3) : "v"(s3), "v"(s0));
                    ++pl;
                } else if (ds < NDS && (ds * kFill <= done * NDS)) {
                    d[ds & 7] = lds[(((it & 3) * 8 + ds) & 31) * 64 + lane];
                    ++ds;
                } else if (vm < NVM && (vm * kFill <= done * NVM)) {
                    do_load(it, vm);
                    ++vm;
                } else if (pk < NPK) {
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p2) : "v"(p2), "v"(p1)); ++pk;
                } else if (pl < NPL) {
                    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s2) : "v"(s2), "v"(s1)); ++pl;
                } else if (ds < NDS) {
                    d[ds & 7] = lds[(((it & 3) * 8 + ds) & 31) * 64 + lane]; ++ds;
                } else if (vm < NVM) {
                    do_load(it, vm); ++vm;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // consume the loads (keeps them alive; one wait per iteration like the kernel's unit)
        if (NDS) {
#pragma unroll
            for (int i = 0; i < (NDS < 8 ? NDS : 8); ++i) asm volatile("" ::"v"(d[i]));
        }
        if (NVM && LK == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (NVM && LK != 2) {
#pragma unroll
            for (int i = 0; i < (NVM < 4 ? NVM : 4); ++i) asm volatile("" ::"v"(gl[i]));
        }
        if (BAR) __syncthreads();
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = p2[0] + p2[1] + p3[0] + p3[1] + s2 + s3 + av + bv;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[(size_t)blockIdx.x * NW * 64 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NW, int NPK, int NPL, int NDS, int NVM, bool CLUSTER, bool BAR, int PRIO = 0, int LK = 0>
static int run(const char* what, const float* g, float* out, long long* cyc) {
    const int iters = 512, blocks = 256, smem = 96 * 1024;        // (one block per CU)
    auto kern = k<NW, NPK, NPL, NDS, NVM, CLUSTER, BAR, PRIO, LK>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(NW * 64), smem, 0, g, out, cyc, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NW * 64), smem, 0, g, out, cyc, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long h[256 * 8];
    CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double lo = 0, hi = 0;
    for (int b = 0; b < blocks; ++b) {
        for (int w = 0; w < 4; ++w) lo += (double)h[b * 8 + w];
        for (int w = 4; w < NW; ++w) hi += (double)h[b * 8 + w];
    }
    lo /= blocks * 4; hi = NW > 4 ? hi / (blocks * 4) : 0;
    const double per_simd_slot = (lo > hi ? lo : hi) / ((double)iters * 16 * (NW / 4));        // cycles per MFMA issued on a SIMD
    const double tf = 2.0 * 32 * 32 * 2 * 16.0 * iters * NW * blocks / (ms * 1e-3) / 1e12;
    printf("{\"what\": \"mfma_coissue\", \"case\": \"%s\", \"waves_per_simd\": %d, \"per_16_mfma\": {\"v_pk_add_f32\": %d, \"v_add_f32\": %d, \"ds_read_b128\": %d, \"load_16B\": %d}, \"clustered\": %s, \"barrier\": %s, \"prio\": %d, \"load_kind\": \"%s\", "
           "\"cycles_per_simd_mfma_slot\": %.1f, \"pipe_busy\": %.3f, \"cycles_waves0_3\": %.0f, \"cycles_waves4_7\": %.0f, \"ms\": %.4f, \"tflops\": %.1f}\n",
           what, NW / 4, NPK, NPL, NDS, NVM, CLUSTER ? "true" : "false", BAR ? "true" : "false", PRIO, LK == 0 ? "global_load_dwordx4 (64-bit address arithmetic per load)" : (LK == 1 ? "buffer_load_dwordx4, scalar offset" : (LK == 2 ? "buffer_load_dwordx4 lds (LDS-DMA)" : "2 x buffer_load_dwordx2")), per_simd_slot, 64.0 / per_simd_slot, lo, hi, ms, tf);
    fflush(stdout);
    return 0;
}


// ---- the loop shape the prices above suggest: per unit ONE cluster [wait for the patch reads issued before the previous burst -> 16 packed adds
//      (the transform) -> 4 buffer loads (next weights) -> 8 ds_read_b128 (next patch)] and ONE burst of 16 MFMAs; a block barrier in the cluster of
//      every fourth unit (the stage boundary).  PRIO: 0 none, 1 waves 4-7 static, 2 alternating by unit parity, 3 raised in the cluster, 4 raised in the burst.
template <int PRIO, int NPKU, bool BAR4>
__global__ __launch_bounds__(512) void k_unit(const float* __restrict__ g, float* __restrict__ out, long long* __restrict__ cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* lds = reinterpret_cast<f32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 512) lds[i] = f32x4{(float)i, 1.f, 2.f, 3.f};
    __syncthreads();
    if (PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    f32x16 acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, 64 * 512 * 16 + 8 * 64 * 16, 0x00020000);
    const int lane_off = (wave * 64 + lane) * 16;
    f32x4 raw[8], v[4], u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { raw[i] = lds[i * 64 + lane]; u[i] = f32x4{0.5f, 0.25f, 0.125f, 1.f}; }
    const long long t0 = __builtin_readcyclecounter();
    for (int it4 = 0; it4 < iters; it4 += 4) {
#pragma unroll
      for (int u4 = 0; u4 < 4; ++u4) {
        const int it = it4 + u4, row = u4 & 1;
        // ---- cluster
        if (PRIO == 2) { if (((u4 ^ (wave >> 2)) & 1) == 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        if (PRIO == 3) __builtin_amdgcn_s_setprio(1);
        if (PRIO == 4) __builtin_amdgcn_s_setprio(0);
        // transform: NPKU packed adds on the patch read before the previous burst
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x2 lo = {raw[j][0], raw[j][1]}, hi = {raw[j][2], raw[j][3]}, lo2 = {raw[4 + j][0], raw[4 + j][1]}, hi2 = {raw[4 + j][2], raw[4 + j][3]};
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(lo2));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(hi2));
            if (NPKU > 8) {
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(hi2));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(lo2));
            }
            v[j] = f32x4{lo[0], lo[1], hi[0], hi[1]};
        }
        // next weights into the row the previous burst released
        const int soff = ((it * 4) & 63) * 8192;
#pragma unroll
        for (int j = 0; j < 4; ++j) u[(row ^ 1) * 4 + j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, soff + j * 8192, 0));
        if (BAR4 && u4 == 3) __syncthreads();
        // next patch
#pragma unroll
        for (int i = 0; i < 8; ++i) raw[i] = lds[((u4 * 8 + i) & 31) * 64 + lane];
        if (PRIO == 3) __builtin_amdgcn_s_setprio(0);
        if (PRIO == 4) __builtin_amdgcn_s_setprio(1);
        __builtin_amdgcn_sched_barrier(0);
        // ---- burst
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[row * 4 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j][q], u[row * 4 + j][q], acc[row * 4 + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[(size_t)blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int PRIO, int NPKU, bool BAR4>
static int run_unit(const char* what, const float* g, float* out, long long* cyc) {
    const int iters = 512, blocks = 256, smem = 96 * 1024;
    auto kern = k_unit<PRIO, NPKU, BAR4>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), smem, 0, g, out, cyc, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), smem, 0, g, out, cyc, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long h[256 * 8];
    CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double lo = 0, hi = 0;
    for (int b = 0; b < blocks; ++b) {
        for (int w = 0; w < 4; ++w) lo += (double)h[b * 8 + w];
        for (int w = 4; w < 8; ++w) hi += (double)h[b * 8 + w];
    }
    lo /= blocks * 4; hi /= blocks * 4;
    const double per_simd_slot = (lo > hi ? lo : hi) / ((double)iters * 32);
    printf("{\"what\": \"mfma_coissue_unit_loop\", \"case\": \"%s\", \"prio\": %d, \"pk_per_unit\": %d, \"barrier_every_4_units\": %s, \"cycles_per_simd_mfma_slot\": %.1f, \"pipe_busy\": %.3f, "
           "\"cycles_waves0_3\": %.0f, \"cycles_waves4_7\": %.0f, \"ms\": %.4f}\n", what, PRIO, NPKU, BAR4 ? "true" : "false", per_simd_slot, 64.0 / per_simd_slot, lo, hi, ms);
    fflush(stdout);
    return 0;
}

int main() {
    float *g, *out; long long* cyc;
    CHECK(hipMalloc(&g, 64 * 512 * 16 + 8 * 64 * 16 + 4096));
    CHECK(hipMemset(g, 0, 64 * 512 * 16 + 8 * 64 * 16 + 4096));
    CHECK(hipMalloc(&out, 256 * 512 * 4));
    CHECK(hipMalloc(&cyc, 256 * 8 * 8));
    //        NW NPK NPL NDS NVM  cluster barrier
    run<4, 0, 0, 0, 0, false, false>("mfma only", g, out, cyc);
    run<8, 0, 0, 0, 0, false, false>("mfma only", g, out, cyc);
    run<8, 0, 0, 0, 0, false, true>("mfma + barrier", g, out, cyc);
    run<4, 16, 0, 0, 0, false, false>("16 pk spread", g, out, cyc);
    run<8, 16, 0, 0, 0, false, false>("16 pk spread", g, out, cyc);
    run<8, 16, 0, 0, 0, true, false>("16 pk clustered", g, out, cyc);
    run<8, 32, 0, 0, 0, false, false>("32 pk spread", g, out, cyc);
    run<4, 0, 32, 0, 0, false, false>("32 plain spread", g, out, cyc);
    run<8, 0, 32, 0, 0, false, false>("32 plain spread", g, out, cyc);
    run<8, 0, 32, 0, 0, true, false>("32 plain clustered", g, out, cyc);
    run<8, 0, 64, 0, 0, false, false>("64 plain spread", g, out, cyc);
    run<4, 0, 0, 8, 0, false, false>("8 ds_read", g, out, cyc);
    run<8, 0, 0, 8, 0, false, false>("8 ds_read", g, out, cyc);
    run<8, 0, 0, 16, 0, false, false>("16 ds_read", g, out, cyc);
    run<4, 0, 0, 0, 4, false, false>("4 loads", g, out, cyc);
    run<8, 0, 0, 0, 4, false, false>("4 loads", g, out, cyc);
    run<8, 0, 0, 0, 8, false, false>("8 loads", g, out, cyc);
    run<8, 16, 0, 8, 4, false, false>("kernel mix (16 pk, 8 ds, 4 loads)", g, out, cyc);
    run<8, 16, 0, 8, 4, false, true>("kernel mix + barrier", g, out, cyc);
    run<8, 16, 0, 8, 4, true, false>("kernel mix clustered", g, out, cyc);
    run<8, 16, 0, 8, 4, false, false, 1>("kernel mix, waves 4-7 prio 1", g, out, cyc);
    run<8, 0, 32, 8, 4, false, false>("kernel mix with plain adds", g, out, cyc);
    run<8, 0, 32, 8, 4, true, false>("kernel mix with plain adds clustered", g, out, cyc);
    run<4, 16, 0, 8, 4, false, false>("kernel mix", g, out, cyc);
    // the load flavours, alone
    run<8, 0, 0, 0, 4, false, false, 0, 1>("4 loads", g, out, cyc);
    run<4, 0, 0, 0, 4, false, false, 0, 1>("4 loads", g, out, cyc);
    run<8, 0, 0, 0, 8, false, false, 0, 1>("8 loads", g, out, cyc);
    run<8, 0, 0, 0, 2, false, false, 0, 1>("2 loads", g, out, cyc);
    run<8, 0, 0, 0, 4, true, false, 0, 1>("4 loads clustered", g, out, cyc);
    run<8, 0, 0, 0, 4, false, false, 0, 2>("4 loads", g, out, cyc);
    run<8, 0, 0, 0, 2, false, false, 0, 2>("2 loads", g, out, cyc);
    run<8, 0, 0, 0, 4, false, false, 0, 3>("4 loads", g, out, cyc);
    run<8, 0, 0, 4, 0, false, false>("4 ds_read", g, out, cyc);
    run<8, 0, 0, 12, 0, false, false>("12 ds_read", g, out, cyc);
    run<8, 0, 0, 12, 0, true, false>("12 ds_read clustered", g, out, cyc);
    run<8, 8, 0, 0, 0, false, false>("8 pk spread", g, out, cyc);
    run<8, 8, 0, 0, 0, true, false>("8 pk clustered", g, out, cyc);
    run<8, 16, 0, 8, 4, false, false, 0, 1>("kernel mix", g, out, cyc);
    run<8, 16, 0, 8, 4, true, false, 0, 1>("kernel mix clustered", g, out, cyc);
    run<8, 16, 0, 12, 0, true, false>("mix with weights through LDS (12 ds_read) clustered", g, out, cyc);
    run<8, 16, 0, 12, 0, false, false>("mix with weights through LDS (12 ds_read)", g, out, cyc);
    run<8, 16, 0, 12, 2, true, false, 0, 2>("mix with weights through LDS + 2 DMA pieces, clustered", g, out, cyc);
    run<8, 8, 0, 8, 4, true, false, 0, 1>("half the transforms, clustered", g, out, cyc);
    run<8, 8, 0, 4, 4, true, false, 0, 1>("half the transforms + half the patch reads, clustered", g, out, cyc);
    run<8, 16, 0, 8, 2, true, false, 0, 1>("half the weight loads, clustered", g, out, cyc);
    run_unit<0, 16, false>("unit loop", g, out, cyc);
    run_unit<0, 16, true>("unit loop", g, out, cyc);
    run_unit<1, 16, true>("unit loop, waves 4-7 prio 1", g, out, cyc);
    run_unit<2, 16, true>("unit loop, prio alternating by unit", g, out, cyc);
    run_unit<3, 16, true>("unit loop, prio raised in the cluster", g, out, cyc);
    run_unit<4, 16, true>("unit loop, prio raised in the burst", g, out, cyc);
    run_unit<2, 16, false>("unit loop, prio alternating by unit", g, out, cyc);
    run_unit<3, 16, false>("unit loop, prio raised in the cluster", g, out, cyc);
    run_unit<4, 16, false>("unit loop, prio raised in the burst", g, out, cyc);
    run_unit<0, 8, true>("unit loop", g, out, cyc);
    return 0;
}
