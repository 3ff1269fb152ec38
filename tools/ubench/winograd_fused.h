// Fused Winograd F(2 x 2, 3 x 3) kernel of the micro-benchmark tools/ubench/polyphase_winograd.hip (NOT part of the product
// library): conv3's 3 x 3 polyphase component -- sub-image [B, 18, 18, C] (16 + halo, NHWC), C -> N channels, outputs
// [B, 16, 16, N] with bias + ReLU -- in ONE launch: the input transform B^T d B and the output transform A^T m A happen on MFMA
// fragments in registers, nothing but the activations, the transformed weights and the outputs touches memory.
//
//   block = 4 waves = one image (its 8 x 8 Winograd tiles) x 64 output channels; wave (mh, nh) = tile rows 4 mh .. 4 mh + 3 (32
//           tiles = the M side of a 32 x 32 MFMA tile) x channels 32 nh .. 32 nh + 31.  Every wave keeps the accumulators of ALL
//           16 Winograd points of its (32 tiles x 32 channels): 16 x 16 = 256 registers, so the output transform is 16 -> 4
//           element-wise combinations of accumulator registers and never leaves the wave.
//   K loop  = stages of 32 input channels.  The image's 18 x 18 x 32-channel slab goes global -> registers -> LDS once per block
//           (double buffered), laid out [channel quad][column parity][row][column / 2] so that the 16 patch positions of the 32
//           tiles are conflict-free ds_read_b128 (row pitch 12 quads: two rows = 24 quads = a different bank group for each of
//           the 16 lanes the hardware serves together) and the fill's eight quads of a pixel fall into different banks (plane
//           pitch + 1 quad).  Per 8-channel group a lane reads its tile's 4 x 4 patch as 16 float4 (4 channels of its K half),
//           transforms it in place (32 adds per channel), loads the 16 points' weight fragments (float4 each, pre-packed in
//           fragment order: 1 KB contiguous per wave and point) and issues 64 MFMAs (4 per point).
//   weights = U[p] = G g G^T per (c, n), packed [32-channel block of N][8-channel group][point][K half][32][4].
//
// Written against the portable subset the CPU fiber emulator (tests/emu/hip_emu.h) models, so tools/ubench/winograd_fused_emu.cpp
// checks the index math on the CPU before a GPU minute is spent.
#pragma once

namespace wf {

constexpr int kSub = 18;                                  // side of the sub-image with its halo
constexpr int kOut = 16;                                  // side of the output
constexpr int kRowUnits = 12;                             // LDS row pitch in 16-B units (9 used)
constexpr int kPlaneUnits = 2 * kSub * kRowUnits + 1;     // one (8-channel group, K half) plane: [parity][row][x / 2], + 1 unit
constexpr int kStageUnits = 8 * kPlaneUnits;              // 32 channels = 8 quads
constexpr int kSmemBytes = 2 * kStageUnits * 16;          // two stages: 110 848 B
constexpr int kStageQuads = kSub * kSub * 8;              // 2592 float4 per stage
constexpr int kFillIters = (kStageQuads + 255) / 256;     // 11

// packed fp32 add / subtract (two values per instruction and lane: the transforms are VALU work that competes with the MFMA issue)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ f32x2 pk_add2(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_sub2(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#else
__host__ __device__ inline f32x2 pk_add2(f32x2 a, f32x2 b) { return a + b; }
__host__ __device__ inline f32x2 pk_sub2(f32x2 a, f32x2 b) { return a - b; }
#endif
__device__ __forceinline__ f32x4 add4(f32x4 a, f32x4 b) {
    const f32x2 lo = pk_add2(a.lo, b.lo), hi = pk_add2(a.hi, b.hi);
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) {
    const f32x2 lo = pk_sub2(a.lo, b.lo), hi = pk_sub2(a.hi, b.hi);
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}

struct Args {
    const float* s;          // [B][18][18][C]
    const float* U;          // packed transformed weights (see above)
    const float* bias;       // [N]
    float* out;              // [B][16][16][N]
    int C, N;                // C % 32 == 0, N % 64 == 0
};

// index of U[point p][channel c][column n] in the packed array
inline __host__ __device__ size_t packed_u_index(int p, int c, int n, int C) {
    const int n32 = n >> 5, nn = n & 31, kg = c >> 3, hh = (c >> 2) & 1, q = c & 3;
    return ((((size_t)n32 * (C / 8) + kg) * 16 + p) * 64 + hh * 32 + nn) * 4 + q;
}

__global__ __launch_bounds__(256) void wino_fused_kernel(Args a) {
    AAE_DYN_SMEM(smem_raw);
    f32x4* lds = reinterpret_cast<f32x4*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mh = wave & 1, nh = wave >> 1, m = lane & 31, h = lane >> 5;
    const int ty = 4 * mh + (m >> 3), tx = m & 7;
    const int nbn = a.N / 64;
    const int nb = blockIdx.x % nbn, b = blockIdx.x / nbn;
    const int n32 = nb * 2 + nh, KG = a.C / 8, nst = a.C / 32, cq_per_pixel = a.C / 4;
    const f32x4* src = reinterpret_cast<const f32x4*>(a.s) + (size_t)b * kSub * kSub * cq_per_pixel;
    const f32x4* up = reinterpret_cast<const f32x4*>(a.U) + (size_t)n32 * KG * 16 * 64 + h * 32 + m;

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    f32x4 stg[kFillIters];
    auto stage_load = [&](int st) {
#pragma unroll
        for (int i = 0; i < kFillIters; ++i) {
            const int idx = tid + 256 * i;
            if (idx < kStageQuads) stg[i] = src[(size_t)(idx >> 3) * cq_per_pixel + st * 8 + (idx & 7)];
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < kFillIters; ++i) {
            const int idx = tid + 256 * i;
            if (idx < kStageQuads) {
                const int pixel = idx >> 3, cq = idx & 7, py = pixel / kSub, px = pixel - py * kSub;
                lds[buf * kStageUnits + cq * kPlaneUnits + ((px & 1) * kSub + py) * kRowUnits + (px >> 1)] = stg[i];
            }
        }
    };
    // this lane's patch origin inside a plane
    const int patch0 = (2 * ty) * kRowUnits + tx;

    stage_load(0);
    stage_store(0);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        if (st + 1 < nst) stage_load(st + 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4* plane = lds + buf * kStageUnits + (2 * g + h) * kPlaneUnits + patch0;
            f32x4 v[16];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int s = 0; s < 4; ++s) v[r * 4 + s] = plane[((s & 1) * kSub + r) * kRowUnits + (s >> 1)];
            f32x4 u[16];
            const f32x4* ug = up + (size_t)(st * 4 + g) * 16 * 64;
#pragma unroll
            for (int p = 0; p < 16; ++p) u[p] = ug[p * 64];
            // B^T d B in place: rows, then columns
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 d0 = v[0 + s], d1 = v[4 + s], d2 = v[8 + s], d3 = v[12 + s];
                v[0 + s] = d0 - d2;
                v[4 + s] = d1 + d2;
                v[8 + s] = d2 - d1;
                v[12 + s] = d1 - d3;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 e0 = v[4 * i], e1 = v[4 * i + 1], e2 = v[4 * i + 2], e3 = v[4 * i + 3];
                v[4 * i] = e0 - e2;
                v[4 * i + 1] = e1 + e2;
                v[4 * i + 2] = e2 - e1;
                v[4 * i + 3] = e1 - e3;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int p = 0; p < 16; ++p) acc[p] = aae::mfma_32x32x2(v[p][q], u[p][q], acc[p]);
        }
        if (st + 1 < nst) stage_store(buf ^ 1);
        __syncthreads();
    }

    // output transform A^T m A (A^T = [[1, 1, 1, 0], [0, 1, -1, -1]]) + bias + ReLU, element-wise on the accumulator registers
    const int n = n32 * 32 + m;
    const float bs = a.bias[n];
    float* ob = a.out + (size_t)b * kOut * kOut * a.N + n;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mt = (r & 3) + 8 * (r >> 2) + 4 * h;           // this register's tile of the wave's 32
        const int oy = 2 * (4 * mh + (mt >> 3)), ox = 2 * (mt & 7);
        float q0[4], q1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            q0[j] = acc[j][r] + acc[4 + j][r] + acc[8 + j][r];
            q1[j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
        }
        const float y00 = q0[0] + q0[1] + q0[2] + bs, y01 = q0[1] - q0[2] - q0[3] + bs;
        const float y10 = q1[0] + q1[1] + q1[2] + bs, y11 = q1[1] - q1[2] - q1[3] + bs;
        ob[((size_t)(oy + 0) * kOut + ox + 0) * a.N] = fmaxf(y00, 0.f);
        ob[((size_t)(oy + 0) * kOut + ox + 1) * a.N] = fmaxf(y01, 0.f);
        ob[((size_t)(oy + 1) * kOut + ox + 0) * a.N] = fmaxf(y10, 0.f);
        ob[((size_t)(oy + 1) * kOut + ox + 1) * a.N] = fmaxf(y11, 0.f);
    }
}

// The same tiling with the 16 points of a (32 tiles x 32 channels) block split over TWO waves (point rows 0-1 / 2-3): 8 waves per
// block, 128 accumulator registers per wave, so two waves share a SIMD and one's patch reads, transform and weight loads run under
// the other's MFMAs (the one-wave form above issues its loads and waits for them in front of every 64 MFMAs).  A wave reads only the
// three patch rows its point rows need (rows 0-2 / 1-3).  The two halves of the output transform meet through LDS once per block.
constexpr int kFillIters8 = (kStageQuads + 511) / 512;    // 6

template <int ABLATE = 0>      // timing experiments (results wrong): 1 = the weight fragments of every group from one address, 2 = the patch of every group from one plane,
                               // + 16 = no patch read / transform after the first group, + 32 = no weight loads, + 64 = no stage fill, + 128 = no barrier in the loop
__global__ __launch_bounds__(512) void wino_fused8_kernel(Args a) {
    AAE_DYN_SMEM(smem_raw);
    f32x4* lds = reinterpret_cast<f32x4*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mh = wave & 1, nh = (wave >> 1) & 1, ph = wave >> 2, m = lane & 31, h = lane >> 5;
    const int ty = 4 * mh + (m >> 3), tx = m & 7;
    const int nbn = a.N / 64;
    const int nb = blockIdx.x % nbn, b = blockIdx.x / nbn;
    const int n32 = nb * 2 + nh, KG = a.C / 8, nst = a.C / 32, cq_per_pixel = a.C / 4;
    const f32x4* src = reinterpret_cast<const f32x4*>(a.s) + (size_t)b * kSub * kSub * cq_per_pixel;
    const f32x4* up = reinterpret_cast<const f32x4*>(a.U) + (size_t)n32 * KG * 16 * 64 + (size_t)ph * 8 * 64 + h * 32 + m;

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // the fill of the next stage in two halves (three float4 per thread each), so that only 12 staging registers are live at a time:
    // its buffer is free for the whole of the current stage (the barrier behind the previous one)
    constexpr int kHalf = kFillIters8 / 2;
    f32x4 stg[kHalf];
    auto stage_load = [&](int st, int half) {
#pragma unroll
        for (int i = 0; i < kHalf; ++i) {
            const int idx = tid + 512 * (half * kHalf + i);
            if (idx < kStageQuads) stg[i] = src[(size_t)(idx >> 3) * cq_per_pixel + st * 8 + (idx & 7)];
        }
    };
    auto stage_store = [&](int buf, int half) {
#pragma unroll
        for (int i = 0; i < kHalf; ++i) {
            const int idx = tid + 512 * (half * kHalf + i);
            if (idx < kStageQuads) {
                const int pixel = idx >> 3, cq = idx & 7, py = pixel / kSub, px = pixel - py * kSub;
                lds[buf * kStageUnits + cq * kPlaneUnits + ((px & 1) * kSub + py) * kRowUnits + (px >> 1)] = stg[i];
            }
        }
    };
    // patch rows this wave reads, in the order (y0, y1, y2) that makes both halves the same arithmetic:
    //   rows of B^T d:  v0 = y0 - y2,  v1 = y2 + sg y1      ph = 0: (row0, row1, row2), sg = +1 -> (d0 - d2, d1 + d2)
    //                                                      ph = 1: (row2, row3, row1), sg = -1 -> (d2 - d1, d1 - d3)
    const int patch0 = (2 * ty) * kRowUnits + tx;
    const int row_of[3] = {ph == 0 ? 0 : 2, ph == 0 ? 1 : 3, ph == 0 ? 2 : 1};
    const float sg = ph == 0 ? 1.f : -1.f;

    f32x4 u[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) u[p] = up[p * 64];
    stage_load(0, 0);
    stage_store(0, 0);
    stage_load(0, 1);
    stage_store(0, 1);
    __syncthreads();
    f32x4 v[8];
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        const bool more = st + 1 < nst && !(ABLATE & 64);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int gi = st * 4 + g;
            if (more && (g & 1) == 0) stage_load(st + 1, g >> 1);
            if (!(ABLATE & 16) || gi == 0) {
            const f32x4* plane = lds + buf * kStageUnits + (2 * (ABLATE == 2 ? 0 : g) + h) * kPlaneUnits + patch0;
            f32x4 d[12];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 4; ++s) d[r * 4 + s] = plane[((s & 1) * kSub + row_of[r]) * kRowUnits + (s >> 1)];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                v[s] = sub4(d[s], d[8 + s]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 + s][e] = fmaf(sg, d[4 + s][e], d[8 + s][e]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 e0 = v[4 * i], e1 = v[4 * i + 1], e2 = v[4 * i + 2], e3 = v[4 * i + 3];
                v[4 * i] = sub4(e0, e2);
                v[4 * i + 1] = add4(e1, e2);
                v[4 * i + 2] = sub4(e2, e1);
                v[4 * i + 3] = sub4(e1, e3);
            }
            }
            // two points at a time (their accumulators alternate), and as soon as a pair is through, ITS weight registers take the next
            // group's fragments: the global loads of group t + 1 fly under the MFMAs of group t without a second set of registers
            const f32x4* un = up + (size_t)(ABLATE == 1 ? 0 : gi + 1) * 16 * 64;
            const bool next = gi + 1 < 4 * nst && !(ABLATE & 32);
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[2 * pp] = aae::mfma_32x32x2(v[2 * pp][q], u[2 * pp][q], acc[2 * pp]);
                    acc[2 * pp + 1] = aae::mfma_32x32x2(v[2 * pp + 1][q], u[2 * pp + 1][q], acc[2 * pp + 1]);
                }
                if (next) {
                    u[2 * pp] = un[(2 * pp) * 64];
                    u[2 * pp + 1] = un[(2 * pp + 1) * 64];
                }
            }
            if (more && (g & 1) == 1) stage_store(buf ^ 1, g >> 1);
        }
        if (!(ABLATE & 128)) __syncthreads();
    }

    // output transform: rows of A^T m split over the two waves (q0 = m0 + m1 | m2, q1 = m1 | -m2 - m3), columns applied by each, the
    // upper half hands its four partial outputs per register over through LDS (the stage buffers are free after the last barrier)
    float* xch = reinterpret_cast<float*>(smem_raw) + (size_t)(wave & 3) * 64 * 64;
    auto partial = [&](int r, float (&y)[4]) {
        float q0[4], q1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            q0[j] = ph == 0 ? acc[j][r] + acc[4 + j][r] : acc[j][r];
            q1[j] = ph == 0 ? acc[4 + j][r] : -acc[j][r] - acc[4 + j][r];
        }
        y[0] = q0[0] + q0[1] + q0[2];
        y[1] = q0[1] - q0[2] - q0[3];
        y[2] = q1[0] + q1[1] + q1[2];
        y[3] = q1[1] - q1[2] - q1[3];
    };
    if (ph == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float y[4];
            partial(r, y);
#pragma unroll
            for (int k = 0; k < 4; ++k) xch[(r * 4 + k) * 64 + lane] = y[k];
        }
    }
    __syncthreads();
    if (ph == 0) {
        const int n = n32 * 32 + m;
        const float bs = a.bias[n];
        float* ob = a.out + (size_t)b * kOut * kOut * a.N + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mt = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int oy = 2 * (4 * mh + (mt >> 3)), ox = 2 * (mt & 7);
            float y[4];
            partial(r, y);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float val = y[k] + xch[(r * 4 + k) * 64 + lane] + bs;
                ob[((size_t)(oy + (k >> 1)) * kOut + ox + (k & 1)) * a.N] = fmaxf(val, 0.f);
            }
        }
    }
}

constexpr int kRows4 = 10;                                  // window rows of four tile rows
constexpr int kPlaneUnits4 = 2 * kRows4 * kRowUnits + 1;
constexpr int kStageUnits4 = 8 * kPlaneUnits4;
constexpr int kSmemBytes4 = 2 * kStageUnits4 * 16;          // 61 696 B: two blocks per CU
constexpr int kStageQuads4 = kRows4 * kSub * 8;             // 1440
constexpr int kFillIters4 = (kStageQuads4 + 255) / 256;     // 6

// The 8-wave form cut in two: block = 4 waves = 32 tiles (four tile rows of the image) x 64 channels, point rows split over wave pairs as
// above; 61 KB of LDS and 256 registers per wave, so TWO independent blocks share a CU -- their prologues, epilogues and barrier waits
// fall under each other's MFMAs.
__global__ __launch_bounds__(256, 2) void wino_fused4x2_kernel(Args a) {
    AAE_DYN_SMEM(smem_raw);
    f32x4* lds = reinterpret_cast<f32x4*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nh = wave & 1, ph = wave >> 1, m = lane & 31, h = lane >> 5;
    const int ty = m >> 3, tx = m & 7;                      // tile inside the block's four tile rows
    const int nbn = a.N / 64;
    const int nb = blockIdx.x % nbn, mb = (blockIdx.x / nbn) & 1, b = blockIdx.x / (2 * nbn);
    const int n32 = nb * 2 + nh, KG = a.C / 8, nst = a.C / 32, cq_per_pixel = a.C / 4;
    const f32x4* src = reinterpret_cast<const f32x4*>(a.s) + ((size_t)b * kSub + 8 * mb) * kSub * cq_per_pixel;
    const f32x4* up = reinterpret_cast<const f32x4*>(a.U) + (size_t)n32 * KG * 16 * 64 + (size_t)ph * 8 * 64 + h * 32 + m;

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // the fill of the next stage in two halves (three float4 per thread each), so that only 12 staging registers are live at a time:
    // its buffer is free for the whole of the current stage (the barrier behind the previous one)
    constexpr int kHalf = kFillIters4 / 2;
    f32x4 stg[kHalf];
    auto stage_load = [&](int st, int half) {
#pragma unroll
        for (int i = 0; i < kHalf; ++i) {
            const int idx = tid + 256 * (half * kHalf + i);
            if (idx < kStageQuads4) stg[i] = src[(size_t)(idx >> 3) * cq_per_pixel + st * 8 + (idx & 7)];
        }
    };
    auto stage_store = [&](int buf, int half) {
#pragma unroll
        for (int i = 0; i < kHalf; ++i) {
            const int idx = tid + 256 * (half * kHalf + i);
            if (idx < kStageQuads4) {
                const int pixel = idx >> 3, cq = idx & 7, py = pixel / kSub, px = pixel - py * kSub;
                lds[buf * kStageUnits4 + cq * kPlaneUnits4 + ((px & 1) * kRows4 + py) * kRowUnits + (px >> 1)] = stg[i];
            }
        }
    };
    // patch rows this wave reads, in the order (y0, y1, y2) that makes both halves the same arithmetic:
    //   rows of B^T d:  v0 = y0 - y2,  v1 = y2 + sg y1      ph = 0: (row0, row1, row2), sg = +1 -> (d0 - d2, d1 + d2)
    //                                                      ph = 1: (row2, row3, row1), sg = -1 -> (d2 - d1, d1 - d3)
    const int patch0 = (2 * ty) * kRowUnits + tx;
    const int row_of[3] = {ph == 0 ? 0 : 2, ph == 0 ? 1 : 3, ph == 0 ? 2 : 1};
    const float sg = ph == 0 ? 1.f : -1.f;

    f32x4 u[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) u[p] = up[p * 64];
    stage_load(0, 0);
    stage_store(0, 0);
    stage_load(0, 1);
    stage_store(0, 1);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        const bool more = st + 1 < nst;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (more && (g & 1) == 0) stage_load(st + 1, g >> 1);
            const f32x4* plane = lds + buf * kStageUnits4 + (2 * g + h) * kPlaneUnits4 + patch0;
            f32x4 d[12];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s = 0; s < 4; ++s) d[r * 4 + s] = plane[((s & 1) * kRows4 + row_of[r]) * kRowUnits + (s >> 1)];
            f32x4 v[8];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                v[s] = sub4(d[s], d[8 + s]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 + s][e] = fmaf(sg, d[4 + s][e], d[8 + s][e]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 e0 = v[4 * i], e1 = v[4 * i + 1], e2 = v[4 * i + 2], e3 = v[4 * i + 3];
                v[4 * i] = sub4(e0, e2);
                v[4 * i + 1] = add4(e1, e2);
                v[4 * i + 2] = sub4(e2, e1);
                v[4 * i + 3] = sub4(e1, e3);
            }
            // two points at a time (their accumulators alternate), and as soon as a pair is through, ITS weight registers take the next
            // group's fragments: the global loads of group t + 1 fly under the MFMAs of group t without a second set of registers
            const int gi = st * 4 + g;
            const f32x4* un = up + (size_t)(gi + 1) * 16 * 64;
            const bool next = gi + 1 < 4 * nst;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[2 * pp] = aae::mfma_32x32x2(v[2 * pp][q], u[2 * pp][q], acc[2 * pp]);
                    acc[2 * pp + 1] = aae::mfma_32x32x2(v[2 * pp + 1][q], u[2 * pp + 1][q], acc[2 * pp + 1]);
                }
                if (next) {
                    u[2 * pp] = un[(2 * pp) * 64];
                    u[2 * pp + 1] = un[(2 * pp + 1) * 64];
                }
            }
            if (more && (g & 1) == 1) stage_store(buf ^ 1, g >> 1);
        }
        __syncthreads();
    }

    // output transform: rows of A^T m split over the two waves (q0 = m0 + m1 | m2, q1 = m1 | -m2 - m3), columns applied by each, the
    // upper half hands its four partial outputs per register over through LDS (the stage buffers are free after the last barrier)
    float* xch = reinterpret_cast<float*>(smem_raw) + (size_t)(wave & 1) * 64 * 64;
    auto partial = [&](int r, float (&y)[4]) {
        float q0[4], q1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            q0[j] = ph == 0 ? acc[j][r] + acc[4 + j][r] : acc[j][r];
            q1[j] = ph == 0 ? acc[4 + j][r] : -acc[j][r] - acc[4 + j][r];
        }
        y[0] = q0[0] + q0[1] + q0[2];
        y[1] = q0[1] - q0[2] - q0[3];
        y[2] = q1[0] + q1[1] + q1[2];
        y[3] = q1[1] - q1[2] - q1[3];
    };
    if (ph == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float y[4];
            partial(r, y);
#pragma unroll
            for (int k = 0; k < 4; ++k) xch[(r * 4 + k) * 64 + lane] = y[k];
        }
    }
    __syncthreads();
    if (ph == 0) {
        const int n = n32 * 32 + m;
        const float bs = a.bias[n];
        float* ob = a.out + (size_t)b * kOut * kOut * a.N + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mt = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int oy = 2 * (4 * mb + (mt >> 3)), ox = 2 * (mt & 7);
            float y[4];
            partial(r, y);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float val = y[k] + xch[(r * 4 + k) * 64 + lane] + bs;
                ob[((size_t)(oy + (k >> 1)) * kOut + ox + (k & 1)) * a.N] = fmaxf(val, 0.f);
            }
        }
    }
}

}  // namespace wf
