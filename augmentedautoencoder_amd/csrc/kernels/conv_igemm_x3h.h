// Implicit-GEMM convolution / dense layer in "f32x3h" split precision: fp32 in, fp32 out,
// contraction on the fp16 matrix cores (v_mfma_f32_32x32x16_f16, 16x the fp32 MFMA rate)
// with every operand carried as a (hi, lo) pair of halves and fp32 accumulation:
//
//     a*b  ~=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (a_lo*b_lo ~ 2^-22 relative, dropped)
//
// hi = rn16(x*2^s), lo = rn16(x*2^s - hi): the pair carries >= 22 significant bits, so the
// result stays inside the fp32-roundoff class the 1e-5 cosine tolerance asks for (measured
// against the fp64 oracle in tests/), at 16/3 = 5.3x the fp32 matrix-core peak.  This is an
// explicitly selected mode (encoder option "precision" = 1); the default path is exact fp32.
//
// Same layer semantics as conv_igemm_f32.h (/root/reference/auto_pose/ae/encoder.py:41-52,62-66).
// Differences in data layout:
//   * activations travel between layers as (hi, lo) pairs of halves of x * 2^act_shift (same bytes as
//     fp32), written by the producer's epilogue: [M][C/32][hi x 32 | lo x 32] (x3h_pair_index) -- one
//     128-B line per row of a K slab; the consumer's operand loads are straight 16-B copies, no
//     conversion in the K loop;
//   * weights are split on the host after scaling by 2^w_shift and packed per K-slab as
//     [8 slots][CoutPad][8 halves], slot = plane*4 + kgroup8.
//   * LDS slab images: A [128 rows][8 slots x 16 B] (hi slots 0-3, lo slots 4-7, XOR swizzle as
//     the fp32 kernel), B [8 slots][128 cols][16 B].
// K-slab = 32 channels of one tap = two MFMA k16-steps; per wave and slab 24 MFMAs.
// Register staging runs two slabs ahead of the LDS ring (global latency ~2 slabs of MFMA time).
#pragma once

#include <type_traits>

namespace aae {

struct ConvIgemmX3hArgs {
    const unsigned short* x;  // activation pairs [Min][Cin/32][hi x 32 | lo x 32] (halves of x*2^act_shift)
    unsigned x_bytes;         // Min*Cin*4
    const unsigned* wp;       // [slabs][8][CoutPad][4 dwords]
    const float* bias;
    const float* bn_scale;
    const float* bn_shift;
    void* out;                // OUT_PLANES: half pairs [M][Cout/32][64]; OUT_F32: float [M][Cout]; split-K: float partials
    int H, W, Cin, Ho, Wo, Cout, CoutPad;
    int KS, S, pt, pl;
    int M;
    int slabs_total, slabs_per_split;
    int num_mt, num_nt, splits;
    int relu;
    float inv_scale;          // 2^-(act_shift_in + w_shift)
    float out_scale;          // 2^act_shift_out (OUT_PLANES)
    unsigned wp_bytes;        // size of wp (LDS-DMA variant reads it through a buffer view)
    int* sat_flag;            // OUT_PLANES: sticky flag "an activation left the fp16 pair range" (or nullptr)
};

enum { X3H_OUT_F32 = 0, X3H_OUT_PLANES = 1, X3H_OUT_PARTIAL = 2 };

struct X3hFrag {              // one k16-step of a 64x64 wave tile
    u32x4 ah[2], al[2], bh[2], bl[2];
};

__device__ __forceinline__ void x3h_frag_load(const float* As, const float* Bs, int a_row0, int b_col0, int lane, int s,
                                              X3hFrag& f) {
    const int i = lane & 31, slot = 2 * s + (lane >> 5);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        f.ah[mi] = __builtin_bit_cast(u32x4, lds_read4(As + a_slab_off(a_row0 + 32 * mi + i, slot)));
        f.al[mi] = __builtin_bit_cast(u32x4, lds_read4(As + a_slab_off(a_row0 + 32 * mi + i, 4 + slot)));
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        f.bh[ni] = __builtin_bit_cast(u32x4, lds_read4(Bs + (slot * 128 + b_col0 + 32 * ni + i) * 4));
        f.bl[ni] = __builtin_bit_cast(u32x4, lds_read4(Bs + ((4 + slot) * 128 + b_col0 + 32 * ni + i) * 4));
    }
}

// the three products of one 32x32 tile (q = 2*mi + ni)
__device__ __forceinline__ void x3h_mfma_tile(const X3hFrag& f, int q, f32x16 (&acc)[2][2]) {
    const int mi = q >> 1, ni = q & 1;
    acc[mi][ni] = mfma_32x32x16_f16(f.al[mi], f.bh[ni], acc[mi][ni]);
    acc[mi][ni] = mfma_32x32x16_f16(f.ah[mi], f.bl[ni], acc[mi][ni]);
    acc[mi][ni] = mfma_32x32x16_f16(f.ah[mi], f.bh[ni], acc[mi][ni]);
}

// bias / ReLU / folded BN and the store of a wave's 64x64 accumulator tile in the layout OUT names
template <int OUT>
__device__ __forceinline__ void x3h_epilogue(const ConvIgemmX3hArgs& p, const f32x16 (&acc)[2][2], int mt, int nt, int split,
                                             int wm, int wn, int lane, int bm = kBM) {
    const int i = lane & 31;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int n = nt * 128 + wn * 64 + ni * 32 + i;
        if (n >= p.Cout) continue;
        float bias = 0.f, sc = 1.f, sh = 0.f;
        if (OUT != X3H_OUT_PARTIAL) {
            bias = p.bias[n];
            if (p.bn_scale) { sc = p.bn_scale[n]; sh = p.bn_shift[n]; }
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt * bm + wm * 64 + mi * 32 + acc_row(r, lane);
                if (m >= p.M) continue;
                float v = acc[mi][ni][r] * p.inv_scale;
                if (OUT == X3H_OUT_PARTIAL) {
                    reinterpret_cast<float*>(p.out)[((long long)split * p.M + m) * p.Cout + n] = v;
                } else {
                    v += bias;
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.bn_scale) v = v * sc + sh;
                    if (OUT == X3H_OUT_F32) {
                        reinterpret_cast<float*>(p.out)[(long long)m * p.Cout + n] = v;
                    } else {
                        unsigned short hi, lo;
                        split_f16_checked(v * p.out_scale, hi, lo, p.sat_flag);
                        unsigned short* o = reinterpret_cast<unsigned short*>(p.out);
                        const long long at = x3h_pair_index((long long)m * p.Cout + n);
                        o[at] = hi;
                        o[at + 32] = lo;
                    }
                }
            }
        }
    }
}

template <int OUT>
__global__ __launch_bounds__(256) void conv_igemm_x3h_kernel(const ConvIgemmX3hArgs p) {
    AAE_DYN_SMEM(smem_raw);
    float* As = reinterpret_cast<float*>(smem_raw);            // [2][128 rows][32 dwords]
    float* Bs = As + 2 * kSlabFloatsA;                         // [2][8 slots][128 cols][4 dwords]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int nblk = p.num_mt * p.num_nt * p.splits;
    const int L = xcd_remap(blockIdx.x, nblk);
    const int nt = L % p.num_nt;
    const int mt = (L / p.num_nt) % p.num_mt;
    const int split = L / (p.num_nt * p.num_mt);
    const int slab0 = (int)((long long)split * p.slabs_total / p.splits);        // balanced split-K ranges (sizes differ by <= 1 slab)
    const int slab1 = (int)((long long)(split + 1) * p.slabs_total / p.splits);
    const int nslab = slab1 - slab0;

    // ---- A loader: thread -> 4 (row, chunk) pairs; chunk = plane*4 + kgroup8 = LDS slot --------
    const int a_slot = tid & 7;
    const int a_row = tid >> 3;
    const buffer_rsrc xbuf = make_buffer(p.x, p.x_bytes);
    unsigned a_off[4];
    int a_ih0[4], a_iw0[4];
    bool a_ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = mt * kBM + a_row + 32 * q;
        a_ok[q] = m < p.M;
        const int mm = a_ok[q] ? m : 0;
        const int b = mm / (p.Ho * p.Wo);
        const int rem = mm - b * (p.Ho * p.Wo);
        const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
        a_ih0[q] = oh * p.S - p.pt;
        a_iw0[q] = ow * p.S - p.pl;
        a_off[q] = (unsigned)((((long long)b * p.H + a_ih0[q]) * p.W + a_iw0[q]) * (long long)p.Cin * 4) +
                   (unsigned)a_slot * 16u;          // hi k-groups 0-3, lo k-groups 4-7 of the 128-B chunk
    }
    // ---- B loader: idx = tid + 256*q -> slot = idx>>7, col = idx&127 ------------------------------
    const unsigned* b_ptr = p.wp + ((long long)nt * 128 + (tid & 127)) * 4 + (long long)(tid >> 7) * p.CoutPad * 4;
    const long long b_slot_stride = (long long)p.CoutPad * 4;      // dwords per slot row

    const int taps = p.KS * p.KS;
    int cc = slab0 / taps;
    const int tap0 = slab0 - cc * taps;
    int kh = tap0 / p.KS, kw = tap0 - kh * p.KS;

    u32x4 ra[2][4], rb[2][4];
    unsigned tap_off = 0;
    auto fetch_piece = [&](auto PAR, int slab, int q) {
        constexpr int P = decltype(PAR)::value;
        if (q == 0) tap_off = (unsigned)(((kh * p.W + kw) * p.Cin + cc * 32) * 4);
        const bool ok = a_ok[q] && (unsigned)(a_ih0[q] + kh) < (unsigned)p.H && (unsigned)(a_iw0[q] + kw) < (unsigned)p.W;
        ra[P][q] = __builtin_bit_cast(u32x4, buffer_load4(xbuf, ok ? a_off[q] + tap_off : kOobOffset));
        rb[P][q] = *reinterpret_cast<const u32x4*>(b_ptr + ((long long)slab * 8 + 2 * q) * b_slot_stride);
        if (q == 3) {
            if (++kw == p.KS) { kw = 0; if (++kh == p.KS) { kh = 0; ++cc; } }
        }
    };
    auto stash_piece = [&](auto PAR, int buf, int q) {
        constexpr int P = decltype(PAR)::value;
        lds_write4(As + buf * kSlabFloatsA + a_slab_off(a_row + 32 * q, a_slot), __builtin_bit_cast(f32x4, ra[P][q]));
        lds_write4(Bs + buf * (8 * 128 * 4) + (tid + 256 * q) * 4, __builtin_bit_cast(f32x4, rb[P][q]));
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    if (nslab > 0) {
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        // prologue: slab 0 -> R[0] -> LDS buffer 0; slab 1 -> R[1] (stays in registers)
#pragma unroll
        for (int q = 0; q < 4; ++q) fetch_piece(P0{}, slab0, q);
#pragma unroll
        for (int q = 0; q < 4; ++q) stash_piece(P0{}, 0, q);
        if (nslab > 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) fetch_piece(P1{}, slab0 + 1, q);
        }
        __syncthreads();
        X3hFrag f0, f1;
        x3h_frag_load(As, Bs, wm * 64, wn * 64, lane, 0, f0);
        x3h_frag_load(As, Bs, wm * 64, wn * 64, lane, 1, f1);

        // iteration `it` (slab it, LDS buffer it&1), P = it&1:
        //   step 0 (12 MFMAs): issue the global loads of slab it+2 into R[P], one piece per tile
        //   step 1 tiles 0,1 : store slab it+1 from R[P^1] (fetched last iteration) into buffer (it+1)&1
        //   barrier; f0 <- (slab it+1, step 0); step 1 tiles 2,3 cover that read; f1 <- (it+1, step 1)
        auto iteration = [&](auto PAR, int it) {
            constexpr int P = decltype(PAR)::value;
            using Pc = std::integral_constant<int, P>;
            using Pn = std::integral_constant<int, P ^ 1>;
            const bool has1 = it + 1 < nslab, has2 = it + 2 < nslab;
            const float* An = As + (P ^ 1) * kSlabFloatsA;
            const float* Btn = Bs + (P ^ 1) * (8 * 128 * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sched_fence();
                x3h_mfma_tile(f0, q, acc);
                sched_fence();
                if (has2) fetch_piece(Pc{}, slab0 + it + 2, q);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                sched_fence();
                x3h_mfma_tile(f1, q, acc);
                sched_fence();
                if (has1) { stash_piece(Pn{}, P ^ 1, 2 * q); stash_piece(Pn{}, P ^ 1, 2 * q + 1); }
            }
            __syncthreads();
            sched_fence();
            if (has1) x3h_frag_load(An, Btn, wm * 64, wn * 64, lane, 0, f0);
            sched_fence();
            x3h_mfma_tile(f1, 2, acc);
            x3h_mfma_tile(f1, 3, acc);
            sched_fence();
            if (has1) x3h_frag_load(An, Btn, wm * 64, wn * 64, lane, 1, f1);
        };
        int it = 0;
        for (; it + 1 < nslab; it += 2) {
            iteration(P0{}, it);
            iteration(P1{}, it + 1);
        }
        if (it < nslab) iteration(P0{}, it);
    }

    x3h_epilogue<OUT>(p, acc, mt, nt, split, wm, wn, lane);
}

// ---- LDS-DMA variant ------------------------------------------------------------------------
// Same tiles, same LDS slab images, same arithmetic; the operand slabs travel global -> LDS with
// buffer_load_dwordx4 ... lds (no staging registers, no ds_write_b128: the 16-B LDS stores are
// what saturates the LDS path in the kernel above -- 13 cycles per wave-store against 4 per
// fragment read).  A DMA piece is lane-linear in LDS (lane l -> base + 16 l), so the XOR swizzle of
// the A image is applied on the source side: the lane that owns physical slot ps of row r fetches
// logical slot ps ^ ((r>>1)&7).
//
// Two LDS buffers, one barrier per slab, the DMA runs two slabs ahead:
//   iteration it (slab it, P = it&1; its fragments are already in registers):
//     step 0 (12 MFMAs on f0)
//     wait vmcnt/lgkmcnt, barrier   -> slab it+1 has landed in buffer P^1 (issued one iteration ago)
//                                      and every wave is done reading buffer P
//     read the fragments of slab it+1 from P^1 (all ds_reads first: nothing the compiler could
//     order behind a DMA), then issue the 8 DMA pieces of slab it+2 into buffer P, all under
//     step 1 (12 MFMAs on f1[P]).
//
// WM = wave rows of the block: 2 -> 4 waves, 128 x 128 tile, 64 KB of LDS (two blocks per CU);
//                              4 -> 8 waves, 256 x 128 tile, 96 KB (one block per CU, still two waves per
// SIMD).  The wide tile moves 25 % fewer operand bytes per MFMA (32 KB per 96 MFMAs -> 48 KB per 192) and
// issues 6 instead of 8 DMA instructions per wave and slab.  Measured at B = 256: neutral (conv2 358 vs 363,
// conv3 388 vs 381, conv4 390 vs 390 TF-equivalent), bit-identical; host option x3h_wide_min_blocks, off by
// default.  (Also measured and dropped: term-major MFMA order, single-MFMA interleave of reads/DMA, an L2
// prefetch two slabs beyond the DMA -- the latter cost 20 %: the DMA/texture-address issue path, at ~100+
// cycles of wave time per 1-KiB piece, is what bounds this kernel near 47 % of the fp16 matrix peak.)
// TAG only makes the symbol unique per encoder layer for rocprofv3 --stats (see conv_igemm_f32.h).
template <int WM> constexpr int x3h_dma_smem() { return 2 * (64 * WM * kBK + 8 * 128 * 4) * 4; }

template <int OUT, int TAG = 0, int WM = 2>
__global__ __launch_bounds__(128 * WM) void conv_igemm_x3h_dma_kernel(const ConvIgemmX3hArgs p) {
    constexpr int T = 128 * WM;                                // threads
    constexpr int BM = 64 * WM;                                // tile rows
    constexpr int kStageA = BM * kBK;                          // floats of one A slab image
    constexpr int NB = 1024 / T;                               // 16-B weight pieces per thread and slab
    AAE_DYN_SMEM(smem_raw);
    float* As = reinterpret_cast<float*>(smem_raw);            // [2][BM rows][32 dwords]
    float* Bs = As + 2 * kStageA;                              // [2][8 slots][128 cols][4 dwords]

    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int nblk = p.num_mt * p.num_nt * p.splits;
    const int L = xcd_remap(blockIdx.x, nblk);
    const int nt = L % p.num_nt;
    const int mt = (L / p.num_nt) % p.num_mt;
    const int split = L / (p.num_nt * p.num_mt);
    const int slab0 = (int)((long long)split * p.slabs_total / p.splits);        // balanced split-K ranges (sizes differ by <= 1 slab)
    const int slab1 = (int)((long long)(split + 1) * p.slabs_total / p.splits);
    const int nslab = slab1 - slab0;

    // ---- A pieces: piece q of wave w fills rows (T/8) q + 8w .. +7 (8 lanes per row) -----------------
    const int a_row = tid >> 3;                                    // + (T/8) q
    const int a_slot = (tid & 7) ^ ((a_row >> 1) & 7);             // logical slot for this LDS position
    const buffer_rsrc xbuf = make_buffer(p.x, p.x_bytes);
    const buffer_rsrc wbuf = make_buffer(p.wp, p.wp_bytes);
    unsigned a_off[4];
    int a_ih0[4], a_iw0[4];
    bool a_ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = mt * BM + a_row + (T / 8) * q;
        a_ok[q] = m < p.M;
        const int mm = a_ok[q] ? m : 0;
        const int b = mm / (p.Ho * p.Wo);
        const int rem = mm - b * (p.Ho * p.Wo);
        const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
        a_ih0[q] = oh * p.S - p.pt;
        a_iw0[q] = ow * p.S - p.pl;
        a_off[q] = (unsigned)((((long long)b * p.H + a_ih0[q]) * p.W + a_iw0[q]) * (long long)p.Cin * 4) +
                   (unsigned)a_slot * 16u;          // hi k-groups 0-3, lo k-groups 4-7 of the 128-B chunk
    }
    // ---- B pieces: idx = tid + T q -> slot = idx>>7, col = idx&127, LDS position idx (q < NB) --------
    const unsigned b_off0 = (unsigned)(((tid >> 7) * p.CoutPad + nt * 128 + (tid & 127)) * 16);
    const unsigned b_piece_stride = (unsigned)((T / 128) * p.CoutPad * 16);   // T/128 slot rows per piece

    const int taps = p.KS * p.KS;
    int cc = slab0 / taps;
    const int tap0 = slab0 - cc * taps;
    int kh = tap0 / p.KS, kw = tap0 - kh * p.KS;
    unsigned tap_off = 0;

    auto dma_piece = [&](int slab, int buf, int q) {
        if (q == 0) tap_off = (unsigned)(((kh * p.W + kw) * p.Cin + cc * 32) * 4);
        const bool ok = a_ok[q] && (unsigned)(a_ih0[q] + kh) < (unsigned)p.H && (unsigned)(a_iw0[q] + kw) < (unsigned)p.W;
        lds_dma16(xbuf, ok ? a_off[q] + tap_off : kOobOffset, As + buf * kStageA + ((T / 8) * q + 8 * wave) * kBK);
        if (q < NB)
            lds_dma16(wbuf, b_off0 + (unsigned)(slab * NB + q) * b_piece_stride, Bs + buf * (8 * 128 * 4) + (T * q + 64 * wave) * 4);
        if (q == 3) {
            if (++kw == p.KS) { kw = 0; if (++kh == p.KS) { kh = 0; ++cc; } }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    if (nslab > 0) {
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
#pragma unroll
        for (int q = 0; q < 4; ++q) dma_piece(slab0, 0, q);
        wait_dma_and_lds();
        __syncthreads();
        X3hFrag f0, f1[2];
        x3h_frag_load(As, Bs, wm * 64, wn * 64, lane, 0, f0);
        x3h_frag_load(As, Bs, wm * 64, wn * 64, lane, 1, f1[0]);
        sched_fence();
        if (nslab > 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dma_piece(slab0 + 1, 1, q);
        }

        auto iteration = [&](auto PAR, int it) {
            constexpr int P = decltype(PAR)::value;
            const bool has1 = it + 1 < nslab, has2 = it + 2 < nslab;
            const float* An = As + (P ^ 1) * kStageA;
            const float* Btn = Bs + (P ^ 1) * (8 * 128 * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sched_fence();
                x3h_mfma_tile(f0, q, acc);
            }
            sched_fence();
            if (has1) {
                wait_dma_and_lds();
                __syncthreads();
                sched_fence();
                x3h_frag_load(An, Btn, wm * 64, wn * 64, lane, 0, f0);
            }
            sched_fence();
            x3h_mfma_tile(f1[P], 0, acc);
            sched_fence();
            if (has1) x3h_frag_load(An, Btn, wm * 64, wn * 64, lane, 1, f1[P ^ 1]);
            sched_fence();
            x3h_mfma_tile(f1[P], 1, acc);
            sched_fence();
            if (has2) { dma_piece(slab0 + it + 2, P, 0); dma_piece(slab0 + it + 2, P, 1); }
            sched_fence();
            x3h_mfma_tile(f1[P], 2, acc);
            sched_fence();
            if (has2) { dma_piece(slab0 + it + 2, P, 2); dma_piece(slab0 + it + 2, P, 3); }
            sched_fence();
            x3h_mfma_tile(f1[P], 3, acc);
        };
        int it = 0;
        for (; it + 1 < nslab; it += 2) {
            iteration(P0{}, it);
            iteration(P1{}, it + 1);
        }
        if (it < nslab) iteration(P0{}, it);
    }
    x3h_epilogue<OUT>(p, acc, mt, nt, split, wm, wn, lane, BM);
}


// ---- 256 x 256 block tile, 8 waves of 64 x 128 ------------------------------------------------------------------
// The 128 x 128 kernel above is bound by the LDS port, not by the matrix pipe: per slab a wave issues 24 MFMAs
// (768 cycles) against 16 KiB of fragment reads and 8 DMA pieces; at two blocks per CU that is 83 B/clk of reads +
// 42 B/clk of DMA writes = 125 of the 128 B/clk an LDS delivers (0.45 of the fp16 matrix peak, every scheduling
// variant measured the same).  A 64 x 128 wave tile re-uses each A fragment for four N tiles instead of two: 24 KiB of
// reads and 8 DMA pieces per 48 MFMAs -> 62 + 21 = 84 B/clk at full MFMA rate.  Eight accumulator tiles leave 128
// registers for operands at two waves per SIMD, so the fragments are not kept as whole k16-step sets: the A fragments
// of the next step are double-buffered (2 x 16 registers), the B fragments of N tile ni are refreshed IN PLACE right
// after the six MFMAs that consumed them (4 x 8 registers) -- by the time the next step reaches that tile again
// 18 MFMAs have passed.
//   step j of slab t:  [group ni = 0..3:  6 MFMAs on (a_cur, b[ni]);  b[ni] <- (next step, ni)],  A_nxt loaded under group 0
//   the ONE barrier of a slab sits in its step 1 after group 0, in front of the first read of slab t+1's buffer:
//   every wave waits for its own DMA pieces of slab t+1 (vmcnt) before it; behind it every wave has finished reading
//   buffer t (its step-1 fragments were all fetched during step 0), so the pieces of slab t+2 go into buffer t.
// Same k-step and product order per accumulator as the 128 x 128 kernels: bit-identical results (tested).
constexpr int kX3hWideSmem = 2 * (256 * kBK + 8 * 256 * 4) * 4;         // 128 KiB: one block per CU

template <int OUT, int TAG = 0>
__global__ __launch_bounds__(512) void conv_igemm_x3h_wide_kernel(const ConvIgemmX3hArgs p) {
    constexpr int T = 512, BM = 256, BN = 256;
    constexpr int kStageA = BM * kBK;                          // floats of one A slab image
    constexpr int kStageB = 8 * BN * 4;
    AAE_DYN_SMEM(smem_raw);
    float* As = reinterpret_cast<float*>(smem_raw);            // [2][256 rows][32 dwords]
    float* Bs = As + 2 * kStageA;                              // [2][8 slots][256 cols][4 dwords]

    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                   // 4 x 2 waves of 64 x 128
    const int i = lane & 31, h = lane >> 5;

    const int nblk = p.num_mt * p.num_nt;
    const int L = xcd_remap(blockIdx.x, nblk);
    const int nt = L % p.num_nt;
    const int mt = L / p.num_nt;
    const int nslab = p.slabs_total;

    // ---- A pieces: piece q of a thread fills row tid/8 + 64 q, physical slot tid%8 (logical slot swizzled on the source side)
    const int a_row = tid >> 3;
    const int a_slot = (tid & 7) ^ ((a_row >> 1) & 7);
    const buffer_rsrc xbuf = make_buffer(p.x, p.x_bytes);
    const buffer_rsrc wbuf = make_buffer(p.wp, p.wp_bytes);
    unsigned a_off[4];
    int a_ihw[4];                                              // (ih0 << 16) | (iw0 & 0xffff): the window origin, may be negative
    bool a_ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int m = mt * BM + a_row + 64 * q;
        a_ok[q] = m < p.M;
        const int mm = a_ok[q] ? m : 0;
        const int b = mm / (p.Ho * p.Wo);
        const int rem = mm - b * (p.Ho * p.Wo);
        const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
        const int ih0 = oh * p.S - p.pt, iw0 = ow * p.S - p.pl;
        a_ihw[q] = (ih0 << 16) | (iw0 & 0xffff);
        a_off[q] = (unsigned)((((long long)b * p.H + ih0) * p.W + iw0) * (long long)p.Cin * 4) +
                   (unsigned)a_slot * 16u;          // hi k-groups 0-3, lo k-groups 4-7 of the 128-B chunk
    }
    // ---- B pieces: idx = tid + 512 q -> slot = idx >> 8, col = idx & 255; LDS position idx
    const unsigned b_off0 = (unsigned)(((tid >> 8) * p.CoutPad + nt * BN + (tid & 255)) * 16);
    const unsigned b_piece_stride = (unsigned)(2 * p.CoutPad * 16);               // two slot rows per piece

    const int taps = p.KS * p.KS;
    int cc = 0, kh = 0, kw = 0;
    unsigned tap_off = 0;
    auto dma_piece = [&](int slab, int buf, int q) {
        if (q == 0) tap_off = (unsigned)(((kh * p.W + kw) * p.Cin + cc * 32) * 4);
        const int ih = (a_ihw[q] >> 16) + kh, iw = (int)(short)(a_ihw[q] & 0xffff) + kw;
        const bool ok = a_ok[q] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        lds_dma16(xbuf, ok ? a_off[q] + tap_off : kOobOffset, As + buf * kStageA + (64 * q + 8 * wave) * kBK);
        lds_dma16(wbuf, b_off0 + (unsigned)(slab * 4 + q) * b_piece_stride, Bs + buf * kStageB + (T * q + 64 * wave) * 4);
        if (q == 3) {
            if (++kw == p.KS) { kw = 0; if (++kh == p.KS) { kh = 0; ++cc; } }
        }
    };
    (void)taps;

    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // fragment registers: A of the step in progress and of the next one, B per N tile (refreshed in place)
    u32x4 ah[2][2], al[2][2], bh[4], bl[4];
    auto load_a = [&](const float* A, int s, int set) {
        const int slot = 2 * s + h;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            ah[set][mi] = __builtin_bit_cast(u32x4, lds_read4(A + a_slab_off(wm * 64 + 32 * mi + i, slot)));
            al[set][mi] = __builtin_bit_cast(u32x4, lds_read4(A + a_slab_off(wm * 64 + 32 * mi + i, 4 + slot)));
        }
    };
    auto load_b = [&](const float* B, int s, int ni) {
        const int slot = 2 * s + h;
        bh[ni] = __builtin_bit_cast(u32x4, lds_read4(B + (slot * BN + wn * 128 + 32 * ni + i) * 4));
        bl[ni] = __builtin_bit_cast(u32x4, lds_read4(B + ((4 + slot) * BN + wn * 128 + 32 * ni + i) * 4));
    };
    auto mfma_group = [&](int set, int ni) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            acc[mi][ni] = mfma_32x32x16_f16(al[set][mi], bh[ni], acc[mi][ni]);
            acc[mi][ni] = mfma_32x32x16_f16(ah[set][mi], bl[ni], acc[mi][ni]);
            acc[mi][ni] = mfma_32x32x16_f16(ah[set][mi], bh[ni], acc[mi][ni]);
        }
    };

    if (nslab > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dma_piece(0, 0, q);
        wait_dma_and_lds();
        __syncthreads();
        load_a(As, 0, 0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) load_b(Bs, 0, ni);
        sched_fence();
        if (nslab > 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dma_piece(1, 1, q);
        }
        for (int t = 0; t < nslab; ++t) {
            const int buf = t & 1;
            const float* A = As + buf * kStageA;
            const float* B = Bs + buf * kStageB;
            const float* An = As + (buf ^ 1) * kStageA;
            const float* Bn = Bs + (buf ^ 1) * kStageB;
            const bool has1 = t + 1 < nslab, has2 = t + 2 < nslab;
            // ---- step 0 (fragments in set 0); the fragments of step 1 of this slab are fetched under it
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                sched_fence();
                mfma_group(0, ni);
                sched_fence();
                if (ni == 0) load_a(A, 1, 1);
                load_b(B, 1, ni);
            }
            // ---- step 1 (set 1); behind the barrier the fragments of step 0 of slab t+1 come from the other buffer
            sched_fence();
            mfma_group(1, 0);
            sched_fence();
            if (has1) {
                wait_dma_and_lds();
                __syncthreads();
                sched_fence();
                load_a(An, 0, 0);
                load_b(Bn, 0, 0);
            }
#pragma unroll
            for (int ni = 1; ni < 4; ++ni) {
                sched_fence();
                mfma_group(1, ni);
                sched_fence();
                if (has1) load_b(Bn, 0, ni);
                if (has2) dma_piece(t + 2, buf, ni - 1);          // buffer t is free: every wave is past its reads of slab t
            }
            sched_fence();
            if (has2) dma_piece(t + 2, buf, 3);
        }
    }

    // ---- epilogue: the wave's 64 x 128 tile
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n = nt * BN + wn * 128 + ni * 32 + i;
        if (n >= p.Cout) continue;
        const float bias = p.bias[n];
        float sc = 1.f, sh = 0.f;
        if (p.bn_scale) { sc = p.bn_scale[n]; sh = p.bn_shift[n]; }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt * BM + wm * 64 + mi * 32 + acc_row(r, lane);
                if (m >= p.M) continue;
                float v = acc[mi][ni][r] * p.inv_scale;
                v += bias;
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.bn_scale) v = v * sc + sh;
                if (OUT == X3H_OUT_F32) {
                    reinterpret_cast<float*>(p.out)[(long long)m * p.Cout + n] = v;
                } else {
                    unsigned short hi, lo;
                    split_f16_checked(v * p.out_scale, hi, lo, p.sat_flag);
                    unsigned short* o = reinterpret_cast<unsigned short*>(p.out);
                    const long long at = x3h_pair_index((long long)m * p.Cout + n);
                    o[at] = hi;
                    o[at + 32] = lo;
                }
            }
        }
    }
}

}  // namespace aae
