// Implicit-GEMM convolution / dense layer on the fp32 matrix cores.
//
// Replaces tf.layers.conv2d(k, stride, 'same', relu) [+ inference batch-norm]
// of /root/reference/auto_pose/ae/encoder.py:41-52 for layers whose Cin is a
// multiple of 32, and (as a 1x1 "conv" over a 1x1 image) tf.layers.dense of
// encoder.py:62-66.
//
//   C[M, N] = A[M, K] * W[K, N],  M = B*Ho*Wo, N = Cout, K = KS*KS*Cin
//   A is never materialised: row m = (b, oh, ow), k = (kh, kw, ci) reads
//   x[b, oh*S-pt+kh, ow*S-pl+kw, ci] (zero outside the image: TF 'SAME').
//   W is pre-packed on the host as [K/4][CoutPad][4] (CoutPad = 128-multiple) in the
//   kernel's K order: k' = (cc*KS*KS + kh*KS + kw)*32 + j for channel ci = cc*32 + j.
//
// Block = 256 threads = 4 waves (2x2), tile 128x128, K-slab 32 (one tap, 32
// channels), each wave 64x64 = 2x2 accumulators of v_mfma_f32_32x32x2_f32.
// Register-staged double buffering: slab t+1 is in flight in VGPRs while slab t
// is consumed from LDS; one barrier per slab.  64 MFMAs (4096 SIMD cycles) per
// slab per wave against 8 global 16-B loads per thread.
#pragma once

namespace aae {

struct ConvIgemmArgs {
    const float* x;         // [B, H, W, Cin] NHWC
    unsigned x_bytes;       // size of x in bytes (< 4 GiB: bounds-checked buffer view)
    const float* wp;        // [K/4][CoutPad][4]
    const float* bias;      // [Cout]            (epilogue mode only)
    const float* bn_scale;  // [Cout] or nullptr
    const float* bn_shift;  // [Cout] or nullptr
    float* out;             // epilogue: [M][Cout]; split-K: [splits][M][Cout]
    int H, W, Cin, Ho, Wo, Cout, CoutPad;
    int KS, S, pt, pl;
    int M;                  // B*Ho*Wo
    int slabs_total;        // KS*KS*Cin/32
    int slabs_per_split;
    int num_mt, num_nt, splits;
    int relu;
    int stagger;            // kcycles of start delay for every second block generation (0 = off)
    unsigned wp_bytes;      // size of wp (the LDS-DMA variant reads it through a buffer view)
    // SCATTER mode (decoder: 2x nearest-neighbour upsampling folded into four phase problems,
    // kernels/decoder_f32.h): the grid's third index is the output phase (py, px) instead of a
    // K split; phase ph reads weights wp + ph*wp_phase_floats, pads by ph_pt[py] / ph_pl[px] and
    // stores row (b, oh, ow) at out[b][2*oh+py][2*ow+px][:] of a [B, 2*Ho, 2*Wo, Cout] tensor.
    long long wp_phase_floats;
    int ph_pt[2], ph_pl[2];
};

constexpr int kConvIgemmSmem = 2 * (kSlabFloatsA + 8 * 128 * 4) * 4;   // 64 KiB

// DMA = true: the operand slabs go global -> LDS with buffer_load_dwordx4 ... lds (lds_dma16)
// instead of through staging registers and ds_write_b128.  A DMA piece is lane-linear in LDS,
// so the A image's XOR swizzle moves to the source side: the lane that owns physical slot ps
// of row r fetches logical slot ps ^ ((r>>1)&7).  Slab t+1 is issued under group 0 of slab t
// into the other buffer (free since the barrier of slab t-1) and is waited for (vmcnt) by
// every wave right before the barrier of slab t.  Same MFMA sequence, bit-identical results.
// TAG only makes the symbol unique: the host instantiates the un-split DMA kernel once per encoder layer
// (TAG = layer index) so that rocprofv3 --stats reports conv2 / conv3 / conv4 as separate rows.
//
// BREG = true (needs DMA): the weight operand skips LDS altogether.  The packed weights [K/4][CoutPad][4] are
// already in MFMA B-fragment order, so lane (i, h) of a wave loads its two 16-B fragments of k-group c
// straight from global memory (L2-resident: every M tile re-reads them) into registers, one slab ahead, right
// after the MFMAs of group c have consumed the previous ones.  Per wave and slab that is 4 DMA pieces (A only)
// + 8 plain loads instead of 8 DMA pieces + 8 LDS fragment reads; an LDS-DMA piece costs the issuing wave
// well over 100 cycles, a plain load a fraction of that.  LDS shrinks to the two A buffers (32 KB).
// The slab barrier is a bare s_barrier behind a counted vmcnt that covers the DMA pieces only.
// NW (BREG only): 32-column MFMA tiles per wave.  2 -> 128 x 128 block tile; 4 -> 128 x 256 (each wave 64 x 128,
// 8 accumulators): half the A bytes, DMA pieces and LDS fragment reads per MFMA, at two instead of three waves
// per SIMD.
template <bool SPLITK, bool DMA = false, bool SCATTER = false, int TAG = 0, bool BREG = false, int NW = 2>
__global__ __launch_bounds__(256, NW == 4 ? 2 : 1) void conv_igemm_f32_kernel(const ConvIgemmArgs p) {
    static_assert(NW == 2 || BREG, "wide wave tiles exist for the weights-to-registers variant only");
    AAE_DYN_SMEM(smem_raw);
    float* As = reinterpret_cast<float*>(smem_raw);            // [2][128*32]
    float* Bs = As + 2 * kSlabFloatsA;                         // [2][8*128*4]

    const int tid = threadIdx.x, lane = tid & 63, wave = DMA ? wave_uniform(tid >> 6) : tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int nblk = p.num_mt * p.num_nt * p.splits;
    const int L = xcd_remap(blockIdx.x, nblk);
    const int nt = L % p.num_nt;
    const int mt = (L / p.num_nt) % p.num_mt;
    const int split = L / (p.num_nt * p.num_mt);
    // split-K: split s walks slabs [s*total/splits, (s+1)*total/splits) -- sizes differ by at most one slab
    const int slab0 = SCATTER ? 0 : (int)((long long)split * p.slabs_total / p.splits);
    const int slab1 = SCATTER ? p.slabs_total : (int)((long long)(split + 1) * p.slabs_total / p.splits);
    const int pt = SCATTER ? p.ph_pt[split >> 1] : p.pt;
    const int pl = SCATTER ? p.ph_pl[split & 1] : p.pl;
    const float* wp = SCATTER ? p.wp + split * p.wp_phase_floats : p.wp;

    // ---- A loader: thread -> 4 (row, slot) pairs, row = tid/8 + 32*q -------
    const int a_row = tid >> 3;
    const int a_slot = DMA ? (tid & 7) ^ ((a_row >> 1) & 7) : tid & 7;
    // The activation tensor is read through a bounds-checked buffer view: taps that fall in
    // the SAME padding (or rows >= M) get the out-of-range offset and the hardware returns
    // zeros -- no branch around the load, so nothing forces an early vmcnt wait.
    const buffer_rsrc xbuf = make_buffer(p.x, p.x_bytes);
    unsigned a_off[4];          // byte offset of (b, ih0, iw0, slot*4); may wrap for ih0/iw0 < 0, only used when in range
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
        a_ih0[q] = oh * p.S - pt;
        a_iw0[q] = ow * p.S - pl;
        a_off[q] = (unsigned)(((((long long)b * p.H + a_ih0[q]) * p.W + a_iw0[q]) * (long long)p.Cin + a_slot * 4) * 4);
    }
    // ---- B loader: thread -> 4 (slot, col) pairs, idx = tid + 256*q ----------
    const float* b_ptr = wp + ((long long)nt * 128 + (tid & 127)) * 4 + (long long)(tid >> 7) * p.CoutPad * 4;
    const long long b_slot_stride = (long long)p.CoutPad * 4;      // floats per k-slot row
    const buffer_rsrc wbuf = make_buffer(wp, DMA ? p.wp_bytes : 0u);
    const unsigned b_off0 = (unsigned)(((tid >> 7) * p.CoutPad + nt * 128 + (tid & 127)) * 16);
    const unsigned b_piece_stride = (unsigned)(2 * p.CoutPad * 16);

    // K order: 32-channel chunk outermost, then kh, then kw (fastest).  For one channel chunk
    // the KS*KS taps of an M tile touch the same few input rows, so the working set the
    // co-resident tiles of an XCD keep re-reading stays inside its 4 MB L2.
    const int taps = p.KS * p.KS;
    int cc = slab0 / taps;
    const int tap = slab0 - cc * taps;
    int kh = tap / p.KS, kw = tap - kh * p.KS;
    // Blocks p and p+256 share a CU (observed dispatch) and would otherwise run in lockstep:
    // identical work, so their load/barrier phases coincide and the matrix pipe idles in both.
    // Delaying every second generation by about half a slab interleaves them.
    if (p.stagger > 0 && ((blockIdx.x >> 8) & 1)) sleep_kcycles(p.stagger);

    // Loader state.  fetch_piece(slab, q) computes one A address + issues one A and one B
    // 16-B load; stash_piece(buf, q) stores that pair into the LDS slab.  They are issued in
    // four pieces so they can sit in the shadow of individual MFMAs (see the main loop).
    f32x4 ra[4], rb[4];
    unsigned tap_off = 0;
    auto fetch_piece = [&](int slab, int q, int buf) {
        if (q == 0) tap_off = (unsigned)(((kh * p.W + kw) * p.Cin + cc * 32) * 4);
        const bool ok = a_ok[q] && (unsigned)(a_ih0[q] + kh) < (unsigned)p.H && (unsigned)(a_iw0[q] + kw) < (unsigned)p.W;
        if (DMA) {
            lds_dma16(xbuf, ok ? a_off[q] + tap_off : kOobOffset, As + buf * kSlabFloatsA + (32 * q + 8 * wave) * kBK);
            lds_dma16(wbuf, b_off0 + (unsigned)(slab * 4 + q) * b_piece_stride, Bs + buf * (8 * 128 * 4) + (256 * q + 64 * wave) * 4);
        } else {
            ra[q] = buffer_load4(xbuf, ok ? a_off[q] + tap_off : kOobOffset);
            rb[q] = *reinterpret_cast<const f32x4*>(b_ptr + ((long long)slab * 8 + 2 * q) * b_slot_stride);
        }
        if (q == 3) {                            // advance the (cc, kh, kw) counters to the next slab
            if (++kw == p.KS) { kw = 0; if (++kh == p.KS) { kh = 0; ++cc; } }
        }
    };
    auto stash_piece = [&](int buf, int q) {
        if (DMA) return;
        lds_write4(As + buf * kSlabFloatsA + a_slab_off(a_row + 32 * q, a_slot), ra[q]);
        lds_write4(Bs + buf * (8 * 128 * 4) + (tid + 256 * q) * 4, rb[q]);
    };

    f32x16 acc[2][NW];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NW; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    if constexpr (BREG) {
      if (slab0 < slab1) {
        const int i = lane & 31, h = lane >> 5;
        const unsigned bw_off = (unsigned)(((h * p.CoutPad) + nt * (64 * NW) + wn * (32 * NW) + i) * 16);
        f32x4 bw[4][NW];                         // B fragments of the 4 k-groups of one slab
        auto load_bw = [&](int slab, int c) {
#pragma unroll
            for (int ni = 0; ni < NW; ++ni)
                bw[c][ni] = buffer_load4(wbuf, bw_off + (unsigned)(((slab * 8 + 2 * c) * p.CoutPad + 32 * ni) * 16));
        };
        auto load_fa = [&](const float* A, int c, f32x4 (&fa)[2]) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) fa[mi] = lds_read4(A + a_slab_off(wm * 64 + 32 * mi + i, 2 * c + h));
        };
        auto dma_a = [&](int q, int buf) {
            if (q == 0) tap_off = (unsigned)(((kh * p.W + kw) * p.Cin + cc * 32) * 4);
            const bool ok = a_ok[q] && (unsigned)(a_ih0[q] + kh) < (unsigned)p.H && (unsigned)(a_iw0[q] + kw) < (unsigned)p.W;
            lds_dma16(xbuf, ok ? a_off[q] + tap_off : kOobOffset, As + buf * kSlabFloatsA + (32 * q + 8 * wave) * kBK);
            if (q == 3) {
                if (++kw == p.KS) { kw = 0; if (++kh == p.KS) { kh = 0; ++cc; } }
            }
        };
#pragma unroll
        for (int q = 0; q < 4; ++q) dma_a(q, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) load_bw(slab0, c);
        wait_dma_and_lds();
        block_barrier();
        int buf = 0;
        f32x4 fa0[2], fa1[2];
        load_fa(As, 0, fa0);
        load_fa(As, 1, fa1);
        for (int t = slab0; t < slab1; ++t) {
            const bool more = (t + 1) < slab1;
            const float* A = As + buf * kSlabFloatsA;
            const float* An = As + (buf ^ 1) * kSlabFloatsA;
#pragma unroll
            for (int q = 0; q < 4; ++q) {        // group 0: the A pieces of slab t+1 ride in its MFMA shadows
                sched_fence();
                frag_mfma_q<2, NW>(fa0, bw[0], q, acc);
                sched_fence();
                // memory instructions issue at raised wave priority: when both waves of the SIMD want the issue port the
                // one feeding the pipeline goes first (+0.3 % end to end, conv4 -1 %; measured A/B on one box)
                if (more) { wave_priority<1>(); dma_a(q, buf ^ 1); wave_priority<0>(); }
            }
            sched_fence();
            if (more) { wave_priority<1>(); load_bw(t + 1, 0); wave_priority<0>(); }         // bw[0] has been consumed
            load_fa(A, 2, fa0);
            sched_fence();
            frag_mfma<2, NW>(fa1, bw[1], acc);   // group 1
            sched_fence();
            if (more) { wave_priority<1>(); load_bw(t + 1, 1); wave_priority<0>(); }
            load_fa(A, 3, fa1);
            sched_fence();
            frag_mfma<2, NW>(fa0, bw[2], acc);   // group 2
            sched_fence();
            if (more) {
                { wave_priority<1>(); load_bw(t + 1, 2); wave_priority<0>(); }
                wait_dma_keep_and_lds<3 * NW>(); // the 4 DMA pieces have landed; the 3*NW weight loads issued after them stay in flight
            } else {
                wait_dma_and_lds();
            }
            block_barrier();
            sched_fence();
            if (more) load_fa(An, 0, fa0);
            sched_fence();
            frag_mfma<2, NW>(fa1, bw[3], acc);   // group 3
            sched_fence();
            if (more) {
                { wave_priority<1>(); load_bw(t + 1, 3); wave_priority<0>(); }
                load_fa(An, 1, fa1);
            }
            buf ^= 1;
        }
      }
    } else if (slab0 < slab1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) fetch_piece(slab0, q, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) stash_piece(0, q);
        if (DMA) wait_dma_and_lds();
        __syncthreads();
        int buf = 0;
        // Software pipeline over the 4 k-groups (4 q-steps x 4 MFMAs each) of every slab with two
        // fragment sets S0/S1 in registers.  The matrix pipe never waits on LDS, on address
        // arithmetic or on the barrier:
        //   g0 (S0): after each q-step one fetch piece of slab t+1 (address VALU + 2 loads)
        //            issues in the shadow of the q-step's last MFMA;   then S0 <- (t, c2)
        //   g1 (S1):                                                    then S1 <- (t, c3)
        //   g2 (S0): after each q-step one stash piece of slab t+1 (2 LDS stores; its global
        //            loads were issued ~3000 cycles earlier)
        //   barrier: slab t+1 visible, every wave has its slab-t fragments in registers
        //   S0 <- (t+1, c0);  g3 (S1) runs while that read is in flight;  S1 <- (t+1, c1)
        f32x4 fa0[2], fb0[2], fa1[2], fb1[2];
        frag_load<2, 2>(As, Bs, 128, wm * 64, wn * 64, lane, 0, fa0, fb0);
        frag_load<2, 2>(As, Bs, 128, wm * 64, wn * 64, lane, 1, fa1, fb1);
        for (int t = slab0; t < slab1; ++t) {
            const bool more = (t + 1) < slab1;
            const float* A = As + buf * kSlabFloatsA;
            const float* Bt = Bs + buf * (8 * 128 * 4);
            const float* An = As + (buf ^ 1) * kSlabFloatsA;
            const float* Btn = Bs + (buf ^ 1) * (8 * 128 * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {        // group 0
                sched_fence();
                frag_mfma_q<2, 2>(fa0, fb0, q, acc);
                sched_fence();
                if (more) fetch_piece(t + 1, q, buf ^ 1);
            }
            sched_fence();
            frag_load<2, 2>(A, Bt, 128, wm * 64, wn * 64, lane, 2, fa0, fb0);
            sched_fence();
            frag_mfma<2, 2>(fa1, fb1, acc);      // group 1
            sched_fence();
            frag_load<2, 2>(A, Bt, 128, wm * 64, wn * 64, lane, 3, fa1, fb1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {        // group 2
                sched_fence();
                frag_mfma_q<2, 2>(fa0, fb0, q, acc);
                sched_fence();
                if (more) stash_piece(buf ^ 1, q);
            }
            if (DMA) wait_dma_and_lds();
            __syncthreads();
            sched_fence();
            if (more) frag_load<2, 2>(An, Btn, 128, wm * 64, wn * 64, lane, 0, fa0, fb0);
            sched_fence();
            frag_mfma<2, 2>(fa1, fb1, acc);      // group 3
            sched_fence();
            if (more) frag_load<2, 2>(An, Btn, 128, wm * 64, wn * 64, lane, 1, fa1, fb1);
            buf ^= 1;
        }
    }

    // ---- epilogue -----------------------------------------------------------
    const int i = lane & 31;
#pragma unroll
    for (int ni = 0; ni < NW; ++ni) {
        const int n = nt * (64 * NW) + wn * (32 * NW) + ni * 32 + i;
        if (n >= p.Cout) continue;
        float bias = 0.f, sc = 1.f, sh = 0.f;
        if (!SPLITK) {
            bias = p.bias[n];
            if (p.bn_scale) { sc = p.bn_scale[n]; sh = p.bn_shift[n]; }
            // wait for these three loads ONCE: every store below sits in its own predicated basic block, and the
            // compiler otherwise puts s_waitcnt vmcnt(0) in front of each of them -- which also waits for the
            // previous STORE's acknowledge: 64-128 serialised round trips per wave.
            bias = pin_value(bias); sc = pin_value(sc); sh = pin_value(sh);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mt * kBM + wm * 64 + mi * 32 + acc_row(r, lane);
                if (m >= p.M) continue;
                float v = acc[mi][ni][r];
                if (SPLITK) {
                    p.out[((long long)split * p.M + m) * p.Cout + n] = v;
                } else {
                    v += bias;
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.bn_scale) v = v * sc + sh;
                    long long orow = m;
                    if (SCATTER) {
                        const int b = m / (p.Ho * p.Wo);
                        const int rem = m - b * (p.Ho * p.Wo);
                        const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
                        orow = ((long long)b * (2 * p.Ho) + 2 * oh + (split >> 1)) * (2 * p.Wo) + 2 * ow + (split & 1);
                    }
                    p.out[orow * p.Cout + n] = v;
                }
            }
        }
    }
}

// out[m][n] = epilogue(sum_s partial[s][m][n] + bias[n]) -- fixed summation
// order s = 0..splits-1, so results are run-to-run reproducible.
struct SplitKReduceArgs {
    const float* partial;   // [splits][M][Cout]
    const float* bias;
    const float* bn_scale;
    const float* bn_shift;
    float* out;             // [M][Cout]
    long long MN;           // M*Cout
    int Cout, splits, relu;
    int out_planes;         // 1: write two fp16 planes of v*out_scale (f32x3h activation format)
    float out_scale;
    int* sat_flag = nullptr;  // out_planes: sticky flag "a value left the fp16 pair range" (or nullptr)
};

// 512 threads = 8 split-groups x 64 consecutive outputs.  Group g sums the partials
// s = g, g+8, g+16, ... (4 independent loads in flight per thread), the 8 group sums are
// then added in order g = 0..7 -- a fixed tree, independent of launch geometry.
constexpr int kReduceGroups = 8;

__global__ __launch_bounds__(512) void splitk_reduce_kernel(const SplitKReduceArgs p) {
    AAE_DYN_SMEM(smem_raw);
    float* part = reinterpret_cast<float*>(smem_raw);            // [8][64]
    const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
    for (long long e0 = (long long)blockIdx.x * 64; e0 < p.MN; e0 += (long long)gridDim.x * 64) {
        const long long e = e0 + l;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        if (e < p.MN) {
            int s = g;
            for (; s + 3 * kReduceGroups < p.splits; s += 4 * kReduceGroups) {
                v0 += p.partial[(long long)s * p.MN + e];
                v1 += p.partial[(long long)(s + kReduceGroups) * p.MN + e];
                v2 += p.partial[(long long)(s + 2 * kReduceGroups) * p.MN + e];
                v3 += p.partial[(long long)(s + 3 * kReduceGroups) * p.MN + e];
            }
            for (; s < p.splits; s += kReduceGroups) v0 += p.partial[(long long)s * p.MN + e];
        }
        part[g * 64 + l] = (v0 + v1) + (v2 + v3);
        __syncthreads();
        if (g == 0 && e < p.MN) {
            float v = part[l];
#pragma unroll
            for (int k = 1; k < kReduceGroups; ++k) v += part[k * 64 + l];
            const int n = (int)(e % p.Cout);
            v += p.bias[n];
            if (p.relu) v = fmaxf(v, 0.f);
            if (p.bn_scale) v = v * p.bn_scale[n] + p.bn_shift[n];
            if (p.out_planes) {
                unsigned short hi, lo;
                split_f16_checked(v * p.out_scale, hi, lo, p.sat_flag);
                unsigned short* op = reinterpret_cast<unsigned short*>(p.out);
                op[x3h_pair_index(e)] = hi;
                op[x3h_pair_index(e) + 32] = lo;
            } else {
                p.out[e] = v;
            }
        }
        __syncthreads();
    }
}

// Few splits (<= kReduceGroups) over a large tile -- the mid-batch case (B = 4 ... 128): four consecutive
// elements per thread, every split's float4 in flight at once, no LDS, no barriers.  Same summation
// order as splitk_reduce_kernel (there group g holds split g alone and the groups are added in order),
// including its "+ 0.f" canonicalisation of each term, so both kernels give the same bits.
// Requires MN % 4 == 0 and Cout % 4 == 0.
// U chunks of four elements per thread (a block covers U consecutive 1024-element segments), all loads of all splits in
// flight before the first add: with one chunk per thread the kernel moved 50 MB in 17 us (B = 16, conv2: two splits), i.e.
// it was bound by the latency of two loads per thread, not by HBM.  SPLITS > 0: exactly that many splits (registers for
// SPLITS x U float4); SPLITS == 0: any number up to kReduceGroups, one chunk.
template <int SPLITS, int U>
__global__ __launch_bounds__(256) void splitk_reduce_small_kernel(const SplitKReduceArgs p) {
    constexpr int NS = SPLITS > 0 ? SPLITS : kReduceGroups;
    f32x4 t[NS][U];
    long long e[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        e[u] = (((long long)blockIdx.x * U + u) * 256 + threadIdx.x) * 4;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            t[s][u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (e[u] < p.MN && s < p.splits) t[s][u] = *reinterpret_cast<const f32x4*>(p.partial + (long long)s * p.MN + e[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (e[u] >= p.MN) continue;
        const int n = (int)(e[u] % p.Cout);
        const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + n);
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (p.bn_scale) {
            sc = *reinterpret_cast<const f32x4*>(p.bn_scale + n);
            sh = *reinterpret_cast<const f32x4*>(p.bn_shift + n);
        }
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = (t[0][u][j] + 0.f) + (0.f + 0.f);
#pragma unroll
            for (int s = 1; s < kReduceGroups; ++s) a += ((s < NS ? t[s < NS ? s : 0][u][j] : 0.f) + 0.f) + (0.f + 0.f);
            a += bias[j];
            if (p.relu) a = fmaxf(a, 0.f);
            if (p.bn_scale) a = a * sc[j] + sh[j];
            v[j] = a;
        }
        if (p.out_planes) {
            u16x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned short h16, l16;
                split_f16_checked(v[j] * p.out_scale, h16, l16, p.sat_flag);
                hi[j] = h16; lo[j] = l16;
            }
            unsigned short* op = reinterpret_cast<unsigned short*>(p.out);
            *reinterpret_cast<u16x4*>(op + x3h_pair_index(e[u])) = hi;           // (e % 4 == 0: the four halves stay inside one chunk)
            *reinterpret_cast<u16x4*>(op + x3h_pair_index(e[u]) + 32) = lo;
        } else {
            *reinterpret_cast<f32x4*>(p.out + e[u]) = v;
        }
    }
}

}  // namespace aae
