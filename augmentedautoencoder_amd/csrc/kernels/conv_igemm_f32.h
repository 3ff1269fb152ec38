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
//   W is pre-packed on the host as [K/4][CoutPad][4] (CoutPad = 128-multiple).
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
};

constexpr int kConvIgemmSmem = 2 * (kSlabFloatsA + 8 * 128 * 4) * 4;   // 64 KiB

template <bool SPLITK>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvIgemmArgs p) {
    AAE_DYN_SMEM(smem_raw);
    float* As = reinterpret_cast<float*>(smem_raw);            // [2][128*32]
    float* Bs = As + 2 * kSlabFloatsA;                         // [2][8*128*4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int nblk = p.num_mt * p.num_nt * p.splits;
    const int L = xcd_remap(blockIdx.x, nblk);
    const int nt = L % p.num_nt;
    const int mt = (L / p.num_nt) % p.num_mt;
    const int split = L / (p.num_nt * p.num_mt);
    const int slab0 = split * p.slabs_per_split;
    const int slab1 = min(slab0 + p.slabs_per_split, p.slabs_total);

    // ---- A loader: thread -> 4 (row, slot) pairs, row = tid/8 + 32*q -------
    const int a_slot = tid & 7;
    const int a_row = tid >> 3;
    const float* a_ptr[4];
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
        a_ptr[q] = p.x + (((long long)b * p.H + a_ih0[q]) * p.W + a_iw0[q]) * (long long)p.Cin + a_slot * 4;
    }
    // ---- B loader: thread -> 4 (slot, col) pairs, idx = tid + 256*q ----------
    const float* b_ptr = p.wp + ((long long)nt * 128 + (tid & 127)) * 4 + (long long)(tid >> 7) * p.CoutPad * 4;
    const long long b_slot_stride = (long long)p.CoutPad * 4;      // floats per k-slot row

    const int cpt = p.Cin >> 5;                   // slabs per tap
    int tap = slab0 / cpt;
    int cc = slab0 - tap * cpt;
    int kh = tap / p.KS, kw = tap - kh * p.KS;

    f32x4 ra[4], rb[4];
    auto fetch = [&](int slab) {
        const int tap_off = (kh * p.W + kw) * p.Cin + cc * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool ok = a_ok[q] && (unsigned)(a_ih0[q] + kh) < (unsigned)p.H &&
                            (unsigned)(a_iw0[q] + kw) < (unsigned)p.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(a_ptr[q] + tap_off);
            ra[q] = v;
        }
        const float* bp = b_ptr + (long long)slab * 8 * b_slot_stride;
#pragma unroll
        for (int q = 0; q < 4; ++q) rb[q] = *reinterpret_cast<const f32x4*>(bp + (long long)(2 * q) * b_slot_stride);
        // advance the (kh, kw, cc) counters to the next slab
        if (++cc == cpt) { cc = 0; if (++kw == p.KS) { kw = 0; ++kh; } }
    };
    auto stash = [&](int buf) {
        float* A = As + buf * kSlabFloatsA;
        float* Bt = Bs + buf * (8 * 128 * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) lds_write4(A + a_slab_off(a_row + 32 * q, a_slot), ra[q]);
#pragma unroll
        for (int q = 0; q < 4; ++q) lds_write4(Bt + (tid + 256 * q) * 4, rb[q]);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    if (slab0 < slab1) {
        fetch(slab0);
        stash(0);
        __syncthreads();
        int buf = 0;
        for (int t = slab0; t < slab1; ++t) {
            const bool more = (t + 1) < slab1;
            if (more) fetch(t + 1);
            mfma_slab<2, 2>(As + buf * kSlabFloatsA, Bs + buf * (8 * 128 * 4), 128, wm * 64, wn * 64, lane, acc);
            if (more) stash(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }

    // ---- epilogue -----------------------------------------------------------
    const int i = lane & 31;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int n = nt * 128 + wn * 64 + ni * 32 + i;
        if (n >= p.Cout) continue;
        float bias = 0.f, sc = 1.f, sh = 0.f;
        if (!SPLITK) {
            bias = p.bias[n];
            if (p.bn_scale) { sc = p.bn_scale[n]; sh = p.bn_shift[n]; }
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
                    p.out[(long long)m * p.Cout + n] = v;
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
};

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const SplitKReduceArgs p) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < p.MN; e += stride) {
        const int n = (int)(e % p.Cout);
        float v = 0.f;
        for (int s = 0; s < p.splits; ++s) v += p.partial[(long long)s * p.MN + e];
        v += p.bias[n];
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.bn_scale) v = v * p.bn_scale[n] + p.bn_shift[n];
        p.out[e] = v;
    }
}

}  // namespace aae
