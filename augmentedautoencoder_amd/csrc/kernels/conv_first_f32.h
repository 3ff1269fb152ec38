// First encoder layer: small-Cin convolution (Cin = 1 or 3) on the fp32 matrix
// cores with the uint8 -> float conversion fused into the load.
//
// Replaces  x = x/255.            (/root/reference/auto_pose/ae/codebook.py:58-59)
//           tf.layers.conv2d(...) (/root/reference/auto_pose/ae/encoder.py:43-50), first iteration.
//
// GEMM view: M = output pixels, N = Cout, K = KS*KS*C with k = kh*(KS*C) + j,
// j = kw*C + ci -- i.e. every kernel row is one contiguous run of KS*C input
// values in NHWC.  A block stages the input rows it needs (full width, zero
// borders = TF 'SAME') as fp32 in LDS, so the A operand of MFMA step s is one
// ds_read_b32 at  pixel_base + kh*rowlen + j.  Each wave owns 32 output
// channels and keeps ALL its weights in VGPRs (ceil(K/2) registers) across a
// run of tiles; M tile = 128 consecutive output pixels of one image.
//
// uint8 input goes through a 256-entry table  float32(v/255.)  -- the exact
// value TensorFlow sees after the float64 division and the float32 feed cast.
#pragma once

namespace aae {

struct ConvFirstArgs {
    const void* x;          // [B,H,W,C] uint8 or float32
    const float* lut;       // [256] (uint8 input only)
    const float* w;         // HWIO [KS][KS][C][Cout] == [K][Cout]
    const float* bias;
    const float* bn_scale;  // or nullptr
    const float* bn_shift;
    float* out;             // [B,Ho,Wo,Cout]
    int H, W, Ho, Wo, Cout;
    int S, pt, pl;
    int rowlen;             // staged floats per input row = ((Wo-1)*S + KS) * C
    int tiles_per_image;    // ceil(Ho*Wo / 128)
    int total_tiles;        // B * tiles_per_image
    int tiles_per_block;
    int relu;
    float out_scale;        // OUT_PLANES: 2^act_shift of the f32x3h activation format
};

// OUT_PLANES: write the activation as two fp16 planes (hi, lo of v*out_scale) for the f32x3h
// implicit GEMM (conv_igemm_x3h.h) instead of fp32.
template <int KS, int C, bool IN_U8, bool OUT_PLANES>
__global__ __launch_bounds__(256) void conv_first_f32_kernel(const ConvFirstArgs p) {
    constexpr int KROW = KS * C;
    constexpr int K = KS * KROW;
    constexpr int NK2 = (K + 1) / 2;

    AAE_DYN_SMEM(smem_raw);
    float* lut_s = reinterpret_cast<float*>(smem_raw);     // [256]
    float* patch = lut_s + 256;                            // [rows][rowlen]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int n = blockIdx.y * 128 + wave * 32 + i;
    const bool n_ok = n < p.Cout;

    if (IN_U8) lut_s[tid] = p.lut[tid];

    // this wave's weights: breg[s] = W[2s+h][n]
    float breg[NK2];
#pragma unroll
    for (int s = 0; s < NK2; ++s) {
        const int k = 2 * s + h;
        breg[s] = (n_ok && k < K) ? p.w[(long long)k * p.Cout + n] : 0.f;
    }
    float bias = 0.f, sc = 1.f, sh = 0.f;
    if (n_ok) {
        bias = p.bias[n];
        if (p.bn_scale) { sc = p.bn_scale[n]; sh = p.bn_shift[n]; }
    }

    const int HoWo = p.Ho * p.Wo;
    const int WC = p.W * C;
    const int t_begin = blockIdx.x * p.tiles_per_block;
    const int t_end = min(t_begin + p.tiles_per_block, p.total_tiles);

    // Input staging is software-pipelined across tiles: the raw bytes/floats of tile T+1 are
    // fetched into registers (kEpt elements per thread) before the MFMA phase of tile T and
    // converted + written to LDS after it, so the global-load latency sits under ~10k cycles
    // of matrix work instead of in front of it.  Elements beyond 256*kEpt (wide images) are
    // staged synchronously.
    constexpr int kEpt = 12;
    unsigned raw[kEpt];
    struct TileGeom { int b, p0, pend, oh_first, in_row0, total; };
    auto geom = [&](int T) {
        TileGeom g;
        g.b = T / p.tiles_per_image;
        g.p0 = (T - g.b * p.tiles_per_image) * 128;
        g.pend = min(g.p0 + 128, HoWo);
        g.oh_first = g.p0 / p.Wo;
        const int oh_last = (g.pend - 1) / p.Wo;
        g.in_row0 = g.oh_first * p.S - p.pt;
        g.total = ((oh_last - g.oh_first) * p.S + KS) * p.rowlen;
        return g;
    };
    // element e of the staged patch -> raw input value (0 for the zero border: lut[0] == 0.f)
    auto fetch = [&](const TileGeom& g, int e) -> unsigned {
        const int r = e / p.rowlen, c = e - r * p.rowlen;
        const int ih = g.in_row0 + r, src = c - p.pl * C;
        if (e < g.total && (unsigned)ih < (unsigned)p.H && (unsigned)src < (unsigned)WC) {
            const long long at = ((long long)g.b * p.H + ih) * WC + src;
            if (IN_U8) return reinterpret_cast<const unsigned char*>(p.x)[at];
            return reinterpret_cast<const unsigned*>(p.x)[at];
        }
        return 0u;
    };
    auto to_float = [&](unsigned v) -> float {
        if (IN_U8) return lut_s[v];
        return __builtin_bit_cast(float, v);
    };
    auto prefetch = [&](const TileGeom& g) {
#pragma unroll
        for (int j = 0; j < kEpt; ++j) raw[j] = fetch(g, tid + 256 * j);
    };
    auto stage = [&](const TileGeom& g) {
#pragma unroll
        for (int j = 0; j < kEpt; ++j) {
            const int e = tid + 256 * j;
            if (e < g.total) patch[e] = to_float(raw[j]);
        }
        for (int e = tid + 256 * kEpt; e < g.total; e += 256) patch[e] = to_float(fetch(g, e));
    };

    if (t_begin >= t_end) return;
    TileGeom g = geom(t_begin);
    prefetch(g);
    __syncthreads();                           // lut_s visible
    stage(g);
    __syncthreads();
    TileGeom gn = g;
    if (t_begin + 1 < t_end) { gn = geom(t_begin + 1); prefetch(gn); }

    for (int T = t_begin; T < t_end; ++T) {
        const int b = g.b, p0 = g.p0, pend = g.pend, oh_first = g.oh_first;

        int abase[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            int px = p0 + mt * 32 + i;
            px = min(px, pend - 1);            // tail lanes recompute the last pixel; masked at the store
            const int oh = px / p.Wo, ow = px - oh * p.Wo;
            abase[mt] = (oh - oh_first) * p.S * p.rowlen + ow * p.S * C;
        }

        f32x16 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

#pragma unroll
        for (int s = 0; s < NK2; ++s) {
            const int k0 = 2 * s, k1 = (2 * s + 1 < K) ? 2 * s + 1 : K - 1;   // k=K (odd K pad) re-reads k=K-1 against a zero weight
            const int off0 = (k0 / KROW) * p.rowlen + (k0 % KROW);
            const int off1 = (k1 / KROW) * p.rowlen + (k1 % KROW);
            const int off = h ? off1 : off0;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma_32x32x2(patch[abase[mt] + off], breg[s], acc[mt]);
        }

        // next tile's patch goes into LDS before this tile's stores are issued (its loads have
        // been in flight for the whole MFMA phase; the stores then never delay a vmcnt wait)
        const bool more = T + 1 < t_end;
        __syncthreads();                       // every wave is done reading this tile's patch
        if (more) stage(gn);
        __syncthreads();

        // (A transposed MFMA would give each lane 4 consecutive channels of one pixel = 16-B
        // stores, but every store instruction then touches 32 partial cache lines instead of 2
        // full ones: measured 0.41 ms against 0.26 ms for this form at B=256.)
        if (n_ok) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int px = p0 + mt * 32 + acc_row(r, lane);
                    if (px < pend) {
                        float v = acc[mt][r] + bias;
                        if (p.relu) v = fmaxf(v, 0.f);
                        if (p.bn_scale) v = v * sc + sh;
                        const long long o = ((long long)b * HoWo + px) * p.Cout + n;
                        if (OUT_PLANES) {
                            unsigned short hi, lo;
                            split_f16(v * p.out_scale, hi, lo);
                            unsigned short* op = reinterpret_cast<unsigned short*>(p.out);
                            op[o] = hi;
                            op[(long long)p.total_tiles / p.tiles_per_image * HoWo * p.Cout + o] = lo;
                        } else {
                            p.out[o] = v;
                        }
                    }
                }
        }
        g = gn;
        if (T + 2 < t_end) { gn = geom(T + 2); prefetch(gn); }
    }
}

}  // namespace aae
