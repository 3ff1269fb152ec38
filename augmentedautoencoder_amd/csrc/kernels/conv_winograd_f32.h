// 5 x 5 stride-2 'SAME' convolution (conv2 ... conv4 of /root/reference/auto_pose/ae/encoder.py:41-52) with FEWER MULTIPLIES in fp32:
// polyphase split + Winograd F(2 x 2, r x s), transforms fused on MFMA fragments.  Opt-in (encoder option "winograd"), large batches.
//
// The arithmetic.  out[y][x] = sum_{kh,kw} in[2y + kh - 1][2x + kw - 1] w[kh][kw] splits by the parity (eh, ew) of the input row / column
// into four stride-1 convolutions over the sub-images X_e[u][v] = in[2u + eh][2v + ew]:
//     odd rows  (e = 1): taps kh = 0, 2, 4 at sub-image offsets -1, 0, +1      (3 taps)
//     even rows (e = 0): taps kh = 1, 3    at sub-image offsets  0, +1         (2 taps)
// 3 x 3 + 3 x 2 + 2 x 3 + 2 x 2 = 25 taps.  Each component is evaluated as Winograd F(2, taps) per dimension: a 2 x 2 output tile from
// (tA + 1) x (tB + 1) element-wise products instead of 4 tA tB -- 16 + 12 + 12 + 9 = 49 products per tile and channel pair instead of 100.
//     F(2, 3):  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//     F(2, 2):  B^T = [1 -1 0; 0 1 0; 0 1 -1]                   G = [1 0; 1 1; 0 1]                       A^T = [1 1 0; 0 1 -1]
// Every product is a GEMM over the input channels: M[p] = V[p] (tiles x Cin) * U[p] (Cin x Cout) on the fp32 matrix cores, V = B^T d B of
// the tile's input patch d, U = G g G^T of the taps (computed once on the host in float64, rounded once), out tile = A^T M A.
// Measured error against float64 is SMALLER than the direct fp32 kernel's (fewer, better-conditioned additions per output: 3.7e-7 vs
// 8.5e-7 of the output scale on conv3, tools/ubench/polyphase_winograd.hip); results differ from the direct kernels by that much.
//
// One launch per phase (the four phases add up in the output buffer: mode 0 stores, 1 adds, 2 adds and applies bias / ReLU / BN):
//   block = 8 waves = 64 tiles x 64 output channels.  GEOM 0: an 8 x 8-tile (16 x 16-pixel) region of one image (conv2, conv3);
//           GEOM 1: the 4 x 4 tiles of four images (conv4: 8 x 8 outputs).  Wave (mh, nh, ph): 32 tiles x 32 channels x HALF the points --
//           the point rows of the split dimension A go to two waves (rows {0, 1} | the rest), so a wave keeps 8 (6, 3) accumulator tiles
//           = at most 128 registers, two waves share a SIMD and one's patch reads, transform and weight loads run under the other's
//           MFMAs.  The two halves of the output transform meet through LDS once per block.  A = rows, or columns when SWAP (the
//           2 x 3-tap phase: the 3-tap dimension is the one that splits evenly).
//   K loop = stages of 32 input channels: the block's window of the sub-image (tiles + halo, zero outside the image = the 'SAME'
//           padding) goes global -> registers -> LDS, double buffered, laid out [channel quad][image][column parity][row][column / 2]
//           with pitches that make the patch reads (ds_read_b128 by 32 tiles) and the fill conflict-free.  Per 8-channel group a lane
//           reads the patch rows its points need as float4 (4 channels of its K half), transforms them with packed fp32 adds, and
//           issues 4 MFMAs per point against weight fragments that were loaded one group ahead into the registers the previous
//           group's finished points released.
//   weights: packed [32-column block][8-channel group][point = a PB + b][K half][32 columns][4 channels] per phase (aae_encoder_plan.h).
#pragma once

namespace aae {

struct ConvWinoArgs {
    const float* x;          // [B][H][W][Cin]  (H = 2 Ho, W = 2 Wo)
    const float* U;          // packed transformed weights of this phase
    const float* bias;       // [Cout]   (mode 2)
    const float* bn_scale;   // [Cout] or nullptr
    const float* bn_shift;
    float* out;              // [B][Ho][Wo][Cout]
    int B, H, W, Cin, Cout, Ho, Wo;
    int eh, ew;              // parity of the phase's input rows / columns
    int mode;                // 0: out = y   1: out += y   2: out = epilogue(out + y + bias)
    int relu;
    int blocks_x, blocks_y;  // GEOM 0: 8 x 8-tile regions per image
};

template <int GEOM>
struct WinoGeom;
template <>
struct WinoGeom<0> {         // one image, 8 x 8 tiles: window 18 x 18
    static constexpr int kImages = 1, kRows = 18, kCols = 18, kRowPitch = 12, kImagePitch = 2 * 18 * 12;
};
template <>
struct WinoGeom<1> {         // four images, 4 x 4 tiles each: windows 10 x 10
    static constexpr int kImages = 4, kRows = 10, kCols = 10, kRowPitch = 6, kImagePitch = 128;
};
template <int GEOM>
constexpr int wino_plane_units() { return WinoGeom<GEOM>::kImages * WinoGeom<GEOM>::kImagePitch + 1; }
template <int GEOM>
constexpr int wino_stage_units() { return 8 * wino_plane_units<GEOM>(); }
template <int GEOM>
constexpr int wino_smem_bytes() { return 2 * wino_stage_units<GEOM>() * 16; }

// packed fp32 add / subtract: two values per instruction and lane (the transforms are vector work beside the MFMA stream)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ f32x2 wino_pk_add(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 wino_pk_sub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#else
__host__ __device__ inline f32x2 wino_pk_add(f32x2 a, f32x2 b) { return a + b; }
__host__ __device__ inline f32x2 wino_pk_sub(f32x2 a, f32x2 b) { return a - b; }
#endif
__device__ __forceinline__ f32x4 wino_add4(f32x4 a, f32x4 b) {
    const f32x2 lo = wino_pk_add(a.lo, b.lo), hi = wino_pk_add(a.hi, b.hi);
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 wino_sub4(f32x4 a, f32x4 b) {
    const f32x2 lo = wino_pk_sub(a.lo, b.lo), hi = wino_pk_sub(a.hi, b.hi);
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}

// TA / TB: taps along the split dimension A / the other dimension B (3 | 2).  SWAP: A = columns.
template <int TA, int TB, bool SWAP, int GEOM>
__global__ __launch_bounds__(512) void conv_wino_phase_kernel(ConvWinoArgs a) {
    using G = WinoGeom<GEOM>;
    constexpr int PB = TB + 1;                                  // points (= patch positions) along B
    constexpr int kPlane = wino_plane_units<GEOM>(), kStage = wino_stage_units<GEOM>();
    constexpr int kParity = G::kRows * G::kRowPitch;            // units between the two column-parity halves of an image
    constexpr int kStageQuads = G::kImages * G::kRows * G::kCols * 8;
    constexpr int kHalf = ((kStageQuads + 511) / 512 + 1) / 2;  // float4 per thread and half stage
    constexpr int kOffA = TA == 2 ? 1 : 0, kOffB = TB == 2 ? 1 : 0;   // a 2-tap dimension starts one sample into the window
    AAE_DYN_SMEM(smem_raw);
    f32x4* lds = reinterpret_cast<f32x4*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mh = wave & 1, nh = (wave >> 1) & 1, ph = wave >> 2, m = lane & 31, h = lane >> 5;
    const int nbn = a.Cout / 64;
    const int nb = blockIdx.x % nbn;
    int rest = blockIdx.x / nbn;
    // block -> image(s) and window origin in sub-image coordinates
    int img0, wy0, wx0;
    if (GEOM == 0) {
        const int bx = rest % a.blocks_x;
        rest /= a.blocks_x;
        const int by = rest % a.blocks_y;
        img0 = rest / a.blocks_y;
        wy0 = 16 * by - 1;
        wx0 = 16 * bx - 1;
    } else {
        img0 = 4 * rest;
        wy0 = wx0 = -1;
    }
    // tile of index t (0 ... 31) of this wave: (image of the block, tile row, tile column)
    auto tile_of = [&](int t, int& ti, int& ty, int& tx) {
        if (GEOM == 0) { ti = 0; ty = 4 * mh + (t >> 3); tx = t & 7; }
        else { ti = 2 * mh + (t >> 4); ty = (t >> 2) & 3; tx = t & 3; }
    };
    int li, lty, ltx;
    tile_of(m, li, lty, ltx);
    const int n32 = nb * 2 + nh, KG = a.Cin / 8, nst = a.Cin / 32, cq_per_pixel = a.Cin / 4;
    constexpr int NP = (TA + 1) * PB;
    const f32x4* src = reinterpret_cast<const f32x4*>(a.x);
    const f32x4* up = reinterpret_cast<const f32x4*>(a.U) + (size_t)n32 * KG * NP * 64 + (size_t)ph * 2 * PB * 64 + h * 32 + m;
    const bool two_rows = TA == 3 || ph == 0;                   // point rows of A this wave owns: 2, or 1 (the third row of F(2, 2))

    f32x16 acc[2 * PB];
#pragma unroll
    for (int p = 0; p < 2 * PB; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // ---- stage fill: two halves, so that few staging registers are live at a time (the buffer of the next stage is free for the whole
    //      of the current one: the barrier behind the previous stage)
    f32x4 stg[kHalf];
    auto stage_load = [&](int st, int half) {
#pragma unroll
        for (int i = 0; i < kHalf; ++i) {
            const int idx = tid + 512 * (half * kHalf + i);
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (idx < kStageQuads) {
                const int pixel = idx >> 3, cq = idx & 7;
                const int wi = pixel / (G::kRows * G::kCols), rem = pixel - wi * (G::kRows * G::kCols);
                const int wy = rem / G::kCols, wx = rem - wy * G::kCols;
                const int u = wy0 + wy, v = wx0 + wx, b = img0 + wi;
                if (u >= 0 && u < a.Ho && v >= 0 && v < a.Wo && b < a.B)
                    val = src[(((size_t)b * a.H + 2 * u + a.eh) * a.W + 2 * v + a.ew) * cq_per_pixel + st * 8 + cq];
            }
            stg[i] = val;
        }
    };
    auto stage_store = [&](int buf, int half) {
#pragma unroll
        for (int i = 0; i < kHalf; ++i) {
            const int idx = tid + 512 * (half * kHalf + i);
            if (idx < kStageQuads) {
                const int pixel = idx >> 3, cq = idx & 7;
                const int wi = pixel / (G::kRows * G::kCols), rem = pixel - wi * (G::kRows * G::kCols);
                const int wy = rem / G::kCols, wx = rem - wy * G::kCols;
                lds[buf * kStage + cq * kPlane + wi * G::kImagePitch + (wx & 1) * kParity + wy * G::kRowPitch + (wx >> 1)] = stg[i];
            }
        }
    };
    // ---- patch addressing.  A sample at window position (wy, wx) = (2 ty + pA, 2 tx + pB) (or with A and B exchanged when SWAP) lies at
    //      lane_base + fA(pA) + fB(pB).  The A positions this wave reads, in the order (y0, y1, y2) that makes both halves the same
    //      arithmetic:   w0 = y0 - y2,  w1 = y2 + sg y1
    //        F(2, 3)  ph 0: positions (0, 1, 2), sg = +1 -> B^T rows 0, 1 (d0 - d2, d1 + d2)     ph 1: (2, 3, 1), sg = -1 -> rows 2, 3 (d2 - d1, d1 - d3)
    //        F(2, 2)  ph 0: positions (0, -, 1)          -> rows 0, 1 (d0 - d1, d1)              ph 1: (1, -, 2) -> row 2 (d1 - d2)
    auto fA = [](int pos) { return SWAP ? (pos & 1) * kParity + (pos >> 1) : pos * G::kRowPitch; };
    auto fB = [](int pos) { return SWAP ? pos * G::kRowPitch : (pos & 1) * kParity + (pos >> 1); };
    const int lane_base = li * G::kImagePitch + 2 * lty * G::kRowPitch + ltx;
    const int posA0 = (TA == 3 ? (ph == 0 ? 0 : 2) : (ph == 0 ? 0 : 1)) + kOffA;
    const int posA1 = (ph == 0 ? 1 : 3) + kOffA;                 // (F(2, 3) only)
    const int posA2 = (TA == 3 ? (ph == 0 ? 2 : 1) : (ph == 0 ? 1 : 2)) + kOffA;
    const int offA0 = fA(posA0), offA1 = fA(posA1), offA2 = fA(posA2);
    const float sg = ph == 0 ? 1.f : -1.f;

    f32x4 u[2 * PB];
#pragma unroll
    for (int p = 0; p < 2 * PB; ++p)
        if (two_rows || p < PB) u[p] = up[p * 64];
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
            const int gi = st * 4 + g;
            if (more && (g & 1) == 0) stage_load(st + 1, g >> 1);
            const f32x4* plane = lds + buf * kStage + (2 * g + h) * kPlane + lane_base;
            f32x4 v[2 * PB];
            // rows of B^T d along A ...
#pragma unroll
            for (int s = 0; s < PB; ++s) {
                const int ob = fB(s + kOffB);
                const f32x4 y0 = plane[offA0 + ob], y2 = plane[offA2 + ob];
                v[s] = wino_sub4(y0, y2);
                if (TA == 3) {
                    const f32x4 y1 = plane[offA1 + ob];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[PB + s][e] = fmaf(sg, y1[e], y2[e]);
                } else {
                    v[PB + s] = y2;
                }
            }
            // ... then along B
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (TB == 3) {
                    const f32x4 e0 = v[PB * i], e1 = v[PB * i + 1], e2 = v[PB * i + 2], e3 = v[PB * i + 3];
                    v[PB * i] = wino_sub4(e0, e2);
                    v[PB * i + 1] = wino_add4(e1, e2);
                    v[PB * i + 2] = wino_sub4(e2, e1);
                    v[PB * i + 3] = wino_sub4(e1, e3);
                } else {
                    const f32x4 e0 = v[PB * i], e1 = v[PB * i + 1], e2 = v[PB * i + 2];
                    v[PB * i] = wino_sub4(e0, e1);
                    v[PB * i + 2] = wino_sub4(e1, e2);
                }
            }
            // two points at a time (their accumulators alternate); as soon as a pair is through, ITS weight registers take the next
            // group's fragments: the global loads of group t + 1 fly under the MFMAs of group t without a second set of registers
            const f32x4* un = up + (size_t)(gi + 1) * NP * 64;
            const bool next = gi + 1 < 4 * nst;
            if (two_rows) {
#pragma unroll
                for (int bb = 0; bb < PB; ++bb) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[bb] = mfma_32x32x2(v[bb][q], u[bb][q], acc[bb]);
                        acc[PB + bb] = mfma_32x32x2(v[PB + bb][q], u[PB + bb][q], acc[PB + bb]);
                    }
                    if (next) {
                        u[bb] = un[bb * 64];
                        u[PB + bb] = un[(PB + bb) * 64];
                    }
                }
            } else {
#pragma unroll
                for (int bb = 0; bb < PB; ++bb) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[bb] = mfma_32x32x2(v[bb][q], u[bb][q], acc[bb]);
                    if (next) u[bb] = un[bb * 64];
                }
            }
            if (more && (g & 1) == 1) stage_store(buf ^ 1, g >> 1);
        }
        __syncthreads();
    }

    // ---- output transform.  Along A the rows of A^T m split over the two waves:
    //        ph 0: q0 = m0 + m1, q1 = m1        ph 1, F(2, 3): q0 = m2, q1 = -m2 - m3        ph 1, F(2, 2): q0 = 0, q1 = -m2
    //      each wave applies A along B to its part; the upper half hands its four partial outputs per accumulator register over through
    //      LDS (the stage buffers are free after the last barrier), the lower half adds, finishes and stores.
    float* xch = reinterpret_cast<float*>(smem_raw) + (size_t)(wave & 3) * 64 * 64;
    auto partial = [&](int r, float (&y)[4]) {
        float q0[PB], q1[PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const float mA = acc[j][r], mB = acc[PB + j][r];
            if (ph == 0) { q0[j] = mA + mB; q1[j] = mB; }
            else if (TA == 3) { q0[j] = mA; q1[j] = -mA - mB; }
            else { q0[j] = 0.f; q1[j] = -mA; }
        }
        if (TB == 3) {
            y[0] = q0[0] + q0[1] + q0[2];
            y[1] = q0[1] - q0[2] - q0[3];
            y[2] = q1[0] + q1[1] + q1[2];
            y[3] = q1[1] - q1[2] - q1[3];
        } else {
            y[0] = q0[0] + q0[1];
            y[1] = q0[1] - q0[2];
            y[2] = q1[0] + q1[1];
            y[3] = q1[1] - q1[2];
        }
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
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float y[4];
            partial(r, y);
#pragma unroll
            for (int k = 0; k < 4; ++k) xch[(r * 4 + k) * 64 + lane] += y[k];
        }
    }
    __syncthreads();
    // ---- the block's 64 tiles x 4 pixels x 64 channels leave as float4 per thread (eight each, all loads of the accumulating modes in
    //      flight together): value (tile t, pixel k = 2 iA + iB, channel c) sits at pair (t / 32, c / 32), register r = (t & 3) + 4 ((t & 31) / 8),
    //      lane 32 ((t / 4) & 1) + c % 32 of the exchange buffer
    const float* xall = reinterpret_cast<const float*>(smem_raw);
    f32x4 val[8];
    float* optr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + 512 * i, cq = idx & 15, pix = idx >> 4, t = pix >> 2, k = pix & 3;
        const int tmh = t >> 5, mt = t & 31, tnh = cq >> 3;
        const int r = (mt & 3) + 4 * (mt >> 3), hh = (mt >> 2) & 1;
        val[i] = *reinterpret_cast<const f32x4*>(xall + (size_t)(tmh + 2 * tnh) * 64 * 64 + (r * 4 + k) * 64 + 32 * hh + (cq & 7) * 4);
        int ti, ty, tx;
        if (GEOM == 0) { ti = 0; ty = 4 * tmh + (mt >> 3); tx = mt & 7; }
        else { ti = 2 * tmh + (mt >> 4); ty = (mt >> 2) & 3; tx = mt & 3; }
        const int b = img0 + ti;
        const int iA = k >> 1, iB = k & 1, dy = SWAP ? iB : iA, dx = SWAP ? iA : iB;
        const int oy = (GEOM == 0 ? wy0 + 1 : 0) + 2 * ty + dy, ox = (GEOM == 0 ? wx0 + 1 : 0) + 2 * tx + dx;
        optr[i] = b < a.B ? a.out + (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.Cout + nb * 64 + cq * 4 : nullptr;
    }
    if (a.mode != 0) {
        f32x4 old[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (optr[i]) old[i] = *reinterpret_cast<const f32x4*>(optr[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (optr[i]) val[i] += old[i];
    }
    if (a.mode == 2) {
        const int n4 = nb * 64 + (tid & 15) * 4;                     // (the channel quad of a thread is the same in all eight rounds)
        const f32x4 bs = *reinterpret_cast<const f32x4*>(a.bias + n4);
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (a.bn_scale) {
            sc = *reinterpret_cast<const f32x4*>(a.bn_scale + n4);
            sh = *reinterpret_cast<const f32x4*>(a.bn_shift + n4);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            val[i] += bs;
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) val[i][e] = fmaxf(val[i][e], 0.f);
            }
            if (a.bn_scale) val[i] = val[i] * sc + sh;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (optr[i]) *reinterpret_cast<f32x4*>(optr[i]) = val[i];
}

}  // namespace aae
