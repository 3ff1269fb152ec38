// 5 x 5 stride-2 'SAME' convolution (conv2 ... conv4 of /root/reference/auto_pose/ae/encoder.py:41-52) with FEWER MULTIPLIES in fp32:
// polyphase split + Winograd F(2 x 2, r x s), transforms fused on MFMA fragments.  The default for every eligible layer whose blocks fill the rounds of blocks they occupy (encoder
// option "winograd", aae_encoder_launch.h: runs_winograd); 0 = the direct implicit-GEMM kernels.
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
// One launch per LAYER (conv_wino_layer_kernel: the four phases run one behind the other inside the block and add up in an LDS buffer; the
// output is written once with bias / ReLU / BN).  The experiments build also has one launch per PHASE (the phases add up in the output
// buffer: measured 2 % slower, tools/wino_ab.py).  Either way:
//   block = 8 waves = 64 tiles x 64 output channels.  GEOM 0: an 8 x 8-tile (16 x 16-pixel) region of one image (conv2, conv3);
//           GEOM 1: the 4 x 4 tiles of four images (conv4: 8 x 8 outputs).  Wave (mh, nh, ph): 32 tiles x 32 channels x HALF the points --
//           the point rows of the split dimension A go to two waves (rows {0, 1} | the rest), so a wave keeps 8 (6, 3) accumulator tiles
//           = at most 128 registers, two waves share a SIMD and one's patch reads, transform and weight loads run under the other's
//           MFMAs.  The two halves of the output transform meet through LDS once per block.  A = rows, or columns when SWAP (the
//           2 x 3-tap phase: the 3-tap dimension is the one that splits evenly).
//   K loop = stages of 16 input channels (32 in the one-launch-per-phase form): the block's window of the sub-image (tiles + halo, zero outside the image = the 'SAME'
//           padding) goes global -> registers -> LDS, double buffered, laid out [channel quad][image][column parity][row][column / 2]
//           with pitches that make the patch reads (ds_read_b128 by 32 tiles) and the fill conflict-free.  The loop runs in units of
//           one point row: a lane reads the patch rows the row needs as float4 (4 channels of its K half), transforms them with packed
//           fp32 adds BETWEEN the MFMAs of the previous unit, and issues 4 MFMAs per point against weight fragments that were loaded
//           one group ahead (raw buffer views, scalar offsets) into the registers the previous group's finished row released.
//   weights: packed [32-column block][8-channel group][point = a PB + b][K half][32 columns][4 channels] per phase (aae_encoder_plan.h).
#pragma once
#include <type_traits>

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
    // (parity pitch 220 instead of 18 * 12 = 216 quads: the fill writes the two column parities of neighbouring pixels from neighbouring
    //  lanes -- 880 dwords apart they fall into different halves of the 32 banks a ds_write_b128 group uses)
    static constexpr int kImages = 1, kRows = 18, kCols = 18, kRowPitch = 12, kParityPitch = 220, kImagePitch = 2 * 220;
};
template <>
struct WinoGeom<1> {         // four images, 4 x 4 tiles each: windows 10 x 10
    static constexpr int kImages = 4, kRows = 10, kCols = 10, kRowPitch = 6, kParityPitch = 60, kImagePitch = 128;
};
template <int GEOM>
constexpr int wino_plane_units() { return WinoGeom<GEOM>::kImages * WinoGeom<GEOM>::kImagePitch + 1; }
template <int GEOM>
constexpr int wino_stage_units() { return 8 * wino_plane_units<GEOM>(); }
template <int GEOM>
constexpr int wino_smem_bytes() { return 2 * wino_stage_units<GEOM>() * 16; }

// arguments of the one-launch-per-layer kernel (below) and its LDS budget: 16-channel stages + a 64 KB exchange buffer of its own
struct ConvWinoLayerArgs {
    ConvWinoArgs c;          // (U, eh, ew, mode unused)
    const float* U4[4];      // index 2 eh + ew
};
template <int GEOM>
constexpr int wino_layer_stage_bytes() { return 2 * 4 * wino_plane_units<GEOM>() * 16; }
template <int GEOM>
constexpr int wino_layer_smem_bytes() { return wino_layer_stage_bytes<GEOM>() + 4 * 64 * 64 * 4; }


#ifndef AAE_WINO_DECLARATIONS_ONLY      // (the product library compiles the kernels below in a translation unit of their own: aae_wino.hip)

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
// a * s + b with one scalar s for all four lanes of the quad
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ f32x4 wino_fma4(float s, f32x4 a, f32x4 b) {
    const f32x2 ss = {s, s};
    f32x2 lo, hi;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(ss), "v"(a.lo), "v"(b.lo));
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(ss), "v"(a.hi), "v"(b.hi));
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
#else
__host__ __device__ inline f32x4 wino_fma4(float s, f32x4 a, f32x4 b) {
    f32x4 r;
    for (int e = 0; e < 4; ++e) r[e] = fmaf(s, a[e], b[e]);
    return r;
}
#endif
__device__ __forceinline__ f32x4 wino_add4(f32x4 a, f32x4 b) {
    const f32x2 lo = wino_pk_add(a.lo, b.lo), hi = wino_pk_add(a.hi, b.hi);
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 wino_sub4(f32x4 a, f32x4 b) {
    const f32x2 lo = wino_pk_sub(a.lo, b.lo), hi = wino_pk_sub(a.hi, b.hi);
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}

// Block geometry shared by the phases of a launch: which images / window a block covers and where a tile of a wave lies.
struct WinoBlock {
    int nb, img0, wy0, wx0;
};
template <int GEOM>
__device__ __forceinline__ WinoBlock wino_block(int block, int nbn, int blocks_x, int blocks_y) {
    WinoBlock w;
    w.nb = block % nbn;
    int rest = block / nbn;
    if (GEOM == 0) {
        const int bx = rest % blocks_x;
        rest /= blocks_x;
        const int by = rest % blocks_y;
        w.img0 = rest / blocks_y;
        w.wy0 = 16 * by - 1;
        w.wx0 = 16 * bx - 1;
    } else {
        w.img0 = 4 * rest;
        w.wy0 = w.wx0 = -1;
    }
    return w;
}
// tile t (0 ... 31) of the wave with tile half mh: (image of the block, tile row, tile column)
template <int GEOM>
__device__ __forceinline__ void wino_tile(int mh, int t, int& ti, int& ty, int& tx) {
    if (GEOM == 0) { ti = 0; ty = 4 * mh + (t >> 3); tx = t & 7; }
    else { ti = 2 * mh + (t >> 4); ty = (t >> 2) & 3; tx = t & 3; }
}

// One polyphase component for one block: the K loop over all input channels and the two-wave output transform.  Leaves the block's
// 64 tiles x 4 pixels x 64 channels in the exchange buffer `xch_all` ([wave pair mh + 2 nh][register r * 4 + pixel k][lane] floats,
// pixel k = 2 iA + iB): stored when !ACCUMULATE, added to what is there otherwise.  Ends behind a block barrier.
// TA / TB: taps along the split dimension A / the other dimension B (3 | 2).  SWAP: A = columns.  STAGE_CH: input channels per LDS stage.
// WIDE: block = 4 waves (mh, ph), each over BOTH 32-channel halves (twice the MFMAs per patch read, transform and barrier; 256 accumulator
// registers: one wave per SIMD).  Otherwise 8 waves (mh, nh, ph), one 32-channel half each, two waves per SIMD.
template <int TA, int TB, bool SWAP, int GEOM, int STAGE_CH, bool ACCUMULATE, bool WIDE>
__device__ __forceinline__ void wino_phase_body(const ConvWinoArgs& a, const float* U, int eh, int ew, const WinoBlock& wb, f32x4* lds, float* xch_all) {
    using G = WinoGeom<GEOM>;
    constexpr int PB = TB + 1;                                  // points (= patch positions) along B
    constexpr int kQuads = STAGE_CH / 4, kGroups = STAGE_CH / 8;                    // channel quads / 8-channel groups per stage
    constexpr int kPlane = wino_plane_units<GEOM>(), kStage = kQuads * kPlane;
    constexpr int kParity = G::kParityPitch;                    // units between the two column-parity halves of an image
    constexpr int kStageQuads = G::kImages * G::kRows * G::kCols * kQuads;
    constexpr int kParts = kGroups / 2;                         // the fill of the next stage happens in this many parts (load at an even group, store at the next)
    constexpr int NT = WIDE ? 256 : 512, NH = WIDE ? 2 : 1;                         // threads of the block, 32-channel halves per wave
    constexpr int kPer = ((kStageQuads + NT - 1) / NT + kParts - 1) / kParts;       // float4 per thread and part
    constexpr int kOffA = TA == 2 ? 1 : 0, kOffB = TB == 2 ? 1 : 0;   // a 2-tap dimension starts one sample into the window
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mh = wave & 1, nh = WIDE ? 0 : (wave >> 1) & 1, ph = WIDE ? wave >> 1 : wave >> 2, m = lane & 31, h = lane >> 5;
    const int img0 = wb.img0, wy0 = wb.wy0, wx0 = wb.wx0;
    int li, lty, ltx;
    wino_tile<GEOM>(mh, m, li, lty, ltx);
    const int n32 = wb.nb * 2 + nh, KG = a.Cin / 8, nst = a.Cin / STAGE_CH, cq_per_pixel = a.Cin / 4;
    constexpr int NP = (TA + 1) * PB;
    // both operands through raw buffer views (scalar base + one 32-bit lane offset; out-of-range = zeros: the 'SAME' padding and the empty
    // image slots of a ragged group need no branches).  The host keeps the activation below 2 GiB for that (0x80000000 marks "outside").
    constexpr uint32_t kOutside = 0x80000000u;
    const buffer_rsrc xrs = make_buffer(a.x, (uint32_t)((size_t)a.B * a.H * a.W * a.Cin * 4));
    const buffer_rsrc urs = make_buffer(U, (uint32_t)((size_t)NP * a.Cin * a.Cout * 4));
    const uint32_t ulane = (uint32_t)(((size_t)n32 * KG * NP + (size_t)ph * 2 * PB) * 64 + h * 32 + m) * 16u;
    const bool two_rows = TA == 3 || ph == 0;                   // point rows of A this wave owns: 2, or 1 (the third row of F(2, 2))

    f32x16 acc[NH][2 * PB];
#pragma unroll
    for (int n2 = 0; n2 < NH; ++n2)
#pragma unroll
        for (int p = 0; p < 2 * PB; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n2][p][r] = 0.f;

    // ---- stage fill in parts, so that few staging registers are live at a time (the buffer of the next stage is free for the whole
    //      of the current one: the barrier behind the previous stage).  A thread's source offsets and LDS slots are the same in every stage.
    f32x4 stg[kPer];
    uint32_t goff[kParts * kPer];
    int lslot[kParts * kPer];
#pragma unroll
    for (int i = 0; i < kParts * kPer; ++i) {
        const int idx = tid + NT * i;
        goff[i] = kOutside;
        lslot[i] = -1;
        if (idx < kStageQuads) {
            const int pixel = idx / kQuads, cq = idx - pixel * kQuads;
            const int wi = pixel / (G::kRows * G::kCols), rem = pixel - wi * (G::kRows * G::kCols);
            const int wy = rem / G::kCols, wx = rem - wy * G::kCols;
            const int uu = wy0 + wy, vv = wx0 + wx, b = img0 + wi;
            if (uu >= 0 && uu < a.Ho && vv >= 0 && vv < a.Wo && b < a.B)
                goff[i] = (uint32_t)(((((size_t)b * a.H + 2 * uu + eh) * a.W + 2 * vv + ew) * cq_per_pixel + cq) * 16);
            lslot[i] = cq * kPlane + wi * G::kImagePitch + (wx & 1) * kParity + wy * G::kRowPitch + (wx >> 1);
        }
    }
    auto stage_load = [&](int st, int part) {
#pragma unroll
        for (int i = 0; i < kPer; ++i) stg[i] = buffer_load4_s(xrs, goff[part * kPer + i], (uint32_t)(st * kQuads * 16));
    };
    auto stage_store = [&](int buf, int part) {
#pragma unroll
        for (int i = 0; i < kPer; ++i)
            if (lslot[part * kPer + i] >= 0) lds[buf * kStage + lslot[part * kPer + i]] = stg[i];
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

    const uint32_t half_stride = (uint32_t)KG * NP * 1024u;      // the next 32-column block of the packed weights lies KG * NP KB on
    f32x4 u[NH][2 * PB];
#pragma unroll
    for (int n2 = 0; n2 < NH; ++n2)
#pragma unroll
        for (int p = 0; p < 2 * PB; ++p)
            if (two_rows || p < PB) u[n2][p] = buffer_load4_s(urs, ulane + p * 1024u, n2 * half_stride);
#pragma unroll
    for (int part = 0; part < kParts; ++part) {
        stage_load(0, part);
        stage_store(0, part);
    }
    __syncthreads();

    // ---- the K loop in UNITS of one point row (PB points = 4 PB MFMAs): the patch rows of unit t + 1 are read and transformed between
    //      the MFMAs of unit t, into registers of their own, so that a wave feeds the matrix pipe by itself (two waves that run the same
    //      instruction stream side by side stall at the same places: the partner does not fill the gaps).  Pinned with scheduling fences:
    //      reads of t + 1 | MFMAs q = 0 | MFMAs q = 1, A-step of t + 1 | MFMAs q = 2, B-step of t + 1 | MFMAs q = 3, weight prefetch.
    //      The first unit behind a stage barrier has nothing to hide behind (once per STAGE_CH channels).
    auto read_unit = [&](const f32x4* plane, int i, f32x4 (&raw)[2 * PB]) {
#pragma unroll
        for (int s2 = 0; s2 < PB; ++s2) {
            const int ob = fB(s2 + kOffB);
            if (i == 0) raw[s2] = plane[offA0 + ob];
            else if (TA == 3) raw[s2] = plane[offA1 + ob];
            raw[PB + s2] = plane[offA2 + ob];
        }
    };
    auto step_a = [&](int i, const f32x4 (&raw)[2 * PB], f32x4 (&w)[PB]) {       // rows of B^T d along A
#pragma unroll
        for (int s2 = 0; s2 < PB; ++s2) {
            if (i == 0) w[s2] = wino_sub4(raw[s2], raw[PB + s2]);
            else if (TA == 3) w[s2] = wino_fma4(sg, raw[s2], raw[PB + s2]);
            else w[s2] = raw[PB + s2];
        }
    };
    auto step_b = [&](f32x4 (&w)[PB]) {                                           // ... then along B, in place
        if (TB == 3) {
            const f32x4 e0 = w[0], e1 = w[1], e2 = w[2], e3 = w[PB - 1];
            w[0] = wino_sub4(e0, e2);
            w[1] = wino_add4(e1, e2);
            w[2] = wino_sub4(e2, e1);
            w[PB - 1] = wino_sub4(e1, e3);
        } else {
            const f32x4 e0 = w[0], e1 = w[1], e2 = w[2];
            w[0] = wino_sub4(e0, e1);
            w[2] = wino_sub4(e1, e2);
        }
    };
    auto run = [&](auto rows_tag) {
        constexpr int ROWS = decltype(rows_tag)::value, NU = kGroups * ROWS;
        f32x4 raw[2 * PB], v[PB], vn[PB];
        for (int st = 0; st < nst; ++st) {
            const int buf = st & 1;
            const bool more = st + 1 < nst;
            const f32x4* stage = lds + buf * kStage + h * kPlane + lane_base;
            read_unit(stage, 0, raw);
            step_a(0, raw, v);
            step_b(v);
#pragma unroll
            for (int t = 0; t < NU; ++t) {
                const int g = t / ROWS, i = t % ROWS, gi = st * kGroups + g;
                if (more && i == 0 && (g & 1) == 0) stage_load(st + 1, g >> 1);
                if (t + 1 < NU) read_unit(stage + 2 * ((t + 1) / ROWS) * kPlane, (t + 1) % ROWS, raw);
                sched_fence();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int j = 0; j < PB; ++j)
#pragma unroll
                        for (int n2 = 0; n2 < NH; ++n2) acc[n2][i * PB + j] = mfma_32x32x2(v[j][q], u[n2][i * PB + j][q], acc[n2][i * PB + j]);
                    sched_fence();
                    if (t + 1 < NU) {
                        if (q == 1) step_a((t + 1) % ROWS, raw, vn);
                        if (q == 2) step_b(vn);
                    }
                    sched_fence();
                }
                // this row's weight registers take the next group's fragments (they fly under the MFMAs of the units in between)
                if (gi + 1 < kGroups * nst) {
                    const uint32_t un = (uint32_t)(gi + 1) * (NP * 1024u) + i * PB * 1024u;        // (wave-uniform: a scalar register)
#pragma unroll
                    for (int n2 = 0; n2 < NH; ++n2)
#pragma unroll
                        for (int j = 0; j < PB; ++j) u[n2][i * PB + j] = buffer_load4_s(urs, ulane + j * 1024u, un + n2 * half_stride);
                }
                if (more && i == ROWS - 1 && (g & 1) == 1) stage_store(buf ^ 1, g >> 1);
                if (t + 1 < NU) {
#pragma unroll
                    for (int j = 0; j < PB; ++j) v[j] = vn[j];
                }
            }
            __syncthreads();
        }
    };
    if (TA == 3 || ph == 0) run(std::integral_constant<int, 2>());
    else run(std::integral_constant<int, 1>());

    // ---- output transform.  Along A the rows of A^T m split over the two waves:
    //        ph 0: q0 = m0 + m1, q1 = m1        ph 1, F(2, 3): q0 = m2, q1 = -m2 - m3        ph 1, F(2, 2): q0 = 0, q1 = -m2
    //      each wave applies A along B to its part and adds its four partial outputs per accumulator register into the exchange
    //      buffer, the upper half first (it STORES when the buffer holds nothing yet), the lower half behind a barrier.
    auto partial = [&](int n2, int r, float (&y)[4]) {
        float q0[PB], q1[PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const float mA = acc[n2][j][r], mB = acc[n2][PB + j][r];
            if (ph == 0) { q0[j] = mA + mB; q1[j] = mB; }
            else if (TA == 3) { q0[j] = mA; q1[j] = -mA - mB; }
            else { q0[j] = 0.f; q1[j] = -mA; }
        }
        // (pixel k = 2 iA + iB of the exchange buffer is output (dy, dx) = (iA, iB), or (iB, iA) when A = columns: stored as 2 dy + dx)
        float t[4];
        if (TB == 3) {
            t[0] = q0[0] + q0[1] + q0[2];
            t[1] = q0[1] - q0[2] - q0[3];
            t[2] = q1[0] + q1[1] + q1[2];
            t[3] = q1[1] - q1[2] - q1[3];
        } else {
            t[0] = q0[0] + q0[1];
            t[1] = q0[1] - q0[2];
            t[2] = q1[0] + q1[1];
            t[3] = q1[1] - q1[2];
        }
        y[0] = t[0];
        y[1] = SWAP ? t[2] : t[1];
        y[2] = SWAP ? t[1] : t[2];
        y[3] = t[3];
    };
    if (ph == 1) {
#pragma unroll
        for (int n2 = 0; n2 < NH; ++n2) {
            float* xch = xch_all + (size_t)(mh + 2 * (nh + n2)) * 64 * 64;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float y[4];
                partial(n2, r, y);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (ACCUMULATE) xch[(r * 4 + k) * 64 + lane] += y[k];
                    else xch[(r * 4 + k) * 64 + lane] = y[k];
                }
            }
        }
    }
    __syncthreads();
    if (ph == 0) {
#pragma unroll
        for (int n2 = 0; n2 < NH; ++n2) {
            float* xch = xch_all + (size_t)(mh + 2 * (nh + n2)) * 64 * 64;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float y[4];
                partial(n2, r, y);
#pragma unroll
                for (int k = 0; k < 4; ++k) xch[(r * 4 + k) * 64 + lane] += y[k];
            }
        }
    }
    __syncthreads();
}

// The block's 64 tiles x 4 pixels x 64 channels leave the exchange buffer as float4 per thread (eight each, all loads of the accumulating
// modes in flight together): value (tile t, pixel k = 2 dy + dx, channel c) sits at pair (t / 32, c / 32), register
// r = (t & 3) + 4 ((t & 31) / 8), lane 32 ((t / 4) & 1) + c % 32.
template <int GEOM, int NT>
__device__ __forceinline__ void wino_store_block(const ConvWinoArgs& a, int mode, const WinoBlock& wb, const float* xall) {
    const int tid = threadIdx.x;
    for (int round = 0; round < 512 / NT; ++round) {
    f32x4 val[8];
    float* optr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + NT * (i + 8 * round), cq = idx & 15, pix = idx >> 4, t = pix >> 2, k = pix & 3;
        const int tmh = t >> 5, mt = t & 31, tnh = cq >> 3;
        const int r = (mt & 3) + 4 * (mt >> 3), hh = (mt >> 2) & 1;
        val[i] = *reinterpret_cast<const f32x4*>(xall + (size_t)(tmh + 2 * tnh) * 64 * 64 + (r * 4 + k) * 64 + 32 * hh + (cq & 7) * 4);
        int ti, ty, tx;
        wino_tile<GEOM>(tmh, mt, ti, ty, tx);
        const int b = wb.img0 + ti;
        const int oy = (GEOM == 0 ? wb.wy0 + 1 : 0) + 2 * ty + (k >> 1), ox = (GEOM == 0 ? wb.wx0 + 1 : 0) + 2 * tx + (k & 1);
        optr[i] = b < a.B ? a.out + (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.Cout + wb.nb * 64 + cq * 4 : nullptr;
    }
    if (mode == 1 || mode == 2) {
        f32x4 old[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (optr[i]) old[i] = *reinterpret_cast<const f32x4*>(optr[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (optr[i]) val[i] += old[i];
    }
    if (mode >= 2) {
        const int n4 = wb.nb * 64 + (tid & 15) * 4;                  // (the channel quad of a thread is the same in all eight rounds)
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
}

#ifdef AAE_EXPERIMENTS
// ---- one launch per PHASE: 32-channel stages, the exchange buffer overlays the stage buffers; the phases add up in the output buffer
//      (a.mode 0 stores, 1 adds, 2 adds and applies bias / ReLU / BN).  a.U / a.eh / a.ew select the phase.
template <int TA, int TB, bool SWAP, int GEOM>
__global__ __launch_bounds__(512) void conv_wino_phase_kernel(ConvWinoArgs a) {
    AAE_DYN_SMEM(smem_raw);
    const WinoBlock wb = wino_block<GEOM>(blockIdx.x, a.Cout / 64, a.blocks_x, a.blocks_y);
    wino_phase_body<TA, TB, SWAP, GEOM, 32, false, false>(a, a.U, a.eh, a.ew, wb, reinterpret_cast<f32x4*>(smem_raw), reinterpret_cast<float*>(smem_raw));
    wino_store_block<GEOM, 512>(a, a.mode, wb, reinterpret_cast<const float*>(smem_raw));
}
#endif

// ---- one launch per LAYER: the four phases one behind the other in the block (3 x 3, 3 x 2, 2 x 3, 2 x 2 taps), their outputs added
//      up in an exchange buffer of its own (64 KB behind the stage buffers, which shrink to 16-channel stages to make room); the
//      output is written once, with bias / ReLU / BN.  No read-modify-write of the output tensor, one prologue / epilogue per four phases.
template <int GEOM, bool WIDE>
__global__ __launch_bounds__(WIDE ? 256 : 512) void conv_wino_layer_kernel(ConvWinoLayerArgs p) {
    AAE_DYN_SMEM(smem_raw);
    const ConvWinoArgs& a = p.c;
    const WinoBlock wb = wino_block<GEOM>(blockIdx.x, a.Cout / 64, a.blocks_x, a.blocks_y);
    f32x4* lds = reinterpret_cast<f32x4*>(smem_raw);
    float* xch = reinterpret_cast<float*>(smem_raw + wino_layer_stage_bytes<GEOM>());
    wino_phase_body<3, 3, false, GEOM, 16, false, WIDE>(a, p.U4[3], 1, 1, wb, lds, xch);
    wino_phase_body<3, 2, false, GEOM, 16, true, WIDE>(a, p.U4[2], 1, 0, wb, lds, xch);
    wino_phase_body<3, 2, true, GEOM, 16, true, WIDE>(a, p.U4[1], 0, 1, wb, lds, xch);
    wino_phase_body<2, 2, false, GEOM, 16, true, WIDE>(a, p.U4[0], 0, 0, wb, lds, xch);
    wino_store_block<GEOM, WIDE ? 256 : 512>(a, 3, wb, xch);
}

#endif  // AAE_WINO_DECLARATIONS_ONLY

}  // namespace aae
