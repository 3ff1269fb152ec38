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
//           = at most 128 registers, two waves share a SIMD and one's cluster (patch reads, transform, loads) runs under the other's
//           MFMA burst.  The two halves of the output transform meet through LDS once per phase.  A = rows, or columns when SWAP (the
//           2 x 3-tap phase: the 3-tap dimension is the one that splits evenly).
//   K loop = stages of 16 input channels (32 in the one-launch-per-phase form): the block's window of the sub-image (tiles + halo, zero outside the image = the 'SAME'
//           padding) goes global -> registers -> LDS, double buffered, laid out [channel quad][image][column parity][row][column / 2]
//           with pitches that make the patch reads (ds_read_b128 by 32 tiles) conflict-free.  The loop runs in units of one point row,
//           each ONE cluster + ONE burst (wino_phase_body: the price list of what an instruction costs next to fp32 MFMAs): a lane transforms
//           the patch rows it read before the previous burst with packed fp32 adds, loads the weight fragments of the unit after next (raw
//           buffer views, scalar offsets) into the register slot the previous burst released, reads the next unit's patch rows as float4
//           (4 channels of its K half), and then issues the unit's 4 MFMAs per point back to back.
//   weights: packed [32-column block][8-channel group][point = a PB + b][K half][32 columns][4 channels] per phase (aae_encoder_plan.h).
#pragma once
#include <type_traits>

#include "multi_launch.h"

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
    int regions;             // window regions of the launch: B * blocks_x * blocks_y (GEOM 0), ceil(B / 4) (GEOM 1)
    int xcd_cols;            // 64-column blocks of one region that run on ONE XCD (wino_block); 0: blocks in plain (column block, region) order
#ifdef AAE_WINO_STAMPS
    long long* stamps;       // [block][wave][kWinoStampSlots] shader-clock stamps (tools/ubench/wino_layer_time.hip)
#endif
};
#ifdef AAE_WINO_STAMPS
constexpr int kWinoStampSlots = 48;
#define AAE_WINO_STAMP(a, slot) do { if ((threadIdx.x & 63) == 0) (a).stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * aae::kWinoStampSlots + (slot)] = aae::clock_ticks(); } while (0)
#else
#define AAE_WINO_STAMP(a, slot) do {} while (0)
#endif
// Blocks of a launch for `regions` window regions x `nbn` column blocks with `xcd_cols` of them per XCD (the grid is padded so that every
// XCD gets whole regions; surplus blocks leave at once).
inline unsigned wino_grid_blocks(int regions, int nbn, int xcd_cols) {
    if (xcd_cols <= 0) return (unsigned)regions * (unsigned)nbn;
    const int region_groups = 8 / (nbn / xcd_cols);
    return 8u * (unsigned)xcd_cols * (unsigned)((regions + region_groups - 1) / region_groups);
}
// the mapping is defined for column-block counts that split evenly over the 8 XCDs
inline bool wino_xcd_cols_valid(int nbn, int xcd_cols) {
    return xcd_cols >= 1 && nbn % xcd_cols == 0 && (nbn / xcd_cols) <= 8 && 8 % (nbn / xcd_cols) == 0;
}

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
// ... and of the same launch over SEVERAL objects (one network shape; each its own activations, Winograd-domain weights and epilogue
// vectors): the grouped mid-batch query of aae_encode_nn_multi.  c holds the shared geometry, c.regions the regions of all objects; regions
// [range.first[o], range.first[o + 1]) belong to object o.  Lives in the kernel-argument segment (multi_launch.h).
struct ConvWinoObject {
    const float* x;
    float* out;
    const float* U4[4];
    const float* bias;
    const float* bn_scale;
    const float* bn_shift;
    int B, pad_;
};
struct ConvWinoMultiArgs {
    ConvWinoArgs c;
    MultiRange range;
    ConvWinoObject obj[kMultiMax];
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
// Which block takes which (region, column block): physical block p runs on XCD p % 8 (observed dispatch order; a wrong guess costs speed
// only), and every XCD has an L2 of its own.  The nbn column blocks of a region read the SAME input window and different weights; the
// regions of a column block read the same weights and different windows.  With xcd_cols = S the XCDs form nbn / S column groups x
// 8 S / nbn region groups: an XCD runs S column blocks of a region side by side (consecutive slots of the XCD), so a window crosses the
// fabric nbn / S times and an XCD's L2 streams S / nbn of the weights.  A region group takes a CONTIGUOUS chunk of the region list, in
// order: the 32 / S regions an XCD works on at a time are neighbours -- in a grouped launch (ConvWinoMultiArgs) regions of ONE object,
// which share their weight fetches.  xcd_cols = 0: plain order p = region * nbn + nb (for nbn = 4 | 8 every column block on its own
// XCD: the window is fetched nbn times -- 2.8 / 2.5 / 1.2 GB per launch for conv2 / conv3 / conv4 at B = 256 against 0.8 / 0.4 / 0.2 GB
// of tensors, profiles/r14/pmc_summary.txt).
struct WinoBlock {
    int nb, region, img0, wy0, wx0;
    bool live;
};
__device__ __forceinline__ void wino_block_index(int block, int nbn, int regions, int xcd_cols, WinoBlock& w) {
    if (xcd_cols > 0) {
        const int col_groups = nbn / xcd_cols, region_groups = 8 / col_groups, chunk = (regions + region_groups - 1) / region_groups;
        const int xcd = block & 7, within = block >> 3, k = within / xcd_cols;
        w.nb = (xcd % col_groups) * xcd_cols + within % xcd_cols;
        w.region = (xcd / col_groups) * chunk + k;
        w.live = k < chunk && w.region < regions;
    } else {
        w.nb = block % nbn;
        w.region = block / nbn;
        w.live = w.region < regions;
    }
}
// region `local` of one object's launch: the images / window it covers
template <int GEOM>
__device__ __forceinline__ void wino_block_geometry(int local, int blocks_x, int blocks_y, WinoBlock& w) {
    if (GEOM == 0) {
        int rest = local;
        const int bx = rest % blocks_x;
        rest /= blocks_x;
        const int by = rest % blocks_y;
        w.img0 = rest / blocks_y;
        w.wy0 = 16 * by - 1;
        w.wx0 = 16 * bx - 1;
    } else {
        w.img0 = 4 * local;
        w.wy0 = w.wx0 = -1;
    }
}
template <int GEOM>
__device__ __forceinline__ WinoBlock wino_block(int block, int nbn, int blocks_x, int blocks_y, int regions, int xcd_cols) {
    WinoBlock w;
    wino_block_index(block, nbn, regions, xcd_cols, w);
    wino_block_geometry<GEOM>(w.region, blocks_x, blocks_y, w);
    return w;
}
// tile t (0 ... 31) of the wave with tile half mh: (image of the block, tile row, tile column)
template <int GEOM>
__device__ __forceinline__ void wino_tile(int mh, int t, int& ti, int& ty, int& tx) {
    if (GEOM == 0) { ti = 0; ty = 4 * mh + (t >> 3); tx = t & 7; }
    else { ti = 2 * mh + (t >> 4); ty = (t >> 2) & 3; tx = t & 3; }
}


// ---- what an instruction costs next to the fp32 matrix stream (tools/ubench/mfma_coissue.hip, profiles/r15/mfma_coissue.jsonl) ----------
// v_mfma_f32_32x32x2_f32 holds a SIMD's matrix pipe for 64 cycles; two waves per SIMD keep it busy (0.999) as long as NOTHING else is
// issued.  Everything else is paid in pipe time: a packed fp32 add 13 cycles when it stands alone between two MFMAs and 5 in a run of its
// kind (VALU and the fp32 matrix stream do not overlap on a SIMD), a buffer load ~20 alone and ~0 in a run, a ds_read_b128 3 ... 8; wave
// priorities change nothing.  The K loop below is built on that price list: per unit ONE cluster of everything that is not an MFMA and
// ONE burst of 4 PB MFMAs (the kernel of round 5 wove the transform between the MFMAs of a unit: 0.74-0.79 of the matrix rate in the K
// loop by in-kernel stamps, profiles/r15/wino_layer_stamps_r14_kernel.jsonl; this form 0.83-0.86).

// A thread's share of a stage fill: source offsets (for the polyphase component (0, 0): the others lie a constant further, added as the
// scalar part of the load) and LDS slots.  The same for every stage and every component of a block: computed once per block.
template <int GEOM, int STAGE_CH, int NT>
struct WinoFillPlan {
    static constexpr int kQuads = STAGE_CH / 4, kParts = STAGE_CH / 16;
    static constexpr int kStageQuads = WinoGeom<GEOM>::kImages * WinoGeom<GEOM>::kRows * WinoGeom<GEOM>::kCols * kQuads;
    static constexpr int kPer = ((kStageQuads + NT - 1) / NT + kParts - 1) / kParts;       // float4 per thread and part
    uint32_t goff[kParts * kPer];
    int lslot[kParts * kPer];
};
constexpr uint32_t kWinoOutside = 0x80000000u;      // a lane offset beyond every buffer view: the load returns zeros ('SAME' padding, empty image slots)
template <int GEOM, int STAGE_CH, int NT>
__device__ __forceinline__ void wino_fill_plan(const ConvWinoArgs& a, const WinoBlock& wb, WinoFillPlan<GEOM, STAGE_CH, NT>& f) {
    using G = WinoGeom<GEOM>;
    using F = WinoFillPlan<GEOM, STAGE_CH, NT>;
    constexpr int kPlane = wino_plane_units<GEOM>();
    const int cq_per_pixel = a.Cin / 4;
#pragma unroll
    for (int i = 0; i < F::kParts * F::kPer; ++i) {
        const int idx = (int)threadIdx.x + NT * i;
        f.goff[i] = kWinoOutside;
        f.lslot[i] = -1;
        if (idx < F::kStageQuads) {
            const int pixel = idx / F::kQuads, cq = idx - pixel * F::kQuads;
            const int wi = pixel / (G::kRows * G::kCols), rem = pixel - wi * (G::kRows * G::kCols);
            const int wy = rem / G::kCols, wx = rem - wy * G::kCols;
            const int uu = wb.wy0 + wy, vv = wb.wx0 + wx, b = wb.img0 + wi;
            if (uu >= 0 && uu < a.Ho && vv >= 0 && vv < a.Wo && b < a.B)
                f.goff[i] = (uint32_t)(((((size_t)b * a.H + 2 * uu) * a.W + 2 * vv) * cq_per_pixel + cq) * 16);
            f.lslot[i] = cq * kPlane + wi * G::kImagePitch + (wx & 1) * G::kParityPitch + wy * G::kRowPitch + (wx >> 1);
        }
    }
}

// One polyphase component for one block: the K loop over all input channels and the two-wave output transform.  Leaves the block's
// 64 tiles x 4 pixels x 64 channels in the exchange buffer `xch_all` ([wave pair mh + 2 nh][accumulator register][lane] float4 = the four
// pixels k = 2 dy + dx of a tile): stored when !ACCUMULATE, added to what is there otherwise.  Ends behind a block barrier.
// TA / TB: taps along the split dimension A / the other dimension B (3 | 2).  SWAP: A = columns.  STAGE_CH: input channels per LDS stage.
// WIDE: block = 4 waves (mh, ph), each over BOTH 32-channel halves (256 accumulator registers: one wave per SIMD).  Otherwise 8 waves
// (mh, nh, ph), one 32-channel half each, two waves per SIMD.
// CHAIN (the one-launch-per-layer kernel): bit 0 -- the first stage of this component already lies in stage buffer `buf0` and `uin` holds the
// weight fragments of units 0 and 1 (the previous component fetched them); bit 1 -- this component does the same for the next one (taps NTA x
// NTB, weights Un, parities neh / new_): its first stage is loaded during the last stage here, its first fragments behind the last burst.
template <int TA, int TB, bool SWAP, int GEOM, int STAGE_CH, bool ACCUMULATE, bool WIDE, int CHAIN = 0, int NTA = 3, int NTB = 3>
__device__ __forceinline__ void wino_phase_body(const ConvWinoArgs& a, const float* U, int eh, int ew, const WinoBlock& wb, f32x4* lds, float* xch_all,
                                                const WinoFillPlan<GEOM, STAGE_CH, WIDE ? 256 : 512>& fill, int& buf0, f32x4 (&uin)[8],
                                                const float* Un, int neh, int new_, f32x4 (&uout)[8]) {
    using G = WinoGeom<GEOM>;
    using F = WinoFillPlan<GEOM, STAGE_CH, WIDE ? 256 : 512>;
    constexpr int PB = TB + 1;                                  // points (= patch positions) along B
    constexpr int kQuads = STAGE_CH / 4, kGroups = STAGE_CH / 8;                    // channel quads / 8-channel groups per stage
    constexpr int kPlane = wino_plane_units<GEOM>(), kStage = kQuads * kPlane;
    constexpr int kParity = G::kParityPitch;                    // units between the two column-parity halves of an image
    constexpr int kParts = F::kParts, kPer = F::kPer;           // the fill of the next stage happens in this many parts (few staging registers live at a time)
    constexpr int NH = WIDE ? 2 : 1;                            // 32-channel halves per wave
    constexpr int kOffA = TA == 2 ? 1 : 0, kOffB = TB == 2 ? 1 : 0;   // a 2-tap dimension starts one sample into the window
    static_assert(CHAIN == 0 || !WIDE, "the chained form passes one wave's weight fragments between the components");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mh = wave & 1, nh = WIDE ? 0 : (wave >> 1) & 1, ph = WIDE ? wave >> 1 : wave >> 2, m = lane & 31, h = lane >> 5;
    int li, lty, ltx;
    wino_tile<GEOM>(mh, m, li, lty, ltx);
    const int n32 = wb.nb * 2 + nh, KG = a.Cin / 8, nst = a.Cin / STAGE_CH;
    constexpr int NP = (TA + 1) * PB;
    constexpr int kPhaseNo = TA == 3 && TB == 3 ? 0 : (TA == 3 && !SWAP ? 1 : (TA == 3 ? 2 : 3));      // (stamps only)
    (void)kPhaseNo;
    AAE_WINO_STAMP(a, 4 * kPhaseNo + 0);
    // both operands through raw buffer views (scalar base + one 32-bit lane offset; out-of-range = zeros: the 'SAME' padding and the empty
    // image slots of a ragged group need no branches).  The host keeps the activation below 2 GiB for that (0x80000000 marks "outside").
    const buffer_rsrc xrs = make_buffer(a.x, (uint32_t)((size_t)a.B * a.H * a.W * a.Cin * 4));
    const buffer_rsrc urs = make_buffer(U, (uint32_t)((size_t)NP * a.Cin * a.Cout * 4));
    const uint32_t ulane = (uint32_t)(((size_t)n32 * KG * NP + (size_t)ph * 2 * PB) * 64 + h * 32 + m) * 16u;
    const bool two_rows = TA == 3 || ph == 0;                   // point rows of A this wave owns: 2, or 1 (the third row of F(2, 2))
    const uint32_t parity_off = (uint32_t)(((size_t)eh * a.W + ew) * a.Cin * 4);          // this component's samples inside the (0, 0) component's offsets
    const uint32_t next_parity_off = (uint32_t)(((size_t)neh * a.W + new_) * a.Cin * 4);

    f32x16 acc[NH][2 * PB];
#pragma unroll
    for (int n2 = 0; n2 < NH; ++n2)
#pragma unroll
        for (int p = 0; p < 2 * PB; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n2][p][r] = 0.f;

    // ---- stage fill: global -> registers -> LDS (the buffer of the next stage is free for the whole of the current one)
    f32x4 stg[kPer];
    auto stage_load = [&](uint32_t soff, int part) {
#pragma unroll
        for (int i = 0; i < kPer; ++i) stg[i] = buffer_load4_s(xrs, fill.goff[part * kPer + i], soff);
    };
    auto stage_store = [&](int buf, int part) {
#pragma unroll
        for (int i = 0; i < kPer; ++i)
            if (fill.lslot[part * kPer + i] >= 0) lds[buf * kStage + fill.lslot[part * kPer + i]] = stg[i];
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

    // ---- weight fragments: register slot s = unit & 1.  Units 0 and 1 are rows 0, 1 of group 0 -- or, for the wave with ONE point row, that
    //      row of groups 0 and 1.
    const uint32_t half_stride = (uint32_t)KG * NP * 1024u;      // the next 32-column block of the packed weights lies KG * NP KB on
    f32x4 u[NH][2 * PB];
    if (CHAIN & 1) {
#pragma unroll
        for (int p = 0; p < 2 * PB; ++p) u[0][p] = uin[p];
    } else {
#pragma unroll
        for (int n2 = 0; n2 < NH; ++n2)
#pragma unroll
            for (int p = 0; p < 2 * PB; ++p)
                u[n2][p] = buffer_load4_s(urs, ulane + (p % PB) * 1024u, (two_rows ? (p / PB) * PB * 1024u : (p / PB) * (NP * 1024u)) + n2 * half_stride);
#pragma unroll
        for (int part = 0; part < kParts; ++part) {
            stage_load(parity_off, part);
            stage_store(buf0, part);
        }
        __syncthreads();
    }
    AAE_WINO_STAMP(a, 4 * kPhaseNo + 1);

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
    // ---- the K loop: per unit (one point row of one 8-channel group) ONE cluster and ONE burst.
    //        cluster: transform the patch that was read before the previous burst -> weight loads for the unit after next (into the slot the
    //                 previous burst released) -> the stage fill's loads / stores -> patch reads of the next unit
    //        burst:   4 PB MFMAs back to back; the partner wave of the SIMD runs its cluster underneath.
    //      The stage barrier stands INSIDE a cluster, between "every read of this stage is issued" and "the first read of the next stage":
    //      nothing waits behind it with an empty pipe.  The two waves of a SIMD run about one burst apart (the older one leads), so the
    //      ph = 1 waves take the barrier one unit earlier in their instruction stream than the ph = 0 waves -- in time the eight arrive
    //      together.  (What the barrier guarantees does not depend on where a wave takes it: a wave arrives behind its own last read of the
    //      stage and its own fill stores, and reads the next stage / refills this stage's buffer only behind it.)
    auto run = [&](auto rows_tag) {
        constexpr int ROWS = decltype(rows_tag)::value, NU = kGroups * ROWS, kPartUnits = NU / kParts;
        static_assert(NU % 2 == 0 && NU % kParts == 0, "units per stage");
        constexpr int kStoreBack = (kParts == 1 && NU >= 4) ? 2 : 1;            // the fill is stored this many units before its part ends
        const bool early = ph == 1 && NU >= 4 && kParts == 1;                   // this wave takes the stage barrier in the cluster of unit NU - 2
        f32x4 raw[2 * PB], v[PB];
        read_unit(lds + buf0 * kStage + h * kPlane + lane_base, 0, raw);
        for (int st = 0; st < nst; ++st) {
            const int buf = (buf0 + st) & 1;
            const bool more = st + 1 < nst;
            const bool filling = more || (CHAIN & 2);                            // a stage (this component's next, or the next component's first) is on its way into buf ^ 1
            const uint32_t fill_off = more ? parity_off + (uint32_t)(st + 1) * kQuads * 16 : next_parity_off;
#ifdef AAE_WINO_STAMPS
            if (kPhaseNo == 0 && st < 8) AAE_WINO_STAMP(a, 20 + st);
#endif
            const f32x4* stage = lds + buf * kStage + h * kPlane + lane_base;
            const f32x4* next_stage = lds + (buf ^ 1) * kStage + h * kPlane + lane_base;
#pragma unroll
            for (int t = 0; t < NU; ++t) {
                const int i = t % ROWS, slot = t & 1;
#ifdef AAE_WINO_STAMPS
                if (kPhaseNo == 0 && (st == 2 || st == 3) && t < 4) AAE_WINO_STAMP(a, 28 + 6 * (st - 2) + t);
#endif
                // ---- cluster
                step_a(i, raw, v);
                step_b(v);
                if ((st > 0 || t > 0) && (more || t + 1 < NU)) {
                    // unit T + 1 = st * NU + t + 1: (group, row) = (T' / ROWS, T' % ROWS); its slot is the one unit T - 1 just released
                    const int tn = (t + 1) % NU, gn = (st + (t + 1) / NU) * kGroups + tn / ROWS;
                    const uint32_t un = (uint32_t)gn * (NP * 1024u) + (tn % ROWS) * PB * 1024u;      // (wave-uniform: a scalar register)
#pragma unroll
                    for (int n2 = 0; n2 < NH; ++n2)
#pragma unroll
                        for (int j = 0; j < PB; ++j) u[n2][(slot ^ 1) * PB + j] = buffer_load4_s(urs, ulane + j * 1024u, un + n2 * half_stride);
                }
                // (the fill's loads BEHIND the weight loads: the burst after next waits for these weights, and the counter retires loads in order)
                if (filling && t % kPartUnits == 0) stage_load(fill_off, t / kPartUnits);
                if (filling && t % kPartUnits == kPartUnits - kStoreBack) stage_store(buf ^ 1, t / kPartUnits);
                if (t + 1 < NU) read_unit(stage + 2 * ((t + 1) / ROWS) * kPlane, (t + 1) % ROWS, raw);
                if (filling && (t == NU - 1 ? !early : (t == NU - 2 && early))) {
#ifdef AAE_WINO_STAMPS
                    if (kPhaseNo == 0 && (st == 2 || st == 3)) AAE_WINO_STAMP(a, 28 + 6 * (st - 2) + 4);
#endif
                    __syncthreads();
#ifdef AAE_WINO_STAMPS
                    if (kPhaseNo == 0 && (st == 2 || st == 3)) AAE_WINO_STAMP(a, 28 + 6 * (st - 2) + 5);
#endif
                }
                if (t == NU - 1 && more) read_unit(next_stage, 0, raw);
                sched_fence();
                // ---- burst
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < PB; ++j)
#pragma unroll
                        for (int n2 = 0; n2 < NH; ++n2)
                            acc[n2][(ROWS == 2 ? i * PB : 0) + j] = mfma_32x32x2(v[j][q], u[n2][slot * PB + j][q], acc[n2][(ROWS == 2 ? i * PB : 0) + j]);
                sched_fence();
            }
        }
    };
    if (TA == 3 || ph == 0) run(std::integral_constant<int, 2>());
    else run(std::integral_constant<int, 1>());
    if (STAGE_CH == 32 && !(CHAIN & 2)) __syncthreads();      // (the one-launch-per-phase form: the exchange buffer overlays the stage buffers -- every wave must be through its last patch reads)
    AAE_WINO_STAMP(a, 4 * kPhaseNo + 2);
    if (CHAIN & 2) {
        // the next component's first weight fragments fly while the accumulators drain and leave
        constexpr int NPB = NTB + 1, NNP = (NTA + 1) * NPB;
        const buffer_rsrc nrs = make_buffer(Un, (uint32_t)((size_t)NNP * a.Cin * a.Cout * 4));
        const uint32_t nlane = (uint32_t)(((size_t)n32 * KG * NNP + (size_t)ph * 2 * NPB) * 64 + h * 32 + m) * 16u;
        const bool next_two_rows = NTA == 3 || ph == 0;
#pragma unroll
        for (int p = 0; p < 2 * NPB; ++p)
            uout[p] = buffer_load4_s(nrs, nlane + (p % NPB) * 1024u, next_two_rows ? (p / NPB) * NPB * 1024u : (p / NPB) * (NNP * 1024u));
    }
    buf0 = (buf0 + nst) & 1;

    // ---- output transform.  Along A the rows of A^T m split over the two waves:
    //        ph 0: q0 = m0 + m1, q1 = m1        ph 1, F(2, 3): q0 = m2, q1 = -m2 - m3        ph 1, F(2, 2): q0 = 0, q1 = -m2
    //      each wave applies A along B to its part: four partial outputs (the pixels of a tile) per accumulator register.
    auto partial = [&](int n2, int r, float (&y)[4]) {
        float q0[PB], q1[PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const float mA = acc[n2][j][r], mB = acc[n2][PB + j][r];
            if (ph == 0) { q0[j] = mA + mB; q1[j] = mB; }
            else if (TA == 3) { q0[j] = mA; q1[j] = -mA - mB; }
            else { q0[j] = 0.f; q1[j] = -mA; }
        }
        // (pixel k = 2 iA + iB is output (dy, dx) = (iA, iB), or (iB, iA) when A = columns: stored as 2 dy + dx)
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
    // Every wave first turns its accumulators into partial outputs, then the two waves of a pair add them into the exchange buffer in two
    // rounds with all eight waves at work in both: round one takes the registers r with r % 2 == ph, round two the others (the order of the
    // two additions into an element is fixed: deterministic sums).
    f32x4 yv[NH][16];
#pragma unroll
    for (int n2 = 0; n2 < NH; ++n2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float y[4];
            partial(n2, r, y);
            yv[n2][r] = f32x4{y[0], y[1], y[2], y[3]};
        }
#pragma unroll
    for (int round = 0; round < 2; ++round) {
#pragma unroll
        for (int n2 = 0; n2 < NH; ++n2) {
            f32x4* xq = reinterpret_cast<f32x4*>(xch_all) + (size_t)(mh + 2 * (nh + n2)) * 16 * 64 + lane;
            if (ACCUMULATE || round == 1) {
                f32x4 old[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) old[e] = ph == round ? xq[(2 * e) * 64] : xq[(2 * e + 1) * 64];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (ph == round) xq[(2 * e) * 64] = old[e] + yv[n2][2 * e];
                    else xq[(2 * e + 1) * 64] = old[e] + yv[n2][2 * e + 1];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (ph == round) xq[(2 * e) * 64] = yv[n2][2 * e];
                    else xq[(2 * e + 1) * 64] = yv[n2][2 * e + 1];
                }
            }
        }
        __syncthreads();
    }
    AAE_WINO_STAMP(a, 4 * kPhaseNo + 3);
}

// The block's 64 tiles x 4 pixels x 64 channels leave the exchange buffer ([pair (t / 32, c / 32)][register r = (t & 3) + 4 ((t & 31) / 8)]
// [lane 32 ((t / 4) & 1) + c % 32] float4 = the four pixels k = 2 dy + dx of tile t, channel c): a thread takes a tile's four pixels for
// four consecutive channels (four float4, 64 contiguous bytes), transposes, and writes one float4 per pixel (mode 0 stores, 1 adds to what
// is there, 2 adds and applies bias / ReLU / BN, 3 stores with bias / ReLU / BN).
template <int GEOM, int NT>
__device__ __forceinline__ void wino_store_block(const ConvWinoArgs& a, int mode, const WinoBlock& wb, const float* xall) {
    const int tid = threadIdx.x;
    const int n4 = wb.nb * 64 + (tid & 15) * 4;                      // (the channel quad of a thread is the same in every round)
    f32x4 bs = {0.f, 0.f, 0.f, 0.f}, sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (mode >= 2) {
        bs = *reinterpret_cast<const f32x4*>(a.bias + n4);
        if (a.bn_scale) {
            sc = *reinterpret_cast<const f32x4*>(a.bn_scale + n4);
            sh = *reinterpret_cast<const f32x4*>(a.bn_shift + n4);
        }
    }
#pragma unroll
    for (int round = 0; round < 1024 / NT; ++round) {
        const int idx = tid + NT * round, cq = idx & 15, t = idx >> 4;
        const int tmh = t >> 5, mt = t & 31, tnh = cq >> 3;
        const int r = (mt & 3) + 4 * (mt >> 3), hh = (mt >> 2) & 1;
        const f32x4* src = reinterpret_cast<const f32x4*>(xall) + (size_t)(tmh + 2 * tnh) * 16 * 64 + r * 64 + 32 * hh + (cq & 7) * 4;
        const f32x4 c0 = src[0], c1 = src[1], c2 = src[2], c3 = src[3];
        int ti, ty, tx;
        wino_tile<GEOM>(tmh, mt, ti, ty, tx);
        const int b = wb.img0 + ti;
        if (b >= a.B) continue;
        const int oy = (GEOM == 0 ? wb.wy0 + 1 : 0) + 2 * ty, ox = (GEOM == 0 ? wb.wx0 + 1 : 0) + 2 * tx;
        float* o = a.out + (((size_t)b * a.Ho + oy) * a.Wo + ox) * a.Cout + wb.nb * 64 + cq * 4;
        float* optr[4] = {o, o + a.Cout, o + (size_t)a.Wo * a.Cout, o + (size_t)a.Wo * a.Cout + a.Cout};
        f32x4 val[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) val[k] = f32x4{c0[k], c1[k], c2[k], c3[k]};
        if (mode == 1 || mode == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) val[k] += *reinterpret_cast<const f32x4*>(optr[k]);
        }
        if (mode >= 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                val[k] += bs;
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) val[k][e] = fmaxf(val[k][e], 0.f);
                }
                if (a.bn_scale) val[k] = val[k] * sc + sh;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(optr[k]) = val[k];
    }
}

#ifdef AAE_EXPERIMENTS
// ---- one launch per PHASE: 32-channel stages, the exchange buffer overlays the stage buffers; the phases add up in the output buffer
//      (a.mode 0 stores, 1 adds, 2 adds and applies bias / ReLU / BN).  a.U / a.eh / a.ew select the phase.
template <int TA, int TB, bool SWAP, int GEOM>
__global__ __launch_bounds__(512) void conv_wino_phase_kernel(ConvWinoArgs a) {
    AAE_DYN_SMEM(smem_raw);
    const WinoBlock wb = wino_block<GEOM>(blockIdx.x, a.Cout / 64, a.blocks_x, a.blocks_y, a.regions, a.xcd_cols);
    if (!wb.live) return;
    WinoFillPlan<GEOM, 32, 512> fill;
    wino_fill_plan(a, wb, fill);
    int buf0 = 0;
    f32x4 unused[8];
    wino_phase_body<TA, TB, SWAP, GEOM, 32, false, false>(a, a.U, a.eh, a.ew, wb, reinterpret_cast<f32x4*>(smem_raw), reinterpret_cast<float*>(smem_raw), fill, buf0, unused,
                                                          nullptr, 0, 0, unused);
    wino_store_block<GEOM, 512>(a, a.mode, wb, reinterpret_cast<const float*>(smem_raw));
}
#endif

// ---- one launch per LAYER: the four phases one behind the other in the block (3 x 3, 3 x 2, 2 x 3, 2 x 2 taps), their outputs added
//      up in an exchange buffer of its own (64 KB behind the stage buffers, which shrink to 16-channel stages to make room); the
//      output is written once, with bias / ReLU / BN.  No read-modify-write of the output tensor, one prologue / epilogue per four phases.
//      The phases are CHAINED: each fetches the next one's first stage and first weight fragments while its own K loop ends, so that only
//      the first phase of a block waits for global memory with an empty matrix pipe.
template <int GEOM, bool WIDE>
__device__ __forceinline__ void wino_layer_block(const ConvWinoArgs& a, const float* const (&U4)[4], const WinoBlock& wb, unsigned char* smem_raw) {
    f32x4* lds = reinterpret_cast<f32x4*>(smem_raw);
    float* xch = reinterpret_cast<float*>(smem_raw + wino_layer_stage_bytes<GEOM>());
    WinoFillPlan<GEOM, 16, WIDE ? 256 : 512> fill;
    wino_fill_plan(a, wb, fill);
    int buf0 = 0;
    f32x4 ua[8], ub[8];
    if (WIDE) {
        wino_phase_body<3, 3, false, GEOM, 16, false, WIDE>(a, U4[3], 1, 1, wb, lds, xch, fill, buf0, ua, nullptr, 0, 0, ub);
        wino_phase_body<3, 2, false, GEOM, 16, true, WIDE>(a, U4[2], 1, 0, wb, lds, xch, fill, buf0, ua, nullptr, 0, 0, ub);
        wino_phase_body<3, 2, true, GEOM, 16, true, WIDE>(a, U4[1], 0, 1, wb, lds, xch, fill, buf0, ua, nullptr, 0, 0, ub);
        wino_phase_body<2, 2, false, GEOM, 16, true, WIDE>(a, U4[0], 0, 0, wb, lds, xch, fill, buf0, ua, nullptr, 0, 0, ub);
    } else {
        constexpr bool W = WIDE;        // (false here: the chained form)
        wino_phase_body<3, 3, false, GEOM, 16, false, W, W ? 0 : 2, 3, 2>(a, U4[3], 1, 1, wb, lds, xch, fill, buf0, ua, U4[2], 1, 0, ub);
        wino_phase_body<3, 2, false, GEOM, 16, true, W, W ? 0 : 3, 3, 2>(a, U4[2], 1, 0, wb, lds, xch, fill, buf0, ub, U4[1], 0, 1, ua);
        wino_phase_body<3, 2, true, GEOM, 16, true, W, W ? 0 : 3, 2, 2>(a, U4[1], 0, 1, wb, lds, xch, fill, buf0, ua, U4[0], 0, 0, ub);
        wino_phase_body<2, 2, false, GEOM, 16, true, W, W ? 0 : 1>(a, U4[0], 0, 0, wb, lds, xch, fill, buf0, ub, nullptr, 0, 0, ua);
    }
    wino_store_block<GEOM, WIDE ? 256 : 512>(a, 3, wb, xch);
}

template <int GEOM, bool WIDE>
__global__ __launch_bounds__(WIDE ? 256 : 512) void conv_wino_layer_kernel(ConvWinoLayerArgs p) {
    AAE_DYN_SMEM(smem_raw);
    const ConvWinoArgs& a = p.c;
    const WinoBlock wb = wino_block<GEOM>(blockIdx.x, a.Cout / 64, a.blocks_x, a.blocks_y, a.regions, a.xcd_cols);
    if (!wb.live) return;
    wino_layer_block<GEOM, WIDE>(a, p.U4, wb, smem_raw);
#ifdef AAE_WINO_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AAE_WINO_STAMP(a, 16);
    if (threadIdx.x == 0) a.stamps[((size_t)blockIdx.x * 8) * kWinoStampSlots + 17] = (long long)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15);
#endif
}

// ---- the same launch over several objects (ConvWinoMultiArgs): a block finds the object its region belongs to and runs exactly the
//      single-object block on that object's tensors -- bit-identical to the object's own launch.
template <int GEOM>
__global__ __launch_bounds__(512) void conv_wino_layer_multi_kernel(ConvWinoMultiArgs p) {
    AAE_DYN_SMEM(smem_raw);
    WinoBlock wb;
    wino_block_index(blockIdx.x, p.c.Cout / 64, p.c.regions, p.c.xcd_cols, wb);
    if (!wb.live) return;
    const int o = multi_find(p.range, wb.region);
    wino_block_geometry<GEOM>(wb.region - p.range.first[o], p.c.blocks_x, p.c.blocks_y, wb);
    ConvWinoArgs a = p.c;
    const ConvWinoObject& ob = p.obj[o];
    a.x = ob.x; a.out = ob.out; a.bias = ob.bias; a.bn_scale = ob.bn_scale; a.bn_shift = ob.bn_shift; a.B = ob.B;
    const float* U4[4] = {ob.U4[0], ob.U4[1], ob.U4[2], ob.U4[3]};
    wino_layer_block<GEOM, false>(a, U4, wb, smem_raw);
}

#endif  // AAE_WINO_DECLARATIONS_ONLY

}  // namespace aae
