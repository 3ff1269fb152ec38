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
//
// Everything around the 152 MFMAs of a tile is kept off the vector ALU as far as it goes (the
// first version spent ~1400 VALU instructions per tile on staging/store address arithmetic, as
// many issue cycles as the matrix work itself):
//  * the (row, column) decomposition of the staged elements a thread owns does not depend on
//    the tile: it is computed once per block and kept packed in registers;
//  * VEC4 (uint8 input, W*C % 4 == 0): the patch rows carry `lead` zero floats in front so that
//    whole aligned global dwords map to aligned LDS float4s -- 3 dword loads per thread and
//    tile instead of 11 byte loads;
//  * full tiles store through a wave-uniform base + one per-lane 32-bit offset (no per-store
//    predicate, no 64-bit vector address arithmetic); ragged tiles keep the predicated form;
//  * the A-operand pixel bases are recomputed only when the tile's position inside an output
//    row changes.
#pragma once

#include <type_traits>

#ifndef AAE_FIRST_CHAINS
#define AAE_FIRST_CHAINS 4     // accumulator tiles in flight per wave: 1, 2 or 4 (see the tile loop)
#endif

namespace aae {

struct ConvFirstCore {
    const void* x;          // [B,H,W,C] uint8 or float32
    const float* lut;       // [256] (uint8 input only)
    const float* w;         // HWIO [KS][KS][C][Cout] == [K][Cout]
    const float* bias;
    const float* bn_scale;  // or nullptr
    const float* bn_shift;
    float* out;             // [B,Ho,Wo,Cout]
    int H, W, Ho, Wo, Cout;
    int S, pt, pl;
    int rowlen;             // staged floats per input row: lead + ((Wo-1)*S + KS) * C (VEC4: rounded up to 4)
    int vec4;               // host: the dword-staged form was planned for this launch
    int lead;               // zero floats in front of every staged row (VEC4: makes pl*C + lead a multiple of 4)
    int tiles_per_image;    // ceil(Ho*Wo / 128)
    int total_tiles;        // B * tiles_per_image
    int tiles_per_block;
    int relu;
    float out_scale;        // OUT_PLANES: 2^act_shift of the f32x3h activation format
    int* sat_flag;          // OUT_PLANES: sticky "a value left the fp16 pair range" flag of the encoder (or nullptr)
};
struct ConvFirstArgs : ConvFirstCore {
    TicketPrep prep;        // ticket words of the later launches of this forward call (n == 0: none), see conv_wavek_f32.h
};

constexpr unsigned kFirstRowShift = 20;                 // packed staging unit: row << 20 | offset from the first staged row
constexpr unsigned kFirstOffMask = (1u << kFirstRowShift) - 1;
constexpr unsigned kFirstNoRow = 0xFFFu;                // "column outside the image": never a valid row

// OUT_PLANES: write the activation as two fp16 planes (hi, lo of v*out_scale) for the f32x3h
// implicit GEMM (conv_igemm_x3h.h) instead of fp32.
// VEC4: uint8 input staged by aligned dwords (host guarantees W*C % 4 == 0, rowlen % 4 == 0,
// (pl*C + lead) % 4 == 0).
// GROUP_SPLIT (per-detection batches): the four 32-pixel groups of a tile go to four blocks (blockIdx.z = group), each staging
// the tile's patch and running ONE accumulator chain of 38 MFMA steps instead of four -- at B = 1 a tile kernel launch is 32
// blocks x 152 MFMA steps on a chip of 256 CUs; split, it is 128 blocks x 38.  Same steps in the same order: bit-identical.
// One block's share of the layer: block (bx, by, bz) = (run of tiles, 128-channel tile, GROUP_SPLIT: 32-pixel group).
template <int KS, int C, bool IN_U8, bool OUT_PLANES, bool VEC4, bool GROUP_SPLIT>
__device__ __forceinline__ void conv_first_block(const ConvFirstCore& p, const int bx, const int by, const int bz) {
    static_assert(!VEC4 || IN_U8, "dword staging is the uint8 path");
    static_assert(!GROUP_SPLIT || !OUT_PLANES, "the group-split form exists for the fp32 per-detection path");
    constexpr int KROW = KS * C;
    constexpr int K = KS * KROW;
    constexpr int NK2 = (K + 1) / 2;

    AAE_DYN_SMEM(smem_raw);
    float* lut_s = reinterpret_cast<float*>(smem_raw);     // [256]
    float* patch = lut_s + 256;                            // [rows][rowlen]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int n = by * 128 + wave * 32 + i;
    const bool n_ok = n < p.Cout;
    const bool wave_n_ok = by * 128 + wave * 32 + 32 <= p.Cout;      // wave-uniform

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
    const int t_begin = bx * p.tiles_per_block;
    const int t_end = min(t_begin + p.tiles_per_block, p.total_tiles);
    if (t_begin >= t_end) return;

    // ---- input staging, software-pipelined across tiles ----
    // A staging unit is one element (generic) or one aligned dword of four uint8 (VEC4).  Unit
    // u = tid + 256 j of the patch is (row r, column q) with u = r * units_per_row + q whatever
    // the tile, so r and the unit's offset from the first staged input row are worked out once.
    // The raw values of tile T+1 are fetched into registers before the MFMA phase of tile T and
    // converted + written to LDS after it: the global-load latency sits under ~10k cycles of
    // matrix work.  Units beyond 256*kUpt (wide images) are staged synchronously.
    constexpr int kUpt = VEC4 ? 3 : 12;
    constexpr int kUnit = VEC4 ? 4 : 1;
    const int units_per_row = p.rowlen / kUnit;
    const int col0 = p.pl * C + p.lead;                    // staged column of image column 0
    auto pack_unit = [&](int u) -> unsigned {
        const int r = u / units_per_row, q = u - r * units_per_row;
        const int src = q * kUnit - col0;
        const long long off = (long long)r * WC + src;
        return ((unsigned)src < (unsigned)WC && r < (int)kFirstNoRow && off <= (long long)kFirstOffMask)
                   ? (((unsigned)r << kFirstRowShift) | (unsigned)off) : (kFirstNoRow << kFirstRowShift);
    };
    unsigned upack[kUpt];
#pragma unroll
    for (int j = 0; j < kUpt; ++j) upack[j] = pack_unit(tid + 256 * j);
    unsigned raw[kUpt];

    struct TileGeom {
        int b, p0, pend, oh_first, nrows;
        int row_lo, row_span;              // staged rows [row_lo, row_lo + row_span) lie inside the image
        long long row0_at;                 // element index of (b, in_row0, 0, 0) -- may point before the image
    };
    auto geom = [&](int T) {
        TileGeom g;
        g.b = T / p.tiles_per_image;
        g.p0 = (T - g.b * p.tiles_per_image) * 128;
        g.pend = min(g.p0 + 128, HoWo);
        g.oh_first = g.p0 / p.Wo;
        const int oh_last = (g.pend - 1) / p.Wo;
        const int in_row0 = g.oh_first * p.S - p.pt;
        g.nrows = (oh_last - g.oh_first) * p.S + KS;
        g.row_lo = max(0, -in_row0);
        g.row_span = max(0, min(g.nrows, p.H - in_row0) - g.row_lo);
        g.row0_at = ((long long)g.b * p.H + in_row0) * WC;
        return g;
    };
    auto fetch_packed = [&](const TileGeom& g, unsigned pk) -> unsigned {
        const unsigned r = pk >> kFirstRowShift;
        if (r - (unsigned)g.row_lo < (unsigned)g.row_span) {
            const unsigned off = pk & kFirstOffMask;
            if (VEC4) return *reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(p.x) + g.row0_at + off);
            if (IN_U8) return (reinterpret_cast<const unsigned char*>(p.x) + g.row0_at)[off];
            return (reinterpret_cast<const unsigned*>(p.x) + g.row0_at)[off];
        }
        return 0u;                                   // zero border: lut[0] == 0.f
    };
    auto put = [&](int u, unsigned v) {
        if (VEC4) {
            f32x4 f;
            f[0] = lut_s[v & 255u]; f[1] = lut_s[(v >> 8) & 255u]; f[2] = lut_s[(v >> 16) & 255u]; f[3] = lut_s[v >> 24];
            *reinterpret_cast<f32x4*>(patch + 4 * u) = f;
        } else {
            patch[u] = IN_U8 ? lut_s[v] : __builtin_bit_cast(float, v);
        }
    };
    auto prefetch = [&](const TileGeom& g) {
#pragma unroll
        for (int j = 0; j < kUpt; ++j) raw[j] = fetch_packed(g, upack[j]);
    };
    auto stage = [&](const TileGeom& g) {
        const int total = g.nrows * units_per_row;
#pragma unroll
        for (int j = 0; j < kUpt; ++j) {
            const int u = tid + 256 * j;
            if (u < total) put(u, raw[j]);
        }
        for (int u = tid + 256 * kUpt; u < total; u += 256) put(u, fetch_packed(g, pack_unit(u)));
    };

    TileGeom g = geom(t_begin);
    prefetch(g);
    __syncthreads();                           // lut_s visible
    stage(g);
    __syncthreads();
    TileGeom gn = g;
    if (t_begin + 1 < t_end) { gn = geom(t_begin + 1); prefetch(gn); }

    int abase[4] = {0, 0, 0, 0};
    int abase_key = -1;
    // element offset of this lane inside a tile's output.  f32x3h pairs (x3h_pair_index): a row is 2*Cout halves,
    // channel chunk n/32 holds 32 hi halves then 32 lo halves
    constexpr int kRowMul = OUT_PLANES ? 2 : 1;
    const unsigned lane_out = OUT_PLANES ? (unsigned)(8 * h * p.Cout + ((n >> 5) << 6) + (n & 31)) : (unsigned)(4 * h * p.Cout + n);

    // one finished value; `at` = this lane's element of the tile row (fp32, or the hi half of the f32x3h pair)
    using out_t = std::conditional_t<OUT_PLANES, unsigned short, float>;
    auto emit = [&](out_t* at, float v) {
        if constexpr (OUT_PLANES) {
            unsigned short hi, lo;
            split_f16_checked(v * p.out_scale, hi, lo, p.sat_flag);
            at[0] = hi;
            at[32] = lo;
        } else {
            at[0] = v;
        }
    };
    auto a_off = [&](int s) {                  // LDS offset of k-step s for this half-wave
        const int k0 = 2 * s, k1 = (2 * s + 1 < K) ? 2 * s + 1 : K - 1;   // k=K (odd K pad) re-reads k=K-1 against a zero weight
        const int off0 = (k0 / KROW) * p.rowlen + (k0 % KROW);
        const int off1 = (k1 / KROW) * p.rowlen + (k1 % KROW);
        return h ? off1 : off0;
    };

    auto run = [&](auto relu_c, auto bn_c) {
        constexpr bool RELU = decltype(relu_c)::value, BN = decltype(bn_c)::value;
        auto finish = [&](float v) {
            v += bias;
            if (RELU) v = fmaxf(v, 0.f);
            if (BN) v = v * sc + sh;
            return v;
        };
        for (int T = t_begin; T < t_end; ++T) {
            const int b = g.b, p0 = g.p0, pend = g.pend, oh_first = g.oh_first;

            // A-operand bases depend only on where the tile starts inside its output row and on a ragged end
            const int key = (p0 - oh_first * p.Wo) * 256 + (pend - p0);
            if (key != abase_key) {
                abase_key = key;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    int px = p0 + mt * 32 + i;
                    px = min(px, pend - 1);        // tail lanes recompute the last pixel; masked at the store
                    const int oh = px / p.Wo, ow = px - oh * p.Wo;
                    abase[mt] = (oh - oh_first) * p.S * p.rowlen + ow * p.S * C + p.lead;
                }
            }
            const long long tile_at = ((long long)b * HoWo + p0) * p.Cout;            // wave-uniform
            const bool more = T + 1 < t_end;

            // The four 32-pixel groups of the tile run in batches of kChains accumulator tiles; the
            // finished values of one batch are stored BETWEEN the MFMAs of the next.  The schedule is
            // pinned by hand (sched_fence after every step): A values come through a ring filled kRing
            // steps ahead, stores and their pointer bumps sit in fixed MFMA gaps.
            // Measured at B=256 on one box (ms): kChains 4 (all groups at once, every store after the
            // barrier) 0.186, 2 (stores of groups 0-1 under the MFMAs of groups 2-3) 0.195, 1 (one
            // chain, 3 waves/SIMD) 0.186 but 0.209 when a wave has its SIMD to itself (an instruction
            // between two MFMAs on the SAME accumulator costs ~43 cycles).  Ablation of the 4-chain
            // form: MFMA phase alone 0.150, stores + staging alone 0.087, everything 0.203 before the
            // VALU diet described in the header -- the in-wave store overlap buys nothing on top of
            // what the SIMD's other wave already hides, so the default keeps four chains.
            constexpr int kChains = GROUP_SPLIT ? 1 : AAE_FIRST_CHAINS;
            constexpr int kBatches = GROUP_SPLIT ? 1 : 4 / kChains;             // batches of kChains groups this block runs
            const int g0 = GROUP_SPLIT ? bz : 0;                                // its first (only) group
            constexpr int kRing = 8;
            constexpr int kVals = 16 * kChains;
            static_assert(NK2 >= kRing && 4 % kChains == 0, "ring deeper than the chain");
            const int lim = pend - p0 - 4 * h;             // tile rows below lim exist (ragged image tail)
            auto tile_body = [&](auto full_c) {
                constexpr bool FULL = decltype(full_c)::value;
                // Rows leave in tile order (+1, +1, +1, +5 rows from one accumulator register to the next),
                // so the address is one running pointer: nothing tile-invariant for the compiler to hoist
                // into 64 register pairs.
                out_t* sp = reinterpret_cast<out_t*>(p.out) + kRowMul * (tile_at + (long long)g0 * 32 * p.Cout) + lane_out;
                const long long step1 = kRowMul * p.Cout, step5 = 5ll * kRowMul * p.Cout;
                auto emit_next = [&](int row, float v) {
                    if (FULL || (n_ok && row < lim)) emit(sp, v);
                    sp += ((row & 7) == 3) ? step5 : step1;
                };
                float av[kChains][kRing];
#pragma unroll
                for (int s = 0; s < kRing; ++s)
#pragma unroll
                    for (int c = 0; c < kChains; ++c) av[c][s] = patch[abase[g0 + c] + a_off(s)];
                float outv[kVals];
#pragma unroll
                for (int bt = 0; bt < kBatches; ++bt) {
                    f32x16 acc[kChains];
#pragma unroll
                    for (int c = 0; c < kChains; ++c)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
#pragma unroll
                    for (int s = 0; s < NK2; ++s) {
                        const int slot = (bt * NK2 + s) % kRing;
#pragma unroll
                        for (int c = 0; c < kChains; ++c) {
                            acc[c] = mfma_32x32x2(av[c][slot], breg[s], acc[c]);
                            if (s + kRing < NK2) av[c][slot] = patch[abase[g0 + bt * kChains + c] + a_off(s + kRing)];
                            else if (bt + 1 < kBatches) av[c][slot] = patch[abase[(bt + 1) * kChains + c] + a_off(s + kRing - NK2)];
                        }
                        if (bt > 0) {
#pragma unroll
                            for (int v = s * kVals / NK2; v < (s + 1) * kVals / NK2; ++v) {
                                const int r = v & 15;
                                emit_next((bt - 1) * kChains * 32 + (v >> 4) * 32 + (r & 3) + 8 * (r >> 2), outv[v]);
                            }
                        }
                        sched_fence();
                    }
#pragma unroll
                    for (int v = 0; v < kVals; ++v) outv[v] = finish(acc[v >> 4][v & 15]);
                    sched_fence();
                }
                // next tile's patch goes into LDS before the last batch's stores are issued (its loads
                // have been in flight for the whole MFMA phase; the stores never delay a vmcnt wait)
                __syncthreads();                   // every wave is done reading this tile's patch
                if (more) stage(gn);
                __syncthreads();
#pragma unroll
                for (int v = 0; v < kVals; ++v) {
                    const int r = v & 15;
                    emit_next((GROUP_SPLIT ? g0 * 32 : 128 - kChains * 32) + (v >> 4) * 32 + (r & 3) + 8 * (r >> 2), outv[v]);
                }
            };
            if (pend - p0 == 128 && wave_n_ok) tile_body(std::true_type{});
            else tile_body(std::false_type{});    // ragged tile (image tail) or a channel tile cut by Cout: predicated stores
            g = gn;
            if (T + 2 < t_end) { gn = geom(T + 2); prefetch(gn); }
        }
    };
    using T1 = std::true_type;
    using T0 = std::false_type;
    if (p.relu) { if (p.bn_scale) run(T1{}, T1{}); else run(T1{}, T0{}); }
    else        { if (p.bn_scale) run(T0{}, T1{}); else run(T0{}, T0{}); }
}

template <int KS, int C, bool IN_U8, bool OUT_PLANES, bool VEC4 = false, bool GROUP_SPLIT = false>
__global__ __launch_bounds__(256) void conv_first_f32_kernel(const ConvFirstArgs p) {
    // ticket preparation for the later launches of this forward: an EXTRA block (the host adds one to the grid) does
    // nothing else, so no working block is delayed (at B = 1 the 32 working blocks leave most CUs free anyway)
    if (p.prep.n > 0 && blockIdx.x == gridDim.x - 1) {
        if (blockIdx.y == 0 && blockIdx.z == 0) ticket_prep_install(p.prep);
        return;
    }
    conv_first_block<KS, C, IN_U8, OUT_PLANES, VEC4, GROUP_SPLIT>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// ---- the same layer for SEVERAL objects in one launch (multi_launch.h): blocks [first[o], first[o + 1]) of grid.x run object o's
// crops with object o's weights; behind them one ticket-preparation block per object.  Per-detection form only (uint8 or float
// crops, fp32 output, one block per 32-pixel group).
struct ConvFirstMultiArgs {
    MultiRange range;                      // working blocks per object
    ConvFirstCore item[kMultiMax];
    MultiTicketPrep prep[kMultiMax];       // (n == 0: nothing to prepare for that object)
    unsigned nonce;                        // one per call: every ticketed launch of the call has its own words
};
// GROUP_SPLIT = true: the per-detection form (one block per 32-pixel group); false: whole 128-pixel tiles per block, runs of tiles per block
// -- the mid-batch groups (objects with five or more detections each, aae_multi_impl.h: launch_mid_group)
template <int KS, int C, bool IN_U8, bool VEC4, bool GROUP_SPLIT = true>
__global__ __launch_bounds__(256) void conv_first_multi_kernel(const ConvFirstMultiArgs m) {
    const int total = m.range.first[m.range.n];
    if ((int)blockIdx.x >= total) {                                   // the preparation blocks: one per object
        const int o = (int)blockIdx.x - total;
        if (blockIdx.y == 0 && blockIdx.z == 0 && o < m.range.n) multi_ticket_prep_install(m.prep[o], m.nonce);
        return;
    }
    const int o = multi_find(m.range, (int)blockIdx.x);
    conv_first_block<KS, C, IN_U8, false, VEC4, GROUP_SPLIT>(m.item[o], (int)blockIdx.x - m.range.first[o], (int)blockIdx.y, (int)blockIdx.z);
}

}  // namespace aae
