// Batched codebook arg-max with the QUERIES resident in registers and the codebook streamed
// through LDS -- the batched (B > 4) form of
//   tf.matmul(q, E, transpose_b=True) + argmax   (/root/reference/auto_pose/ae/codebook.py:50-51, 63-64)
// for fp32 rows (exact fp32 MFMA) and bf16 rows (kBf16QueryTerms = 2 bf16 MFMA terms per product, fp32 accumulate).
//
// scan_mfma_kernel / scan_bf16_kernel keep one 128-row codebook tile in LDS and walk the query
// chunks through LDS as well: one block per CU (128 KB of LDS), every chunk pays a staging round
// trip, two barriers and an epilogue, and each MFMA needs more than one 16-byte LDS read.  Here a
// block of 8 waves owns 128 (or 256) queries for its whole life: a wave keeps the MFMA B fragments
// of its query group (32 queries) in registers (64 VGPRs) and accumulates its rows
// of every 32-KB row tile (64 fp32 rows / 128 bf16 rows) that streams through three LDS images:
// one barrier and one 16-byte LDS read per 4 (fp32) or 2 (bf16) MFMAs, the LDS-DMA pieces of tiles
// t+1 and t+2 in flight under the MFMAs of tile t, a running (best score, first row) pair per lane, and
// one partial per (row range, query) at the very end.  The per-accumulator MFMA
// order is the one of the tile-resident kernels, so scores and indices are bit-identical to theirs.
//
// Restrictions (the tile-resident kernels remain for the rest): J == 128, top-1, col_stride == 1,
// no similarity output.
#pragma once

// timing experiments only (tools/ubench/scan_resident_ablate.hip builds the kernel with parts of a tile's work removed;
// results are then wrong): 1 no arg-max / list fold, 2 no LDS fragment reads, 4 no MFMAs, 8 no codebook stream, 16 (top-k) the
// list insertion replaced by one maximum, 32 (top-k) accumulator tiles that pass the pretest are not looked at value by value,
// 64 (top-k) nothing is published to the shared bound words, 128 (top-k) they are never read
#ifndef AAE_SCAN_RESIDENT_ABLATE
#define AAE_SCAN_RESIDENT_ABLATE 0
#endif

namespace aae {

struct ScanResidentArgs {
    const void* E;          // [N][128] fp32 or bf16, row-major
    unsigned e_bytes;
    const void* qp;         // fp32: float [32 slots][Bpad][4]; bf16: ushort [3 terms][16 slots][Bpad][8]
    float* pval;            // [gridDim.x][Bstride] partial best score per row range
    int* pidx;              // [gridDim.x][Bstride] partial best row
    int N, B, Bpad, Bstride;
    int tiles_per_block;    // row tiles one block walks
    // top-k form (K > 0): per (row block, query) the block's best k rows in canonical order (score descending, lower
    // row first on ties), sentinel row 0x7fffffff where a block has fewer -- the candidate lists topk_merge_kernel takes
    float* cand_v = nullptr;    // [B][gridDim.x][k]
    int* cand_i = nullptr;
    int k = 0;
    // top-k pruning across blocks (optional): [kPruneReplicas][Bpad][kPruneGroups] score keys, reset by the normalise kernel in front of the
    // scan; block b raises word b % kPruneGroups of a query to the best score it has seen for it
    int* prune = nullptr;
    // NORM = true (arg-max form): the raw latent codes [B][128] fp32 (16-byte aligned) -- the block normalises its own queries,
    // no l2norm_pack launch in front of the scan; qp is not read then
    const float* z = nullptr;
    // arg-max form, gridDim.y == 1 (opt-in, AAE_SCAN_AUTO_FIN): when tickets != nullptr the last row block to arrive merges the block
    // partials itself and writes (index, score) -- no argmax_reduce launch behind the scan
    unsigned long long* tickets = nullptr;   // kTicketSlotWords words
    unsigned nonce = 0;
    long long* idx_out = nullptr;            // [B] int64
    float* score_out = nullptr;              // [B]
    int idx_scale = 1;
#ifdef AAE_SCAN_COUNT
    int* dbg = nullptr;         // [32 steps][2]: accumulator tiles that passed the pretest, value slots that ran the insertion
#endif
};

constexpr int kScanResidentThreads = 512;
// queries one block owns: 8 waves = RH row parts x (8 / RH) groups of 32 queries
template <int RH>
constexpr int scan_resident_queries() { return 32 * (8 / RH); }
// Geometry of the codebook stream.  A row tile is 32 KB -- 64 fp32 rows / 128 bf16 rows -- in three LDS images (one in use, two in
// flight), except fp32 with RH = 4 (at most 64 queries per block): there a tile is 128 rows = 64 KB in two images (one in flight), so
// that FOUR waves have a 32-row MFMA tile of their own per query group instead of two.
template <bool BF16, int RH>
constexpr int scan_resident_tile_rows() { return BF16 ? 128 : (RH == 4 ? 128 : 64); }
template <bool BF16, int RH>
constexpr int scan_resident_tile_floats() { return scan_resident_tile_rows<BF16, RH>() * (BF16 ? 64 : 128); }
template <bool BF16, int RH>
constexpr int scan_resident_stages() { return scan_resident_tile_floats<BF16, RH>() > 8192 ? 2 : 3; }
template <bool BF16, int RH>
constexpr int scan_resident_smem() { return scan_resident_stages<BF16, RH>() * scan_resident_tile_floats<BF16, RH>() * 4 + 2 * 2 * 256 * 4; }

// K == 0: arg-max (one partial per row block and query).  K > 0: top-k for k <= K WITHOUT the [B][N] similarity matrix:
// every lane keeps the K best (score, row) pairs of the rows it sees, sorted; a new score enters in front of the first
// entry it beats strictly (rows arrive in ascending order per lane, so equal scores keep their row order); the lists that
// share a query (row parts x two lane halves) are merged at the end.  The insertion is ~5 VALU instructions per list slot
// and candidate, skipped when no lane of the wave has a score above its K-th best.
//
// Top-k pruning across blocks.  The per-lane lists of a block are YOUNG -- a lane sees 64 rows per tile, a dozen tiles -- so in
// almost every value slot some lane of the wave has a candidate and the whole wave walks the insertion (config 5: 125 us for
// top-5 against 45 for the arg-max).  But most of those candidates cannot be in the final answer: every block raises one of
// kPruneGroups shared words per query to the best score it has seen; the maxima of different words belong to different rows,
// so the K-th largest word is a LOWER BOUND of the query's final K-th best score, and a value below it is dropped before it
// costs anything (a value equal to it is kept: ties are decided by row later).  After the first tile of every block the bound
// is about the 8th best of 30 000 rows; one value slot in 60 still has a candidate.  The words are read without any
// ordering -- whatever has been published so far is a valid bound -- so the result does not depend on timing.
//
// RH = 4 (B <= 32, arg-max; round 4): ONE query group per block and four waves that share the rows of a tile -- with RH = 2 a batch of
// at most 32 queries kept two of a CU's four matrix pipes idle (20.7 -> 15.8 us per query of the 92 232-row fp32 codebook); fp32 rows
// then come in 128-row tiles = 64 KB, held in two LDS images instead of three.
// RH = 2 (B <= 128): wave (rh, qg) accumulates half the rows of a tile for query group qg -- four query groups per block.
// RH = 1 (B > 128): every wave takes ALL rows of a tile for its own query group -- eight groups = 256 queries per block, so
// the codebook is streamed ONCE for 256 queries instead of once per 128 (config 5, B = 256: 189 -> 94 MB per scan; the kernel
// was bound by that stream at one 32-KB tile in flight per CU: 2.6 TB/s, profiles/r11_small).
// Round 3, all forms: (a) the stream goes global -> LDS by LDS-DMA into three images -- two tiles in flight behind the one in
// use, no staging registers, no LDS writes (was: one tile ahead through registers); (b) top-k: a tile is looked at value by
// value only if its maximum beats some lane's K-th best; (c) arg-max with two or more accumulator tiles per wave: the fold of
// finished accumulators and the LDS fragment reads sit between the MFMAs of the next ones (below).  Same MFMA order per
// accumulator: bit-identical scores and indices.  (A running maximum per accumulator position -- 3 instead of 5 vector
// instructions per value -- was tried for the arg-max: 128 more registers at four accumulator tiles, 300 spilled: dropped.)
// NORM: the query fragments come straight from the raw latent codes.  Lane (i, h) of a wave holds 64 of the 128 elements of
// query q0 + i -- exactly the fragments it multiplies -- and the norm is added up in the order of l2norm_pack_kernel (element
// j and j + 64 by one fma chain, then the butterfly over j % 64 with masks 32 ... 1): the steps of that butterfly that pair
// elements of the same lane are plain adds, the one that pairs the lane halves is an exchange with lane ^ 32.  Same sum, same
// 1 / sqrt, same products: the fragments have the bits the packed planes had (tests/test_emu_kernels.py compares them).
// Every block repeats this for its queries (128 KB of hot L2 reads and ~200 vector instructions per wave against a kernel of
// 40 us); what it buys is the launch in front: config 5 arg-max 53 -> 50 us, the B = 5 ... 256 calls 3 us each.
// (two steps: the raw pieces are requested in front of the first codebook tiles, the arithmetic runs while those are in flight)
template <bool BF16>
__device__ __forceinline__ void scan_resident_load_raw(const float* z, const int query, const bool real, const int h, f32x4 (&zs)[16]) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {   // fp32: slot 2c + h (4 elements); bf16: slot 2s + h = pieces 2s, 2s + 1 (8 elements), c = 2s + piece
        const int piece = BF16 ? (2 * (c >> 1) + h) * 2 + (c & 1) : 2 * c + h;       // 16-byte piece of the query row
        zs[c] = real ? *reinterpret_cast<const f32x4*>(z + (long long)query * 128 + piece * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
template <bool BF16>
__device__ __forceinline__ void scan_resident_normalise(const f32x4 (&zs)[16], f32x4 (&bq)[BF16 ? 8 * kBf16QueryTerms : 16]) {
    // no fused multiply-adds the packing kernel does not have: there the product z * inv is rounded before the first bf16 term is
    // taken off it (on the GPU the contracted form changed the second term's last bit in a few elements per batch)
#pragma clang fp contract(off)
    // element j of the row sits in (register r, component e) with j % 64 = 8 r' ... : the first 8 registers hold j < 64, register
    // r + 8 the partner j + 64 of register r (fp32: j = 8 c + 4 h + e; bf16: j = 16 s + 8 h + 4 (c & 1) + e, c = 2 s + (c & 1))
    f32x4 t[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) t[r][e] = fmaf(zs[r + 8][e], zs[r + 8][e], fmaf(zs[r][e], zs[r][e], 0.f));
    float ss;
    if (BF16) {
        // j % 64 = 16 (s & 3) + 8 h + 4 (c & 1) + e, register r = 2 (s & 3) + (c & 1): masks 32, 16 pair registers r ^ 4, r ^ 2;
        // mask 8 the lane halves; mask 4 registers r ^ 1; masks 2, 1 components
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] += t[r + 4];
#pragma unroll
        for (int r = 0; r < 2; ++r) t[r] += t[r + 2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) t[r][e] += shfl_xor(t[r][e], 32);
        t[0] += t[1];
    } else {
        // j % 64 = 8 c + 4 h + e, register r = c: masks 32, 16, 8 pair registers r ^ 4, r ^ 2, r ^ 1; mask 4 the lane halves
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] += t[r + 4];
#pragma unroll
        for (int r = 0; r < 2; ++r) t[r] += t[r + 2];
        t[0] += t[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) t[0][e] += shfl_xor(t[0][e], 32);
    }
    ss = (t[0][0] + t[0][2]) + (t[0][1] + t[0][3]);
    const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    if (BF16) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            // the first two terms of split_bf16x3, two elements per conversion: t0 = bf16(v), t1 = bf16(v - t0)
            static_assert(kBf16QueryTerms == 2, "the fused normalisation forms two bf16 terms per element");
            u32x4 w0, w1;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float v_lo = zs[2 * s + (d >> 1)][(2 * d) & 3] * inv, v_hi = zs[2 * s + (d >> 1)][(2 * d + 1) & 3] * inv;
                w0[d] = bf16_rn_pack2(v_lo, v_hi);
                w1[d] = bf16_rn_pack2(v_lo - __builtin_bit_cast(float, w0[d] << 16), v_hi - __builtin_bit_cast(float, w0[d] & 0xFFFF0000u));
            }
            bq[s] = __builtin_bit_cast(f32x4, w0);
            bq[8 + s] = __builtin_bit_cast(f32x4, w1);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) bq[c] = zs[c] * inv;
    }
}

// One block of the scan: row block `bx` of `gx`, query chunk `by` (the single-codebook kernel passes its block / grid indices; the grouped kernel below the
// position inside the object's own grid).
template <bool BF16, int K, int RH, bool NORM>
__device__ __forceinline__ void scan_resident_block(const ScanResidentArgs& p, const unsigned bx, const unsigned by, const unsigned gx) {
    constexpr int kTileRows = scan_resident_tile_rows<BF16, RH>();
    constexpr int kMi = kTileRows / (32 * RH);     // 32-row accumulator tiles per wave
    constexpr int kSlots = BF16 ? 16 : 32;         // 16-byte pieces per codebook row
    constexpr int kRowBytes = kSlots * 16;
    constexpr int kTileFloats = scan_resident_tile_floats<BF16, RH>();
    constexpr int kStages = scan_resident_stages<BF16, RH>();
    constexpr int kPieces = kTileFloats / (4 * 64 * 8);                    // LDS-DMA instructions per wave and tile (4 or 8)
    constexpr int QB = scan_resident_queries<RH>();
    static_assert(kMi >= 1 && RH * kMi * 32 == kTileRows, "row parts x accumulator tiles cover the row tile");
    AAE_DYN_SMEM(smem_raw);
    float* Et = reinterpret_cast<float*>(smem_raw);                        // [kStages][tile]
    float* red_v = Et + kStages * kTileFloats;                             // [RH row parts][QB queries]
    int* red_i = reinterpret_cast<int*>(red_v + 2 * 256);
    float* tau = red_v;                                                    // top-k: [QB] pruning bound per query of the block

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int rh = wave % RH, qg = wave / RH;
    const int q0 = by * QB + qg * 32;
    const bool active = q0 < p.Bpad && q0 < p.B;   // wave-uniform (a query group of padding only has nothing to do)
    const int query = q0 + i;

    // this wave's query fragments, for good
    f32x4 bq[BF16 ? 8 * kBf16QueryTerms : 16];
    [[maybe_unused]] f32x4 zraw[16];                      // (NORM only; dead otherwise)
    if (NORM) {
        if (active) scan_resident_load_raw<BF16>(p.z, query, query < p.B, h, zraw);
    } else if (active) {
        if (BF16) {
            const unsigned short* qp3 = reinterpret_cast<const unsigned short*>(p.qp);
            const long long qplane = (long long)16 * p.Bpad * 8;
#pragma unroll
            for (int term = 0; term < kBf16QueryTerms; ++term)
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    bq[term * 8 + s] = *reinterpret_cast<const f32x4*>(qp3 + term * qplane + ((long long)(2 * s + h) * p.Bpad + query) * 8);
        } else {
            const float* qp = reinterpret_cast<const float*>(p.qp);
#pragma unroll
            for (int c = 0; c < 16; ++c) bq[c] = *reinterpret_cast<const f32x4*>(qp + ((long long)(2 * c + h) * p.Bpad + query) * 4);
        }
    }

    const int ntiles = (p.N + kTileRows - 1) / kTileRows;
    const int tile0 = bx * p.tiles_per_block;
    const int tile1 = min(tile0 + p.tiles_per_block, ntiles);

    // staging by LDS-DMA: the codebook goes global -> LDS without passing through registers.  A tile is 2048 16-byte
    // pieces = 32 wave instructions of 64 lanes x 16 B (lane-linear in LDS), four per wave.  The XOR swizzle of the image is
    // applied on the SOURCE side: the lane that owns piece position row * kSlots + ps fetches logical slot ps ^ (row & 15)
    // of that row.  Rows past N lie beyond the buffer's size: the hardware delivers zeros.
    const buffer_rsrc ebuf = make_buffer(p.E, p.e_bytes);
    constexpr int kAblate = AAE_SCAN_RESIDENT_ABLATE;
    unsigned dma_off[kPieces];
#pragma unroll
    for (int j = 0; j < kPieces; ++j) {
        const int pos = (wave * kPieces + j) * 64 + lane;
        const int r = pos / kSlots, ps = pos % kSlots;
        dma_off[j] = (unsigned)(r * kRowBytes + ((ps ^ (r & 15)) << 4));
    }
    auto dma = [&](int t, float* img) {
        if (kAblate & 8) return;
        const unsigned tile_off = (unsigned)t * (unsigned)(kTileRows * kRowBytes);
#pragma unroll
        for (int j = 0; j < kPieces; ++j) lds_dma16(ebuf, t < tile1 ? tile_off + dma_off[j] : kOobOffset, img + (wave * kPieces + j) * 256);
    };

    // arg-max state: running (best score, first row) of this lane
    // arg-max state: running best of this lane, remembered as (score, tile, accumulator slot mi * 16 + r) -- the slot an
    // inline constant of the select -- and decoded to a row once at the end: a compare and two selects per value
    float bv = kNegInf;
    int bt = tile0, bs = 0;
    float tau_l = kNegInf;                         // top-k: this lane's pruning bound (a lower bound of its query's final K-th best)
    auto publish = [&](float best) {               // top-k: raise this block's word of the lane's query, in every replica
#pragma unroll
        for (int rep = 0; rep < kPruneReplicas; ++rep)
            shared_word_max(p.prune + ((long long)rep * p.Bpad + query) * kPruneGroups + (bx & (kPruneGroups - 1)), score_key(best));
    };
    constexpr int KL = K > 0 ? K : 1;
    float tv[KL];
    int ti[KL];
#pragma unroll
    for (int j = 0; j < KL; ++j) { tv[j] = kNegInf; ti[j] = 0x7fffffff; }

    // one tile of work for this wave: accumulate, then fold the accumulators into the running state
    auto consume = [&](const float* Eb, int t) {
        f32x16 acc[kMi];
#pragma unroll
        for (int mi = 0; mi < kMi; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
        if (BF16) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                u32x4 a[kMi];
#pragma unroll
                for (int mi = 0; mi < kMi; ++mi) {
                    if (kAblate & 2) a[mi] = u32x4{(unsigned)(lane + mi), (unsigned)s, 0x3f803f80u, (unsigned)t};
                    else a[mi] = __builtin_bit_cast(u32x4, lds_read4(Eb + e16_tile_off(rh * (32 * kMi) + mi * 32 + i, 2 * s + h)));
                }
#pragma unroll
                for (int term = kBf16QueryTerms - 1; term >= 0; --term)       // smallest term first; the accumulator tiles take
#pragma unroll                                                                  // turns, so no MFMA sits behind the one it depends on
                    for (int mi = 0; mi < kMi; ++mi) {
                        if (kAblate & 4) acc[mi][s] += (float)a[mi][term] * bq[term * 8 + s][mi];
                        else acc[mi] = mfma_32x32x16_bf16(a[mi], __builtin_bit_cast(u32x4, bq[term * 8 + s]), acc[mi]);
                    }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                f32x4 a[kMi];
#pragma unroll
                for (int mi = 0; mi < kMi; ++mi) a[mi] = lds_read4(Eb + e_tile_off(rh * (32 * kMi) + mi * 32 + i, 2 * c + h));
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int mi = 0; mi < kMi; ++mi) acc[mi] = mfma_32x32x2(a[mi][q], bq[c][q], acc[mi]);
            }
        }
        const int row_base = t * kTileRows + rh * 32 * kMi;
        const bool inside = (t + 1) * kTileRows <= p.N;
        if constexpr ((kAblate & 1) != 0) {
#pragma unroll
            for (int mi = 0; mi < kMi; ++mi)
                if (acc[mi][0] + acc[mi][7] > bv) { bv = acc[mi][0]; bs = mi; bt = t; }
            if (K > 0) tv[0] = bv;
        } else if constexpr (K == 0) {
            // running (max, first row): rows ascend with mi, r for a fixed lane, tiles ascend with t.  The row itself is not
            // formed per value: the winner is remembered as (tile, accumulator slot) -- compare + two selects per value,
            // the slot an inline constant -- and decoded once at the end.
            const float before = bv;
            if (inside) {
#pragma unroll
                for (int mi = 0; mi < kMi; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (acc[mi][r] > bv) { bv = acc[mi][r]; bs = mi * 16 + r; }
            } else {
#pragma unroll
                for (int mi = 0; mi < kMi; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (row_base + mi * 32 + acc_row(r, lane) < p.N && acc[mi][r] > bv) { bv = acc[mi][r]; bs = mi * 16 + r; }
            }
            bt = bv > before ? t : bt;
        } else {
            // Does any value of an accumulator tile (16 values per lane) enter any lane's list?  Its maximum against the lane's
            // K-th best and the query's bound decides for the wave; only the few accumulator tiles that pass are looked at value
            // by value, and there the K-slot insertion runs only for value slots in which some lane has a candidate.
#pragma unroll
            for (int mi = 0; mi < kMi; ++mi) {
                float m16 = acc[mi][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m16 = fmaxf(m16, acc[mi][r]);
                if (mi == 0 && t == tile0 && p.prune != nullptr && bx < 4 * kPruneGroups && !(kAblate & 64)) {
                    // the first word of the bound: the best of this wave's first 32 rows, from four blocks per word -- in place
                    // long before the end of step 0, where every block reads the words for the first time
                    const float best = fmaxf(m16, shfl_xor(m16, 32));
                    if (h == 0 && (inside || row_base + 32 <= p.N)) publish(best);
                }
                if (!wave_any(m16 > tv[K - 1] && m16 >= tau_l)) continue;
                if (kAblate & 32) { tv[0] = fmaxf(tv[0], m16); continue; }
#if defined(AAE_SCAN_COUNT) && AAE_SCAN_COUNT == 2
                if (lane == 0 && t - tile0 < 32) atomicAdd(p.dbg + (t - tile0) * 2, 1);
#endif
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_base + mi * 32 + acc_row(r, lane);
                    const float v = acc[mi][r];
                    const bool enters = (inside || row < p.N) && v > tv[K - 1] && v >= tau_l;
                    if (wave_any(enters)) {
                        if (kAblate & 16) { tv[0] = fmaxf(tv[0], v); continue; }
#if defined(AAE_SCAN_COUNT) && AAE_SCAN_COUNT == 2
                        if (lane == 0 && t - tile0 < 32) atomicAdd(p.dbg + (t - tile0) * 2 + 1, 1);
#endif
                        float cv = enters ? v : kNegInf;
                        int ci = row;
                        bool ins = false;
#pragma unroll
                        for (int j = 0; j < K; ++j) {          // from the first entry it beats on, everything moves down one slot
                            ins = ins || cv > tv[j];
                            const float ov = tv[j];
                            const int oi = ti[j];
                            tv[j] = ins ? cv : ov; ti[j] = ins ? ci : oi;
                            cv = ins ? ov : cv; ci = ins ? oi : ci;
                        }
                    }
                }
            }
        }
    };

    // The stream: tile t is used from LDS image t % 3 while the DMA pieces of tiles t + 1 and t + 2 are in flight (64 KB per
    // CU).  Tile t + 2 is requested right after the barrier of step t into the image tile t - 1 was read from -- every wave
    // has passed that barrier, so nobody reads it any more.  One barrier per tile; a wave waits only for its OWN pieces of
    // tile t (vmcnt counts in issue order: the four of tile t + 1 stay in flight), the barrier covers the other waves'.
    dma(tile0, Et);
    if (kStages == 3) dma(tile0 + 1, Et + kTileFloats);
    if constexpr (K == 0) {
        // in-launch finish: block 0 gives the ticket words this launch's nonce now, with its first tiles in flight -- the arrivals come
        // microseconds later and then all take the one-atomic path (round 4 measured the finish WITHOUT this: every arrival of a
        // stand-alone query met a foreign word and queued behind the install, 20.7 against 16.1 us)
        if (p.tickets != nullptr && bx == 0) ticket_prepare_slot(p.tickets, p.nonce, gx);
    }
    if constexpr (NORM) {
        if (active) scan_resident_normalise<BF16>(zraw, bq);
    }
    int img = 0;
    const bool prune = K > 0 && p.prune != nullptr;
    if (K > 0 && tid < QB) tau[tid] = kNegInf;                 // (in place at the first barrier)
    float published = kNegInf;
    // The schedule of the shared bound (measured at config 5, tools/ubench/scan_resident_ablate.hip; 148 us unpruned):
    // publications inside step 0, after a quarter of the first tile, by a quarter of the blocks (consume()), and after steps
    // 1, 2, 4, 7 by every block whose best can still move the bound; re-reads by the first QB threads at the END of steps 0, 1,
    // 2, 4, 7 (in place for everybody behind the next barrier): 96 us.  Dense early, when the bound moves.  What lost: a
    // publication by every block at the end of step 0 -- a burst of 60 000 atomics per replica right in front of the first
    // re-read of the same lines -- 115 us in any schedule that had it; only two re-reads (end of step 0 + a pipelined one in
    // step 2) 120 us: the early bound is weak (7th best of 2000 rows), it has to be followed up; re-reading every step 116 us.
    // Requesting the later re-reads at the start of their step (in front of the DMA pieces) and using them at its end, or
    // dropping the re-reads after step 2: 97-99 us, the same.  Publishing at the START of the next step, in front of its DMA pieces (so that the
    // next wait for pieces does not stand behind fresh atomics): the publishing step gets 6 us shorter, the two behind it 7 us
    // longer (the bound arrives a step later) -- 92.7 against 91.6 us.
    constexpr unsigned kPublishSteps = 0x96u, kRefreshSteps = 0x12fu;      // after steps {1, 2, 4, 7} / at the end of steps {0, 1, 2, 3, 5, 8}
    for (int t = tile0; t < tile1; ++t) {
        const float* Eb = Et + img * kTileFloats;
        wait_dma_keep_and_lds<(kStages - 2) * kPieces>();      // (two images: nothing else is in flight at this point)
        block_barrier();                                       // image of tile t complete
        [[maybe_unused]] const int step = t - tile0;
#ifdef AAE_SCAN_COUNT
        if (tid == 0 && bx == 100 && step < 16) reinterpret_cast<long long*>(p.dbg + 64)[step] = (long long)wall_ticks();
#endif
        [[maybe_unused]] u32x4 pw[kPruneGroups / 4];
        [[maybe_unused]] const bool reader = prune && tid < QB && by * QB + tid < p.Bpad && !(kAblate & 128);
        [[maybe_unused]] auto load_words = [&]() {
            const buffer_rsrc pb = make_buffer(p.prune, (unsigned)(kPruneReplicas * p.Bpad) * kPruneGroups * 4u);
            const int rq = (int)(bx & (kPruneReplicas - 1)) * p.Bpad + by * QB + tid;      // this block's replica
#pragma unroll
            for (int w4 = 0; w4 < kPruneGroups / 4; ++w4)      // (whole-vector casts: a bit_cast of one element indexed by a loop
                pw[w4] = __builtin_bit_cast(u32x4, coherent_load4(pb, (unsigned)(rq * kPruneGroups + w4 * 4) * 4u));   // variable picked element 0 under clang -O2)
        };
        dma(t + kStages - 1, Et + (img == 0 ? kStages - 1 : img - 1) * kTileFloats);      // into the image tile t - 1 was read from
        img = img == kStages - 1 ? 0 : img + 1;
        if constexpr (K > 0) {
            if (active) {
                tau_l = tau[qg * 32 + i];                      // (a bound of any age is valid)
                consume(Eb, t);
            }
            if (reader && step < 32 && ((kRefreshSteps >> step) & 1)) {
                // K-th largest of the query's shared words (K slots kept sorted by max / min exchanges): in place for everybody
                // behind the next barrier
                load_words();
                int top[K];
#pragma unroll
                for (int j = 0; j < K; ++j) top[j] = kScoreKeyEmpty;
#pragma unroll
                for (int w4 = 0; w4 < kPruneGroups / 4; ++w4)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int x = (int)pw[w4][e];
#pragma unroll
                        for (int j = 0; j < K; ++j) { const int hi = max(top[j], x); x = min(top[j], x); top[j] = hi; }
                    }
                tau[tid] = score_of_key(top[K - 1]);
            }
            if (active) {                                  // (published AFTER this block's own re-read: its loads do not queue behind its atomics)
                if (prune && step < 32 && ((kPublishSteps >> step) & 1) && !(kAblate & 64) &&
                    ((bx >> 6) & 3) == (step == 1 ? 1u : step == 2 ? 2u : step == 4 ? 3u : 0u)) {     // a quarter of the blocks per step
                    // one lane per query and wave, only what can move the bound (a score at or below it cannot become one of
                    // the K largest words), and a quarter of the blocks per step: a burst of device-scope atomics on a few thousand
                    // words right in front of the re-read of the same lines stalls it (every block publishing after step 1 made
                    // that step 30 us long instead of 8, in-kernel stamps; every lane of every block at five steps cost 60 us)
                    const float best = fmaxf(tv[0], shfl_xor(tv[0], 32));
                    if (h == 0 && best > published && best > tau_l) {
                        published = best;
                        publish(best);
                    }
                }
            }
        } else {
            if (active) consume(Eb, t);
        }
    }

#ifdef AAE_SCAN_COUNT
    if (tid == 0 && bx == 100) reinterpret_cast<long long*>(p.dbg + 64)[tile1 - tile0 < 16 ? tile1 - tile0 : 15] = (long long)wall_ticks();
#endif
    if constexpr (K > 0) {
        // ---- the sorted lists of a query (row part rh, lane half h) meet in LDS (the tile images are free) and
        // are merged by one thread per query: k rounds over the list heads, canonical order
        constexpr int NL = 2 * RH;
        wait_dma_and_lds();                                                   // (the zero pieces of the tiles past the end are still landing)
        __syncthreads();
        float* lv = Et;                                                       // [NL lists][QB queries][K]
        int* li = reinterpret_cast<int*>(Et + NL * QB * K);
        if (active) {
            const int at = ((rh * 2 + h) * QB + qg * 32 + i) * K;
#pragma unroll
            for (int j = 0; j < K; ++j) { lv[at + j] = tv[j]; li[at + j] = ti[j]; }
        }
        __syncthreads();
        const int qo = by * QB + tid;
        if (tid < QB && qo < p.B) {
            int head[NL];
#pragma unroll
            for (int l = 0; l < NL; ++l) head[l] = 0;
            const long long obase = ((long long)qo * gx + bx) * p.k;
            for (int j = 0; j < p.k; ++j) {
                float wv = kNegInf;
                int wi = 0x7fffffff, wl = 0;
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    const int at = (l * QB + tid) * K + head[l];
                    const float v = head[l] < K ? lv[at] : kNegInf;
                    const int ix = head[l] < K ? li[at] : 0x7fffffff;
                    if (better(v, ix, wv, wi)) { wv = v; wi = ix; wl = l; }
                }
#pragma unroll
                for (int l = 0; l < NL; ++l) head[l] += (l == wl && wi != 0x7fffffff) ? 1 : 0;
                p.cand_v[obase + j] = wv;
                p.cand_i[obase + j] = wi;
            }
        }
#ifdef AAE_SCAN_COUNT
        __syncthreads();
        if (tid == 0 && bx == 100) { reinterpret_cast<long long*>(p.dbg + 64)[13] = (long long)wall_ticks(); }
#endif
        return;
    }
    if (active) {
        int bi = bt * kTileRows + rh * 32 * kMi + (bs >> 4) * 32 + acc_row(bs & 15, lane);
        const float ov = shfl_xor(bv, 32);
        const int oi = shfl_xor(bi, 32);
        if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        if (h == 0) { red_v[rh * QB + qg * 32 + i] = bv; red_i[rh * QB + qg * 32 + i] = bi; }
    }
    __syncthreads();
    if constexpr (K == 0) {
        if (p.tickets != nullptr) {                            // (block-uniform)
            // ---- in-launch finish.  This block's partial row leaves as 16-byte device-coherent pieces (four queries each, scores in
            // pval, rows in pidx); the last block to arrive reads every block's pieces -- all of a thread's loads in flight at once --
            // and merges them with the tie rule of argmax_reduce_kernel (higher score, then lower row: a total order, so the grouping
            // of the merge does not matter).
            constexpr int Q4 = QB / 4;                         // pieces per partial row
            const int nblk = (int)gx;
            const buffer_rsrc vbuf = make_buffer(p.pval, (unsigned)nblk * (unsigned)p.Bstride * 4u);
            const buffer_rsrc ibuf = make_buffer(p.pidx, (unsigned)nblk * (unsigned)p.Bstride * 4u);
            if (tid < Q4) {
                f32x4 pv;
                u32x4 pi;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = 4 * tid + e;
                    float v = red_v[q];
                    int ix = red_i[q];
#pragma unroll
                    for (int part = 1; part < RH; ++part)
                        if (better(red_v[part * QB + q], red_i[part * QB + q], v, ix)) { v = red_v[part * QB + q]; ix = red_i[part * QB + q]; }
                    const bool real = q < p.B && (q >> 5) * 32 < p.B;            // (a query group of padding only wrote nothing into red_v)
                    pv[e] = real ? v : kNegInf;
                    pi[e] = real ? (uint32_t)ix : 0x7fffffffu;
                }
                const bool in_row = 4 * tid < p.Bstride;
                coherent_store4(vbuf, in_row ? (unsigned)(bx * p.Bstride + 4 * tid) * 4u : kOobOffset, pv);
                coherent_store4(ibuf, in_row ? (unsigned)(bx * p.Bstride + 4 * tid) * 4u : kOobOffset, __builtin_bit_cast(f32x4, pi));
            }
            int* flag = red_i + 2 * 256 - 4;                   // (the tail of the reduction area: red_i holds RH * QB = 256 entries)
            if (!block_ticket_arrive(p.tickets, p.nonce, (unsigned)nblk, bx, flag)) return;
            constexpr int PARTS = kScanResidentThreads / Q4;   // 32 (QB = 64) ... 8 (QB = 256)
            constexpr int KP = 8;                              // partial rows per thread and round: 256 row blocks in one round at QB = 64
            const int q4 = tid % Q4, part = tid / Q4;
            float bvq[4];
            int biq[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { bvq[e] = kNegInf; biq[e] = 0x7fffffff; }
            for (int k0 = part; k0 < nblk; k0 += PARTS * KP) {
                f32x4 tv[KP], ti[KP];
#pragma unroll
                for (int j = 0; j < KP; ++j) {
                    const int k = k0 + PARTS * j;
                    const unsigned at = (k < nblk && 4 * q4 < p.Bstride) ? (unsigned)(k * p.Bstride + 4 * q4) * 4u : kOobOffset;
                    tv[j] = coherent_load4(vbuf, at);
                    ti[j] = coherent_load4(ibuf, at);
                }
#pragma unroll
                for (int j = 0; j < KP; ++j) {
                    const bool live = k0 + PARTS * j < nblk && 4 * q4 < p.Bstride;
                    const u32x4 iw = __builtin_bit_cast(u32x4, ti[j]);
                    const uint32_t i0 = iw[0], i1 = iw[1], i2 = iw[2], i3 = iw[3];
                    const float fv[4] = {tv[j][0], tv[j][1], tv[j][2], tv[j][3]};
                    const int fi[4] = {(int)i0, (int)i1, (int)i2, (int)i3};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (live && better(fv[e], fi[e], bvq[e], biq[e])) { bvq[e] = fv[e]; biq[e] = fi[e]; }
                }
            }
            wait_dma_and_lds();                                // (the zero pieces of the tiles past the end may still be landing in Et)
            __syncthreads();
            float* fin_v = Et;                                 // [PARTS][QB]
            int* fin_i = reinterpret_cast<int*>(Et + PARTS * QB);
#pragma unroll
            for (int e = 0; e < 4; ++e) { fin_v[part * QB + 4 * q4 + e] = bvq[e]; fin_i[part * QB + 4 * q4 + e] = biq[e]; }
            __syncthreads();
            if (tid < QB && tid < p.B) {
                float v = fin_v[tid];
                int ix = fin_i[tid];
                for (int w = 1; w < PARTS; ++w)
                    if (better(fin_v[w * QB + tid], fin_i[w * QB + tid], v, ix)) { v = fin_v[w * QB + tid]; ix = fin_i[w * QB + tid]; }
                if (ix == 0x7fffffff) ix = 0;                  // all-NaN scores: np.argmax would also answer 0
                p.idx_out[tid] = (long long)ix * p.idx_scale;
                p.score_out[tid] = v;
            }
            return;
        }
    }
    const int qout = by * QB + tid;
    if (tid < QB && qout < p.B) {
        float v = red_v[tid];
        int ix = red_i[tid];
#pragma unroll
        for (int part = 1; part < RH; ++part)                  // row parts in ascending row order
            if (better(red_v[part * QB + tid], red_i[part * QB + tid], v, ix)) {
                v = red_v[part * QB + tid];
                ix = red_i[part * QB + tid];
            }
        p.pval[(long long)bx * p.Bstride + qout] = v;
        p.pidx[(long long)bx * p.Bstride + qout] = ix;
    }
}

template <bool BF16, int K = 0, int RH = 2, bool NORM = false>
__global__ __launch_bounds__(kScanResidentThreads) void scan_resident_kernel(const ScanResidentArgs p) {
    scan_resident_block<BF16, K, RH, NORM>(p, blockIdx.x, blockIdx.y, gridDim.x);
}

// ---- several codebooks in one launch (the arg-max form with the block normalising its own queries): object o owns blocks range.first[o] ... of the 1-D grid, laid out
//      as its own (row blocks x query chunks) grid; every block is exactly the single-codebook launch's block -- bit-identical partials.
struct ScanResidentMultiArgs {
    MultiRange range;
    int row_blocks[kMultiMax];             // gridDim.x of object o's own launch
    ScanResidentArgs item[kMultiMax];
};
static_assert(sizeof(ScanResidentMultiArgs) <= 4096, "kernel arguments: 4 KB");
template <int RH>
__global__ __launch_bounds__(kScanResidentThreads) void scan_resident_multi_kernel(const ScanResidentMultiArgs m) {
    const int o = multi_find(m.range, (int)blockIdx.x);
    const unsigned local = blockIdx.x - (unsigned)m.range.first[o], gx = (unsigned)m.row_blocks[o];
    scan_resident_block<false, 0, RH, true>(m.item[o], local % gx, local / gx, gx);
}

}  // namespace aae
