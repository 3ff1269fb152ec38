// Batched codebook arg-max with the QUERIES resident in registers and the codebook streamed
// through LDS -- the batched (B > 4) form of
//   tf.matmul(q, E, transpose_b=True) + argmax   (/root/reference/auto_pose/ae/codebook.py:50-51, 63-64)
// for fp32 rows (exact fp32 MFMA) and bf16 rows (three bf16 MFMA terms per product, fp32 accumulate).
//
// scan_mfma_kernel / scan_bf16_kernel keep one 128-row codebook tile in LDS and walk the query
// chunks through LDS as well: one block per CU (128 KB of LDS), every chunk pays a staging round
// trip, two barriers and an epilogue, and each MFMA needs more than one 16-byte LDS read.  Here a
// block of 8 waves owns 128 (or 256) queries for its whole life: a wave keeps the MFMA B fragments
// of its query group (32 queries) in registers (64 VGPRs fp32, 96 bf16) and accumulates its rows
// of every 32-KB row tile (64 fp32 rows / 128 bf16 rows) that streams through two LDS images:
// one barrier and one 16-byte LDS read per 4 (fp32) or 3 (bf16) MFMAs, the global loads of tiles
// t+1 and t+2 in flight under the MFMAs of tile t, a running (best score, first row) pair per lane, and
// one partial per (row range, query) at the very end.  The per-accumulator MFMA
// order is the one of the tile-resident kernels, so scores and indices are bit-identical to theirs.
//
// Restrictions (the tile-resident kernels remain for the rest): J == 128, top-1, col_stride == 1,
// no similarity output.
#pragma once

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
};

constexpr int kScanResidentThreads = 512;
constexpr int kScanResidentTileFloats = 8192;                               // 32 KB
constexpr int kScanResidentStages = 2;                                      // LDS images of the codebook stream
// queries one block owns: 8 waves = RH row parts x (8 / RH) groups of 32 queries
template <int RH>
constexpr int scan_resident_queries() { return 32 * (8 / RH); }
constexpr int kScanResidentSmem = kScanResidentStages * kScanResidentTileFloats * 4 + 2 * 2 * 256 * 4;

// K == 0: arg-max (one partial per row block and query).  K > 0: top-k for k <= K WITHOUT the [B][N] similarity matrix:
// every lane keeps the K best (score, row) pairs of the rows it sees, sorted; a new score enters in front of the first
// entry it beats strictly (rows arrive in ascending order per lane, so equal scores keep their row order); the lists that
// share a query (row parts x two lane halves) are merged at the end.  The insertion is ~5 VALU instructions per list slot
// and candidate, skipped when no lane of the wave has a score above its K-th best.
//
// RH = 2 (B <= 128): wave (rh, qg) accumulates half the rows of a tile for query group qg -- four query groups per block.
// RH = 1 (B > 128): every wave takes ALL rows of a tile for its own query group -- eight groups = 256 queries per block, so
// the codebook is streamed ONCE for 256 queries instead of once per 128 (config 5, B = 256: 189 -> 94 MB per scan; the kernel
// was bound by that stream at one 32-KB tile in flight per CU: 2.6 TB/s, profiles/r11_small).
// Round 3, all forms: (a) two tiles in flight behind the one in use (a second register set) instead of one; (b) top-k: a
// tile is looked at value by value only if its maximum beats some lane's K-th best.  Same MFMA order per accumulator:
// bit-identical scores and indices.  (A running maximum per accumulator position -- 3 instead of 5 vector instructions per
// value -- was tried for the arg-max: 128 more registers at four accumulator tiles, 300 spilled: dropped.)
template <bool BF16, int K = 0, int RH = 2>
__global__ __launch_bounds__(kScanResidentThreads) void scan_resident_kernel(const ScanResidentArgs p) {
    constexpr int kTileRows = BF16 ? 128 : 64;
    constexpr int kMi = (BF16 ? 4 : 2) / RH;       // 32-row accumulator tiles per wave
    constexpr int kSlots = BF16 ? 16 : 32;         // 16-byte pieces per codebook row
    constexpr int kRowBytes = kSlots * 16;
    constexpr int QB = scan_resident_queries<RH>();
    AAE_DYN_SMEM(smem_raw);
    float* Et = reinterpret_cast<float*>(smem_raw);                        // [kScanResidentStages][32 KB]
    float* red_v = Et + kScanResidentStages * kScanResidentTileFloats;     // [RH row parts][QB queries]
    int* red_i = reinterpret_cast<int*>(red_v + 2 * 256);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int rh = RH == 2 ? (wave & 1) : 0, qg = RH == 2 ? (wave >> 1) : wave;
    const int q0 = blockIdx.y * QB + qg * 32;
    const bool active = q0 < p.Bpad;               // wave-uniform
    const int query = q0 + i;

    // this wave's query fragments, for good
    f32x4 bq[BF16 ? 8 * kBf16QueryTerms : 16];
    if (active) {
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
    const int tile0 = blockIdx.x * p.tiles_per_block;
    const int tile1 = min(tile0 + p.tiles_per_block, ntiles);

    // staging: 2048 16-byte pieces per tile, 4 per thread, coalesced along the row; two register sets = two tiles in flight
    const buffer_rsrc ebuf = make_buffer(p.E, p.e_bytes);
    f32x4 st[2][4];
    auto fetch = [&](int t, int set) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + kScanResidentThreads * u;
            const int r = idx / kSlots, slot = idx % kSlots;
            const int row = t * kTileRows + r;
            st[set][u] = buffer_load4(ebuf, (t < tile1 && row < p.N) ? (unsigned)row * kRowBytes + slot * 16 : kOobOffset);
        }
    };
    auto put = [&](float* dst, int set) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = tid + kScanResidentThreads * u;
            const int r = idx / kSlots, slot = idx % kSlots;
            lds_write4(dst + (BF16 ? e16_tile_off(r, slot) : e_tile_off(r, slot)), st[set][u]);
        }
    };

    // arg-max state: running (best score, first row) of this lane
    float bv = kNegInf;
    int bi = tile0 * kTileRows + rh * 32 * kMi + acc_row(0, lane);
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
                for (int mi = 0; mi < kMi; ++mi)
                    a[mi] = __builtin_bit_cast(u32x4, lds_read4(Eb + e16_tile_off(rh * (32 * kMi) + mi * 32 + i, 2 * s + h)));
#pragma unroll
                for (int term = kBf16QueryTerms - 1; term >= 0; --term)       // smallest term first; the accumulator tiles take
#pragma unroll                                                                  // turns, so no MFMA sits behind the one it depends on
                    for (int mi = 0; mi < kMi; ++mi)
                        acc[mi] = mfma_32x32x16_bf16(a[mi], __builtin_bit_cast(u32x4, bq[term * 8 + s]), acc[mi]);
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
        if constexpr (K == 0) {
            // running (max, first row): rows ascend with mi, r for a fixed lane, tiles ascend with t
#pragma unroll
            for (int mi = 0; mi < kMi; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_base + mi * 32 + acc_row(r, lane);
                    const float v = acc[mi][r];
                    if ((inside || row < p.N) && v > bv) { bv = v; bi = row; }
                }
        } else {
            // does any value of this tile enter any lane's list?  (the tile maximum against the K-th best: two vector
            // instructions per three values; the value-by-value insertion below runs for the few tiles that pass)
            float tmax = kNegInf;
#pragma unroll
            for (int mi = 0; mi < kMi; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, acc[mi][r]);
            if (!wave_any(tmax > tv[K - 1])) return;
#pragma unroll
            for (int mi = 0; mi < kMi; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_base + mi * 32 + acc_row(r, lane);
                    const float v = acc[mi][r];
                    const bool enters = (inside || row < p.N) && v > tv[K - 1];
                    if (wave_any(enters)) {
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
    };

    // The stream: tile t is used from LDS image t % 2 while tile t + 1 waits in one register set and tile t + 2 is requested
    // into the other -- two tiles (64 KB per CU) in flight.  Tile t + 1 goes to its image at the END of step t: that image
    // was last read in step t - 1, and every wave has passed the barrier of step t since.  One barrier per tile.
    // (bf16 top-k lists at four accumulator tiles per wave leave no room for the second register set: one tile ahead there)
    constexpr bool kDeep = !(BF16 && RH == 1 && K > 0);
    fetch(tile0, 0);
    if (kDeep) fetch(tile0 + 1, 1);
    if (tile0 < tile1) put(Et, 0);
    for (int t = tile0; t < tile1; ++t) {
        const int step = t - tile0;
        float* Eb = Et + (step & 1) * kScanResidentTileFloats;
        float* En = Et + ((step + 1) & 1) * kScanResidentTileFloats;
        __syncthreads();                                       // image of tile t complete
        if (!kDeep) {
            fetch(t + 1, 0);
            if (active) consume(Eb, t);
            if (t + 1 < tile1) put(En, 0);
        } else if (step & 1) {
            fetch(t + 2, 1);
            if (active) consume(Eb, t);
            if (t + 1 < tile1) put(En, 0);
        } else {
            fetch(t + 2, 0);
            if (active) consume(Eb, t);
            if (t + 1 < tile1) put(En, 1);
        }
    }

    if constexpr (K > 0) {
        // ---- the sorted lists of a query (row part rh, lane half h) meet in LDS (the tile images are free) and
        // are merged by one thread per query: k rounds over the list heads, canonical order
        constexpr int NL = 2 * RH;
        __syncthreads();
        float* lv = Et;                                                       // [NL lists][QB queries][K]
        int* li = reinterpret_cast<int*>(Et + NL * QB * K);
        if (active) {
            const int at = ((rh * 2 + h) * QB + qg * 32 + i) * K;
#pragma unroll
            for (int j = 0; j < K; ++j) { lv[at + j] = tv[j]; li[at + j] = ti[j]; }
        }
        __syncthreads();
        const int qo = blockIdx.y * QB + tid;
        if (tid < QB && qo < p.B) {
            int head[NL];
#pragma unroll
            for (int l = 0; l < NL; ++l) head[l] = 0;
            const long long obase = ((long long)qo * gridDim.x + blockIdx.x) * p.k;
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
        return;
    }
    if (active) {
        const float ov = shfl_xor(bv, 32);
        const int oi = shfl_xor(bi, 32);
        if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        if (h == 0) { red_v[rh * QB + qg * 32 + i] = bv; red_i[rh * QB + qg * 32 + i] = bi; }
    }
    __syncthreads();
    const int qout = blockIdx.y * QB + tid;
    if (tid < QB && qout < p.B) {
        float v = red_v[tid];
        int ix = red_i[tid];
        if (RH == 2 && better(red_v[QB + tid], red_i[QB + tid], v, ix)) {
            v = red_v[QB + tid];
            ix = red_i[QB + tid];
        }
        p.pval[(long long)blockIdx.x * p.Bstride + qout] = v;
        p.pidx[(long long)blockIdx.x * p.Bstride + qout] = ix;
    }
}

}  // namespace aae
