// Codebook stage: L2-normalise + cosine similarity against the rotation codebook
// + arg-max / top-k, without ever materialising the [B, N] similarity matrix
// (the reference copies it to the host and arg-maxes in NumPy).
//
// Replaces (paths relative to /root/reference/auto_pose/ae/):
//   tf.nn.l2_normalize(z, 1) ......................... codebook.py:27
//   tf.matmul(q, embedding_normalized, transpose_b) .. codebook.py:50
//   np.argmax(cs, axis=1) / upright stride ........... codebook.py:64-68
//   argpartition + argsort top-n ..................... codebook.py:69-71
//
// Two scan kernels share one partial-result format ([ceil(N/128)] blocks x B
// queries of (best score, best row)):
//   scan_gemv : B <= 4.  Pure HBM stream: each half-wave reads one 512-B
//               codebook row per load instruction (float4 per lane), dots it
//               with the LDS/VGPR-resident queries, butterfly-reduces over the
//               32 lanes and keeps a running (max, first index).
//   scan_mfma : any B.  128 codebook rows x (32*NT) queries per pass on
//               v_mfma_f32_32x32x2_f32; the 128 x J codebook tile is staged
//               once in LDS (swizzled, coalesced 16-B loads) and reused for
//               every query tile; the arg-max runs on the accumulators.
// A final one-wave-per-query kernel reduces the block partials.  Ties always
// resolve to the LOWEST row index (np.argmax semantics) at every level.
#pragma once

namespace aae {

constexpr float kNegInf = -__builtin_huge_valf();

// ---------------------------------------------------------------- l2 normalise
// One wave per query.  q[b][:] = z[b][:] * (1/sqrt(max(sum z^2, 1e-12)));
// also emits the MFMA B-operand packing qp[J/4][Bpad][4] (zero rows for b >= B).
struct L2NormArgs {
    const float* z;   // [B][J]
    float* q;         // [B][J]   (may be nullptr)
    float* qp;        // [Jpad/4][Bpad][4] (may be nullptr)
    int B, J, Jpad, Bpad;
    int* prune = nullptr;   // optional [kPruneReplicas][Bpad][kPruneGroups]: reset for the top-k scan that follows (codebook_scan_resident.h)
};

__global__ __launch_bounds__(256) void l2norm_pack_kernel(const L2NormArgs p) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= p.Bpad) return;                      // whole wave exits together
    if (p.prune)
        for (int w = lane; w < kPruneReplicas * kPruneGroups; w += 64) p.prune[((w / kPruneGroups) * p.Bpad + b) * kPruneGroups + w % kPruneGroups] = kScoreKeyEmpty;
    const bool real = b < p.B;
    float ss = 0.f;
    if (real)
        for (int j = lane; j < p.J; j += 64) { const float v = p.z[(long long)b * p.J + j]; ss = fmaf(v, v, ss); }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) ss += shfl_xor(ss, m);
    const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    for (int j = lane; j < p.Jpad; j += 64) {
        const float v = (real && j < p.J) ? p.z[(long long)b * p.J + j] * inv : 0.f;
        if (p.q && real && j < p.J) p.q[(long long)b * p.J + j] = v;
        if (p.qp) p.qp[((long long)(j >> 2) * p.Bpad + b) * 4 + (j & 3)] = v;
    }
}

// ------------------------------------------------------------------- scan args
struct ScanArgs {
    const float* E;     // [N][J] fp32, row-major
    const float* q;     // [B][J]      normalised queries        (gemv)
    const float* qp;    // [Jpad/4][Bpad][4] packed queries      (mfma)
    float* pval;        // [nblk][Bstride] block-partial best score
    int* pidx;          // [nblk][Bstride] block-partial best row
    float* cs;          // optional full [B][N] similarity (parity / top-k path)
    int N, J, Jpad, B, Bpad, Bstride;
    int col_stride;     // 1, or num_cyclo for the reference's `upright` mode
    const float* z;     // [B][J] raw latents                    (stream: normalisation fused)
    unsigned e_bytes;   // N*J*4                                 (stream: bounds-checked view of E)
    // stream kernels, top-1: when tickets != nullptr the last block to arrive merges the block partials itself
    // (scan_ticket_finish) and writes the answers -- the whole nearest-neighbour query is ONE launch
    unsigned long long* tickets = nullptr;   // kTicketSlotWords words
    unsigned nonce = 0;
    long long* idx_out = nullptr;   // [B] int64
    float* score_out = nullptr;     // [B]
    int idx_scale = 1;              // row ids are multiplied by this (upright search on the compacted copy)
};

// block-wide (value, index) arg-best in np.argmax order; result broadcast to every thread.  red: 10 dwords of LDS.
__device__ __forceinline__ void block_best(float& bv, int& bi, float* red) {
    int* red_i = reinterpret_cast<int*>(red + 4);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = shfl_xor(bv, m);
        const int oi = shfl_xor(bi, m);
        if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { red[wave] = bv; red_i[wave] = bi; }
    __syncthreads();
    bv = red[0]; bi = red_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (better(red[w], red_i[w], bv, bi)) { bv = red[w]; bi = red_i[w]; }
    __syncthreads();
}

// Block partials of a stream scan: thread 0 folds the four waves' (best score, best row) of every query and stores them.
// In-launch finish (p.tickets): all queries of the block as 16-byte device-coherent pieces (score, row, score, row) at the
// front of the block's row of p.pval -- one or two stores per block and, for the finishing block, one or two loads per
// partial row instead of 2 NQ dword accesses each (round 3's form: at B = 4 the finish cost 6 us against 3 at B = 1,
// tools/ubench/scan_stream_ablate.hip).  Separate reduce launch (no tickets): the [block][query] score and row arrays
// argmax_reduce_kernel reads.
template <int NQ>
__device__ __forceinline__ void scan_store_block_partials(const ScanArgs& p, const float* red_v, const int* red_i, const unsigned blk, const unsigned nblk) {
    if (threadIdx.x != 0) return;
    float v[NQ];
    int ix[NQ];
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
        v[b] = red_v[b];
        ix[b] = red_i[b];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (better(red_v[w * NQ + b], red_i[w * NQ + b], v[b], ix[b])) { v[b] = red_v[w * NQ + b]; ix[b] = red_i[w * NQ + b]; }
    }
    if (p.tickets) {
        const buffer_rsrc pbuf = make_buffer(p.pval, nblk * (unsigned)p.Bstride * 4u);
#pragma unroll
        for (int h = 0; h < (NQ + 1) / 2; ++h) {
            u32x4 piece;
            piece[0] = __builtin_bit_cast(uint32_t, v[2 * h]);
            piece[1] = (uint32_t)ix[2 * h];
            piece[2] = 2 * h + 1 < NQ ? __builtin_bit_cast(uint32_t, v[(2 * h + 1) % NQ]) : 0u;
            piece[3] = 2 * h + 1 < NQ ? (uint32_t)ix[(2 * h + 1) % NQ] : 0u;
            coherent_store4(pbuf, (blk * (unsigned)p.Bstride + 4u * h) * 4u, __builtin_bit_cast(f32x4, piece));
        }
    } else {
#pragma unroll
        for (int b = 0; b < NQ; ++b)
            if (b < p.B) {
                p.pval[(long long)blk * p.Bstride + b] = v[b];
                p.pidx[(long long)blk * p.Bstride + b] = ix[b];
            }
    }
}

// Called by every thread of a stream-scan block after its block partial is written.  The last of the gridDim.x
// blocks merges all partials (the arg-max reduce of argmax_reduce_kernel: same order, lowest row wins ties) and
// writes (index, score) per query.  red: >= 12 + 8 NQ dwords of LDS.
// Every partial of every query is requested before the first one is looked at: a device-coherent load is a round trip to
// the memory side (~0.7 us), and one query after the other (round 3) the finish of a B = 4 scan cost four of them plus
// eight block barriers.
constexpr int kScanTicketSmem = 64 + 8 * 4 * 4;
template <int NQ>
__device__ __forceinline__ void scan_ticket_finish(const ScanArgs& p, float* red, const unsigned blk, const unsigned nblk_u) {
    int* flag = reinterpret_cast<int*>(red) + 10;
    if (!block_ticket_arrive(p.tickets, p.nonce, nblk_u, blk, flag)) return;
    float* fin_v = red + 16;                                     // [4][NQ]
    int* fin_i = reinterpret_cast<int*>(fin_v + 4 * NQ);         // [4][NQ]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nblk = (int)nblk_u;
    const buffer_rsrc pbuf = make_buffer(p.pval, (unsigned)nblk * p.Bstride * 4u);
    float bv[NQ];
    int bi[NQ];
#pragma unroll
    for (int b = 0; b < NQ; ++b) { bv[b] = kNegInf; bi[b] = 0x7fffffff; }
    constexpr int KP = 3;                                        // partial rows per thread and round: 768 blocks in one round
    constexpr int PIECES = (NQ + 1) / 2;                         // 16-byte pieces (score, row, score, row) per partial row
    for (int k0 = tid; k0 < nblk; k0 += 256 * KP) {
        f32x4 t[KP][PIECES];
#pragma unroll
        for (int j = 0; j < KP; ++j)
#pragma unroll
            for (int h = 0; h < PIECES; ++h) {
                const int k = k0 + 256 * j;
                t[j][h] = coherent_load4(pbuf, k < nblk ? (unsigned)(k * p.Bstride + 4 * h) * 4u : kOobOffset);
            }
#pragma unroll
        for (int j = 0; j < KP; ++j)
#pragma unroll
            for (int b = 0; b < NQ; ++b) {
                const bool live = k0 + 256 * j < nblk;
                const u32x4 piece = __builtin_bit_cast(u32x4, t[j][b / 2]);      // (whole vector, then scalars: see device_intrinsics.h)
                const uint32_t w0 = piece[0], w1 = piece[1], w2 = piece[2], w3 = piece[3];
                const float fv = live ? __builtin_bit_cast(float, (b & 1) ? w2 : w0) : kNegInf;
                const int fi = live ? (int)((b & 1) ? w3 : w1) : 0x7fffffff;
                if (better(fv, fi, bv[b], bi[b])) { bv[b] = fv; bi[b] = fi; }
            }
    }
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ov = shfl_xor(bv[b], m);
            const int oi = shfl_xor(bi[b], m);
            if (better(ov, oi, bv[b], bi[b])) { bv[b] = ov; bi[b] = oi; }
        }
        if (lane == 0) { fin_v[wave * NQ + b] = bv[b]; fin_i[wave * NQ + b] = bi[b]; }
    }
    __syncthreads();
    if (tid < NQ && tid < p.B) {
        float v = fin_v[tid];
        int ix = fin_i[tid];
        for (int w = 1; w < 4; ++w)
            if (better(fin_v[w * NQ + tid], fin_i[w * NQ + tid], v, ix)) { v = fin_v[w * NQ + tid]; ix = fin_i[w * NQ + tid]; }
        if (ix == 0x7fffffff) ix = 0;              // all-NaN scores: np.argmax would also answer 0
        p.idx_out[tid] = (long long)ix * p.idx_scale;
        p.score_out[tid] = v;
    }
}

// ----------------------------------------------------------------- scan_stream
// B <= 4, the reference's real usage (one crop per detection).  One launch does the
// L2-normalisation AND the scan; the codebook is the only HBM stream:
//   * the <= 4 query rows are requested FIRST (16 B per lane), then every wave issues all 16 of its 1-KiB codebook
//     loads -- 32 consecutive rows, one 512-B row per half-wave and load: half rs reads rows 16 rs ... 16 rs + 15 --
//     before it touches any of them, so the whole 47 MB codebook is in flight at once across the chip; vmcnt counts in
//     order, so the queries are normalised (replicated in every lane) while the rows still fly;
//   * loads go through a bounds-checked buffer view (rows >= N read zeros): no branches, no early vmcnt waits;
//   * per query: 16 x 4 fmas give the lane's share of its half-wave's 16 rows, half_wave_reduce_scatter16 sums them over
//     the half-wave in 38 cross-lane instructions and leaves row (l >> 1) of the wave's 32 in lane l -- rows in lane
//     order -- and wave_max_first_lane finds (max, first row) with one compare and a scalar bit search.  (Round 3 finished
//     every (row, query) dot product with its own five-step DPP tree and kept a running best per lane: 80 cross-lane adds
//     and 48 compare/select steps per query -- and, with the optional similarity store as a branch inside that loop, 64
//     basic blocks whose DPP chains the compiler could not interleave: B = 4 took 23.5 us for the 47 MB that B = 1 streams
//     in 13.3.)  WITH_CS: the similarity row is wanted as well (parity tests, the B <= 4 top-k path): the even lanes store
//     their rows, 128 B contiguous per half-wave.
// The same three helpers serve the scan phase of the persistent per-detection launch (detect_chain.h): same bits per row.
template <int NQ>
__device__ __forceinline__ void scan_normalise_queries(const f32x4 (&zv)[NQ], f32x4 (&qv)[NQ]) {
    // tf.nn.l2_normalize(z, 1) (codebook.py:27), per query, replicated in every lane
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
        float ss = zv[b].x * zv[b].x;
        ss = fmaf(zv[b].y, zv[b].y, ss);
        ss = fmaf(zv[b].z, zv[b].z, ss);
        ss = fmaf(zv[b].w, zv[b].w, ss);
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) ss += shfl_xor(ss, m);
        const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        qv[b] = zv[b] * inv;
    }
}
// rows [row0, row0 + 32) of this wave, clipped at row_end: lane (rs = l >> 5, kq = l & 31) requests columns 4 kq ... + 3 of rows row0 + 16 rs + u
__device__ __forceinline__ void scan_issue32(const ScanArgs& p, const buffer_rsrc& ebuf, int row0, int row_end, f32x4 (&e)[16]) {
    const int lane = threadIdx.x & 63, rs = lane >> 5, col = (lane & 31) * 4;
    const bool col_ok = col < p.J;                               // J <= 128, J % 4 == 0
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int r = row0 + 16 * rs + u;
        e[u] = buffer_load4(ebuf, (r < row_end && col_ok) ? (unsigned)(r * p.J + col) * 4u : kOobOffset);
    }
}
// cosine of the wave's 32 rows with one query: lane l gets row row0 + (l >> 1)
__device__ __forceinline__ float scan_scores32(const f32x4 (&e)[16], const f32x4& q) {
    float d[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        float t = e[u].x * q.x;
        t = fmaf(e[u].y, q.y, t);
        t = fmaf(e[u].z, q.z, t);
        d[u] = fmaf(e[u].w, q.w, t);
    }
    return half_wave_reduce_scatter16(d);
}

// Arg-max of a block: every wave leaves the scores of its rows for every query in LDS ([NQ][ROWS], -inf where a row is no
// candidate); after ONE barrier wave b finds (max, first row) of query b over the whole block -- one key maximum per query
// and BLOCK instead of one per query and wave (at B = 4 the four per-wave searches cost as much as the whole reduce-scatter,
// tools/ubench/scan_stream_ablate.hip) -- and its lane 0 stores the block partial: 8 device-coherent bytes (score, row) into
// the query's slot of the packed partial row (in-launch finish) or the [block][query] arrays (separate reduce launch).
template <int NQ, int ROWS>
__device__ __forceinline__ void scan_block_argmax_store(const ScanArgs& p, const float* sc, int row_base, const unsigned blk, const unsigned nblk) {
    const int lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6);
    if (wave >= NQ || wave >= p.B) return;                       // wave-uniform
    float v[ROWS / 64];
#pragma unroll
    for (int j = 0; j < ROWS / 64; ++j) v[j] = sc[wave * ROWS + 64 * j + lane];
    int first;
    const float m = wave_max_first_position<ROWS / 64>(v, first);
    if (lane != 0) return;
    const int ix = m > kNegInf ? row_base + first : 0x7fffffff;
    if (p.tickets)
        coherent_store2(make_buffer(p.pval, nblk * (unsigned)p.Bstride * 4u), (blk * (unsigned)p.Bstride + 2u * wave) * 4u,
                        __builtin_bit_cast(uint32_t, m), (uint32_t)ix);
    else {
        p.pval[(long long)blk * p.Bstride + wave] = m;
        p.pidx[(long long)blk * p.Bstride + wave] = ix;
    }
}

// block `blk` of the `nblk` blocks that scan this codebook (sm: NQ * 128 floats of scores + kScanTicketSmem bytes)
template <int NQ, bool UPRIGHT, bool WITH_CS>
__device__ __forceinline__ void scan_stream_block(const ScanArgs& p, const int blk, const int nblk, unsigned char* smem_raw) {
    float* sc = reinterpret_cast<float*>(smem_raw);              // [NQ][128] scores of the block's rows

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = (lane & 31) * 4;
    const bool col_ok = col < p.J;
    const int row_first = blk * 128 + wave * 32;

    f32x4 zv[NQ];
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
        zv[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (b < p.B && col_ok) zv[b] = *reinterpret_cast<const f32x4*>(p.z + (long long)b * p.J + col);
    }
    const buffer_rsrc ebuf = make_buffer(p.E, p.e_bytes);
    f32x4 e[16];
    scan_issue32(p, ebuf, row_first, p.N, e);

    if (p.tickets && blk == 0) ticket_prepare_slot(p.tickets, p.nonce, (unsigned)nblk);   // the block's loads are in flight; arrivals come microseconds later

    f32x4 qv[NQ];
    scan_normalise_queries<NQ>(zv, qv);

    const int row = row_first + (lane >> 1);
    bool cand = row < p.N;
    if (UPRIGHT) cand = cand && (row % p.col_stride == 0);
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
        const float d = scan_scores32(e, qv[b]);
        if (WITH_CS) {
            if (!(lane & 1) && row < p.N && b < p.B) p.cs[(long long)b * p.N + row] = d;
        }
        if (!(lane & 1)) sc[b * 128 + wave * 32 + (lane >> 1)] = cand ? d : kNegInf;
    }
    __syncthreads();
    scan_block_argmax_store<NQ, 128>(p, sc, blk * 128, (unsigned)blk, (unsigned)nblk);
    if (p.tickets) scan_ticket_finish<NQ>(p, sc + NQ * 128, (unsigned)blk, (unsigned)nblk);
}

template <int NQ, bool UPRIGHT, bool WITH_CS>
__global__ __launch_bounds__(256) void scan_stream_kernel(const ScanArgs p) {
    AAE_DYN_SMEM(smem_raw);
    scan_stream_block<NQ, UPRIGHT, WITH_CS>(p, (int)blockIdx.x, (int)gridDim.x, smem_raw);
}

// ---- SEVERAL codebooks in one launch (multi_launch.h): blocks [first[o], first[o + 1]) stream codebook o against object o's
// (<= 4) latent codes and answer through object o's ticket -- C codebooks of 47 MB behind ONE launch ramp.  Top-1, stride 1 (the
// upright search runs on its compacted copy), answers inside the launch.
struct ScanMultiArgs {
    MultiRange range;
    ScanArgs item[kMultiMax];
};
template <int NQ>
__global__ __launch_bounds__(256) void scan_stream_multi_kernel(const ScanMultiArgs m) {
    AAE_DYN_SMEM(smem_raw);
    const int o = multi_find(m.range, (int)blockIdx.x);
    scan_stream_block<NQ, false, false>(m.item[o], (int)blockIdx.x - m.range.first[o], m.range.first[o + 1] - m.range.first[o], smem_raw);
}

// ------------------------------------------------------------ scan_stream_walk
// The same query with compute UNDER the stream (round 4).  In scan_stream_kernel every wave owns exactly one 32-row batch
// and the whole codebook is requested at t = 0: the memory system then serves 2884 waves round-robin, nearly every wave
// receives its last row near the END of the 7.5 us stream, and all the arithmetic -- per wave ~130 vector instructions per
// query -- runs after it, the waves of a SIMD one behind the other: B = 4 took 20.8 us for the bytes B = 1 streams in 12.8
// although the instructions were already halved.  Here a block per CU walks the codebook: wave gw of nw takes batches gw,
// gw + nw, gw + 2 nw, ... with TWO batches in flight (two 16-load register rings), scores batch i while batch i + 1 lands
// and requests batch i + 2 into the ring it just freed; the running (max, first row) of a wave lives in scalar registers
// (batches ascend: a later row wins only when strictly larger).  A third of the arithmetic is left behind the last byte
// instead of all of it, the launch has 256 blocks instead of 721, and the in-launch finish merges 256 partials.
// Same bits per row as scan_stream_kernel (scan_scores32), same tie rule.
template <int NQ, bool UPRIGHT, bool WITH_CS>
__device__ __forceinline__ void scan_walk_consume(const ScanArgs& p, int row0, const f32x4 (&e)[16], const f32x4 (&qv)[NQ], float (&best_v)[NQ], int (&best_i)[NQ]) {
    const int lane = threadIdx.x & 63;
    const int row = row0 + (lane >> 1);
    bool cand = row < p.N;
    if (UPRIGHT) cand = cand && (row % p.col_stride == 0);
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
        const float d = scan_scores32(e, qv[b]);
        if (WITH_CS) {
            if (!(lane & 1) && row < p.N && b < p.B) p.cs[(long long)b * p.N + row] = d;
        }
        int first;
        const float m = wave_max_first_lane(cand ? d : kNegInf, first);
        if (m > best_v[b]) { best_v[b] = m; best_i[b] = row0 + (first >> 1); }
    }
}

template <int NQ, bool UPRIGHT, bool WITH_CS>
__global__ __launch_bounds__(256) void scan_stream_walk_kernel(const ScanArgs p) {
    AAE_DYN_SMEM(smem_raw);
    float* red_v = reinterpret_cast<float*>(smem_raw);           // [4][NQ]
    int* red_i = reinterpret_cast<int*>(red_v + 4 * NQ);         // [4][NQ]

    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int col = (lane & 31) * 4;
    const bool col_ok = col < p.J;
    const int nbatch = (p.N + 31) >> 5;
    const int gw = (int)blockIdx.x * 4 + wave, nw = (int)gridDim.x * 4;

    f32x4 zv[NQ];
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
        zv[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (b < p.B && col_ok) zv[b] = *reinterpret_cast<const f32x4*>(p.z + (long long)b * p.J + col);
    }
    const buffer_rsrc ebuf = make_buffer(p.E, p.e_bytes);
    f32x4 ea[16], eb[16];
    // (a batch index beyond the codebook turns into rows >= N: every load out of range, no traffic)
    scan_issue32(p, ebuf, gw < nbatch ? gw * 32 : p.N, p.N, ea);
    scan_issue32(p, ebuf, gw + nw < nbatch ? (gw + nw) * 32 : p.N, p.N, eb);

    if (p.tickets && blockIdx.x == 0) ticket_prepare_slot(p.tickets, p.nonce, gridDim.x);

    f32x4 qv[NQ];
    scan_normalise_queries<NQ>(zv, qv);

    float best_v[NQ];
    int best_i[NQ];
#pragma unroll
    for (int b = 0; b < NQ; ++b) { best_v[b] = kNegInf; best_i[b] = 0x7fffffff; }
    for (int i = gw; i < nbatch; i += 2 * nw) {                   // wave-uniform trip count
        scan_walk_consume<NQ, UPRIGHT, WITH_CS>(p, i * 32, ea, qv, best_v, best_i);
        scan_issue32(p, ebuf, i + 2 * nw < nbatch ? (i + 2 * nw) * 32 : p.N, p.N, ea);
        if (i + nw < nbatch) scan_walk_consume<NQ, UPRIGHT, WITH_CS>(p, (i + nw) * 32, eb, qv, best_v, best_i);
        scan_issue32(p, ebuf, i + 3 * nw < nbatch ? (i + 3 * nw) * 32 : p.N, p.N, eb);
    }
    if (lane == 0) {
#pragma unroll
        for (int b = 0; b < NQ; ++b) { red_v[wave * NQ + b] = best_v[b]; red_i[wave * NQ + b] = best_i[b]; }
    }
    __syncthreads();
    scan_store_block_partials<NQ>(p, red_v, red_i, blockIdx.x, gridDim.x);
    if (p.tickets) scan_ticket_finish<NQ>(p, red_v + 8 * NQ, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------- scan_gemv
template <int NQ, bool UPRIGHT>
__global__ __launch_bounds__(256) void scan_gemv_kernel(const ScanArgs p) {
    AAE_DYN_SMEM(smem_raw);
    float* red_v = reinterpret_cast<float*>(smem_raw);           // [4][NQ]
    int* red_i = reinterpret_cast<int*>(red_v + 4 * NQ);         // [4][NQ]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rs = lane >> 5, kq = lane & 31;
    const int col = kq * 4;
    const bool col_ok = col < p.J;                               // J <= 128, J % 4 == 0

    f32x4 qv[NQ];
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
        qv[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (b < p.B && col_ok) qv[b] = *reinterpret_cast<const f32x4*>(p.q + (long long)b * p.J + col);
    }
    float best_v[NQ];
    int best_i[NQ];
    const int row_first = blockIdx.x * 128 + wave * 32;
#pragma unroll
    for (int b = 0; b < NQ; ++b) { best_v[b] = kNegInf; best_i[b] = row_first; }

#pragma unroll
    for (int it0 = 0; it0 < 16; it0 += 8) {
        f32x4 e[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = row_first + (it0 + u) * 2 + rs;
            e[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < p.N && col_ok) e[u] = *reinterpret_cast<const f32x4*>(p.E + (long long)row * p.J + col);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = row_first + (it0 + u) * 2 + rs;
            bool cand = row < p.N;
            if (UPRIGHT) cand = cand && (row % p.col_stride == 0);
#pragma unroll
            for (int b = 0; b < NQ; ++b) {
                float d = e[u].x * qv[b].x;
                d = fmaf(e[u].y, qv[b].y, d);
                d = fmaf(e[u].z, qv[b].z, d);
                d = fmaf(e[u].w, qv[b].w, d);
#pragma unroll
                for (int m = 16; m >= 1; m >>= 1) d += shfl_xor(d, m);
                if (p.cs && kq == 0 && row < p.N && b < p.B) p.cs[(long long)b * p.N + row] = d;
                if (cand && d > best_v[b]) { best_v[b] = d; best_i[b] = row; }
            }
        }
    }
    // even rows (rs=0) vs odd rows (rs=1), then the 4 waves
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
        const float ov = shfl_xor(best_v[b], 32);
        const int oi = shfl_xor(best_i[b], 32);
        if (better(ov, oi, best_v[b], best_i[b])) { best_v[b] = ov; best_i[b] = oi; }
        if (lane == 0) { red_v[wave * NQ + b] = best_v[b]; red_i[wave * NQ + b] = best_i[b]; }
    }
    __syncthreads();
    if (tid < NQ && tid < p.B) {
        float v = red_v[tid];
        int ix = red_i[tid];
        for (int w = 1; w < 4; ++w)
            if (better(red_v[w * NQ + tid], red_i[w * NQ + tid], v, ix)) { v = red_v[w * NQ + tid]; ix = red_i[w * NQ + tid]; }
        p.pval[(long long)blockIdx.x * p.Bstride + tid] = v;
        p.pidx[(long long)blockIdx.x * p.Bstride + tid] = ix;
    }
}

// ------------------------------------------------------------------- scan_mfma
// LDS: E tile [128 rows][128 k] (row stride 128 floats, 16-B slot XOR (row&15))
//      q tile [32 slots][32*NT cols][4]
__device__ __forceinline__ int e_tile_off(int row, int slot) { return row * 128 + ((slot ^ (row & 15)) << 2); }

template <int NT>
constexpr int scan_mfma_smem() { return (128 * 128 + 32 * 32 * NT * 4 + 2 * 4 * 32 * NT) * 4; }

template <int NT, bool UPRIGHT>
__global__ __launch_bounds__(256) void scan_mfma_kernel(const ScanArgs p) {
    constexpr int QC = 32 * NT;                                   // queries per pass
    AAE_DYN_SMEM(smem_raw);
    float* Et = reinterpret_cast<float*>(smem_raw);               // [128][128]
    float* Qt = Et + 128 * 128;                                   // [32][QC][4]
    float* red_v = Qt + 32 * QC * 4;                              // [4][QC]
    int* red_i = reinterpret_cast<int*>(red_v + 4 * QC);          // [4][QC]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * 128;

    // stage the codebook tile once (J <= 128): 128 rows x 32 slots, 16 float4 per
    // thread, each half-wave reads one contiguous 512-B row
    {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = tid + 256 * u;
            const int r = idx >> 5, slot = idx & 31;
            const int row = row0 + r, c = slot * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (row < p.N && c < p.J) v = *reinterpret_cast<const f32x4*>(p.E + (long long)row * p.J + c);
            lds_write4(Et + e_tile_off(r, slot), v);
        }
        for (int qt = 0; qt < p.Bpad; qt += QC) {
            __syncthreads();                                       // Qt / red free again
            for (int idx = tid; idx < 32 * QC; idx += 256) {
                const int slot = idx / QC, c = idx - slot * QC;
                const f32x4 v = *reinterpret_cast<const f32x4*>(p.qp + ((long long)slot * p.Bpad + qt + c) * 4);
                lds_write4(Qt + idx * 4, v);
            }
            __syncthreads();

            f32x16 acc[NT];
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
#pragma unroll 4
            for (int c = 0; c < 16; ++c) {
                const int slot = 2 * c + h;
                const f32x4 a = lds_read4(Et + e_tile_off(wave * 32 + i, slot));
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) {
                    const f32x4 b = lds_read4(Qt + (slot * QC + ni * 32 + i) * 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[ni] = mfma_32x32x2(a[q], b[q], acc[ni]);
                }
            }

            // running (max, first row) per query column; rows ascend with r for fixed h
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                const int query = qt + ni * 32 + i;
                float bv = kNegInf;
                int bi = row0 + wave * 32 + acc_row(0, lane);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + wave * 32 + acc_row(r, lane);
                    const float v = acc[ni][r];
                    bool cand = row < p.N;
                    if (UPRIGHT) cand = cand && (row % p.col_stride == 0);
                    if (p.cs && row < p.N && query < p.B) p.cs[(long long)query * p.N + row] = v;
                    if (cand && v > bv) { bv = v; bi = row; }
                }
                const float ov = shfl_xor(bv, 32);
                const int oi = shfl_xor(bi, 32);
                if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
                if (h == 0) { red_v[wave * QC + ni * 32 + i] = bv; red_i[wave * QC + ni * 32 + i] = bi; }
            }
            __syncthreads();
            if (tid < QC && qt + tid < p.B) {
                float v = red_v[tid];
                int ix = red_i[tid];
                for (int w = 1; w < 4; ++w)
                    if (better(red_v[w * QC + tid], red_i[w * QC + tid], v, ix)) { v = red_v[w * QC + tid]; ix = red_i[w * QC + tid]; }
                p.pval[(long long)blockIdx.x * p.Bstride + qt + tid] = v;
                p.pidx[(long long)blockIdx.x * p.Bstride + qt + tid] = ix;
            }
        }
    }
}

// ------------------------------------------------- final reduction over blocks
// dst[r][:] = src[r * stride][:], rows of row_bytes (a multiple of 16): the every-stride-th-row copy the upright
// search (codebook.py:65-66: arg-max over columns 0, k, 2k, ...) scans instead of masking (k-1)/k of a full scan.
struct GatherRowsArgs {
    const void* src;
    void* dst;
    int rows_out, stride, pieces_per_row;      // 16-byte pieces
};
__global__ __launch_bounds__(256) void gather_rows_kernel(const GatherRowsArgs p) {
    const long long total = (long long)p.rows_out * p.pieces_per_row;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long r = idx / p.pieces_per_row, c = idx - r * p.pieces_per_row;
        reinterpret_cast<f32x4*>(p.dst)[idx] = reinterpret_cast<const f32x4*>(p.src)[r * p.stride * p.pieces_per_row + c];
    }
}

// (index, score) answers -> the fixed-capacity int64 pair buffer the multi-GPU gather exchanges (dist.py): row pos[i]
// (or i) of `packed` receives (idx[i], sign-extended float bits of score[i]); rows nobody writes keep their sentinel.
struct PackPairsArgs {
    const long long* idx;   // [n * stride]
    const float* score;     // [n * stride]
    const int* pos;         // [n] destination rows, or nullptr for 0..n-1
    long long* packed;      // [capacity][2]
    int n, stride;          // stride: elements between consecutive answers (top-k buffers: k)
};
__global__ __launch_bounds__(256) void pack_pairs_kernel(const PackPairsArgs p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const long long row = p.pos ? p.pos[i] : i;
    p.packed[2 * row] = p.idx[(long long)i * p.stride];
    p.packed[2 * row + 1] = (long long)__builtin_bit_cast(int, p.score[(long long)i * p.stride]);
}

// ... and back: after the all_gather every rank holds world x rows_per_rank pairs; answer i of the batch is row i of its
// owner's block (dist.py).  One launch instead of a fancy-index gather + two element-wise conversions.
struct UnpackPairsArgs {
    const long long* gathered;   // [world * rows_per_rank][2]
    const int* owner;            // [n] rank that answered batch row i, or nullptr (single block: owner 0)
    long long* idx;              // [n]
    float* score;                // [n]
    int n, rows_per_rank;
};
__global__ __launch_bounds__(256) void unpack_pairs_kernel(const UnpackPairsArgs p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const long long row = (long long)(p.owner ? p.owner[i] : 0) * p.rows_per_rank + i;
    p.idx[i] = p.gathered[2 * row];
    p.score[i] = __builtin_bit_cast(float, (int)p.gathered[2 * row + 1]);
}

struct ArgmaxReduceArgs {
    const float* pval;
    const int* pidx;
    long long* idx_out;   // [B] int64 (np.argmax dtype)
    float* score_out;     // [B]
    int nblk, B, Bstride;
    int idx_scale;        // row ids are multiplied by this (upright search on the compacted every-k-th-row copy)
};

__device__ __forceinline__ void argmax_reduce_block(const ArgmaxReduceArgs& p, const int query) {
    AAE_DYN_SMEM(smem_raw);
    float* red_v = reinterpret_cast<float*>(smem_raw);      // [4]
    int* red_i = reinterpret_cast<int*>(red_v + 4);         // [4]
    const int b = query, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float bv = kNegInf;
    int bi = 0x7fffffff;
    for (int k0 = tid; k0 < p.nblk; k0 += 4 * 256) {        // 4 independent loads in flight per thread
        float v[4];
        int ix[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * 256;
            v[u] = kNegInf;
            ix[u] = 0x7fffffff;
            if (k < p.nblk) { v[u] = p.pval[(long long)k * p.Bstride + b]; ix[u] = p.pidx[(long long)k * p.Bstride + b]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (better(v[u], ix[u], bv, bi)) { bv = v[u]; bi = ix[u]; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = shfl_xor(bv, m);
        const int oi = shfl_xor(bi, m);
        if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (better(red_v[w], red_i[w], bv, bi)) { bv = red_v[w]; bi = red_i[w]; }
        if (bi == 0x7fffffff) bi = 0;          // all-NaN scores: np.argmax would also answer 0
        p.idx_out[b] = (long long)bi * p.idx_scale;
        p.score_out[b] = bv;
    }
}

__global__ __launch_bounds__(256) void argmax_reduce_kernel(const ArgmaxReduceArgs p) { argmax_reduce_block(p, (int)blockIdx.x); }

// ... for several codebooks in one launch (behind scan_resident_multi_kernel): block range.first[o] + q answers query q of object o
struct ArgmaxReduceMultiArgs {
    MultiRange range;
    ArgmaxReduceArgs item[kMultiMax];
};
__global__ __launch_bounds__(256) void argmax_reduce_multi_kernel(const ArgmaxReduceMultiArgs m) {
    const int o = multi_find(m.range, (int)blockIdx.x);
    argmax_reduce_block(m.item[o], (int)blockIdx.x - m.range.first[o]);
}

// --------------------------------------------------------------------- top-k
// One block per query over a materialised similarity row.  k rounds; round j
// selects the best element strictly after round j-1's winner in the order
// (score descending, index ascending) -- no masking writes, deterministic.
// ---- top-k (codebook.py:69-71: argpartition + sort, here in canonical order: score descending,
// lower index first on ties) over a materialised similarity matrix, in two parallel levels:
//   topk_chunks_kernel : grid (chunks, B); a block selects the k best of its 2048-entry chunk
//                        (8 values per thread in registers, k argmax passes with exclusion);
//   topk_merge_kernel  : grid B; the same selection over the chunks*k candidates.
// Selection pass j picks the best entry strictly after winner j-1 in the canonical order, so
// no state other than the previous winner is carried.
struct TopKArgs {
    const float* cs;       // [B][N]
    float* cand_v;         // [B][chunks][k]
    int* cand_i;
    long long* idx_out;    // [B][k]
    float* score_out;      // [B][k]
    int N, k, chunks;
};

constexpr int kTopKChunk = 2048;

__global__ __launch_bounds__(256) void topk_chunks_kernel(const TopKArgs p) {
    AAE_DYN_SMEM(smem_raw);
    float* red = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x;
    const int n0 = blockIdx.x * kTopKChunk;
    const float* row = p.cs + (long long)blockIdx.y * p.N;
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int n = n0 + tid + 256 * u;
        v[u] = n < p.N ? row[n] : kNegInf;
    }
    float pv = __builtin_huge_valf();
    int pi = -1;
    const long long obase = ((long long)blockIdx.y * p.chunks + blockIdx.x) * p.k;
    for (int j = 0; j < p.k; ++j) {
        float bv = kNegInf;
        int bi = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int n = n0 + tid + 256 * u;
            const bool after = n < p.N && ((v[u] < pv) || (v[u] == pv && n > pi));
            if (after && better(v[u], n, bv, bi)) { bv = v[u]; bi = n; }
        }
        block_best(bv, bi, red);
        if (tid == 0) { p.cand_v[obase + j] = bv; p.cand_i[obase + j] = bi; }
        pv = bv; pi = bi;
        if (bi == 0x7fffffff) pv = kNegInf;          // chunk exhausted: later passes emit sentinels too
    }
}

// REGS: the candidates of a query (chunks * k <= 256 * kTopKMergeSlots) are read ONCE into registers and the k selection passes
// run over those -- each pass of the plain form re-reads them from memory (B = 256, 241 x 5 candidates: 12.5 -> 6.9 us).
constexpr int kTopKMergeSlots = 8;
template <bool REGS>
__global__ __launch_bounds__(256) void topk_merge_kernel(const TopKArgs p) {
    AAE_DYN_SMEM(smem_raw);
    float* red = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x;
    const int total = p.chunks * p.k;
    const float* cv = p.cand_v + (long long)blockIdx.x * total;
    const int* ci = p.cand_i + (long long)blockIdx.x * total;
    float rv[REGS ? kTopKMergeSlots : 1];
    int rn[REGS ? kTopKMergeSlots : 1];
    if (REGS) {
#pragma unroll
        for (int u = 0; u < kTopKMergeSlots; ++u) {
            const int c = tid + 256 * u;
            rv[u] = c < total ? cv[c] : kNegInf;
            rn[u] = c < total ? ci[c] : 0x7fffffff;
        }
    }
    float pv = __builtin_huge_valf();
    int pi = -1;
    for (int j = 0; j < p.k; ++j) {
        float bv = kNegInf;
        int bi = 0x7fffffff;
        if (REGS) {
#pragma unroll
            for (int u = 0; u < kTopKMergeSlots; ++u) {
                const bool after = rn[u] != 0x7fffffff && ((rv[u] < pv) || (rv[u] == pv && rn[u] > pi));
                if (after && better(rv[u], rn[u], bv, bi)) { bv = rv[u]; bi = rn[u]; }
            }
        } else {
            for (int c = tid; c < total; c += 256) {
                const float v = cv[c];
                const int n = ci[c];
                const bool after = n != 0x7fffffff && ((v < pv) || (v == pv && n > pi));
                if (after && better(v, n, bv, bi)) { bv = v; bi = n; }
            }
        }
        block_best(bv, bi, red);
        if (tid == 0) {
            p.idx_out[(long long)blockIdx.x * p.k + j] = (bi == 0x7fffffff) ? 0 : bi;
            p.score_out[(long long)blockIdx.x * p.k + j] = bv;
        }
        pv = bv; pi = bi;
    }
}

}  // namespace aae
