// Dense layer for tiny batches (B <= 8; B <= 4 is the reference's per-detection usage): z = flatten(x) . W + b as a
// weight-streaming GEMV (/root/reference/auto_pose/ae/encoder.py:58-68).  At M <= 4 a 128 x 128 MFMA tile is
// 97 % padding and the layer is the 16.8 MB read of W: each block streams a 128-k chunk of the packed weights
// [K/4][CoutPad][4] -- thread n reads 16 B = four consecutive k of ITS output column, 2 KB contiguous per
// slot row across the block -- against the activation chunk staged in LDS, and writes one partial row per
// batch element.  The chunks are added up (fixed order), with bias and the optional BN, either by
// splitk_reduce_kernel in a second launch or -- TICKET = true, the default -- inside this launch by the last
// block to arrive (block_ticket_arrive): one launch less on the per-detection path.
// B >= 2 (round 4): ONE finishing block had to read chunks x B x Cout x 4 bytes of partials (512 KB at B = 4 of the default
// net) at the ~100 GB/s a single block gets out of handed-off data, one batch row after the other: 13.3 us for the 16.8 MB
// that B = 1 streams in 7.5.  Now the finish is a two-level tree that follows the two-level ticket: the last arriver of
// each of the 16 chunk groups adds ITS group's chunk rows (chunks g, g + 16, ...: 8 KB x B) and publishes a group row; the
// last of those 16 adds the group rows, bias and batch-norm.  Partials leave as 16-byte pieces (a 4-byte device-coherent
// store costs about six times a 16-byte one per byte).  Fixed orders: the same bits whichever blocks arrive last.
#pragma once

namespace aae {

struct DenseGemvArgs {
    const float* x;        // [B][K]
    const float* wp;       // [K/4][CoutPad][4]
    unsigned wp_bytes;
    float* partial;        // [chunks][B][Cout] (+ [kGemvGroups][B][Cout] group rows behind them: the two-level finish)
    unsigned partial_bytes;
    int B, K, Cout, CoutPad;
    // TICKET mode: the last block of a column tile finishes z = sum(chunks) + bias [, BN]
    const float* bias;
    const float* bn_scale;
    const float* bn_shift;
    float* out;                    // [B][Cout]
    unsigned long long* tickets;   // [CoutPad / 128][kTicketSlotWords]
    unsigned nonce;
    int relu;
};

constexpr int kGemvChunk = 128;            // k per block = 32 slot rows: 64 KB of weights at CoutPad = 128
constexpr int kGemvGroups = kTicketGroups; // chunk groups of the two-level finish (= the group words of a ticket slot)

constexpr int kGemvTicketSmem = 8 * 128 * 4 + 16;                // the finishing block's 8 row-group sums + the ticket flag
constexpr int kGemvMaxBatch = 8;                                 // batch rows one block carries (MQ): 8 KB of LDS for the activation chunk + half sums
// the ticket flag sits behind whatever the block keeps in LDS: xs + red (2 x MQ x 512 B) or the 8 row-group sums (4 KB)
template <int MQ>
constexpr int gemv_flag_offset() { return 2 * MQ * kGemvChunk * 4 > 8 * 128 * 4 ? 2 * MQ * kGemvChunk * 4 : 8 * 128 * 4; }

// this thread's 16 weight pieces of chunk bx, column tile by (all in flight at once; out of range = zeros, no traffic)
__device__ __forceinline__ void dense_gemv_load_weights(const DenseGemvArgs& p, int bx, int by, bool live, f32x4 (&w)[16]) {
    const int tid = threadIdx.x;
    const int n = by * 128 + (tid & 127), half = tid >> 7;
    const int k0 = bx * kGemvChunk;
    const buffer_rsrc wbuf = make_buffer(p.wp, p.wp_bytes);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int slot = (k0 >> 2) + half * 16 + j;
        w[j] = buffer_load4(wbuf, (live && slot * 4 < p.K && n < p.CoutPad) ? (unsigned)((slot * p.CoutPad + n) * 16) : kOobOffset);
    }
}

// One (chunk bx of nbx, column tile by) work item.  CHAIN: a phase of the persistent per-detection kernel (detect_chain.h):
// z is read by other blocks of the same launch after a grid barrier -> device-coherent stores; `wpre` (use_wpre): the
// weights were requested ahead of time.
template <int MQ, bool TICKET, bool CHAIN>
__device__ __forceinline__ void dense_gemv_block(const DenseGemvArgs& p, const int bx, const int by, const int nbx, unsigned char* smem_raw,
                                                 const f32x4 (&wpre)[16], const bool use_wpre) {
    float* xs = reinterpret_cast<float*>(smem_raw);              // [MQ][128] activation chunk
    float* red = xs + MQ * kGemvChunk;                           // [MQ][128] second half's sums
    const int tid = threadIdx.x;
    const int n = by * 128 + (tid & 127), half = tid >> 7;       // two k-halves of 16 slot rows each
    const int k0 = bx * kGemvChunk;
    for (int e = tid; e < MQ * kGemvChunk; e += 256) {
        const int m = e / kGemvChunk, k = e - m * kGemvChunk;
        xs[e] = (m < p.B && k0 + k < p.K) ? p.x[(long long)m * p.K + k0 + k] : 0.f;
    }
    f32x4 w[16];
    if (CHAIN && use_wpre) {
#pragma unroll
        for (int j = 0; j < 16; ++j) w[j] = wpre[j];
    } else {
        dense_gemv_load_weights(p, bx, by, true, w);             // all 16 loads in flight before the first use
    }
    if (TICKET && bx == 0) ticket_prepare_slot(p.tickets + by * kTicketSlotWords, p.nonce, nbx);   // loads in flight; arrivals come later
    __syncthreads();
    float acc[MQ];
#pragma unroll
    for (int m = 0; m < MQ; ++m) acc[m] = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int kk = (half * 16 + j) * 4;
#pragma unroll
        for (int m = 0; m < MQ; ++m) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + m * kGemvChunk + kk);   // broadcast read
            acc[m] = fmaf(xv.x, w[j].x, acc[m]);
            acc[m] = fmaf(xv.y, w[j].y, acc[m]);
            acc[m] = fmaf(xv.z, w[j].z, acc[m]);
            acc[m] = fmaf(xv.w, w[j].w, acc[m]);
        }
    }
    if (half == 1) {
#pragma unroll
        for (int m = 0; m < MQ; ++m) red[m * 128 + (tid & 127)] = acc[m];
    }
    __syncthreads();
    const buffer_rsrc pbuf = make_buffer(p.partial, p.partial_bytes);
    const bool tree = TICKET && MQ > 1 && nbx > (int)kTicketSingleLevelMax;      // two-level finish (the ticket has its group words)
    if (!tree) {
        if (half == 0 && n < p.Cout) {
#pragma unroll
            for (int m = 0; m < MQ; ++m)
                if (m < p.B) {
                    const float sum = acc[m] + red[m * 128 + (tid & 127)];
                    const unsigned at = (unsigned)(((bx * p.B + m) * p.Cout + n) * 4);
                    if (TICKET) coherent_store1(pbuf, at, __builtin_bit_cast(uint32_t, sum));      // read back by another block of this launch
                    else p.partial[((long long)bx * p.B + m) * p.Cout + n] = sum;
                }
        }
    }
    if constexpr (TICKET) {
        if (tree) {
            // ---- the chunk row of every batch element as 16-byte pieces: thread (m = tid / 32, n4 = tid % 32) ----------------
            if (half == 0) {
#pragma unroll
                for (int m = 0; m < MQ; ++m) red[m * 128 + (tid & 127)] += acc[m];              // (own element: the k-half pair's sum)
            }
            __syncthreads();
            const int pm = tid >> 5, pn = by * 128 + (tid & 31) * 4;
            const bool piece = pm < p.B && pm < MQ && pn < p.Cout;                               // (Cout % 4 == 0: the host checks)
            if (piece) coherent_store4(pbuf, (unsigned)(((bx * p.B + pm) * p.Cout + pn) * 4), *reinterpret_cast<const f32x4*>(red + pm * 128 + (tid & 31) * 4));
            int* flag = reinterpret_cast<int*>(smem_raw + gemv_flag_offset<MQ>());
            unsigned long long* words = p.tickets + by * kTicketSlotWords;
            const int groups = kGemvGroups, g = bx % groups, members = (nbx - g + groups - 1) / groups;
            // level 1: the last arriver of chunk group g adds chunks g, g + groups, ... (in that order)
            block_ticket_publish();
            if (tid == 0) {
                const bool last = ticket_count(words + (1 + g) * kTicketGroupStride, p.nonce) == (unsigned)members;
                if (last) ticket_clear(words + (1 + g) * kTicketGroupStride);
                *flag = last ? 1 : 0;
            }
            __syncthreads();
            if (*flag == 0) return;
            const unsigned group_base = (unsigned)nbx * (unsigned)(p.B * p.Cout) * 4u;          // group rows sit behind the chunk rows
            {
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
                constexpr int kInFlight = 16;
                for (int j0 = 0; j0 < members; j0 += kInFlight) {
                    f32x4 t[kInFlight];
#pragma unroll
                    for (int u = 0; u < kInFlight; ++u) {
                        const int c = g + groups * (j0 + u);
                        t[u] = coherent_load4(pbuf, (piece && j0 + u < members) ? (unsigned)(((c * p.B + pm) * p.Cout + pn) * 4) : kOobOffset);
                    }
#pragma unroll
                    for (int u = 0; u < kInFlight; ++u) s += t[u];
                }
                if (piece) coherent_store4(pbuf, group_base + (unsigned)(((g * p.B + pm) * p.Cout + pn) * 4), s);
            }
            // level 2: the last of the group finishers adds the group rows in group order, then bias / ReLU / batch-norm
            block_ticket_publish();
            if (tid == 0) {
                const bool last = ticket_count(words, p.nonce) == (unsigned)groups;
                if (last) ticket_clear(words);
                *flag = last ? 1 : 0;
            }
            __syncthreads();
            if (*flag == 0) return;
            f32x4 t[kGemvGroups];
#pragma unroll
            for (int u = 0; u < kGemvGroups; ++u)
                t[u] = coherent_load4(pbuf, piece ? group_base + (unsigned)(((u * p.B + pm) * p.Cout + pn) * 4) : kOobOffset);
            f32x4 e_bias = {0.f, 0.f, 0.f, 0.f}, e_sc = {1.f, 1.f, 1.f, 1.f}, e_sh = {0.f, 0.f, 0.f, 0.f};
            if (piece) {
                e_bias = *reinterpret_cast<const f32x4*>(p.bias + pn);
                if (p.bn_scale) { e_sc = *reinterpret_cast<const f32x4*>(p.bn_scale + pn); e_sh = *reinterpret_cast<const f32x4*>(p.bn_shift + pn); }
            }
            f32x4 v = t[0];
#pragma unroll
            for (int u = 1; u < kGemvGroups; ++u) v += t[u];
            v += e_bias;
            if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
            if (p.bn_scale) v = v * e_sc + e_sh;
            if (piece) {
                const buffer_rsrc zbuf = make_buffer(p.out, (unsigned)(p.B * p.Cout * 4));
                if (CHAIN) coherent_store4(zbuf, (unsigned)((pm * p.Cout + pn) * 4), v);
                else *reinterpret_cast<f32x4*>(p.out + (long long)pm * p.Cout + pn) = v;
            }
            return;
        }
        // The last of the nbx chunk blocks of this column tile adds the chunk rows: thread (group = tid / 32,
        // n4 = tid % 32) sums chunks group, group + 8, ... of four neighbouring columns (16-B loads, all of a batch
        // row's loads in flight), the 8 group sums meet in LDS and are added in group order -- one fixed tree,
        // whichever block happens to finish last.  (Needs Cout % 4 == 0; the host checks.)
        float* gsum = reinterpret_cast<float*>(smem_raw);        // [8][128] (xs / red are dead by now)
        int* flag = reinterpret_cast<int*>(smem_raw + gemv_flag_offset<MQ>());
        __syncthreads();
        if (!block_ticket_arrive(p.tickets + by * kTicketSlotWords, p.nonce, nbx, bx, flag)) return;
        const int group = tid >> 5, n4 = by * 128 + (tid & 31) * 4;
        const int chunks = nbx;
        const int nn_e = by * 128 + (tid & 127);                // epilogue constants first: their latency hides under the partial loads
        const bool nn_ok = tid < 128 && nn_e < p.Cout;
        const float e_bias = nn_ok ? p.bias[nn_e] : 0.f;
        const float e_sc = (nn_ok && p.bn_scale) ? p.bn_scale[nn_e] : 1.f, e_sh = (nn_ok && p.bn_scale) ? p.bn_shift[nn_e] : 0.f;
        const buffer_rsrc zbuf = make_buffer(p.out, (unsigned)(p.B * p.Cout * 4));
        // A coherent load is a round trip to the memory side (~0.7 us): up to 32 of a thread's chunk rows are in flight at
        // once (rows beyond `chunks` read out of range = zeros), added in chunk order.  With 8 in flight the default net's
        // 256 chunks cost four round trips per batch row: 2.5 us of the 7.6 us kernel at B = 1, 10 of 15 us at B = 4.
        constexpr int kInFlight = 32;
        for (int m = 0; m < p.B; ++m) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int c0 = group; c0 < chunks; c0 += 8 * kInFlight) {
                f32x4 t[kInFlight];
#pragma unroll
                for (int u = 0; u < kInFlight; ++u) {
                    const int c = c0 + 8 * u;
                    t[u] = coherent_load4(pbuf, (c < chunks && n4 < p.Cout) ? (unsigned)(((c * p.B + m) * p.Cout + n4) * 4) : kOobOffset);
                }
#pragma unroll
                for (int u = 0; u < kInFlight; ++u) s += t[u];
            }
            *reinterpret_cast<f32x4*>(gsum + group * 128 + (tid & 31) * 4) = s;
            __syncthreads();
            if (tid < 128 && (int)(by * 128 + tid) < p.Cout) {
                const int nn = by * 128 + tid;
                float v = gsum[tid];
#pragma unroll
                for (int k = 1; k < 8; ++k) v += gsum[k * 128 + tid];
                v += e_bias;
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.bn_scale) v = v * e_sc + e_sh;
                if (CHAIN) coherent_store1(zbuf, (unsigned)((m * p.Cout + nn) * 4), __builtin_bit_cast(uint32_t, v));
                else p.out[(long long)m * p.Cout + nn] = v;
            }
            __syncthreads();
        }
    }
}

template <int MQ, bool TICKET = false>
__global__ __launch_bounds__(256) void dense_gemv_f32_kernel(const DenseGemvArgs p) {
    AAE_DYN_SMEM(smem_raw);
    f32x4 none[16];
    dense_gemv_block<MQ, TICKET, false>(p, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, smem_raw, none, false);
}

// ---- the dense layer of SEVERAL objects in one launch (multi_launch.h): grid.x = the objects' chunk ranges one behind the other
struct DenseGemvMultiArgs {
    MultiRange range;
    DenseGemvArgs item[kMultiMax];
};
template <int MQ>
__global__ __launch_bounds__(256) void dense_gemv_multi_kernel(const DenseGemvMultiArgs m) {
    AAE_DYN_SMEM(smem_raw);
    const int o = multi_find(m.range, (int)blockIdx.x);
    f32x4 none[16];
    dense_gemv_block<MQ, true, false>(m.item[o], (int)blockIdx.x - m.range.first[o], (int)blockIdx.y, m.range.first[o + 1] - m.range.first[o], smem_raw, none, false);
}

}  // namespace aae
