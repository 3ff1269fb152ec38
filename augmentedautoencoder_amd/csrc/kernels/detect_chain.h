// The per-detection query as ONE persistent launch behind conv1: every later layer of the encoder, the dense layer and
// the codebook scan are phases that the SAME resident blocks (one per CU) walk, separated by grid-wide barriers.
//
// Replaces, for batches of 1 ... 4 crops, what the reference does once per detected box
// (/root/reference/auto_pose/m3_interface/ae_pose_estimator.py:143-170: one session.run of encoder + cos_similarity per
// box, then np.argmax): encoder.py:41-68 from the second convolution on, codebook.py:27 (l2_normalize), :50 (matmul
// against the codebook) and :64-68 (arg-max / upright stride on the compacted copy).
//
// Why.  As six launches the query costs 79-81 us at B = 1 (profiles/r11_small): 27 us of fp32 MFMA work, 107 MB of
// weights + codebook, and per launch ~2.5 us of boundary plus a start-up in which nothing useful is in flight -- kernel
// arguments, index arithmetic, then the first operand round trip to HBM.  A grid barrier costs about what a kernel boundary
// costs, so nothing is won by merely removing launches; what a persistent kernel CAN do is ask for data that does not
// depend on the previous phase BEFORE the barrier closes: the next layer's first weight fragments, the dense layer's whole
// weight chunk, the first rows of the codebook.  They fly while the barrier's atomics travel (arrive -> prefetch -> wait).
//
// Phases (every phase = a grid-stride loop over the work items of the six-launch plan, same decomposition, same
// summation orders -> bit-identical to the six launches, which stay as the reference path and serve every other case):
//   conv layer l = 2 ... L : conv_wavek_block<MT, NT, 4, 2, CHAIN> (wave-split-K implicit GEMM; split layers finish inside
//                            the phase by the last block of a tile, no waiting)                    -> barrier
//   dense                  : dense_gemv_block (weight-streaming GEMV, chunk rows added by the last block)  -> barrier
//   scan                   : every wave streams its own contiguous range of codebook rows (two row pairs of 16 loads in
//                            flight), block partial -> coherent store -> ticket -> the last block merges and answers.
// Coherence: everything another block reads is stored device-coherently (sc1), every buffer is written once per launch and
// never read before its barrier (device_intrinsics.h, grid barrier).  Residency: grid = min(CUs, 256) blocks of 256
// threads with > 80 KB of LDS each, so a CU never holds two and every block is resident; spins are bounded.
#pragma once

namespace aae {

constexpr int kChainMaxConv = AAE_MAX_LAYERS;
constexpr int kChainSmem = 96 * 1024;            // > half a CU's LDS: one block per CU (also covers the 64 x 64 tile's 64 KB)

struct DetectChainArgs {
    ConvWaveKArgs conv[kChainMaxConv];           // layers 2 ... L of the encoder, planned as for the stand-alone launches
    int shape[kChainMaxConv];                    // wave tile of each: 0 = 32x32, 1 = 64x32, 2 = 64x64
    int nconv;
    DenseGemvArgs dense;
    int dense_tiles;                             // CoutPad / 128
    int dense_chunks;                            // K / 128
    ScanArgs scan;                               // tickets / idx_out / score_out set: the scan answers inside the launch
    int has_scan;
    GridBarrier barrier;
};

__device__ __forceinline__ int chain_shape_nt(int shape) { return shape == 2 ? 2 : 1; }

// codebook rows [row, row + 32) of this wave: 16 loads of two rows each (one 512-B row per half-wave), clipped at row_end
__device__ __forceinline__ void chain_scan_issue(const ScanArgs& p, const buffer_rsrc& ebuf, int row, int row_end, f32x4 (&e)[16]) {
    const int lane = threadIdx.x & 63, rs = lane >> 5, col = (lane & 31) * 4;
    const bool col_ok = col < p.J;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int r = row + 2 * u + rs;
        e[u] = buffer_load4(ebuf, (r < row_end && col_ok) ? (unsigned)(r * p.J + col) * 4u : kOobOffset);
    }
}

template <int NQ>
__device__ __forceinline__ void chain_scan_consume(const ScanArgs& p, int row, int row_end, const f32x4 (&e)[16], const f32x4 (&qv)[NQ],
                                                   float (&best_v)[NQ], int (&best_i)[NQ]) {
    const int rs = (threadIdx.x & 63) >> 5;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int r = row + 2 * u + rs;
        const bool cand = r < row_end;
#pragma unroll
        for (int b = 0; b < NQ; ++b) {
            float d = e[u].x * qv[b].x;                          // the fma chain + DPP tree of scan_stream_kernel: same bits per row
            d = fmaf(e[u].y, qv[b].y, d);
            d = fmaf(e[u].z, qv[b].z, d);
            d = fmaf(e[u].w, qv[b].w, d);
            d = half_wave_sum(d);                                // total valid in lanes 16-31 / 48-63
            if (cand && d > best_v[b]) { best_v[b] = d; best_i[b] = r; }
        }
    }
}

template <int MQ>
__global__ __launch_bounds__(256, 1) void detect_chain_kernel(const DetectChainArgs p) {
    AAE_DYN_SMEM(smem_raw);
    float* red = reinterpret_cast<float*>(smem_raw);
    int* flag = reinterpret_cast<int*>(smem_raw + kChainSmem - 16);
    const int G = (int)gridDim.x, blk = (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    unsigned phase = 0;

    WaveKPrefetch pf;
    bool have_pf = false;
    f32x4 wdense[16];
    bool have_wdense = false;

    // ---------------------------------------------------------------- conv layers 2 ... L
    for (int li = 0; li < p.nconv; ++li) {
        const ConvWaveKArgs& a = p.conv[li];
        const int nblk = a.num_mt * a.num_nt * a.gsplits;
        for (int L = blk; L < nblk; L += G) {
            const bool use = have_pf && L == blk;
            if (p.shape[li] == 0) conv_wavek_block<1, 1, 4, 2, true>(a, L, nblk, red, flag, pf, use);
            else if (p.shape[li] == 1) conv_wavek_block<2, 1, 4, 2, true>(a, L, nblk, red, flag, pf, use);
            else conv_wavek_block<2, 2, 4, 2, true>(a, L, nblk, red, flag, pf, use);
        }
        grid_barrier_arrive(p.barrier, (unsigned)G, (unsigned)blk, ++phase);
        // what the next phase needs and this one does not produce, requested while the barrier closes
        have_pf = false;
        if (li + 1 < p.nconv) {
            const ConvWaveKArgs& nx = p.conv[li + 1];
            conv_wavek_prefetch_b(nx, blk, nx.num_mt * nx.num_nt * nx.gsplits, chain_shape_nt(p.shape[li + 1]), 4, pf);
            have_pf = true;
        } else {
            const int items = p.dense_chunks * p.dense_tiles;
            dense_gemv_load_weights(p.dense, blk % p.dense_chunks, blk / p.dense_chunks, blk < items, wdense);
            have_wdense = blk < items;
        }
        grid_barrier_wait(p.barrier, (unsigned)G, (unsigned)blk, phase);
    }

    // ---------------------------------------------------------------- dense layer
    {
        const int items = p.dense_chunks * p.dense_tiles;
        for (int it = blk; it < items; it += G)
            dense_gemv_block<MQ, true, true>(p.dense, it % p.dense_chunks, it / p.dense_chunks, p.dense_chunks, smem_raw, wdense,
                                             have_wdense && it == blk);
    }
    if (!p.has_scan) return;

    // ---------------------------------------------------------------- codebook scan
    // wave gw of 4 G owns rows [gw N / (4G), (gw + 1) N / (4G)): any split gives the same answer (per-row scores do not
    // depend on it, the arg-max is exact with its lowest-row tie rule)
    const ScanArgs& s = p.scan;
    const int gw = blk * 4 + wave, nw = 4 * G;
    const int row_begin = (int)((long long)gw * s.N / nw), row_end = (int)((long long)(gw + 1) * s.N / nw);
    const buffer_rsrc ebuf = make_buffer(s.E, s.e_bytes);
    f32x4 ea[16], eb[16];
    grid_barrier_arrive(p.barrier, (unsigned)G, (unsigned)blk, ++phase);
    chain_scan_issue(s, ebuf, row_begin, row_end, ea);          // the first 64 KB of this block's rows fly while the barrier closes
    if (blk == 0) ticket_prepare_slot(s.tickets, s.nonce, (unsigned)G);
    grid_barrier_wait(p.barrier, (unsigned)G, (unsigned)blk, phase);

    // tf.nn.l2_normalize(z, 1) (codebook.py:27), per query, replicated in every lane -- as scan_stream_kernel does it
    const int kq = lane & 31, col = kq * 4;
    const bool col_ok = col < s.J;
    const buffer_rsrc zbuf = make_buffer(s.z, (unsigned)(s.B * s.J * 4));
    f32x4 qv[MQ];
#pragma unroll
    for (int b = 0; b < MQ; ++b) {
        f32x4 zv = {0.f, 0.f, 0.f, 0.f};
        if (b < s.B && col_ok) zv = coherent_load4(zbuf, (unsigned)((b * s.J + col) * 4));
        float ss = zv.x * zv.x;
        ss = fmaf(zv.y, zv.y, ss);
        ss = fmaf(zv.z, zv.z, ss);
        ss = fmaf(zv.w, zv.w, ss);
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) ss += shfl_xor(ss, m);
        const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        qv[b] = zv * inv;
    }
    float best_v[MQ];
    int best_i[MQ];
#pragma unroll
    for (int b = 0; b < MQ; ++b) { best_v[b] = kNegInf; best_i[b] = row_begin + (lane >> 5); }
    for (int row = row_begin; row < row_end; row += 64) {
        chain_scan_issue(s, ebuf, row + 32, row_end, eb);
        chain_scan_consume<MQ>(s, row, row_end, ea, qv, best_v, best_i);
        chain_scan_issue(s, ebuf, row + 64, row_end, ea);
        chain_scan_consume<MQ>(s, row + 32, row_end, eb, qv, best_v, best_i);
    }
    // lane 31: even offsets, lane 63: odd offsets -> lane 63 combines, then the 4 waves, then the blocks (ticket)
    float* red_v = red;                                          // [4][MQ]
    int* red_i = reinterpret_cast<int*>(red_v + 4 * MQ);
    __syncthreads();                                             // (the dense phase's LDS use is over)
#pragma unroll
    for (int b = 0; b < MQ; ++b) {
        const float ov = shfl_xor(best_v[b], 32);
        const int oi = shfl_xor(best_i[b], 32);
        if (better(ov, oi, best_v[b], best_i[b])) { best_v[b] = ov; best_i[b] = oi; }
        if (lane == 63) { red_v[wave * MQ + b] = best_v[b]; red_i[wave * MQ + b] = best_i[b]; }
    }
    __syncthreads();
    if (tid < MQ && tid < s.B) {
        float v = red_v[tid];
        int ix = red_i[tid];
        for (int w = 1; w < 4; ++w)
            if (better(red_v[w * MQ + tid], red_i[w * MQ + tid], v, ix)) { v = red_v[w * MQ + tid]; ix = red_i[w * MQ + tid]; }
        scan_store_block_partial(s, tid, v, ix);
    }
    scan_ticket_finish(s, red_v + 32);
}

}  // namespace aae
