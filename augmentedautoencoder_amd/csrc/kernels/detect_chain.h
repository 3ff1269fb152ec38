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
// One kernel per (batch class, wave-tile shape of each of the three conv phases): the shapes are TEMPLATE parameters and the phases
// are written out one after the other, so every layer's arguments sit at fixed offsets of the kernel-argument segment and
// each slab loop is compiled as in its stand-alone kernel.  (A first version chose layer and shape at run time inside one
// loop: under that kernel's register pressure the compiler re-read the slab loop's scalars from the argument segment in
// every slab and kept the wave-uniform K cursor in vector registers -- conv2 at B = 1 took 23.5 us instead of 16,
// profiles/r11_small/chain_first_version_timeline.jsonl.)  Instantiated for the shape sequences the planner produces for
// B = 1 ... 4 of the reference network (4 conv layers: train_template.cfg:50); every other network keeps the six launches.
//
// Phases (every phase = a grid-stride loop over the work items of the six-launch plan, same decomposition, same
// summation orders -> bit-identical to the six launches, which stay as the reference path and serve every other case):
//   conv layer l = 2 ... L : conv_wavek_block<MT, NT, 4, 2, CHAIN> (wave-split-K implicit GEMM; split layers finish inside
//                            the phase by the last block of a tile, no waiting)                    -> barrier
//   dense                  : dense_gemv_block (weight-streaming GEMV, chunk rows added by the last block)  -> barrier
//   scan                   : every wave streams its own contiguous range of codebook rows (two 32-row groups of 16 loads in
//                            flight), block partial -> coherent store -> ticket -> the last block merges and answers.
// Coherence: everything another block reads is stored device-coherently (sc1), every buffer is written once per launch and
// never read before its barrier (device_intrinsics.h, grid barrier).  Residency: grid = min(CUs, 256) blocks of 256
// threads with > 80 KB of LDS each, so a CU never holds two and every block is resident; spins are bounded.
#pragma once

namespace aae {

constexpr int kChainConv = 3;                    // conv phases of the persistent launch: layers 2, 3, 4 of a four-layer encoder
constexpr int kChainSmem = 96 * 1024;            // > half a CU's LDS: one block per CU (also covers the 64 x 64 tile's 64 KB)

struct DetectChainArgs {
    ConvWaveKArgs conv[kChainConv];              // layers 2 ... 4 of the encoder, planned as for the stand-alone launches
    DenseGemvArgs dense;
    int dense_tiles;                             // CoutPad / 128
    int dense_chunks;                            // K / 128
    ScanArgs scan;                               // tickets / idx_out / score_out set: the scan answers inside the launch
    int has_scan;
    GridBarrier barrier;
    long long* timeline;                         // optional [blocks][kChainStamps] wall-clock stamps (100 MHz) of thread 0 per phase edge (tools/chain_timeline.py)
};

constexpr int kChainStamps = 40;

// wave-tile shape codes: 0 = 32 x 32, 1 = 64 x 32, 2 = 64 x 64
template <int SHAPE>
struct ChainShape {
    static constexpr int MT = SHAPE == 0 ? 1 : 2, NT = SHAPE == 2 ? 2 : 1;
};

// one conv phase: this block's work items of layer `a` (the first one may find its first B fragments in pf)
template <int SHAPE>
__device__ __forceinline__ void chain_conv_phase(const ConvWaveKArgs& a, int blk, int G, float* red, int* flag, const WaveKPrefetch& pf, bool have_pf) {
    const int nblk = a.num_mt * a.num_nt * a.gsplits;
    for (int L = blk; L < nblk; L += G)
        conv_wavek_block<ChainShape<SHAPE>::MT, ChainShape<SHAPE>::NT, 4, 2, true, SHAPE != 1>(a, L, nblk, red, flag, pf, have_pf && L == blk);     // (the schedules of the stand-alone launches at their defaults: spread for 32 x 32 and 64 x 64 tiles)
}

// scan phase: rows [row, row + 32) of a wave against the MQ queries -- scan_issue32 / scan_scores32 / wave_max_first_lane of
// codebook_scan_f32.h, i.e. the same bits per row as the stand-alone stream scan; the running (max, first row) of a wave is
// wave-uniform (rows arrive in ascending order: a later row replaces the best only when it is strictly larger)
template <int NQ>
__device__ __forceinline__ void chain_scan_consume(int row, int row_end, const f32x4 (&e)[16], const f32x4 (&qv)[NQ],
                                                   float (&best_v)[NQ], int (&best_i)[NQ]) {
    const int lane = threadIdx.x & 63;
    const bool cand = row + (lane >> 1) < row_end;
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
        const float d = scan_scores32(e, qv[b]);
        int first;
        const float m = wave_max_first_lane(cand ? d : kNegInf, first);
        if (first >= 0 && m > best_v[b]) { best_v[b] = m; best_i[b] = row + (first >> 1); }
    }
}

template <int MQ, int S0, int S1, int S2>
__global__ __launch_bounds__(256, 1) void detect_chain_kernel(const DetectChainArgs p) {
    AAE_DYN_SMEM(smem_raw);
    float* red = reinterpret_cast<float*>(smem_raw);
    int* flag = reinterpret_cast<int*>(smem_raw + kChainSmem - 16);
    const int G = (int)gridDim.x, blk = (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    unsigned phase = 0;
    int stamp_no = 0;
    auto stamp = [&]() {                                         // phase edges of this block, in order (profiling aid; nullptr in production)
        if (p.timeline && tid == 0 && stamp_no < kChainStamps) p.timeline[(long long)blk * kChainStamps + stamp_no] = wall_ticks();
        ++stamp_no;
    };
    stamp();
    // ---------------------------------------------------------------- conv layers 2, 3, 4
    {
        WaveKPrefetch pf;
        chain_conv_phase<S0>(p.conv[0], blk, G, red, flag, pf, false);
        stamp();                                                 // work of the phase done
        grid_barrier_arrive(p.barrier, (unsigned)G, (unsigned)blk, ++phase);
        stamp();                                                 // arrived
        // what the next phase needs and this one does not produce, requested while the barrier closes
        conv_wavek_prefetch_b(p.conv[1], blk, p.conv[1].num_mt * p.conv[1].num_nt * p.conv[1].gsplits, ChainShape<S1>::NT, 4, pf);
        grid_barrier_wait(p.barrier, (unsigned)G, (unsigned)blk, phase);
        stamp();                                                 // released
        chain_conv_phase<S1>(p.conv[1], blk, G, red, flag, pf, true);
        stamp();
        grid_barrier_arrive(p.barrier, (unsigned)G, (unsigned)blk, ++phase);
        stamp();
        conv_wavek_prefetch_b(p.conv[2], blk, p.conv[2].num_mt * p.conv[2].num_nt * p.conv[2].gsplits, ChainShape<S2>::NT, 4, pf);
        grid_barrier_wait(p.barrier, (unsigned)G, (unsigned)blk, phase);
        stamp();
        chain_conv_phase<S2>(p.conv[2], blk, G, red, flag, pf, true);
        stamp();
    }

    // ---------------------------------------------------------------- dense layer
    {
        const int items = p.dense_chunks * p.dense_tiles;
        f32x4 wdense[16];                                        // (live only from here to the first work item: not across the conv loops)
        grid_barrier_arrive(p.barrier, (unsigned)G, (unsigned)blk, ++phase);
        stamp();
        dense_gemv_load_weights(p.dense, blk % p.dense_chunks, blk / p.dense_chunks, blk < items, wdense);
        grid_barrier_wait(p.barrier, (unsigned)G, (unsigned)blk, phase);
        stamp();
        for (int it = blk; it < items; it += G)
            dense_gemv_block<MQ, true, true>(p.dense, it % p.dense_chunks, it / p.dense_chunks, p.dense_chunks, smem_raw, wdense, it == blk);
        stamp();                                                 // dense work done
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
    stamp();
    scan_issue32(s, ebuf, row_begin, row_end, ea);              // the first 64 KB of this block's rows fly while the barrier closes
    if (blk == 0) ticket_prepare_slot(s.tickets, s.nonce, (unsigned)G);
    grid_barrier_wait(p.barrier, (unsigned)G, (unsigned)blk, phase);
    stamp();

    // tf.nn.l2_normalize(z, 1) (codebook.py:27), per query, replicated in every lane -- as scan_stream_kernel does it
    const int kq = lane & 31, col = kq * 4;
    const bool col_ok = col < s.J;
    const buffer_rsrc zbuf = make_buffer(s.z, (unsigned)(s.B * s.J * 4));
    f32x4 zv[MQ], qv[MQ];
#pragma unroll
    for (int b = 0; b < MQ; ++b) {
        zv[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (b < s.B && col_ok) zv[b] = coherent_load4(zbuf, (unsigned)((b * s.J + col) * 4));
    }
    scan_normalise_queries<MQ>(zv, qv);
    float best_v[MQ];
    int best_i[MQ];
#pragma unroll
    for (int b = 0; b < MQ; ++b) { best_v[b] = kNegInf; best_i[b] = 0x7fffffff; }
    for (int row = row_begin; row < row_end; row += 64) {
        scan_issue32(s, ebuf, row + 32, row_end, eb);
        chain_scan_consume<MQ>(row, row_end, ea, qv, best_v, best_i);
        scan_issue32(s, ebuf, row + 64, row_end, ea);
        chain_scan_consume<MQ>(row + 32, row_end, eb, qv, best_v, best_i);
    }
    // the 4 waves, then the blocks (ticket)
    float* red_v = red;                                          // [4][MQ]
    int* red_i = reinterpret_cast<int*>(red_v + 4 * MQ);
    __syncthreads();                                             // (the dense phase's LDS use is over)
    if (lane == 0) {
#pragma unroll
        for (int b = 0; b < MQ; ++b) { red_v[wave * MQ + b] = best_v[b]; red_i[wave * MQ + b] = best_i[b]; }
    }
    __syncthreads();
    scan_store_block_partials<MQ>(s, red_v, red_i, blockIdx.x, gridDim.x);
    stamp();                                                     // this block's rows scanned
    scan_ticket_finish<MQ>(s, red_v + 32, blockIdx.x, gridDim.x);
    stamp();
}

}  // namespace aae
