// Small-batch implicit GEMM on the fp32 matrix cores: K split across the WAVES of a block (and, when the
// layer has fewer output tiles than the chip has CUs, across blocks), operands straight from global memory
// into MFMA fragments -- no LDS staging, no slab barriers -- and the cross-block sum finished inside the same
// launch by the last block to arrive.
//
// Replaces tf.layers.conv2d(k, stride, 'same', relu) [+ inference batch-norm] of
// /root/reference/auto_pose/ae/encoder.py:41-52 for the reference's REAL usage: one crop per detection
// (m3_interface/ae_pose_estimator.py:143-170, one session.run per box), i.e. M = B*Ho*Wo of 64 ... 4096 rows.
//
// Why a second kernel.  conv_igemm_f32.h tiles 128 x 128 outputs per block and shares each staged slab among
// four waves.  At B = 1 a layer has 8 ... 64 such tiles for 256 CUs, so K has to be cut 32 ... 64 ways: every
// block then runs 3-4 slabs behind a full pipeline prologue, writes a whole fp32 partial tile to HBM (33 MB per
// layer at B = 1) and a second launch adds the partials up: 25 us per layer against 11 us of MFMA time.
// Here the unit of work is ONE WAVE with a (32*MT) x (32*NT) accumulator tile and its own K range:
//   * both operands arrive in fragment order by plain 16-B buffer loads: packed weights [K/4][CoutPad][4] are the
//     MFMA B fragments as they lie in memory; an A fragment is 16 B (4 channels) of one input pixel, read through
//     the bounds-checked view (TF 'SAME' padding = out-of-range offset = zeros, no branches);
//   * DEPTH slabs of fragments are in flight per wave (static register ring), the MFMAs of slab t run under the
//     loads of slabs t+1 ... t+DEPTH-1; nothing is shared between waves, so there is no barrier in the K loop;
//   * the WAVES partial tiles of a block meet in LDS and are added in wave order; with gsplits > 1 each block
//     writes its tile partial (lane-linear 16-B device-coherent stores) and takes a ticket (block_ticket_arrive):
//     the last of the gsplits blocks re-reads all partials in split order, applies bias / ReLU / BN and stores the
//     outputs.  Fixed orders everywhere -> run-to-run bit-identical results; 8x fewer partial bytes than the
//     128 x 128 split-K and no second launch.
//   * mid-size batches (round 4): when whole tiles leave the last round of blocks partly empty, the tiles of that round alone are
//     cut in K (tail_tiles / tail_gsplits below) -- smaller blocks, dispatched last, that fill every CU; their partials meet
//     through the same tickets.
// K order inside a slab and the (c, q) MFMA pairing are those of tile_f32.h, so each wave's partial is the same
// k-ordered fma chain the large kernel would compute over that K range.
#pragma once

namespace aae {

struct ConvWaveKArgs {
    const float* x;              // [B, H, W, Cin] NHWC
    unsigned x_bytes;            // < 0xFFFFFF00
    const float* wp;             // [K/4][CoutPad][4]
    unsigned wp_bytes;
    const float* bias;           // [Cout]
    const float* bn_scale;       // [Cout] or nullptr
    const float* bn_shift;
    float* out;                  // [M][Cout]
    float* partial;              // [tiles][gsplits][pieces per thread][threads][4]  (gsplits > 1)
    unsigned partial_bytes;
    unsigned long long* tickets; // [tiles]                              (gsplits > 1; gsplits <= kTicketSingleLevelMax)
    unsigned nonce;              // unique per launch, never 0
    int H, W, Cin, Ho, Wo, Cout, CoutPad;
    int KS, S, pt, pl;
    int M;                       // B*Ho*Wo
    int slabs_total;             // KS*KS*Cin/32
    int num_mt, num_nt, gsplits;
    // Tail split (batches whose tile count is not a multiple of what the chip holds at once): the LAST tail_tiles tiles of the
    // layer are cut tail_gsplits ways in K while the tiles in front of them are whole (gsplits == 1 then), so that the last,
    // partly filled round of blocks is made of SMALLER blocks that fill every CU: 576 tiles of 64 x 64 on 512 block slots cost
    // three tile times as 512 + 64 whole tiles, 2.3 as 512 whole + 64 x 4 quarter blocks.  0 = off.  Tickets and partials of the
    // tail tiles are indexed from the first tail tile (at most kLayerTicketWords of them).
    int tail_tiles = 0, tail_gsplits = 1;
    int relu;
    long long* timeline;         // optional [blocks][8] shader-clock stamps of wave 0 per phase (tools/ablate_wavek.py); nullptr in production
    int ablate;                  // timing experiments only (results are then wrong): 1 no A loads, 2 no B loads, 4 no MFMAs, 8 no cross-block hand-off
    int pingpong;                // 8-wave blocks: the two waves of a SIMD alternate load issue and MFMAs, a block barrier between the half-steps
    int spread;                  // wave tiles with >= 2 accumulators: the operand loads of the next slab issue one per q-step BETWEEN the MFMAs of this one
};

// Ticket preparation.  A launch whose blocks all arrive at a clean ticket word queue up behind the nonce install
// (measured at B = 1: 4 / 7 / 9 us for 4 / 8 / 16 simultaneous arrivals, against ~1 us when the nonce is already
// there and every arrival is one fetch-add).  The FIRST kernel of a forward call (conv1, whose launch precedes all
// ticketed launches in stream order) therefore installs (nonce, 0 arrivals) in the words of every later ticketed
// launch of that call: block 0 writes them while the other blocks already compute.  Purely an accelerator: words
// that were not prepared (another first-layer kernel, a stand-alone scan) are handled by the install path.
constexpr int kMaxTicketPrep = 10;
struct TicketPrep {
    unsigned long long* words[kMaxTicketPrep];
    int count[kMaxTicketPrep];
    unsigned nonce[kMaxTicketPrep];
    int n;
};
__device__ __forceinline__ void ticket_prep_install(const TicketPrep& t) {
    for (int e = 0; e < t.n; ++e)
        for (int i = threadIdx.x; i < t.count[e]; i += blockDim.x) t.words[e][i] = (unsigned long long)t.nonce[e] << 32;
}

// Out-of-range byte offset that stays out of range after the +96 a fragment's k-group adds.
constexpr uint32_t kOobBase = 0xFFFFFF00u;

template <int MT, int NT, int WAVES>
constexpr int conv_wavek_smem() { return WAVES * MT * NT * 16 * 64 * 4 + 16; }

// B fragments of a wave's first slab, fetched ahead of time (detect_chain.h: the weights of the NEXT layer are requested
// before the grid barrier that ends the current one; they do not depend on it)
struct WaveKPrefetch {
    f32x4 b[2][4];           // [ni][k-group]  (NT <= 2)
};

// this lane's B-fragment offset of slab t, column tile tn (see the B fragment comment in conv_wavek_block)
__device__ __forceinline__ void conv_wavek_prefetch_b(const ConvWaveKArgs& p, int L, int nblk, int NT, int WAVES, WaveKPrefetch& pf) {
    const int lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int tiles = p.num_mt * p.num_nt;
    const bool live = L < nblk;
    const int Lr = live ? xcd_remap(L, nblk) : 0;
    const int tn = (Lr / p.num_mt) % p.num_nt;
    const int g = Lr / tiles;
    const int b0 = (int)((long long)g * p.slabs_total / p.gsplits);
    const int b1 = (int)((long long)(g + 1) * p.slabs_total / p.gsplits);
    const int s0 = b0 + wave * (b1 - b0) / WAVES, s1 = b0 + (wave + 1) * (b1 - b0) / WAVES;
    const buffer_rsrc wbuf = make_buffer(p.wp, p.wp_bytes);
    const unsigned bo = (unsigned)((h * p.CoutPad + tn * (32 * NT) + i) * 16) + (unsigned)s0 * (unsigned)(8 * p.CoutPad * 16);
    const unsigned bw_group = (unsigned)(2 * p.CoutPad * 16);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            pf.b[ni][c] = buffer_load4(wbuf, (live && s0 < s1 && ni < NT) ? bo + c * bw_group + 512u * ni : kOobBase);
}

// One block's share of a layer: logical block L of nblk = tiles x gsplits.  red: conv_wavek_smem<MT, NT, WAVES>() bytes of LDS.
//   CHAIN = false: the stand-alone kernel below (plain output stores; the next launch is the consumer);
//   CHAIN = true : a phase of the persistent per-detection kernel (detect_chain.h): the outputs are read by OTHER blocks of
//                  the same launch after a grid barrier, so they leave as device-coherent write-through (sc1) stores, and
//                  `pf` (use_pf) carries this wave's first B fragments, requested before the barrier.
// The epilogue hands every thread whole 16-byte pieces of an output row: thread t finishes float4 number t + T j of the
// tile, i.e. sub-tile (t + T j) / 256, row q / 8, columns 4 (q % 8) ... + 3 of it (q = (t + T j) % 256).  In the lane-linear
// LDS image of an accumulator (register r of lane l at [r][l]) those four values are neighbours -- register
// (row & 3) + 4 (row >> 3), lanes 4 (q % 8) + 32 ((row >> 2) & 1) ... + 3 -- so the cross-wave sum reads one ds_read_b128
// per wave, conflict-free, and the row leaves as ONE 16-byte store instead of four 4-byte ones (a 4-byte sc1 store costs
// about six times a 16-byte one per byte).  Sums run in wave order, then split order: the bits do not depend on who finishes.
template <int MT, int NT, int WAVES, int DEPTH, bool CHAIN, bool SPREAD = false>
__device__ __forceinline__ void conv_wavek_block(const ConvWaveKArgs& p, const int Lphys, const int nblk, float* red, int* flag,
                                                 const WaveKPrefetch& pf, const bool use_pf) {
    constexpr int COMBOS = MT * NT * 16;                       // accumulator registers per lane
    constexpr int T = 64 * WAVES;
    constexpr int PIECES = 256 * MT * NT;                      // float4 pieces of the tile
    constexpr int NF4 = (PIECES + T - 1) / T;                  // ... a thread finishes (32 x 32 tiles on 8 waves: the first 256 threads one each, the rest none)
    static_assert(PIECES % T == 0 || PIECES < T, "every finishing thread finishes whole float4 pieces");

    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
#ifdef AAE_EXPERIMENTS
    auto stamp = [&](int k) {                                   // (CHAIN: the 100 MHz wall clock of the launch's other stamps)
        if (p.timeline && tid == 0) p.timeline[(long long)Lphys * 8 + k] = CHAIN ? wall_ticks() : clock_ticks();
    };
#else
    auto stamp = [](int) {};                                    // (phase stamps: the experiments build's profiling aid)
#endif
    stamp(0);

    const int pH = p.H, pW = p.W, pCin = p.Cin, pKS = p.KS;
#ifdef AAE_EXPERIMENTS
    const int ablate = CHAIN ? 0 : p.ablate;                    // (timing experiments exist for the stand-alone launches only)
#else
    constexpr int ablate = 0;                                   // (the product build has no option that makes results wrong)
#endif
    const int tiles = p.num_mt * p.num_nt;
    // logical block -> (tile, K part g of gs).  Head tiles [0, head): gsplits parts each, part-major; tail tiles: tail_gsplits parts.
    // Head and tail are remapped to the XCDs separately (each XCD gets a contiguous chunk of BOTH), and the tail blocks -- the
    // small ones -- are the last to be dispatched: they fill in behind the whole tiles.
    const int head = tiles - p.tail_tiles, head_blocks = head * p.gsplits;
    const bool in_tail = Lphys >= head_blocks;
    const int Lr = in_tail ? xcd_remap(Lphys - head_blocks, nblk - head_blocks) : xcd_remap(Lphys, head_blocks);
    const int span = in_tail ? p.tail_tiles : head;
    const int g = Lr / span;
    const int tile = (in_tail ? head : 0) + (Lr - g * span);
    const int gs = in_tail ? p.tail_gsplits : p.gsplits;
    const int tm = tile % p.num_mt;                            // M tiles of one (N tile, K split) are neighbours: they share its weights in L2
    const int tn = tile / p.num_mt;
    const int slot0 = in_tail ? head_blocks + (tile - head) * p.tail_gsplits : tile * p.gsplits;    // first partial slot of the tile
    const int ticket_word = p.tail_tiles > 0 ? tile - head : tile;
    // K ranges: block g of the tile walks slabs [b0, b1), its wave w the w-th part of that; sizes differ by at most one slab
    const int b0 = (int)((long long)g * p.slabs_total / gs);
    const int b1 = (int)((long long)(g + 1) * p.slabs_total / gs);
    const int s0 = b0 + wave * (b1 - b0) / WAVES;
    const int s1 = b0 + (wave + 1) * (b1 - b0) / WAVES;

    // ---- A fragments: lane (i, h) of M sub-tile mi owns output row m and reads 4 channels (slot 2c + h) per k-group c
    const buffer_rsrc xbuf = make_buffer(p.x, p.x_bytes);
    unsigned a_off[MT];
    int a_ih0[MT], a_iw0[MT];
    bool a_ok[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int m = tm * (32 * MT) + 32 * mi + i;
        a_ok[mi] = m < p.M;
        const int mm = a_ok[mi] ? m : 0;
        const int b = mm / (p.Ho * p.Wo);
        const int rem = mm - b * (p.Ho * p.Wo);
        const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
        a_ih0[mi] = oh * p.S - p.pt;
        a_iw0[mi] = ow * p.S - p.pl;
        a_off[mi] = (unsigned)(((((long long)b * pH + a_ih0[mi]) * pW + a_iw0[mi]) * (long long)pCin + h * 4) * 4);
    }
    // ---- B fragments: lane (i, h) reads column n0 + 32*ni + i of slot row slab*8 + 2c + h
    const buffer_rsrc wbuf = make_buffer(p.wp, p.wp_bytes);
    const unsigned bw_off = (unsigned)((h * p.CoutPad + tn * (32 * NT) + i) * 16);
    const unsigned bw_slab = (unsigned)(8 * p.CoutPad * 16);
    const unsigned bw_group = (unsigned)(2 * p.CoutPad * 16);

    // (cc, kh, kw) of the next slab to load; K order = 32-channel chunk, kh, kw (pack_weights)
    const int taps = pKS * pKS;
    int cc = s0 / taps;
    int kh = (s0 - cc * taps) / pKS;
    int kw = (s0 - cc * taps) - kh * pKS;
    int t_load = s0;

    f32x4 fa[DEPTH][MT][4], fb[DEPTH][NT][4];
    // with_b = false: this slab's B fragments are already there (prefetched)
    auto load_stage = [&](int d, bool with_b) {
        const bool live = t_load < s1;                          // past the wave's range: every load is forced out of range (zeros, no traffic)
        const bool live_a = live && !(ablate & 1), live_b = live && !(ablate & 2);
        const unsigned tap_off = (unsigned)(((kh * pW + kw) * pCin + cc * 32) * 4);
        unsigned ao[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const bool ok = live_a && a_ok[mi] && (unsigned)(a_ih0[mi] + kh) < (unsigned)pH && (unsigned)(a_iw0[mi] + kw) < (unsigned)pW;
            ao[mi] = ok ? a_off[mi] + tap_off : kOobBase;
        }
        const unsigned bo = live_b ? bw_off + (unsigned)t_load * bw_slab : kOobBase;
#pragma unroll
        for (int c = 0; c < 4; ++c) {                           // k-group order = consumption order
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) fa[d][mi][c] = buffer_load4(xbuf, ao[mi] + 32u * c);
            if (with_b) {
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) fb[d][ni][c] = buffer_load4(wbuf, live_b ? bo + c * bw_group + 512u * ni : kOobBase);
            }
        }
        ++t_load;
        if (++kw == pKS) { kw = 0; if (++kh == pKS) { kh = 0; ++cc; } }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    // 32 x 32 wave tile in the spread form: a second accumulator takes the odd q-steps, so that consecutive MFMAs are independent and
    // the next slab's loads can sit between them (the two chains are added once, behind the K loop: a different -- fixed --
    // summation order than the one-chain form)
    constexpr bool TWIN = SPREAD && MT * NT == 1 && DEPTH == 2;
    f32x16 acc_odd;
    if (TWIN) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_odd[r] = 0.f;
    }

    auto mfma_stage = [&](int d) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = mfma_32x32x2(fa[d][mi][c][q], fb[d][ni][c][q], acc[mi][ni]);
    };

    stamp(1);                                                   // index arithmetic done
    if (CHAIN && use_pf) {                                      // first slab: B came in ahead of the barrier, A now
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int c = 0; c < 4; ++c) fb[0][ni][c] = pf.b[ni][c];
        load_stage(0, false);
#pragma unroll
        for (int d = 1; d < DEPTH - 1; ++d) load_stage(d, true);
    } else {
#pragma unroll
        for (int d = 0; d < DEPTH - 1; ++d) load_stage(d, true);
    }
    // epilogue constants of this thread's output pieces, fetched now: the block that finishes the tile must not start its
    // epilogue with a round trip to L2 (2.5 -> 0.5 us on the critical path of every split layer)
    const bool vec_ok = (p.Cout & 3) == 0;                      // rows of the output are whole float4 pieces
    auto piece_mn = [&](int j, int& m, int& n) {                // output row / first column of this thread's j-th piece
        const int f = tid + T * j, sub = f >> 8, q = f & 255;
        const int mi = sub / NT, ni = sub - mi * NT;
        m = tm * (32 * MT) + 32 * mi + (q >> 3);
        n = tn * (32 * NT) + 32 * ni + 4 * (q & 7);
    };
    const bool finisher = PIECES >= T || tid < PIECES;          // (uniform per wave: PIECES is a multiple of 64)
    f32x4 ep_bias[NF4];
#pragma unroll
    for (int j = 0; j < NF4; ++j) {
        int m, n;
        piece_mn(j, m, n);
#pragma unroll
        for (int e = 0; e < 4; ++e) ep_bias[j][e] = (finisher && n + e < p.Cout) ? p.bias[n + e] : 0.f;
    }
    // (logical block 0, its first loads in flight) the tile words get this launch's nonce long before the first arrival
    if ((p.gsplits > 1 || p.tail_tiles > 0) && Lphys == 0)
        for (int w = tid; w < (p.tail_tiles > 0 ? p.tail_tiles : tiles); w += T) ticket_prepare_word(p.tickets + w, p.nonce);
    if constexpr (SPREAD && DEPTH == 2) {
        // Spread schedule (wave tiles with two or four accumulators).  The burst form below issues the ~100 instructions that
        // address and request the next slab in one piece, fenced in front of the slab's 64 (32) MFMAs: for ~450 cycles per slab
        // the matrix pipe of a one-wave-per-SIMD block has nothing to do (64 x 64 tiles: 0.70 of the MFMA peak at B = 4).
        // Consecutive MFMAs of a q-step go to DIFFERENT accumulators, so an instruction placed between them costs next to nothing
        // (it issues while the pipe is busy with the MFMA just sent): here the address arithmetic of slab t + 1 follows the first
        // q-step of slab t and its MT + NT loads of k-group c follow one per q-step of group c.  Same MFMA order per accumulator:
        // bit-identical results.  (With ONE accumulator every MFMA depends on its predecessor and anything between two of them
        // costs a full dependent-issue bubble: the 32 x 32 tiles keep the burst.)
        for (int t = s0; t < s1; t += 2) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                if (t + d < s1) {                               // wave-uniform
                    const int dn = d ^ 1;
                    unsigned ao[MT], bo = 0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            sched_fence();
                            if (TWIN && (q & 1)) acc_odd = mfma_32x32x2(fa[d][0][c][q], fb[d][0][c][q], acc_odd);
                            else {
#pragma unroll
                                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                                    for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = mfma_32x32x2(fa[d][mi][c][q], fb[d][ni][c][q], acc[mi][ni]);
                            }
                            sched_fence();
                            if (c == 0 && q == 0) {             // addresses of slab t + d + 1 (dead once past s1: out of range, no traffic)
                                const bool live = t_load < s1;
                                const unsigned tap_off = (unsigned)(((kh * pW + kw) * pCin + cc * 32) * 4);
#pragma unroll
                                for (int mi = 0; mi < MT; ++mi) {
                                    const bool ok = live && a_ok[mi] && (unsigned)(a_ih0[mi] + kh) < (unsigned)pH && (unsigned)(a_iw0[mi] + kw) < (unsigned)pW;
                                    ao[mi] = ok ? a_off[mi] + tap_off : kOobBase;
                                }
                                bo = live ? bw_off + (unsigned)t_load * bw_slab : kOobBase;
                                ++t_load;
                                if (++kw == pKS) { kw = 0; if (++kh == pKS) { kh = 0; ++cc; } }
                            }
                            if (TWIN) {                         // (one A and one B load per k-group: behind the even q-steps)
                                if (q == 0) fa[dn][0][c] = buffer_load4(xbuf, ao[0] + 32u * c);
                                if (q == 2) fb[dn][0][c] = buffer_load4(wbuf, bo == kOobBase ? kOobBase : bo + c * bw_group);
                            } else if (q < MT) fa[dn][q < MT ? q : 0][c] = buffer_load4(xbuf, ao[q < MT ? q : 0] + 32u * c);
                            else if (q - MT < NT) fb[dn][q - MT < NT ? q - MT : 0][c] = buffer_load4(wbuf, bo == kOobBase ? kOobBase : bo + c * bw_group + 512u * (q - MT));
                        }
                    }
                }
            }
        }
    } else if (WAVES == 8 && DEPTH == 2 && MT * NT == 1 && !CHAIN && p.pingpong) {     // (kept for the 32 x 32 tiles only: beside the free-running loop it made the 64 x 64 / 8-wave form spill)
        // Two waves per SIMD (w and w + 4: tools/ubench/wave_simd_map.hip), made to ALTERNATE.  A wave's slab is ~300 cycles of
        // address arithmetic + load issue, during which its dependent MFMA chain stands still, then 16 x 64 cycles of MFMAs.  Left
        // alone, the two waves of a SIMD fall into step -- they share the matrix pipe, so they finish their MFMAs together and
        // then both issue loads while the pipe idles: eight waves measured exactly like four (B = 1 conv2 20.1 vs 19.5 us).  With a
        // block barrier between the half-steps one wave's load issue always sits under the other's MFMAs:
        //     waves 0-3:  L(t+1) | M(t)   | L(t+2) | M(t+1) ...
        //     waves 4-7:  M(t)   | L(t+1) | M(t+1) | L(t+2) ...
        // Every wave runs the same number of half-steps (K ranges differ by at most one slab; a wave past its range issues
        // out-of-range loads -- no traffic).  Same per-wave fma chains: the same values as the free-running loop.
        const int nmax = (b1 - b0 + WAVES - 1) / WAVES;
        // ONE loop body for both halves -- { L(t+1); barrier; M(t); barrier } -- the second half enters it one barrier late (and
        // the first half pays that barrier back behind the loop).  (A first version branched on the half inside the loop: the
        // compiler then kept the accumulators in VGPRs and copied all 16 across every join -- each copy waits for the MFMA that
        // produced its source: conv2 at B = 1 24.7 us instead of 19.8.)  A stage past the wave's K range multiplies zeros.
        if (wave >= WAVES / 2) block_barrier();
        for (int t = 0; t < nmax; t += 2) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                sched_fence();
                load_stage(d ^ 1, true);
                sched_fence();
                block_barrier();
                sched_fence();
                mfma_stage(d);
                sched_fence();
                block_barrier();
            }
        }
        if (wave < WAVES / 2) block_barrier();
    } else {
        for (int t = s0; t < s1; t += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (t + d < s1) {                                   // wave-uniform
                    sched_fence();
                    load_stage((d + DEPTH - 1) % DEPTH, true);      // slab t + d + DEPTH - 1 (dead loads once past s1)
                    sched_fence();
                    if (!(ablate & 4)) mfma_stage(d);
                }
            }
        }
    }

    if (TWIN) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += acc_odd[r];
    }
    stamp(2);                                                   // K loop done (MFMA results may still be in the pipe)
    // ---- the WAVES partial tiles meet in LDS, lane-linear (conflict-free), and are added in wave order ----------
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * COMBOS + (mi * NT + ni) * 16 + r) * 64 + lane] = acc[mi][ni][r];
    __syncthreads();
    f32x4 v[NF4];
#pragma unroll
    for (int j = 0; j < NF4; ++j) {
        const int f = finisher ? tid + T * j : 0, sub = f >> 8, q = f & 255, row = q >> 3;
        const int at = (sub * 16 + (row & 3) + 4 * (row >> 3)) * 64 + 4 * (q & 7) + 32 * ((row >> 2) & 1);
        f32x4 s = lds_read4(red + at);
#pragma unroll
        for (int w = 1; w < WAVES; ++w) s += lds_read4(red + w * COMBOS * 64 + at);
        v[j] = s;
    }
    __syncthreads();                                            // (CHAIN: the next work item of this block reuses red)

    stamp(3);                                                   // cross-wave sum done
    if (gs > 1 && !(ablate & 8)) {
        // this block's tile partial: one 16-B coherent store per piece, thread-linear (coalesced)
        const buffer_rsrc pbuf = make_buffer(p.partial, p.partial_bytes);
        const unsigned tile_base = (unsigned)slot0 * PIECES * 16u;
        const unsigned mine = finisher ? tile_base + (unsigned)g * PIECES * 16u + (unsigned)tid * 16u : kOobBase;
#pragma unroll
        for (int j = 0; j < NF4; ++j) coherent_store4(pbuf, finisher ? mine + j * T * 16u : kOobBase, v[j]);
        block_ticket_publish();
        stamp(4);                                               // partial stores complete (device scope)
        const bool last_block = block_ticket_take(p.tickets + ticket_word, p.nonce, (unsigned)gs, (unsigned)g, flag);
        stamp(5);                                               // ticket taken
        if (!last_block) return;
#pragma unroll
        for (int j = 0; j < NF4; ++j) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // all partials of up to 8 splits in flight at once (a coherent load is a round trip to the memory side: one at a
        // time they cost more than the K loop); splits beyond gsplits read out of range = zeros.  Split order: the same
        // sum whichever block arrives last.
        constexpr int kBatch = 8;
        const unsigned split_stride = PIECES * 16u;
        for (int sb = 0; sb < gs; sb += kBatch) {
            f32x4 t[kBatch][NF4];
#pragma unroll
            for (int u = 0; u < kBatch; ++u)
#pragma unroll
                for (int j = 0; j < NF4; ++j)
                    t[u][j] = coherent_load4(pbuf, (finisher && sb + u < gs) ? tile_base + (unsigned)(sb + u) * split_stride + (unsigned)tid * 16u + j * T * 16u
                                                                                    : kOobBase);
#pragma unroll
            for (int u = 0; u < kBatch; ++u)
#pragma unroll
                for (int j = 0; j < NF4; ++j) v[j] += t[u][j];
        }
    }

    stamp(6);                                                   // (last block) all partials summed
    // ---- epilogue: bias, ReLU, folded BN; one output row piece (m, n ... n + 3) per v[j] ------------------------------
    const buffer_rsrc obuf = make_buffer(p.out, (unsigned)((unsigned long long)p.M * p.Cout * 4ull));
#pragma unroll
    for (int j = 0; j < NF4; ++j) {
        int m, n;
        piece_mn(j, m, n);
        if (!finisher || m >= p.M || n >= p.Cout) continue;
        f32x4 o = v[j] + ep_bias[j];
        if (p.relu) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
        if (p.bn_scale) {                                       // (inference batch-norm after the ReLU: encoder.py:51-52)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < p.Cout) o[e] = o[e] * p.bn_scale[n + e] + p.bn_shift[n + e];
        }
        if (vec_ok) {                                           // (n % 4 == 0 and Cout % 4 == 0: the piece lies inside the row, 16-B aligned)
            if (CHAIN) coherent_store4(obuf, (unsigned)(((long long)m * p.Cout + n) * 4), o);
            else *reinterpret_cast<f32x4*>(p.out + (long long)m * p.Cout + n) = o;
        } else {
            const u32x4 ob = __builtin_bit_cast(u32x4, o);       // (whole vector: a bit_cast of ONE element indexed by the loop variable was
#pragma unroll                                                    //  seen to pick element 0 under clang -O2 on the host)
            for (int e = 0; e < 4; ++e)
                if (n + e < p.Cout) {
                    if (CHAIN) coherent_store1(obuf, (unsigned)(((long long)m * p.Cout + n + e) * 4), ob[e]);
                    else p.out[(long long)m * p.Cout + n + e] = o[e];
                }
        }
    }
    stamp(7);
}

// (64 x 64 wave tiles: two blocks per CU asked for -- 209 instead of 288 registers, no spills -- so that the second round of a
// 257 ... 512-tile layer overlaps the first: conv2 at B = 7 / 8 118.6 / 120.6 -> 112.9 / 114.6 us; the smaller tiles lose 8 % under the
// same bound and keep the whole register file)
template <int MT, int NT, int WAVES, int DEPTH, int TAG = 0, bool SPREAD = false>
__global__ __launch_bounds__(64 * WAVES, (MT * NT == 4 && WAVES == 4 && DEPTH == 2 ? 2 : 1)) void conv_wavek_f32_kernel(const ConvWaveKArgs p) {
    AAE_DYN_SMEM(smem_raw);
    float* red = reinterpret_cast<float*>(smem_raw);           // [WAVES][COMBOS][64]
    int* flag = reinterpret_cast<int*>(red + WAVES * MT * NT * 16 * 64);
    WaveKPrefetch none;
    conv_wavek_block<MT, NT, WAVES, DEPTH, false, SPREAD>(p, (int)blockIdx.x, (int)gridDim.x, red, flag, none, false);
}

// ---- the same layer of SEVERAL objects in one launch (multi_launch.h): blocks [first[o], first[o] + nblk[o]) run object o's layer --
// its activations, its weights, its partial buffer, its ticket words -- exactly as its own launch would (logical block index and
// block count of that object), so a frame with eight classes fills the chip eight times over behind one launch ramp.
struct ConvWaveKMultiArgs {
    MultiRange range;                      // (padded to multiples of 8 blocks per object: xcd_remap counts from the object's first block)
    int nblk[kMultiMax];                   // real block count of each object's layer
    // > 0: every object has this many (padded) blocks and the number of objects is a multiple of 8 -- then ALL blocks of an object run on
    // ONE XCD (physical block p runs on XCD p % 8: object = p % 8 + 8 * ((p / 8) / xcd_affine), block (p / 8) % xcd_affine of it), so that an
    // object's weights and activations live in one 4 MB L2 instead of being streamed through all eight
    int xcd_affine;
    ConvWaveKArgs item[kMultiMax];
};
template <int MT, int NT, int WAVES, int DEPTH, int TAG = 0, bool SPREAD = false>
__global__ __launch_bounds__(64 * WAVES, (MT * NT == 4 && WAVES == 4 && DEPTH == 2 ? 2 : 1)) void conv_wavek_multi_kernel(const ConvWaveKMultiArgs m) {
    AAE_DYN_SMEM(smem_raw);
    float* red = reinterpret_cast<float*>(smem_raw);
    int* flag = reinterpret_cast<int*>(red + WAVES * MT * NT * 16 * 64);
    int o, L;
    if (m.xcd_affine > 0) {
        const int p = (int)blockIdx.x, idx = p >> 3;
        o = (p & 7) + 8 * (idx / m.xcd_affine);
        L = idx % m.xcd_affine;
    } else {
        o = multi_find(m.range, (int)blockIdx.x);
        L = (int)blockIdx.x - m.range.first[o];
    }
    if (L >= m.nblk[o]) return;                                  // (padding block)
    WaveKPrefetch none;
    conv_wavek_block<MT, NT, WAVES, DEPTH, false, SPREAD>(m.item[o], L, m.nblk[o], red, flag, none, false);
}

}  // namespace aae
