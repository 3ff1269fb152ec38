// The launch planner of the encoder: weight packing, split-K choice, the wave-split-K plan per layer (tile-count thresholds for the
// per-detection batches, estimated time for mid batches, the tail cut), the ticket-word layout and the workspace layout.
// Replaces nothing of the reference by itself -- it decides HOW the kernels of encoder.py:41-68's layers are launched.
// Part of aae_hip_impl.h.
#pragma once

namespace aae_host {

// ------------------------------------------------------------------ helpers
// Kernel records (label, algorithmic flops) of a forward call are collected in a call-local list and
// published to the handle when the call returns, so concurrent forwards on one handle (distinct streams and
// workspaces) never touch shared state while they launch.
static thread_local int* t_x3h_flag = nullptr;        // range flag of the f32x3h forward this thread is launching
static thread_local int t_x3h_last_slot = -1;         // its slot (-1: the last forward of this thread ran exact fp32)
static thread_local std::vector<KernelRecord>* t_records = nullptr;
static void note_kernel(KernelRecord r) {
    if (t_records) t_records->push_back(std::move(r));
}
struct RecordScope {
    aae_encoder* owner;
    std::vector<KernelRecord> local;
    explicit RecordScope(aae_encoder* e) : owner(e) { t_records = &local; }
    ~RecordScope() {
        t_records = nullptr;
        std::lock_guard<std::mutex> lk(owner->rec_mu);
        owner->records.swap(local);
    }
};

static int upload(aae_encoder* enc, const float* host, size_t count, float** dev) {
    void* p = nullptr;
    AAE_HIP_TRY(hipMalloc(&p, count * sizeof(float)));
    enc->allocations.push_back(p);
    AAE_HIP_TRY(hipMemcpy(p, host, count * sizeof(float), hipMemcpyHostToDevice));
    *dev = static_cast<float*>(p);
    return AAE_OK;
}

// HWIO / [F][J] kernel -> [K/4][CoutPad][4]; k = (kh*KS + kw)*Cin + ci is already
// the row index of the HWIO array flattened to [K][Cout].
// The kernel walks K as (32-channel chunk, kh, kw, channel-in-chunk): packed row
// k' = (cc*taps + tap)*32 + j holds HWIO row k = tap*Cin + cc*32 + j.
static std::vector<float> pack_weights(const float* w, int taps, int Cin, int Cout, int CoutPad) {
    const long long K = (long long)taps * Cin;
    std::vector<float> out((size_t)K * CoutPad, 0.f);
    for (int cc = 0; cc < Cin / 32; ++cc)
        for (int tap = 0; tap < taps; ++tap)
            for (int j = 0; j < 32; ++j) {
                const long long k = (long long)tap * Cin + cc * 32 + j;
                const long long kp = ((long long)cc * taps + tap) * 32 + j;
                for (int n = 0; n < Cout; ++n)
                    out[((size_t)(kp >> 2) * CoutPad + n) * 4 + (kp & 3)] = w[(size_t)k * Cout + n];
            }
    return out;
}

// Winograd-domain weights of ONE polyphase component (conv_winograd_f32.h) of a 5 x 5 stride-2 layer: the taps kh = 2 k + (eh ? 0 : 1)
// (3 taps on the odd rows, 2 on the even ones; the same along the columns) are transformed with G g G^T in float64, rounded once, and
// laid out as the kernel's B fragments: [32-column block][8-channel group][point a PB + b][K half][32 columns][4 channels], a = point
// index along the split dimension A (rows, or columns when swap), b along the other one.
static std::vector<float> pack_weights_winograd(const float* w_hwio, int KS, int Cin, int Cout, int eh, int ew, bool swap) {
    static const double G3[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    static const double G2[3][2] = {{1, 0}, {1, 1}, {0, 1}};
    const int th = eh ? 3 : 2, tw = ew ? 3 : 2;                      // taps along rows / columns
    const int tA = swap ? tw : th, tB = swap ? th : tw, PA = tA + 1, PB = tB + 1, NP = PA * PB;
    auto gA = [&](int a, int k) { return tA == 3 ? G3[a][k] : G2[a][k]; };
    auto gB = [&](int b, int k) { return tB == 3 ? G3[b][k] : G2[b][k]; };
    std::vector<float> out((size_t)NP * Cin * Cout);
    for (int c = 0; c < Cin; ++c)
        for (int n = 0; n < Cout; ++n) {
            double g[3][3], t[4][3];                                  // taps [along A][along B], G_A g
            for (int ka = 0; ka < tA; ++ka)
                for (int kb = 0; kb < tB; ++kb) {
                    const int kr = swap ? kb : ka, kc = swap ? ka : kb;              // tap index along rows / columns
                    const int kh = 2 * kr + (eh ? 0 : 1), kw = 2 * kc + (ew ? 0 : 1);
                    g[ka][kb] = (double)w_hwio[((size_t)(kh * KS + kw) * Cin + c) * Cout + n];
                }
            for (int a = 0; a < PA; ++a)
                for (int kb = 0; kb < tB; ++kb) {
                    double acc = 0.0;
                    for (int ka = 0; ka < tA; ++ka) acc += gA(a, ka) * g[ka][kb];
                    t[a][kb] = acc;
                }
            const int n32 = n >> 5, nn = n & 31, kg = c >> 3, hh = (c >> 2) & 1, q = c & 3;
            float* dst = &out[((((size_t)n32 * (Cin / 8) + kg) * NP) * 64 + hh * 32 + nn) * 4 + q];
            for (int a = 0; a < PA; ++a)
                for (int b = 0; b < PB; ++b) {
                    double acc = 0.0;
                    for (int kb = 0; kb < tB; ++kb) acc += t[a][kb] * gB(b, kb);
                    dst[(size_t)(a * PB + b) * 64 * 4] = (float)acc;
                }
        }
    return out;
}

// ... which layers can take it: 5 x 5, stride 2, 'SAME' with one row / column of padding in front (even input size), 32 | Cin, 64 | Cout,
// and an output of whole 16 x 16-pixel regions (geometry 0) or exactly 8 x 8 (geometry 1: four images per block)
static int winograd_geometry(const Layer& L) {
    if (L.index < 1 || L.kind != KIND_IGEMM) return -1;
    if (L.KS != 5 || L.S != 2 || L.pt != 1 || L.pl != 1 || (L.H & 1) || (L.W & 1) || L.H != 2 * L.Ho || L.W != 2 * L.Wo) return -1;
    if (L.Cin % 32 != 0 || L.Cout % 64 != 0) return -1;
    if (L.Ho % 16 == 0 && L.Wo % 16 == 0) return 0;
    if (L.Ho == 8 && L.Wo == 8) return 1;
    return -1;
}

// f32x3h weights: w*2^shift split into (hi, lo) halves, packed per K-slab as
// [8 slots][CoutPad][8 halves] with slot = plane*4 + kgroup8 (kernel K order, see pack_weights).
static std::vector<unsigned short> pack_weights_x3h(const float* w, int taps, int Cin, int Cout, int CoutPad, int* shift_out) {
    const long long K = (long long)taps * Cin;
    float maxw = 0.f;
    for (long long i = 0; i < K * Cout; ++i) maxw = fmaxf(maxw, fabsf(w[i]));
    int e = 0;
    if (maxw > 0.f) (void)frexpf(maxw, &e);                  // maxw = m * 2^e, m in [0.5, 1)
    const int shift = 10 - e;                                // max |w| * 2^shift in [512, 1024)
    *shift_out = shift;
    const long long slabs = K / 32;
    std::vector<unsigned short> out((size_t)slabs * 8 * CoutPad * 8, 0);
    for (int cc = 0; cc < Cin / 32; ++cc)
        for (int tap = 0; tap < taps; ++tap) {
            const long long slab = (long long)cc * taps + tap;
            for (int j = 0; j < 32; ++j) {
                const long long k = (long long)tap * Cin + cc * 32 + j;
                const int kg = j >> 3, el = j & 7;
                for (int n = 0; n < Cout; ++n) {
                    const float v = ldexpf(w[(size_t)k * Cout + n], shift);
                    const _Float16 h = (_Float16)v;
                    const _Float16 l = (_Float16)(v - (float)h);
                    unsigned short hb, lb;
                    memcpy(&hb, &h, 2);
                    memcpy(&lb, &l, 2);
                    out[(((size_t)slab * 8 + kg) * CoutPad + n) * 8 + el] = hb;
                    out[(((size_t)slab * 8 + 4 + kg) * CoutPad + n) * 8 + el] = lb;
                }
            }
        }
    return out;
}

// split-K partial sums -> layer output: few splits over a large tile take the barrier-free float4 kernel
static void launch_splitk_reduce(const aae::SplitKReduceArgs& r, hipStream_t stream, bool allow_small = true) {
    if (allow_small && r.splits <= aae::kReduceGroups && r.MN % 4 == 0 && r.Cout % 4 == 0 && r.MN >= 16384) {
        const long long chunks = (r.MN / 4 + 255) / 256;           // 1024-element segments
        if (r.splits == 2) AAE_LAUNCH((aae::splitk_reduce_small_kernel<2, 4>), dim3((unsigned)((chunks + 3) / 4)), dim3(256), 0, stream, r);
        else if (r.splits == 3) AAE_LAUNCH((aae::splitk_reduce_small_kernel<3, 4>), dim3((unsigned)((chunks + 3) / 4)), dim3(256), 0, stream, r);
        else if (r.splits == 4) AAE_LAUNCH((aae::splitk_reduce_small_kernel<4, 4>), dim3((unsigned)((chunks + 3) / 4)), dim3(256), 0, stream, r);
        else AAE_LAUNCH((aae::splitk_reduce_small_kernel<0, 1>), dim3((unsigned)chunks), dim3(256), 0, stream, r);
        return;
    }
    long long blocks = (r.MN + 63) / 64;
    if (blocks > 4096) blocks = 4096;
    AAE_LAUNCH((aae::splitk_reduce_kernel), dim3((unsigned)blocks), dim3(512), aae::kReduceGroups * 64 * (int)sizeof(float), stream, r);
}

static bool first_layer_instantiated(int KS, int C) { return KS == 5 && (C == 3 || C == 1); }

static void plan_first_layer(Layer& L) {
    L.rowlen = ((L.Wo - 1) * L.S + L.KS) * L.Cin;
    int max_out_rows = 127 / L.Wo + 2;
    if (max_out_rows > L.Ho) max_out_rows = L.Ho;
    const int max_in_rows = (max_out_rows - 1) * L.S + L.KS;
    int widest = L.rowlen;
    L.rowlen4 = L.lead4 = 0;
    if ((L.W * L.Cin) % 4 == 0) {          // uint8 rows can be staged as aligned dwords
        L.lead4 = (4 - (L.pl * L.Cin) % 4) % 4;
        L.rowlen4 = (L.rowlen + L.lead4 + 3) / 4 * 4;
        widest = L.rowlen4;
    }
    L.first_smem = (256 + max_in_rows * widest) * (int)sizeof(float);
    // the staging units keep (row, offset from the first staged row) packed in 12 + 20 bits
    L.first_packable = (long long)(max_in_rows + 1) * L.W * L.Cin < (1ll << 20);
}

// split-K factor: aim for >= ~512 resident-able blocks without splitting finer than one slab
static void choose_splits(const aae_encoder* enc, int base_blocks, int slabs, int* splits, int* per_split) {
    int s = 1;
    if (base_blocks < enc->splitk_min_base_blocks) {
        s = ceil_div(enc->splitk_target_blocks, base_blocks);
        if (s < 1) s = 1;
        if (s > slabs) s = slabs;
    }
    // the kernels give split i the slabs [i*slabs/s, (i+1)*slabs/s): every requested split exists and the sizes
    // differ by at most one slab (uniform ceil-sized splits left e.g. 400 of 512 requested blocks at B=1)
    *per_split = ceil_div(slabs, s);
    *splits = s;
}

// block_ticket_arrive() nonces: unique per launch within the process, never 0
static unsigned next_nonce() {
    static std::atomic<unsigned> counter{1};
    unsigned n = counter.fetch_add(1, std::memory_order_relaxed);
    while (n == 0) n = counter.fetch_add(1, std::memory_order_relaxed);
    return n;
}

// Ticket words at the front of every encoder workspace: a range of single words per layer (conv layers 0..7, then
// the dense layer) for the wave-split-K tiles, then one two-level slot per 128-column tile of the dense GEMV.  Every
// ticketed launch has its own words, so the first kernel of a forward can prepare all of them (TicketPrep).
constexpr int kChainMaxBlocks = 1024;       // upper bound of the persistent per-detection launch's grid (option detect_chain_blocks)
constexpr int kGemvTicketSlots = 8;
constexpr int kLayerTicketWords = 256;     // per layer: one word per output tile of a split layer (split => at most 128 tiles)
constexpr size_t kConvTicketBytes = (size_t)(AAE_MAX_LAYERS + 1) * kLayerTicketWords * 8;
constexpr size_t kGemvTicketBytes = (size_t)kGemvTicketSlots * aae::kTicketSlotWords * 8;
constexpr size_t kTicketBytes = kConvTicketBytes + kGemvTicketBytes + (size_t)aae::kGridBarrierWords * 8;    // ... then the grid barrier of the persistent per-detection launch

constexpr int kWaveKTileCap = 8192;        // 64 x 64 output tiles the wave-split-K kernel is ever asked to walk (option wavek_max_tiles is clamped to it)

// Launch plan of the wave-split-K igemm (conv_wavek_f32.h) for a layer at M rows, or use == false.
struct WaveKPlan {
    bool use = false;
    int MT = 2, NT = 2, waves = 4, depth = 3;
    int num_mt = 0, num_nt = 0, gsplits = 1;
    int tail_tiles = 0, tail_g = 1;        // the last tail_tiles tiles cut tail_g ways in K (gsplits == 1 then): conv_wavek_f32.h
    size_t partial_bytes = 0;
    int blocks() const { return (num_mt * num_nt - tail_tiles) * gsplits + tail_tiles * tail_g; }
};

// Does a forward of batch B run in f32x3h?  precision 1: always.  precision 2 ("where it is faster"): only when the first
// implicit-GEMM layer has at least x3h_min_tiles 64 x 64 output tiles -- below that the layers do not fill the chip, the
// exact-fp32 wave-split-K path with its in-launch reductions is the faster one (B = 1: 82 us against 156 us for the
// split-precision split-K igemm + reduce launches; break-even at B = 4 of the default net) and it is at least as accurate.
static bool runs_split(const aae_encoder* enc, int B) {
    if (enc->precision == 1) return true;
    if (enc->precision != 2 || enc->layers.size() < 2) return false;
    const Layer& L = enc->layers[1];
    const long long M = (long long)B * L.Ho * L.Wo;
    return ((M + 63) / 64) * (L.CoutPad / 64) >= enc->x3h_min_tiles;
}

// blocks the chip runs at once, one per compute unit: the planner's round size
static int wavek_round_blocks(const aae_encoder* enc) {
    if (enc->wavek_target_blocks > 0) return enc->wavek_target_blocks;
    return enc->cu_count > 0 ? std::min(enc->cu_count, 2 * kLayerTicketWords) : 256;
}

// K splits of a wave-split-K layer of `tiles` output tiles: one block per CU, never a second round of blocks; every wave keeps
// at least two slabs; one ticket word per tile
static int wavek_gsplits(const aae_encoder* enc, int tiles, int slabs, int waves, int boost = 1) {
    int g = wavek_round_blocks(enc) * boost / tiles;
    const int gmax = slabs / (2 * waves);
    if (g > gmax) g = gmax;
    if (g > (int)aae::kTicketSingleLevelMax) g = (int)aae::kTicketSingleLevelMax;
    if (g < 1) g = 1;
    if (tiles > kLayerTicketWords) g = 1;
    return g;
}

// ---- planner by cost (B >= 5) ------------------------------------------------------------------------------------------
// Which implicit-GEMM family, which wave tile?  Both families lose time to block-count quantisation, in different places:
// the 128 x 128 (x 256) tiles of conv_igemm_f32.h come in few large blocks (conv2 at B = 40: 640 blocks = 2.5 "rounds" of the
// chip, paid as 3 or 4), the wave-split-K kernel's tiles are 4 ... 16 times smaller but move 2 ... 4 times the operand bytes per
// MFMA.  Tile-count thresholds cannot see that; an estimate of each candidate's time can:
//     rounds = ceil(blocks / CUs);   t = rounds * (slabs one wave walks [+ pipeline fill]) * (MFMA time of its tile per slab) / efficiency + fixed
// with the efficiencies and fixed costs fitted to per-layer HIP-event times of every candidate at B = 5 ... 128 on MI355X
// (tools/sweep_planner.py -> profiles/r11/planner_sweep_*.jsonl: rms error 3-7 %, and the candidate it picks is the measured
// best at 40 of 42 (layer, batch) points): wave-split-K 32 x 32 0.71, 64 x 32 0.72, 64 x 64 0.88 (+ 5 us), each with 4 slabs
// of fill per block, + 3 us when K is split across blocks; 128 x 128 igemm 0.86 with 2 slabs of fill, + 5 us, + 10 us for the
// split-K reduce launch; its 128 x 256 form sits two blocks to a CU and is paid in rounds of two (0.90).
// A 32 x 32 x 2 fp32 MFMA occupies its pipe for 64 cycles: 16 per slab and 32 x 32 sub-tile = 0.4267 us at 2.4 GHz.
constexpr double kSlabUs = 16.0 * 64.0 / 2400.0;

static double wavek_cost_us(const aae_encoder* enc, int tiles, int g, int slabs, int mt, int nt) {
    const double eff_burst[3] = {0.71, enc->wavek_eff64x32_pct / 100.0, 0.88}, eff_spread[3] = {0.71, enc->wavek_eff64x32_pct / 100.0, 0.96};
    static const double fixed[3] = {0.0, 0.0, 5.0};
    const double* eff = (enc->wavek_spread & 1) ? eff_spread : eff_burst;          // (64 x 64 tiles with the spread schedule: +9 % measured, round 4)
    const int shape = mt == 1 ? 0 : (nt == 1 ? 1 : 2);
    const int cus = wavek_round_blocks(enc);
    return (double)ceil_div(tiles * g, cus) * (ceil_div(slabs, 4 * g) + 4) * (mt * nt) * kSlabUs / eff[shape] + fixed[shape] + (g > 1 ? 3.0 : 0.0);
}

// Tail split.  `tiles` whole tiles (no K split) leave the last round of blocks partly empty whenever tiles is not a multiple of
// what the chip runs at once: 576 tiles of 64 x 64 at B = 9 cost three tile times on 256 CUs, B = 12 is slower than B = 16.  The
// tiles beyond the last full round (a round = one tile per CU) can be cut g ways in K instead -- smaller blocks that fill every CU:
//     t = [full rounds * (slabs / 4 + fill) + ceil(tail * g / CUs) * (slabs / (4 g) + fill)] * tile time per slab / efficiency
// Returns the best g (1 = leave the layer alone) and its estimate.
static int wavek_tail_split(const aae_encoder* enc, int tiles, int slabs, int waves, int mt, int nt, int* tail_tiles, double* cost) {
    const double eff_burst[3] = {0.71, enc->wavek_eff64x32_pct / 100.0, 0.88}, eff_spread[3] = {0.71, enc->wavek_eff64x32_pct / 100.0, 0.96};
    static const double fixed[3] = {0.0, 0.0, 5.0};
    const double* eff = (enc->wavek_spread & 1) ? eff_spread : eff_burst;
    const int shape = mt == 1 ? 0 : (nt == 1 ? 1 : 2);
    const int cus = wavek_round_blocks(enc);
    const int tail = tiles % cus, full_rounds = tiles / cus;
    *tail_tiles = 0;
    *cost = wavek_cost_us(enc, tiles, 1, slabs, mt, nt);
    if (!enc->wavek_tail_split || tail == 0 || full_rounds == 0 || tail > kLayerTicketWords) return 1;
    const int gmax = std::min(slabs / (2 * waves), (int)aae::kTicketSingleLevelMax);
    const double per_slab = (mt * nt) * kSlabUs / eff[shape];
    int best = 1;
    for (int g = 2; g <= gmax && g <= 16; ++g) {
        const double c = ((double)full_rounds * (ceil_div(slabs, 4) + 4) + (double)ceil_div(tail * g, cus) * (ceil_div(slabs, 4 * g) + 4)) * per_slab + fixed[shape] + 3.0;
        if (c < 0.97 * *cost) { *cost = c; best = g; }
    }
    if (best > 1) *tail_tiles = tail;
    return best;
}

static double igemm_cost_us(const aae_encoder* enc, const Layer& L, long long M) {
    const int mt = ceil_div((int)M, 128), nt = L.CoutPad / 128, slabs = (int)(L.K() / 32);
    int s, per;
    choose_splits(enc, mt * nt, slabs, &s, &per);
    const bool wide = s == 1 && enc->igemm_dma && enc->igemm_breg && enc->igemm_breg_wide && (L.index == 1 || L.index == 2) && L.CoutPad % 256 == 0 &&
                      mt * (L.CoutPad / 256) >= enc->igemm_breg_wide_min_blocks;          // (launch_igemm's 128 x 256 tiles)
    const int blocks = wide ? mt * (L.CoutPad / 256) : mt * nt * s;
    const int cus = wavek_round_blocks(enc);
    if (wide) return (double)ceil_div(blocks, 2 * cus) * 2.0 * (slabs + 2) * (8 * kSlabUs) / 0.90 + 5.0;
    return (double)ceil_div(blocks, cus) * (ceil_div(slabs, s) + 2) * (4 * kSlabUs) / 0.86 + (s > 1 ? 10.0 : 0.0) + 5.0;
}

static WaveKPlan plan_wavek_core(const aae_encoder* enc, const Layer& L, long long M, bool split);
static WaveKPlan plan_wavek(const aae_encoder* enc, const Layer& L, long long M, bool split) {
    WaveKPlan w = plan_wavek_core(enc, L, M, split);
    if (w.use && enc->wavek_force_tail_tiles > 0 && w.gsplits == 1 && w.tail_tiles == 0) {       // (tests)
        const int tiles = w.num_mt * w.num_nt, slabs = (int)(L.K() / 32);
        const int gmax = std::min(slabs / (2 * w.waves), (int)aae::kTicketSingleLevelMax);
        const int g = std::min(enc->wavek_force_tail_g, gmax);
        if (g >= 2) {
            w.tail_tiles = std::min(std::min(enc->wavek_force_tail_tiles, tiles), kLayerTicketWords);
            w.tail_g = g;
            w.partial_bytes = (size_t)w.blocks() * (w.MT * w.NT * 16) * 64 * sizeof(float);
        }
    }
    return w;
}
static WaveKPlan plan_wavek_core(const aae_encoder* enc, const Layer& L, long long M, bool split) {
    WaveKPlan w;
    if (!enc->wavek || split || L.kind != KIND_IGEMM) return w;
    const long long tiles22 = ((M + 63) / 64) * (L.CoutPad / 64);
    // (5 <= B < 256: at the headline batch every layer keeps its measured choice -- the big igemm tiles; conv4 would cost the same
    //  on 64 x 64 wave tiles, 1.542 vs 1.547 ms, profiles/r12)
    const long long batch_of = L.index >= 0 ? M / ((long long)L.Ho * L.Wo) : M;
    // (B = 3 too: its layers are 0.75 / 1.5 rounds under the thresholds of the per-detection path -- 174 -> 160 us with the estimate and
    //  the tail split; B = 2 and 4 fill their rounds exactly and measured equal / 1 % slower under the estimate: they keep the thresholds)
    const bool by_cost = enc->planner_cost_model && L.index >= 0 && (batch_of >= enc->planner_cost_min_batch || (batch_of == 3 && enc->planner_cost_batch3)) &&
                         batch_of < 256 && enc->wavek_waves != 8;
    if (tiles22 > kWaveKTileCap || (!by_cost && tiles22 > enc->wavek_max_tiles)) return w;
    const unsigned long long x_bytes = (unsigned long long)(M / (L.Ho * L.Wo)) * L.H * L.W * L.Cin * sizeof(float);
    if (x_bytes >= 0xFFFFFF00ull) return w;
    w.use = true;
    w.waves = enc->wavek_waves == 8 ? 8 : 4;
    w.depth = (enc->wavek_depth == 2 || w.waves == 8) ? 2 : 3;    // 8 waves share the register file two per SIMD: two slabs in flight each
    if (by_cost && w.waves == 4) {
        // conv layers of batches beyond the per-detection regime: cheapest of {igemm, wave-split-K 32 x 32 | 64 x 32 | 64 x 64} by estimate
        const int slabs = (int)(L.K() / 32);
        double best = igemm_cost_us(enc, L, M);
        int best_mt = 0, best_nt = 0;
        static const int shapes[3][2] = {{2, 2}, {2, 1}, {1, 1}};
        for (const auto& sh : shapes) {
            const long long tiles = ((M + 32 * sh[0] - 1) / (32 * sh[0])) * (long long)(L.CoutPad / (32 * sh[1]));
            if (tiles > (1 << 20)) continue;
            const int g = wavek_gsplits(enc, (int)tiles, slabs, w.waves, enc->wavek_g_boost);
            double c = wavek_cost_us(enc, (int)tiles, g, slabs, sh[0], sh[1]);
            if (g == 1) {                                        // whole tiles: the part beyond the last full round may be cut in K
                int tt;
                double ct;
                if (wavek_tail_split(enc, (int)tiles, slabs, w.waves, sh[0], sh[1], &tt, &ct) > 1) c = ct;
            }
            if (c < best) { best = c; best_mt = sh[0]; best_nt = sh[1]; }
        }
        if (best_mt == 0) { w.use = false; return w; }           // the 128-row igemm (+ reduce launch) is estimated faster
        w.MT = best_mt; w.NT = best_nt;
        w.num_mt = (int)((M + 32 * w.MT - 1) / (32 * w.MT));
        w.num_nt = L.CoutPad / (32 * w.NT);
        const int tiles = w.num_mt * w.num_nt;
        w.gsplits = wavek_gsplits(enc, tiles, slabs, w.waves, enc->wavek_g_boost);
        if (w.gsplits == 1) {
            double ct;
            w.tail_g = wavek_tail_split(enc, tiles, slabs, w.waves, w.MT, w.NT, &w.tail_tiles, &ct);
        }
        if (w.gsplits > 1 || w.tail_tiles > 0) w.partial_bytes = (size_t)w.blocks() * (w.MT * w.NT * 16) * 64 * sizeof(float);
        return w;
    }
    w.NT = tiles22 <= enc->wavek_narrow_max_tiles ? 1 : 2;
    w.MT = (tiles22 <= enc->wavek_tiny_max_tiles && w.waves == 4) ? 1 : 2;       // 32 x 32 wave tiles (NT = 1 then: the narrow threshold is the larger one)
    if (w.MT == 1) {
        w.NT = 1;
        if (enc->wavek_tiny_waves == 8) { w.waves = 8; w.depth = 2; }
    }
    // Balance: when the chosen tile shape needs no K split but leaves CUs idle in its last round of blocks (192 blocks of 64 x 64 on
    // 256 CUs: B = 3 conv2), a smaller wave tile can win although it moves more operand bytes per MFMA.  Blocks that share a CU share
    // its matrix pipe, so a layer costs about  ceil(tiles / CUs) * (MT * NT) / efficiency  -- efficiencies from the per-layer A/B
    // runs at B = 2 ... 4 (profiles/r09_small): 64 x 64 1.0, 64 x 32 0.97, 32 x 32 0.88.  (Layers that split K are left alone: there
    // the hand-off cost decides, and the thresholds above were set by measuring it.)
    if (enc->wavek_balance && w.waves == 4 && w.MT * w.NT > 1) {
        const int cus = wavek_round_blocks(enc);
        auto tiles_of = [&](int mt, int nt) { return ((M + 32 * mt - 1) / (32 * mt)) * (long long)(L.CoutPad / (32 * nt)); };
        auto cost_of = [&](int mt, int nt, double eff) { return (double)((tiles_of(mt, nt) + cus - 1) / cus) * (mt * nt) / eff; };
        if (tiles_of(w.MT, w.NT) >= cus / 2) {                    // (fewer tiles than that: the layer splits K)
            double best = cost_of(w.MT, w.NT, w.NT == 2 ? 1.0 : 0.97);
            if (w.NT == 2 && cost_of(2, 1, 0.97) < 0.97 * best) { best = cost_of(2, 1, 0.97); w.NT = 1; }
            if (cost_of(1, 1, 0.88) < 0.97 * best) { w.MT = 1; w.NT = 1; if (enc->wavek_tiny_waves == 8) { w.waves = 8; w.depth = 2; } }
        }
    }
    w.num_mt = (int)((M + 32 * w.MT - 1) / (32 * w.MT));
    w.num_nt = L.CoutPad / (32 * w.NT);
    const int tiles = w.num_mt * w.num_nt;
    const int slabs = (int)(L.K() / 32);
    const int g = wavek_gsplits(enc, tiles, slabs, w.waves);
    w.gsplits = g;
    if (g == 1 && L.index >= 0) {                                // (B = 3 of the default net: 384 tiles of 64 x 32 on 256 CUs)
        double ct;
        w.tail_g = wavek_tail_split(enc, tiles, slabs, w.waves, w.MT, w.NT, &w.tail_tiles, &ct);
    }
    if (g > 1 || w.tail_tiles > 0) w.partial_bytes = (size_t)w.blocks() * (w.MT * w.NT * 16) * 64 * sizeof(float);
    return w;
}

// partial rows of the GEMV form of the dense layer (B <= 4): one per 128-k chunk, then the group rows of its two-level finish
static size_t gemv_partial_bytes(const Layer& D, int B) {
    return (size_t)(ceil_div((int)D.K(), aae::kGemvChunk) + aae::kGemvGroups) * B * D.Cout * sizeof(float);
}

struct Workspace {
    std::vector<size_t> act_off;   // per conv layer
    size_t ticket_off = 0;
    size_t partial_off = 0, partial_bytes = 0;
    // B <= 4: one partial region PER split layer (conv layers, then the dense layer) for the persistent per-detection launch --
    // inside one launch no buffer may be written twice (detect_chain.h)
    std::vector<size_t> chain_partial_off;
    size_t total = 0;
};

static Workspace plan_workspace(const aae_encoder* enc, int B) {
    Workspace ws;
    size_t off = kTicketBytes;                               // ticket words first (offset 0 of the workspace)
    size_t partial = 0;
    auto need_partial = [&](const Layer& L, int M) {
        if (L.kind != KIND_IGEMM) return;
        const WaveKPlan wk = plan_wavek(enc, L, M, runs_split(enc, B));
        if (wk.use) {
            if (wk.partial_bytes > partial) partial = wk.partial_bytes;
            return;
        }
        int splits, per;
        choose_splits(enc, ceil_div(M, 128) * (L.CoutPad / 128), (int)(L.K() / 32), &splits, &per);
        if (splits > 1) {
            const size_t bytes = (size_t)splits * M * L.Cout * sizeof(float);
            if (bytes > partial) partial = bytes;
        }
    };
    if (enc->compact_workspace) {
        // two alternating activation buffers (layer i writes buffer i % 2 while reading the other): at B = 256 of the default
        // net 805 MB instead of 973 MB; only the last two layers' outputs survive a forward
        size_t sz[2] = {0, 0};
        for (size_t li = 0; li < enc->layers.size(); ++li) {
            const Layer& L = enc->layers[li];
            sz[li & 1] = std::max(sz[li & 1], align_up((size_t)B * L.Ho * L.Wo * L.Cout * sizeof(float), 256));
            need_partial(L, B * L.Ho * L.Wo);
        }
        for (size_t li = 0; li < enc->layers.size(); ++li) ws.act_off.push_back(off + ((li & 1) ? sz[0] : 0));
        off += sz[0] + sz[1];
    } else {
        for (const Layer& L : enc->layers) {
            ws.act_off.push_back(off);
            off += align_up((size_t)B * L.Ho * L.Wo * L.Cout * sizeof(float), 256);
            need_partial(L, B * L.Ho * L.Wo);
        }
    }
    need_partial(enc->dense, B);                             // (sized for either dense variant)
    if (enc->dense.kind == KIND_IGEMM && !enc->wavek_dense) {   // ... including the split-K igemm when the wave-split-K form is switched off
        int splits, per;
        choose_splits(enc, ceil_div(B, 128) * (enc->dense.CoutPad / 128), (int)(enc->dense.K() / 32), &splits, &per);
        const size_t bytes = splits > 1 ? (size_t)splits * B * enc->dense.Cout * sizeof(float) : 0;
        if (bytes > partial) partial = bytes;
    }
    if (B <= aae::kGemvMaxBatch && enc->dense.kind == KIND_IGEMM) {   // the GEMV form of the dense layer: one partial row per 128-k chunk
        const size_t gemv = gemv_partial_bytes(enc->dense, B);
        if (gemv > partial) partial = gemv;
    }
    ws.partial_off = off;
    ws.partial_bytes = partial;
    off += align_up(partial, 256);
#ifdef AAE_EXPERIMENTS
    if (B <= 4) {
        for (size_t li = 0; li <= enc->layers.size(); ++li) {
            const bool dense = li == enc->layers.size();
            const Layer& L = dense ? enc->dense : enc->layers[li];
            size_t bytes = 0;
            if (L.kind == KIND_IGEMM) {
                const WaveKPlan wk = plan_wavek(enc, L, dense ? B : (long long)B * L.Ho * L.Wo, false);
                if (wk.use) bytes = wk.partial_bytes;
                if (dense) bytes = std::max(bytes, gemv_partial_bytes(L, B));
            }
            ws.chain_partial_off.push_back(off);
            off += align_up(bytes, 256);
        }
    }
#endif
    ws.total = off;
    return ws;
}


}  // namespace aae_host
