// Handles and shared host-side types of libaae_hip.so: error reporting, the Layer record of one convolution, the encoder and
// codebook handles with every launch-planning knob (include/aae_hip_tuning.h names them).  Part of aae_hip_impl.h.
#pragma once

namespace aae_host {

static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define AAE_HIP_TRY(expr)                                                                       \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return aae_host::fail(AAE_ERR_RUNTIME, "%s failed: %s (%s:%d)", #expr,              \
                                  hipGetErrorString(e__), __FILE__, __LINE__);                  \
    } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// [TF-semantics] 'SAME': out = ceil(in/s); total = max((out-1)*s + k - in, 0); before = total/2.
static inline void same_pad(int in, int k, int s, int* out, int* before) {
    const int o = ceil_div(in, s);
    int total = (o - 1) * s + k - in;
    if (total < 0) total = 0;
    *out = o;
    *before = total / 2;
}

enum LayerKind { KIND_FIRST_MFMA = 0, KIND_IGEMM = 1, KIND_GENERIC = 2 };

struct Layer {
    int H = 0, W = 0, Cin = 0, Ho = 0, Wo = 0, Cout = 0, CoutPad = 0;
    int KS = 0, S = 0, pt = 0, pl = 0;
    int relu = 1;
    int index = -1;             // position among the conv layers (0 = first); -1: the dense layer
    LayerKind kind = KIND_GENERIC;
    float* w_hwio = nullptr;    // device [KS*KS*Cin][Cout]
    float* wp = nullptr;        // device [K/4][CoutPad][4]      (igemm)
    unsigned* wp16 = nullptr;   // device [slabs][8][CoutPad][4 dwords]: (hi, lo) halves of w*2^w_shift (f32x3h)
    int w_shift = 0;
    float* wino[4] = {nullptr, nullptr, nullptr, nullptr};   // Winograd-domain weights of the four polyphase components (conv_winograd_f32.h), index 2 eh + ew;
    int wino_geom = -1;         // ... and the block geometry the layer runs with (-1: not eligible / not prepared)
    float* bias = nullptr;
    float* bn_scale = nullptr;  // folded inference BN: x*scale + shift
    float* bn_shift = nullptr;
    // first-layer staging geometry
    int rowlen = 0, first_smem = 0;       // conv1 (conv_first_f32.h): staged floats per input row, LDS bytes
    bool first_packable = true;
    int rowlen4 = 0, lead4 = 0;           // same for the dword-staged uint8 form (0 = not applicable)
    long long K() const { return (long long)KS * KS * Cin; }
};

struct KernelRecord {
    std::string label;
    double flops;
};

constexpr int kX3hRing = 256;          // range-flag slots of eager f32x3h forwards (reused round-robin)
constexpr int kX3hCaptured = 64;       // ... of forwards recorded into HIP graphs (one each, never reused)

}  // namespace aae_host

struct aae_encoder {
    aae_encoder_desc desc;
    std::vector<aae_host::Layer> layers;   // conv layers
    aae_host::Layer dense;                 // 1x1 "conv" over the flattened activation
    float* lut = nullptr;                  // device [256] float32(v/255.)
    // f32x3h range flags: "an activation left the range its fp16 (hi, lo) pair carries exactly".  One int per forward, taken
    // round-robin from a ring (eager forwards) or, for forwards recorded into a HIP graph, from a region that is never recycled
    // (a graph bakes the address).  Nobody has to wait for the stream after a forward: the flags of many forwards are polled
    // together when their results are consumed (aae_encoder_x3h_poll).
    int* x3h_sat = nullptr;                // device [kX3hRing + kX3hCaptured]
    std::atomic<unsigned long long> x3h_seq{0};
    int x3h_captured = 0;                  // slots of the captured region handed out so far (under x3h_mu) ...
    std::vector<int> x3h_free;             // ... and the ones given back (aae_encoder_x3h_release_slot)
    std::vector<hipEvent_t> x3h_release_ev; // per captured slot: recorded behind the flag clear of its release -- a slot is handed out again only once that clear has executed
    std::mutex x3h_mu;
    std::mutex wino_mu;                    // guards the one-time preparation of the Winograd-domain weights (ensure_winograd_weights)
    std::vector<void*> allocations;
    std::vector<aae_host::KernelRecord> records;   // of the most recent completed forward (swapped in under rec_mu)
    std::mutex rec_mu;
    int splitk_min_base_blocks = 384;      // split K only when the un-split grid is smaller than this
    int splitk_target_blocks = 512;        // ... and then aim for about this many blocks
    int reduce_small = 1;                  // <= 8 splits over >= 16k outputs: barrier-free float4 reduce kernel
    int precision = 0;                     // 0: exact fp32 MFMA; 1: f32x3h split-precision igemm (explicit opt-in)
    int winograd = 1;                      // 1: conv layers behind the first one as polyphase Winograd F(2 x 2) on the fp32 matrix cores (2.04 x fewer multiplies,
                                           // results differ from the direct kernels by fp32 rounding: conv_winograd_f32.h) for batches >= winograd_min_batch
    int winograd_wide = 0;                 // 1: blocks of 4 waves, each over both 32-channel halves (one wave per SIMD) instead of 8 waves (two per SIMD)
    int winograd_min_batch = 8;            // ... and layers whose blocks (64 tiles x 64 channels each) fill at least winograd_min_fill_pct per cent of the
    int winograd_min_fill_pct = 56;        // rounds of blocks they occupy (runs_winograd: break-even measured at 0.50-0.56, profiles/r15); winograd_min_blocks > 0 replaces that rule by a plain
    int winograd_min_blocks = 0;           // block count (tests, A/B)
    int multi_mid_group = 1;               // grouped multi-object query: objects with 5 or more detections each share ONE Winograd launch per conv layer where the GROUP's blocks fill
                                           // the rounds they occupy (aae_multi_impl.h); 0 = such objects one after the other
    int multi_split_items = 1;             // aae_encode_nn_multi: a class with 5 ... 8 boxes (up to 12 when it is the frame's only class beyond 4) is answered as items of <= 4 boxes inside the
                                           // per-detection group of the frame (measured 3-19 % faster than a mid-batch group / its own call, profiles/r15/split_items_ab.jsonl); needs multi_group_plan = 1
    int multi_group_winograd = 1;          // per-detection groups (n <= 4 per object): a conv layer runs as ONE Winograd launch across the objects where the group's blocks pass the fill rule
                                           // (conv2 from ~9 detections in a frame, conv3 from ~18); needs multi_group_plan = 1 (the answers carry the Winograd form's rounding)
    int multi_mid_scan = 1;                // ... and the objects' codebook scans (query-resident arg-max form, fp32) run as one launch per row-part count + one reduce launch; 0 = per object
    int multi_mid_ragged = 1;              // ... and a layer of four-image blocks (8 x 8 outputs) hands the objects' LAST 1-3 images to one grouped wave-split-K launch when the ragged
                                           // blocks would open one more round of blocks (config 4: 67 groups = 536 blocks = 3 rounds -> 61 groups + 12 images); 0 = ragged blocks
    int winograd_xcd_cols = -1;            // column blocks of a region that share an XCD (aae_encoder_launch.h: wino_xcd_cols); -1 = per-layer default
    int first_target_blocks = 512;         // conv1 grid size aimed at (x N tiles); 2 blocks fit a CU
    int first_group_split_max_tiles = 128; // conv1: batches of at most this many 128-pixel tiles (B <= 4 of the default net) run one block per 32-pixel group
    int first_vec4 = 1;                    // conv1: stage uint8 rows as aligned dwords when W*C % 4 == 0
    int first_max_tiles_per_block = 16;    // conv1: consecutive 128-pixel tiles one block walks with its weights in registers
    int igemm_breg_min_blocks = 768;       // ... with the 32 KB footprint only for grids of at least this many blocks
    int igemm_breg_wide = 1;               // BREG conv2/conv3: 128 x 256 block tiles (each wave 64 x 128) when the layer is wide enough (+0.9 %)
    int igemm_breg_wide_min_blocks = 512;
    int dense_gemv = 1;                    // B <= dense_gemv_max_batch: dense layer as a weight-streaming GEMV instead of a split-K MFMA tile
    int dense_gemv_max_batch = 8;          // (1 ... 8; 4 = rounds 2-3: beyond it the wave-split-K MFMA tile, 15 us at any mid batch)
    int igemm_breg = 1;                    // conv layers: weight fragments straight from global memory to registers (A-only LDS-DMA, 32 KB LDS)
    int igemm_dma = 1;                     // fp32 igemm operand slabs by LDS-DMA (buffer_load ... lds); 0 = register-staged variant
    int x3h_wide_min_blocks = 0;           // > 0: f32x3h conv layers use 256x128 tiles (8 waves) when that still yields this many blocks; measured neutral (+-1.5 %), off by default
    int x3h_wide256 = 1;                   // f32x3h conv layers with Cout % 256 == 0: 256 x 256 tiles, 8 waves of 64 x 128 ...
    int x3h_wide256_min_blocks = 256;      // ... when that still gives every CU a block
    int x3h_min_tiles = 256;               // precision 2: f32x3h only for batches whose first igemm layer has at least this many 64 x 64 tiles
    int x3h_dma = 1;                       // f32x3h operand slabs by LDS-DMA (buffer_load ... lds); 0 = register-staged variant
    int x3h_act_shift = 4;                 // activations travel as halves of x*2^shift in f32x3h mode (|x| < 4094 exact range)
    int igemm_stagger = 0;                 // kcycles of start delay for every 2nd block generation of the igemm (0 = off)
    // small batches (the reference's one-crop-per-detection usage): wave-split-K igemm with the in-launch ticketed reduce
    int wavek = 1;                         // 0: always the 128 x 128 split-K igemm + reduce launch
    int wavek_max_tiles = 512;             // used while the layer has at most this many 64 x 64 output tiles (two rounds of one block per CU; 256 until the tile shape was balanced: B = 5 ... 12 gain 7-9 %)
    int wavek_tiny_max_tiles = 64;         // <= this many 64 x 64 tiles: 32 x 32 wave tiles (four times the tiles: K is split across fewer blocks or none);
                                           // measured: wins up to 64 tiles (B=1: 94 -> 87 us), loses from 128 on (twice the operand loads per MFMA)
    int wavek_target_blocks = 0;           // blocks of one "round" of the chip = blocks a split layer is cut into (tiles x K splits): 0 = one per compute unit of the
                                           // device (wavek_round_blocks(): 256 on MI355X, the value the cost model was fitted with); > 0 pins it (tests, A/B)
    int wavek_waves = 4;                   // waves per block (4 | 8), each with its own K range
    int wavek_eff64x32_pct = 74;           // cost model: efficiency of the 64 x 32 wave tile, per cent (0.72 in round 3's fit; with the tail cut it
                                           // wins more often than that predicted: 74 takes conv4 at B = 24 from 198 to 175 us and changes nothing
                                           // else at B = 5 ... 48; 75 also moves conv4 at B = 6 to a 64 x 32 tile that is 5 us slower, 78 and more
                                           // lose at B = 5, 28 as well)
    int wavek_g_boost = 2;                 // planner by cost (B = 3, B >= 5): layers that split K split it for this many blocks per CU (two co-resident
                                           // blocks hide each other's load stalls: B = 5 236 -> 226 us, 8: 316 -> 309, 16: 560 -> 552; the per-detection
                                           // batches B = 1, 2, 4 measured 4-10 % SLOWER that way and keep one block per CU)
    int planner_cost_batch3 = 1;           // ... and at B = 3
    int planner_cost_min_batch = 5;        // planner by cost from this batch on (below: the measured thresholds of the per-detection path)
    int wavek_tail_split = 1;              // planner by cost: tiles beyond the last full round of whole tiles are cut in K (wavek_tail_split())
    int wavek_force_tail_tiles = 0;        // tests: cut the last n tiles of every un-split wave-split-K layer ...
    int wavek_force_tail_g = 2;            // ... this many ways
    int wavek_spread = 3;                  // bit 0: 64 x 64 wave tiles (four accumulators): next-slab loads between the MFMAs instead of a burst in front of them (conv_wavek_f32.h):
                                           // conv2 at B = 8 128 -> 115 us, B = 24 encoder 910 -> 830 us; measured neutral-to-worse for 64 x 32 tiles, not used there.
                                           // bit 1: 32 x 32 tiles with a second accumulator for the odd q-steps (two fma chains, added once): B = 1 80.2 -> 78.0 us
    int wavek_pingpong = 0;                // 8-wave blocks: the two waves of a SIMD alternate load issue and MFMAs behind block barriers (conv_wavek_f32.h);
                                           // measured SLOWER than free-running waves (B = 1: 95 vs 82 us): kept as an option with its measurement, off
    int wavek_tiny_waves = 4;              // ... of the 32 x 32 wave tiles (per-detection batches): 8 = two waves per SIMD, so that one wave's operand-load issue
                                           // (~250 cycles per slab in which its dependent MFMA chain stands still) runs under the other wave's MFMAs
    // measured per layer with rocprofv3 at B = 1 ... 8 (profiles/r09_small/variants_*.txt): depth 2 beats 3 by 0.5-1 us per launch
    // (208 instead of 272 registers, the second slab in flight is enough); 64 x 32 wave tiles win up to 128 tiles of 64 x 64 --
    // fewer blocks per tile to hand over, smaller partials -- and lose beyond (conv2 at B = 4: 68.6 vs 61.5 us)
    int wavek_depth = 2;                   // slabs of fragments in flight per wave (2 | 3)
    int wavek_narrow_max_tiles = 128;      // <= this many 64 x 64 tiles: 64 x 32 wave tiles (twice the tiles, half the splits to add up)
    long long* wavek_timeline = nullptr;   // device [3 layers][512 blocks][8] phase stamps when option wavek_timeline is on (profiling tools)
    int compact_workspace = 0;             // 1: two alternating activation buffers instead of one per layer (layer outputs are then not inspectable)
    int ticket_prep = 1;                   // conv1 installs the nonces of the later ticketed launches of its forward call (0: every launch installs its own)
    int wavek_balance = 1;                 // wave-split-K tile shape: prefer a smaller wave tile when the larger one leaves CUs idle in its last round of blocks (plan_wavek)
    int planner_cost_model = 1;            // B >= 5: kernel family and wave-tile shape of every conv layer by estimated time (plan_by_cost) instead of tile-count thresholds
    int wavek_ablate = 0;                  // timing experiments (conv_wavek_f32.h ConvWaveKArgs::ablate); results are wrong when != 0
    int gemv_ticket = 1;                   // dense GEMV (B <= 4): chunk sums finished by the last block instead of a reduce launch
    int wavek_dense = 1;                   // dense layer (B > 4) on the wave-split-K kernel instead of split-K igemm + reduce launch
    // per-detection batches (B <= 4): everything behind conv1 as ONE persistent launch (detect_chain.h).  Opt-in: measured on MI355X it
    // is SLOWER than the six launches it replaces (B = 1: 92 vs 82 us, B = 4: 215 vs 200 -- a grid barrier costs 3-4 us in there, more
    // than the 1.5-2 us kernel boundary it removes, and the cross-barrier prefetch wins back less; profiles/r11_small/chain_*).
    int detect_chain = 0;
    int detect_chain_blocks = 256;         // its grid: one block per CU, never more than the device has (every block must be resident)
    int cu_count = 0;                      // compute units of the device the handle lives on
    int multi_xcd_affine = 1;              // grouped query, 8 | 16 equal-sized objects: all blocks of an object on one XCD (conv_wavek_f32.h, ConvWaveKMultiArgs)
    int multi_force_depth = 0;             // A/B: slabs in flight of the group plan's 64 x 32 layers (nibble per conv layer)
    int multi_force_shape = 0, multi_force_g = 0;   // A/B of plan_wavek_group: wave tile (nibble per conv layer) / K split (byte per conv layer) forced
    int multi_group_plan = 1;              // aae_encode_nn_multi: a group of objects runs ONE launch plan chosen for the group's total tile count
                                           // (aae_multi_impl.h, plan_wavek_group); 0 = every object its own plan: bit-identical to aae_encode_nn
    int chain_timeline = 0;                // profiling aid: the persistent launch stamps its phase edges into the wavek_timeline buffer
};

struct aae_codebook {
    float* E = nullptr;    // device [N][J] (fp32 codebook), or the bf16 rows when dtype == AAE_DTYPE_BF16
    void* E_alloc = nullptr;   // the allocation E lies in (E is aligned up to kCodebookAlign inside it)
    int dtype = AAE_DTYPE_F32;
    int N = 0, J = 0;
    int scan_mode = AAE_SCAN_AUTO;
    int cu_count = 256;    // compute units of the device the handle lives on: the query-resident scan puts one block on each
    int topk_prune = 1;    // top-k inside the query-resident scan: drop candidates below the bound the blocks publish (AAE_SCAN_AUTO_NO_PRUNE: 0)
    // upright search (col_stride k > 1): a compacted copy of rows 0, k, 2k, ... prepared by
    // aae_codebook_prepare_upright; the scan then runs over N/k rows and the winning row id is scaled by k
    aae_codebook* upright = nullptr;   // the copy for the stride asked for last (one of upright_copies)
    int upright_stride = 0;
    // every compacted copy ever prepared, one per stride, kept until the handle is destroyed: a captured HIP graph may
    // hold the address of a copy made for another stride than the one in use now
    std::vector<std::pair<int, aae_codebook*>> upright_copies;
    // B <= 4, top-1 on a stream kernel: arg-max over the block partials inside the scan launch (last block to arrive)
    // instead of a separate argmax_reduce launch.  0: never (AAE_SCAN_STREAM_2L); otherwise always
    int scan_ticket = 1;
    // fp32 stream scan (B <= 4): 0 = one 32-row batch per wave, the whole codebook requested at once (scan_stream_kernel); 1
    // (AAE_SCAN_STREAM_WALK) = one block per CU walks the codebook with two batches in flight per wave (scan_stream_walk_kernel:
    // measured level at B = 1, slower at B = 4 inside the fused query -- 18.4 vs 16.9 us)
    int scan_walk = 0;
    // B > 4, top-1 on the query-resident kernel: 1 = the scan normalises the raw latent codes in its prologue (one launch less);
    // 0 (AAE_SCAN_AUTO_PACKED) = l2norm_pack launch in front, the scan reads the packed planes -- identical bits
    int scan_fused_norm = 1;
    int scan_resident_fin = 0;   // AAE_SCAN_AUTO_FIN: the B <= 32 resident scan answers inside its launch (ticket finish) instead of an argmax_reduce launch
    int scan_rh4 = 1;      // B <= 32, top-1 on the query-resident kernel: rows of a tile over four waves per query group (AAE_SCAN_AUTO_RH2: 0 = two, A/B)
};
