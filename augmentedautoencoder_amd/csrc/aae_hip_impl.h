// Host side of libaae_hip.so: handle management, weight packing, launch planning.
// Compiled by hipcc for gfx950 through aae_hip.hip.  (tests/emu/ compiles the
// same text against a CPU fiber emulator to unit-test the launch logic; that
// build is test infrastructure and is never loaded by the package.)
//
// The including translation unit provides: the HIP runtime API, the wave
// primitives of device_intrinsics.h and
//   AAE_LAUNCH(kernel, grid, block, smem_bytes, stream, args...)
#pragma once

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/aae_hip.h"
#include "kernels/tile_f32.h"
#include "kernels/multi_launch.h"
#include "kernels/conv_igemm_f32.h"
#include "kernels/conv_wavek_f32.h"
#include "kernels/conv_igemm_x3h.h"
#include "kernels/conv_first_f32.h"
#include "kernels/conv_direct_generic.h"
#include "kernels/dense_gemv_f32.h"
#include "kernels/codebook_scan_f32.h"
#include "kernels/codebook_scan_bf16.h"
#include "kernels/codebook_scan_resident.h"
#include "kernels/crop_resize_u8.h"
#include "kernels/detect_chain.h"

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
    std::mutex x3h_mu;
    std::vector<void*> allocations;
    std::vector<aae_host::KernelRecord> records;   // of the most recent completed forward (swapped in under rec_mu)
    std::mutex rec_mu;
    int splitk_min_base_blocks = 384;      // split K only when the un-split grid is smaller than this
    int splitk_target_blocks = 512;        // ... and then aim for about this many blocks
    int reduce_small = 1;                  // <= 8 splits over >= 16k outputs: barrier-free float4 reduce kernel
    int precision = 0;                     // 0: exact fp32 MFMA; 1: f32x3h split-precision igemm (explicit opt-in)
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
    int multi_force_shape = 0, multi_force_g = 0;   // A/B of plan_wavek_group: wave tile (nibble per conv layer) / K split (byte per conv layer) forced
    int multi_group_plan = 1;              // aae_encode_nn_multi: a group of objects runs ONE launch plan chosen for the group's total tile count
                                           // (aae_multi_impl.h, plan_wavek_group); 0 = every object its own plan: bit-identical to aae_encode_nn
    int chain_timeline = 0;                // profiling aid: the persistent launch stamps its phase edges into the wavek_timeline buffer
};

struct aae_codebook {
    float* E = nullptr;    // device [N][J] (fp32 codebook), or the bf16 rows when dtype == AAE_DTYPE_BF16
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
    ws.total = off;
    return ws;
}

// ------------------------------------------------------------ layer launches
struct Timer {
    bool on = false;
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> ev;
    int mark() {
        if (!on) return AAE_OK;
        hipEvent_t e;
        AAE_HIP_TRY(hipEventCreate(&e));
        ev.push_back(e);
        AAE_HIP_TRY(hipEventRecord(e, stream));
        return AAE_OK;
    }
};

static int launch_igemm(aae_encoder* enc, const Layer& L, const float* x, int M, float* out, float* partial,
                        hipStream_t stream, Timer& tm, const char* name, int tag = 0) {
    aae::ConvIgemmArgs a;
    a.x = x; a.wp = L.wp; a.bias = L.bias; a.bn_scale = L.bn_scale; a.bn_shift = L.bn_shift;
    a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Ho = L.Ho; a.Wo = L.Wo; a.Cout = L.Cout; a.CoutPad = L.CoutPad;
    a.KS = L.KS; a.S = L.S; a.pt = L.pt; a.pl = L.pl; a.M = M; a.relu = L.relu; a.stagger = enc->igemm_stagger;
    const unsigned long long x_bytes = (unsigned long long)(M / (L.Ho * L.Wo)) * L.H * L.W * L.Cin * sizeof(float);
    if (x_bytes >= 0xFFFFFFF0ull)
        return fail(AAE_ERR_UNSUPPORTED, "%s: input activation of %llu bytes exceeds the 4 GiB buffer view; use a smaller batch", name, x_bytes);
    a.x_bytes = (unsigned)x_bytes;
    a.slabs_total = (int)(L.K() / 32);
    a.wp_bytes = (unsigned)((unsigned long long)a.slabs_total * 8ull * L.CoutPad * 16ull);
    a.num_mt = ceil_div(M, 128);
    a.num_nt = L.CoutPad / 128;
    choose_splits(enc, a.num_mt * a.num_nt, a.slabs_total, &a.splits, &a.slabs_per_split);
    const int nblk = a.num_mt * a.num_nt * a.splits;
    const double flops = 2.0 * (double)M * (double)L.K() * (double)L.Cout;
    const bool dma = enc->igemm_dma != 0;
    const char* kname = dma ? "conv_igemm_f32_dma" : "conv_igemm_f32";
    char label[96];
    if (a.splits == 1) {
        a.out = out;
        // A buffers only (32 KB: three blocks per CU); grids too small to give every CU three blocks keep the
        // 64 KB footprint so that the blocks spread two per CU instead of 3/2/1
        const int kBregSmem = nblk >= enc->igemm_breg_min_blocks ? 2 * aae::kSlabFloatsA * 4 : aae::kConvIgemmSmem;
        const bool breg = dma && enc->igemm_breg && tag >= 1 && tag <= 3;
        if (breg) kname = "conv_igemm_f32_dma_breg";
        // 128 x 256 block tiles (each wave 64 x 128) where the layer is wide enough and the grid stays large
        if (breg && enc->igemm_breg_wide && (tag == 1 || tag == 2) && L.CoutPad % 256 == 0 &&
            a.num_mt * (L.CoutPad / 256) >= enc->igemm_breg_wide_min_blocks) {
            a.num_nt = L.CoutPad / 256;
            const int wide_blocks = a.num_mt * a.num_nt;
            constexpr int smem = 2 * aae::kSlabFloatsA * 4;
            if (tag == 1) AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true, false, 1, true, 4>), dim3(wide_blocks), dim3(256), smem, stream, a);
            else AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true, false, 2, true, 4>), dim3(wide_blocks), dim3(256), smem, stream, a);
            snprintf(label, sizeof(label), "%s:conv_igemm_f32_dma_breg_n256 M=%d N=%d K=%lld", name, M, L.Cout, L.K());
            note_kernel({label, flops});
            AAE_HIP_TRY(hipGetLastError());
            return tm.mark();
        }
        if (breg && tag == 1) AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true, false, 1, true>), dim3(nblk), dim3(256), kBregSmem, stream, a);
        else if (breg && tag == 2) AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true, false, 2, true>), dim3(nblk), dim3(256), kBregSmem, stream, a);
        else if (breg && tag == 3) AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true, false, 3, true>), dim3(nblk), dim3(256), kBregSmem, stream, a);
        else if (dma && tag == 1) AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true, false, 1>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
        else if (dma && tag == 2) AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true, false, 2>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
        else if (dma && tag == 3) AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true, false, 3>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
        else if (dma) AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, true>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
        else AAE_LAUNCH((aae::conv_igemm_f32_kernel<false, false>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
        snprintf(label, sizeof(label), "%s:%s M=%d N=%d K=%lld", name, kname, M, L.Cout, L.K());
        note_kernel({label, flops});
        AAE_HIP_TRY(hipGetLastError());
        return tm.mark();
    }
    a.out = partial;
    if (dma) AAE_LAUNCH((aae::conv_igemm_f32_kernel<true, true>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
    else AAE_LAUNCH((aae::conv_igemm_f32_kernel<true, false>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
    snprintf(label, sizeof(label), "%s:%s_splitk%d M=%d N=%d K=%lld", name, kname, a.splits, M, L.Cout, L.K());
    note_kernel({label, flops});
    AAE_HIP_TRY(hipGetLastError());
    if (int rc = tm.mark()) return rc;
    aae::SplitKReduceArgs r;
    r.partial = partial; r.bias = L.bias; r.bn_scale = L.bn_scale; r.bn_shift = L.bn_shift; r.out = out;
    r.MN = (long long)M * L.Cout; r.Cout = L.Cout; r.splits = a.splits; r.relu = L.relu;
    r.out_planes = 0; r.out_scale = 1.f;
    launch_splitk_reduce(r, stream, enc->reduce_small != 0);
    snprintf(label, sizeof(label), "%s:splitk_reduce", name);
    note_kernel({label, 0.0});
    AAE_HIP_TRY(hipGetLastError());
    return tm.mark();
}

// wave-split-K igemm (conv_wavek_f32.h): small M -- the per-detection batches
template <int MT, int NT, int WAVES, int DEPTH, bool SPREAD = false>
static void launch_wavek_t(const aae::ConvWaveKArgs& a, int tag, int nblk, hipStream_t stream) {
    constexpr int smem = aae::conv_wavek_smem<MT, NT, WAVES>();
    // TAG only makes the symbol unique per encoder layer (separate rows in rocprofv3 --stats)
    if (tag == 1) {
        (void)hipFuncSetAttribute((const void*)aae::conv_wavek_f32_kernel<MT, NT, WAVES, DEPTH, 1, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        AAE_LAUNCH((aae::conv_wavek_f32_kernel<MT, NT, WAVES, DEPTH, 1, SPREAD>), dim3(nblk), dim3(64 * WAVES), smem, stream, a);
    } else if (tag == 2) {
        (void)hipFuncSetAttribute((const void*)aae::conv_wavek_f32_kernel<MT, NT, WAVES, DEPTH, 2, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        AAE_LAUNCH((aae::conv_wavek_f32_kernel<MT, NT, WAVES, DEPTH, 2, SPREAD>), dim3(nblk), dim3(64 * WAVES), smem, stream, a);
    } else if (tag == 3) {
        (void)hipFuncSetAttribute((const void*)aae::conv_wavek_f32_kernel<MT, NT, WAVES, DEPTH, 3, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        AAE_LAUNCH((aae::conv_wavek_f32_kernel<MT, NT, WAVES, DEPTH, 3, SPREAD>), dim3(nblk), dim3(64 * WAVES), smem, stream, a);
    } else {
        (void)hipFuncSetAttribute((const void*)aae::conv_wavek_f32_kernel<MT, NT, WAVES, DEPTH, 0, SPREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        AAE_LAUNCH((aae::conv_wavek_f32_kernel<MT, NT, WAVES, DEPTH, 0, SPREAD>), dim3(nblk), dim3(64 * WAVES), smem, stream, a);
    }
}

static aae::ConvWaveKArgs wavek_args(const aae_encoder* enc, const Layer& L, const WaveKPlan& w, const float* x, int M, float* out, float* partial,
                                     unsigned long long* tickets, unsigned nonce, int tag) {
    aae::ConvWaveKArgs a;
    a.x = x; a.wp = L.wp; a.bias = L.bias; a.bn_scale = L.bn_scale; a.bn_shift = L.bn_shift; a.out = out;
    a.partial = partial; a.partial_bytes = (unsigned)w.partial_bytes; a.tickets = tickets; a.nonce = nonce;
    a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Ho = L.Ho; a.Wo = L.Wo; a.Cout = L.Cout; a.CoutPad = L.CoutPad;
    a.KS = L.KS; a.S = L.S; a.pt = L.pt; a.pl = L.pl; a.M = M; a.relu = L.relu; a.ablate = enc->wavek_ablate; a.pingpong = enc->wavek_pingpong; a.spread = enc->wavek_spread;
    a.timeline = (enc->wavek_timeline && tag >= 1 && tag <= 3 && w.blocks() <= 512)       // (the debug buffer holds 512 blocks per layer)
                     ? enc->wavek_timeline + (size_t)(tag - 1) * 512 * 8 : nullptr;
    a.x_bytes = (unsigned)((unsigned long long)(M / (L.Ho * L.Wo)) * L.H * L.W * L.Cin * sizeof(float));
    a.slabs_total = (int)(L.K() / 32);
    a.wp_bytes = (unsigned)((unsigned long long)a.slabs_total * 8ull * L.CoutPad * 16ull);
    a.num_mt = w.num_mt; a.num_nt = w.num_nt; a.gsplits = w.gsplits;
    a.tail_tiles = w.tail_tiles; a.tail_gsplits = w.tail_g;
    return a;
}

static int launch_wavek(aae_encoder* enc, const Layer& L, const WaveKPlan& w, const float* x, int M, float* out, float* partial,
                        unsigned long long* tickets, unsigned nonce, hipStream_t stream, Timer& tm, const char* name, int tag) {
    const aae::ConvWaveKArgs a = wavek_args(enc, L, w, x, M, out, partial, tickets, nonce, tag);
    const int nblk = w.blocks();
    const int key = (w.MT == 1 ? 1000 : 0) + w.NT * 100 + w.waves * 10 + w.depth;
    switch (key) {
        case 243: launch_wavek_t<2, 2, 4, 3>(a, tag, nblk, stream); break;
        case 242: if (a.spread & 1) launch_wavek_t<2, 2, 4, 2, true>(a, tag, nblk, stream); else launch_wavek_t<2, 2, 4, 2>(a, tag, nblk, stream); break;
        case 282: launch_wavek_t<2, 2, 8, 2>(a, tag, nblk, stream); break;
        case 143: launch_wavek_t<2, 1, 4, 3>(a, tag, nblk, stream); break;
        case 142: launch_wavek_t<2, 1, 4, 2>(a, tag, nblk, stream); break;
        case 1142: if (a.spread & 2) launch_wavek_t<1, 1, 4, 2, true>(a, tag, nblk, stream); else launch_wavek_t<1, 1, 4, 2>(a, tag, nblk, stream); break;
        case 1143: launch_wavek_t<1, 1, 4, 3>(a, tag, nblk, stream); break;
        case 182: launch_wavek_t<2, 1, 8, 2>(a, tag, nblk, stream); break;
        case 1182: launch_wavek_t<1, 1, 8, 2>(a, tag, nblk, stream); break;
        default: return fail(AAE_ERR_RUNTIME, "%s: no wave-split-K instantiation for NT=%d waves=%d depth=%d", name, w.NT, w.waves, w.depth);
    }
    char label[128], tail[24] = "";
    if (w.tail_tiles > 0) snprintf(tail, sizeof(tail), "t%dx%d", w.tail_tiles, w.tail_g);       // (e.g. g1t64x4: the last 64 tiles cut four ways)
    snprintf(label, sizeof(label), "%s:conv_wavek_f32_%dx%d_w%d_d%d_g%d%s M=%d N=%d K=%lld", name, 32 * w.MT, 32 * w.NT, w.waves, w.depth,
             w.gsplits, tail, M, L.Cout, L.K());
    note_kernel({label, 2.0 * (double)M * (double)L.K() * (double)L.Cout});
    AAE_HIP_TRY(hipGetLastError());
    return tm.mark();
}

// f32x3h variant: x and (unless out_f32) out are fp16 (hi, lo) pairs of value * 2^x3h_act_shift (x3h_pair_index layout).
static int launch_igemm_x3h(aae_encoder* enc, const Layer& L, const void* x, int M, void* out, bool out_f32, float* partial,
                            hipStream_t stream, Timer& tm, const char* name, int tag = 0) {
    aae::ConvIgemmX3hArgs a;
    a.x = static_cast<const unsigned short*>(x); a.wp = L.wp16; a.bias = L.bias; a.bn_scale = L.bn_scale; a.bn_shift = L.bn_shift;
    a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Ho = L.Ho; a.Wo = L.Wo; a.Cout = L.Cout; a.CoutPad = L.CoutPad;
    a.KS = L.KS; a.S = L.S; a.pt = L.pt; a.pl = L.pl; a.M = M; a.relu = L.relu;
    const unsigned long long in_bytes = (unsigned long long)(M / (L.Ho * L.Wo)) * L.H * L.W * L.Cin * 4;
    if (in_bytes >= 0xFFFFFFF0ull)
        return fail(AAE_ERR_UNSUPPORTED, "%s: input activation of %llu bytes exceeds the 4 GiB buffer view; use a smaller batch", name, in_bytes);
    a.x_bytes = (unsigned)in_bytes;
    a.inv_scale = ldexpf(1.f, -(enc->x3h_act_shift + L.w_shift));
    a.out_scale = ldexpf(1.f, enc->x3h_act_shift);
    a.sat_flag = out_f32 ? nullptr : t_x3h_flag;
    a.slabs_total = (int)(L.K() / 32);
    a.wp_bytes = (unsigned)((unsigned long long)a.slabs_total * 8ull * L.CoutPad * 16ull);
    a.num_mt = ceil_div(M, 128);
    a.num_nt = L.CoutPad / 128;
    choose_splits(enc, a.num_mt * a.num_nt, a.slabs_total, &a.splits, &a.slabs_per_split);
    const int nblk = a.num_mt * a.num_nt * a.splits;
    const double flops = 2.0 * (double)M * (double)L.K() * (double)L.Cout;
    const bool dma = enc->x3h_dma != 0;
    const char* kname = dma ? "conv_igemm_x3h_dma" : "conv_igemm_x3h";
    char label[96];
    // 256 x 256 tiles, 8 waves of 64 x 128 (LDS traffic per MFMA -33 %): layers with Cout % 256 == 0 whose grid still fills the chip
    if (dma && !out_f32 && enc->x3h_wide256 && L.CoutPad % 256 == 0 && tag >= 1 && tag <= 3 &&
        ceil_div(M, 256) * (L.CoutPad / 256) >= enc->x3h_wide256_min_blocks) {
        a.num_mt = ceil_div(M, 256);
        a.num_nt = L.CoutPad / 256;
        a.splits = 1;
        a.slabs_per_split = a.slabs_total;
        a.out = out;
        const int wide_blocks = a.num_mt * a.num_nt;
        constexpr int smem = aae::kX3hWideSmem;
        if (tag == 1) {
            (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_wide_kernel<aae::X3H_OUT_PLANES, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            AAE_LAUNCH((aae::conv_igemm_x3h_wide_kernel<aae::X3H_OUT_PLANES, 1>), dim3(wide_blocks), dim3(512), smem, stream, a);
        } else if (tag == 2) {
            (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_wide_kernel<aae::X3H_OUT_PLANES, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            AAE_LAUNCH((aae::conv_igemm_x3h_wide_kernel<aae::X3H_OUT_PLANES, 2>), dim3(wide_blocks), dim3(512), smem, stream, a);
        } else {
            (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_wide_kernel<aae::X3H_OUT_PLANES, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            AAE_LAUNCH((aae::conv_igemm_x3h_wide_kernel<aae::X3H_OUT_PLANES, 3>), dim3(wide_blocks), dim3(512), smem, stream, a);
        }
        snprintf(label, sizeof(label), "%s:conv_igemm_x3h_wide256 M=%d N=%d K=%lld", name, M, L.Cout, L.K());
        note_kernel({label, flops});
        AAE_HIP_TRY(hipGetLastError());
        return tm.mark();
    }
    // 256 x 128 tiles (8 waves, one block per CU) when they still give every CU a block
    if (dma && !out_f32 && tag >= 1 && tag <= 3 && enc->x3h_wide_min_blocks > 0 &&
        ceil_div(M, 256) * a.num_nt >= enc->x3h_wide_min_blocks) {
        a.num_mt = ceil_div(M, 256);
        a.splits = 1;
        a.slabs_per_split = a.slabs_total;
        a.out = out;
        const int wide_blocks = a.num_mt * a.num_nt;
        constexpr int smem = aae::x3h_dma_smem<4>();
        if (tag == 1) AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 1, 4>), dim3(wide_blocks), dim3(512), smem, stream, a);
        else if (tag == 2) AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 2, 4>), dim3(wide_blocks), dim3(512), smem, stream, a);
        else AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 3, 4>), dim3(wide_blocks), dim3(512), smem, stream, a);
        snprintf(label, sizeof(label), "%s:conv_igemm_x3h_dma256 M=%d N=%d K=%lld", name, M, L.Cout, L.K());
        note_kernel({label, flops});
        AAE_HIP_TRY(hipGetLastError());
        return tm.mark();
    }
    if (a.splits == 1) {
        a.out = out;
        if (dma) {
            if (out_f32) AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_F32>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
            else if (tag == 1) AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 1>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
            else if (tag == 2) AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 2>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
            else if (tag == 3) AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 3>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
            else AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
        } else {
            if (out_f32) AAE_LAUNCH((aae::conv_igemm_x3h_kernel<aae::X3H_OUT_F32>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
            else AAE_LAUNCH((aae::conv_igemm_x3h_kernel<aae::X3H_OUT_PLANES>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
        }
        snprintf(label, sizeof(label), "%s:%s M=%d N=%d K=%lld", name, kname, M, L.Cout, L.K());
        note_kernel({label, flops});
        AAE_HIP_TRY(hipGetLastError());
        return tm.mark();
    }
    a.out = partial;
    if (dma) AAE_LAUNCH((aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PARTIAL>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
    else AAE_LAUNCH((aae::conv_igemm_x3h_kernel<aae::X3H_OUT_PARTIAL>), dim3(nblk), dim3(256), aae::kConvIgemmSmem, stream, a);
    snprintf(label, sizeof(label), "%s:%s_splitk%d M=%d N=%d K=%lld", name, kname, a.splits, M, L.Cout, L.K());
    note_kernel({label, flops});
    AAE_HIP_TRY(hipGetLastError());
    if (int rc = tm.mark()) return rc;
    aae::SplitKReduceArgs r;
    r.partial = partial; r.bias = L.bias; r.bn_scale = L.bn_scale; r.bn_shift = L.bn_shift; r.out = static_cast<float*>(out);
    r.MN = (long long)M * L.Cout; r.Cout = L.Cout; r.splits = a.splits; r.relu = L.relu;
    r.out_planes = out_f32 ? 0 : 1; r.out_scale = a.out_scale; r.sat_flag = out_f32 ? nullptr : t_x3h_flag;
    launch_splitk_reduce(r, stream, enc->reduce_small != 0);
    snprintf(label, sizeof(label), "%s:splitk_reduce", name);
    note_kernel({label, 0.0});
    AAE_HIP_TRY(hipGetLastError());
    return tm.mark();
}

template <int KS, int C>
static void launch_first_t(const aae::ConvFirstArgs& a, bool u8, bool planes, dim3 grid, int smem, hipStream_t stream, bool group_split = false) {
    const bool vec4 = u8 && a.vec4;
    if (group_split && !planes) {                 // per-detection batches: one block per 32-pixel group (grid.z = 4)
        const dim3 g4(grid.x, grid.y, 4);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<KS, C, true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<KS, C, true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<KS, C, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (vec4) AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, true, false, true, true>), g4, dim3(256), smem, stream, a);
        else if (u8) AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, true, false, false, true>), g4, dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, false, false, false, true>), g4, dim3(256), smem, stream, a);
        return;
    }
    if (planes) {
        if (vec4) AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, true, true, true>), grid, dim3(256), smem, stream, a);
        else if (u8) AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, true, true>), grid, dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, false, true>), grid, dim3(256), smem, stream, a);
    } else {
        if (vec4) AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, true, false, true>), grid, dim3(256), smem, stream, a);
        else if (u8) AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, true, false>), grid, dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::conv_first_f32_kernel<KS, C, false, false>), grid, dim3(256), smem, stream, a);
    }
}

// arguments of the first-layer kernel for a batch of B crops; returns the number of tile runs (blocks along grid.x)
static int first_core_args(const aae_encoder* enc, const Layer& L, const void* x, bool u8, int B, float* out, bool planes, aae::ConvFirstCore& a) {
    a.x = x; a.lut = enc->lut; a.w = L.w_hwio; a.bias = L.bias; a.bn_scale = L.bn_scale; a.bn_shift = L.bn_shift;
    a.out = out; a.H = L.H; a.W = L.W; a.Ho = L.Ho; a.Wo = L.Wo; a.Cout = L.Cout;
    a.S = L.S; a.pt = L.pt; a.pl = L.pl; a.relu = L.relu;
    a.vec4 = (u8 && L.rowlen4 > 0 && enc->first_vec4 && (reinterpret_cast<uintptr_t>(x) & 3) == 0) ? 1 : 0;   // dword loads want a 4-byte aligned batch
    a.rowlen = a.vec4 ? L.rowlen4 : L.rowlen;
    a.lead = a.vec4 ? L.lead4 : 0;
    a.out_scale = ldexpf(1.f, enc->x3h_act_shift);
    a.sat_flag = planes ? t_x3h_flag : nullptr;
    a.tiles_per_image = ceil_div(L.Ho * L.Wo, 128);
    a.total_tiles = B * a.tiles_per_image;
    int tpb = ceil_div(a.total_tiles, enc->first_target_blocks);
    if (tpb < 1) tpb = 1;
    if (tpb > enc->first_max_tiles_per_block) tpb = enc->first_max_tiles_per_block;
    a.tiles_per_block = tpb;
    return ceil_div(a.total_tiles, tpb);
}

static int launch_first(aae_encoder* enc, const Layer& L, const void* x, bool u8, int B, float* out, bool planes,
                        hipStream_t stream, Timer& tm, const aae::TicketPrep* prep = nullptr) {
    aae::ConvFirstArgs a;
    if (prep) a.prep = *prep;
    else a.prep.n = 0;
    const int runs = first_core_args(enc, L, x, u8, B, out, planes, a);
    const dim3 grid(runs + (a.prep.n > 0 ? 1 : 0), ceil_div(L.Cout, 128));    // + the ticket-preparation block
    // per-detection batches: the four 32-pixel groups of every tile go to four blocks (10.6 -> ? us at B = 1)
    const bool group_split = !planes && a.total_tiles <= enc->first_group_split_max_tiles;
    if (L.Cin == 3) launch_first_t<5, 3>(a, u8, planes, grid, L.first_smem, stream, group_split);
    else launch_first_t<5, 1>(a, u8, planes, grid, L.first_smem, stream, group_split);
    char label[96];
    snprintf(label, sizeof(label), "conv1:conv_first_f32%s M=%d N=%d K=%lld", group_split ? "_g4" : "", B * L.Ho * L.Wo, L.Cout, L.K());
    note_kernel({label, 2.0 * (double)B * L.Ho * L.Wo * (double)L.K() * L.Cout});
    AAE_HIP_TRY(hipGetLastError());
    return tm.mark();
}

static int launch_generic(aae_encoder* enc, const Layer& L, const void* x, bool u8, long long B, float* out,
                          hipStream_t stream, Timer& tm, const char* name) {
    aae::ConvDirectArgs a;
    a.x = x; a.lut = enc->lut; a.w = L.w_hwio; a.bias = L.bias; a.bn_scale = L.bn_scale; a.bn_shift = L.bn_shift;
    a.out = out; a.H = L.H; a.W = L.W; a.Cin = L.Cin; a.Ho = L.Ho; a.Wo = L.Wo; a.Cout = L.Cout;
    a.KS = L.KS; a.S = L.S; a.pt = L.pt; a.pl = L.pl; a.relu = L.relu;
    a.total = B * L.Ho * L.Wo * L.Cout;
    long long blocks = (a.total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (u8) AAE_LAUNCH((aae::conv_direct_generic_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else AAE_LAUNCH((aae::conv_direct_generic_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    char label[96];
    snprintf(label, sizeof(label), "%s:conv_direct_generic", name);
    note_kernel({label, 2.0 * (double)B * L.Ho * L.Wo * (double)L.K() * L.Cout});
    AAE_HIP_TRY(hipGetLastError());
    return tm.mark();
}

// dense layer at B <= 8: weight-streaming GEMV + the fixed-order chunk reduction
static int gemv_max_batch(const aae_encoder* enc) {
    return enc->dense_gemv_max_batch < 1 ? 1 : (enc->dense_gemv_max_batch > aae::kGemvMaxBatch ? aae::kGemvMaxBatch : enc->dense_gemv_max_batch);
}
static bool gemv_uses_ticket(const aae_encoder* enc, const Layer& D) {
    return enc->gemv_ticket && D.Cout % 4 == 0 && D.CoutPad / 128 <= kGemvTicketSlots;
}

static aae::DenseGemvArgs gemv_args(const Layer& D, const float* x, int B, float* partial) {
    aae::DenseGemvArgs a;
    a.x = x; a.wp = D.wp; a.partial = partial; a.B = B; a.K = (int)D.K(); a.Cout = D.Cout; a.CoutPad = D.CoutPad;
    a.wp_bytes = (unsigned)((unsigned long long)(D.K() / 4) * D.CoutPad * 16ull);
    a.partial_bytes = (unsigned)gemv_partial_bytes(D, B);
    a.bias = nullptr; a.bn_scale = nullptr; a.bn_shift = nullptr; a.out = nullptr; a.tickets = nullptr; a.nonce = 0; a.relu = 0;
    return a;
}

static int launch_dense_gemv(aae_encoder* enc, const Layer& D, const float* x, int B, float* out, float* partial,
                             unsigned long long* tickets, unsigned nonce, hipStream_t stream, Timer& tm) {
    aae::DenseGemvArgs a = gemv_args(D, x, B, partial);
    const int chunks = ceil_div(a.K, aae::kGemvChunk);
    const dim3 grid(chunks, D.CoutPad / 128);
    const int MQ = B <= 4 ? B : 8;
    int smem = 2 * MQ * aae::kGemvChunk * (int)sizeof(float) + 16;
    const bool ticket = tickets && gemv_uses_ticket(enc, D);
    char label[96];
    if (ticket) {
        a.bias = D.bias; a.bn_scale = D.bn_scale; a.bn_shift = D.bn_shift; a.out = out;
        a.tickets = tickets;
        a.nonce = nonce; a.relu = D.relu;
        if (smem < aae::kGemvTicketSmem) smem = aae::kGemvTicketSmem;
        if (MQ == 1) AAE_LAUNCH((aae::dense_gemv_f32_kernel<1, true>), grid, dim3(256), smem, stream, a);
        else if (MQ == 2) AAE_LAUNCH((aae::dense_gemv_f32_kernel<2, true>), grid, dim3(256), smem, stream, a);
        else if (MQ == 3) AAE_LAUNCH((aae::dense_gemv_f32_kernel<3, true>), grid, dim3(256), smem, stream, a);
        else if (MQ == 4) AAE_LAUNCH((aae::dense_gemv_f32_kernel<4, true>), grid, dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::dense_gemv_f32_kernel<8, true>), grid, dim3(256), smem, stream, a);
        snprintf(label, sizeof(label), "dense:dense_gemv_f32_ticket chunks=%d M=%d N=%d K=%d", chunks, B, D.Cout, a.K);
        note_kernel({label, 2.0 * B * (double)D.K() * D.Cout});
        AAE_HIP_TRY(hipGetLastError());
        return tm.mark();
    }
    a.bias = nullptr; a.bn_scale = nullptr; a.bn_shift = nullptr; a.out = nullptr; a.tickets = nullptr; a.nonce = 0; a.relu = 0;
    if (MQ == 1) AAE_LAUNCH((aae::dense_gemv_f32_kernel<1>), grid, dim3(256), smem, stream, a);
    else if (MQ == 2) AAE_LAUNCH((aae::dense_gemv_f32_kernel<2>), grid, dim3(256), smem, stream, a);
    else if (MQ == 3) AAE_LAUNCH((aae::dense_gemv_f32_kernel<3>), grid, dim3(256), smem, stream, a);
    else if (MQ == 4) AAE_LAUNCH((aae::dense_gemv_f32_kernel<4>), grid, dim3(256), smem, stream, a);
    else AAE_LAUNCH((aae::dense_gemv_f32_kernel<8>), grid, dim3(256), smem, stream, a);
    snprintf(label, sizeof(label), "dense:dense_gemv_f32 chunks=%d M=%d N=%d K=%d", chunks, B, D.Cout, a.K);
    note_kernel({label, 2.0 * B * (double)D.K() * D.Cout});
    AAE_HIP_TRY(hipGetLastError());
    if (int rc = tm.mark()) return rc;
    aae::SplitKReduceArgs r;
    r.partial = partial; r.bias = D.bias; r.bn_scale = D.bn_scale; r.bn_shift = D.bn_shift; r.out = out;
    r.MN = (long long)B * D.Cout; r.Cout = D.Cout; r.splits = chunks; r.relu = D.relu; r.out_planes = 0; r.out_scale = 1.f;
    launch_splitk_reduce(r, stream, enc->reduce_small != 0);
    note_kernel({"dense:splitk_reduce", 0.0});
    AAE_HIP_TRY(hipGetLastError());
    return tm.mark();
}

// One extra ticket range the first kernel of the forward prepares for a launch that FOLLOWS the encoder on the same
// stream (the single-launch codebook scan of aae_encode_nn).
struct ExtraTicketPrep {
    unsigned long long* words = nullptr;
    int count = 0;
    unsigned nonce = 0;
    // aae_encode_nn, B <= 4, top-1 on the fp32 stream scan: the scan itself, ready to run as the last phase of the persistent
    // per-detection launch (tickets / nonce = the words above)
    bool scan_ready = false;
    aae::ScanArgs scan;
};

// The persistent per-detection launch (detect_chain.h) serves a forward when every layer behind the first runs the
// wave-split-K kernel in one of its three 4-wave / depth-2 shapes, the dense layer is the ticketed GEMV, and each layer
// output has its own buffer.
static int chain_shape_code(const WaveKPlan& w) { return w.MT == 1 ? 0 : (w.NT == 1 ? 1 : 2); }

// the instantiated (batch class, shape sequence) kernels: what plan_wavek gives the reference network at B = 1, 2, 3, 4
struct ChainVariant { int mq, s0, s1, s2; };
static const ChainVariant kChainVariants[] = {{1, 0, 0, 0}, {2, 1, 0, 0}, {4, 0, 1, 0}, {4, 2, 1, 0}};

static bool chain_eligible(const aae_encoder* enc, int B, const std::vector<WaveKPlan>& plans, bool dense_gemv_ticket) {
    const size_t nl = enc->layers.size();
    if (!enc->detect_chain || B > 4 || nl != (size_t)aae::kChainConv + 1 || enc->compact_workspace || !dense_gemv_ticket) return false;
    if (enc->wavek_ablate || (enc->wavek_timeline && !enc->chain_timeline)) return false;   // (profiling aids of the stand-alone launches)
    if (enc->wavek_spread != 3) return false;                  // (the phases are compiled with the default schedules)
    for (size_t li = 1; li < nl; ++li) {
        const WaveKPlan& w = plans[li];
        if (!w.use || w.waves != 4 || w.depth != 2 || w.tail_tiles > 0 || enc->layers[li].Cout % 4 != 0) return false;
        if (!((w.MT == 1 && w.NT == 1) || (w.MT == 2 && w.NT == 1) || (w.MT == 2 && w.NT == 2))) return false;
    }
    if (enc->dense.Cout % 4 != 0) return false;
    const int mq = B <= 2 ? B : 4;
    for (const ChainVariant& v : kChainVariants)
        if (v.mq == mq && v.s0 == chain_shape_code(plans[1]) && v.s1 == chain_shape_code(plans[2]) && v.s2 == chain_shape_code(plans[3])) return true;
    return false;
}

template <int MQ, int S0, int S1, int S2>
static void launch_chain_t(const aae::DetectChainArgs& a, int grid, hipStream_t stream) {
    (void)hipFuncSetAttribute((const void*)aae::detect_chain_kernel<MQ, S0, S1, S2>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kChainSmem);
    AAE_LAUNCH_RESIDENT((aae::detect_chain_kernel<MQ, S0, S1, S2>), dim3(grid), dim3(256), aae::kChainSmem, stream, a);
}

static int launch_detect_chain(aae_encoder* enc, int B, const std::vector<WaveKPlan>& plans, const std::vector<unsigned>& nonces, unsigned gemv_nonce,
                               unsigned barrier_nonce, const float* act0, unsigned char* base, const Workspace& ws, unsigned long long* tickets, float* z_out,
                               const ExtraTicketPrep* extra, hipStream_t stream, Timer& tm) {
    const size_t nl = enc->layers.size();
    const Layer& D = enc->dense;
    aae::DetectChainArgs a;
    memset(&a, 0, sizeof(a));
    const float* cur = act0;
    double flops = 0.0;
    for (size_t li = 1; li < nl; ++li) {
        const Layer& L = enc->layers[li];
        const WaveKPlan& w = plans[li];
        float* out = reinterpret_cast<float*>(base + ws.act_off[li]);
        a.conv[li - 1] = wavek_args(enc, L, w, cur, B * L.Ho * L.Wo, out, reinterpret_cast<float*>(base + ws.chain_partial_off[li]),
                                    tickets + li * kLayerTicketWords, nonces[li], 0);          // (tag 0: no per-layer stamps ...)
        // ... unless option chain_timeline = 1 + layer asks for the phase stamps of ONE conv layer, kept behind the launch's own stamps
        if (enc->chain_timeline == 1 + (int)li && enc->wavek_timeline && w.num_mt * w.num_nt * w.gsplits <= 256)
            a.conv[li - 1].timeline = enc->wavek_timeline + 256 * aae::kChainStamps;
        flops += 2.0 * (double)B * L.Ho * L.Wo * (double)L.K() * L.Cout;
        cur = out;
    }
    a.dense = gemv_args(D, cur, B, reinterpret_cast<float*>(base + ws.chain_partial_off[nl]));
    a.dense.bias = D.bias; a.dense.bn_scale = D.bn_scale; a.dense.bn_shift = D.bn_shift; a.dense.out = z_out; a.dense.relu = D.relu;
    a.dense.tickets = tickets + kConvTicketBytes / 8; a.dense.nonce = gemv_nonce;
    a.dense_tiles = D.CoutPad / 128;
    a.dense_chunks = ceil_div((int)D.K(), aae::kGemvChunk);
    flops += 2.0 * B * (double)D.K() * D.Cout;
    a.has_scan = (extra && extra->scan_ready) ? 1 : 0;
    if (a.has_scan) {
        a.scan = extra->scan;
        flops += 2.0 * B * (double)a.scan.N * a.scan.J;
    }
    a.barrier.words = tickets + (kConvTicketBytes + kGemvTicketBytes) / 8;
    a.barrier.nonce = barrier_nonce;
    a.timeline = (enc->chain_timeline && enc->wavek_timeline) ? enc->wavek_timeline : nullptr;      // (3 * 512 * 8 stamps: up to 307 blocks x 40)
    if (a.timeline && (size_t)std::min(enc->detect_chain_blocks, enc->cu_count > 0 ? enc->cu_count : enc->detect_chain_blocks) * aae::kChainStamps > 3u * 512u * 8u) a.timeline = nullptr;
    int grid = enc->detect_chain_blocks;
    if (enc->cu_count > 0 && grid > enc->cu_count) grid = enc->cu_count;
    if (grid < 1) grid = 1;
    const int key = (B <= 2 ? B : 4) * 1000 + chain_shape_code(plans[1]) * 100 + chain_shape_code(plans[2]) * 10 + chain_shape_code(plans[3]);
    switch (key) {                                              // (kChainVariants)
        case 1000: launch_chain_t<1, 0, 0, 0>(a, grid, stream); break;
        case 2100: launch_chain_t<2, 1, 0, 0>(a, grid, stream); break;
        case 4010: launch_chain_t<4, 0, 1, 0>(a, grid, stream); break;
        case 4210: launch_chain_t<4, 2, 1, 0>(a, grid, stream); break;
        default: return fail(AAE_ERR_RUNTIME, "no persistent per-detection kernel for batch %d / wave-tile shapes %d", B, key % 1000);
    }
    char label[128];
    snprintf(label, sizeof(label), "chain:detect_chain_f32 B=%d blocks=%d shapes=%d%d%d phases=conv2..conv%zu+dense%s", B, grid, chain_shape_code(plans[1]),
             chain_shape_code(plans[2]), chain_shape_code(plans[3]), nl, a.has_scan ? "+scan" : "");
    note_kernel({label, flops});
    AAE_HIP_TRY(hipGetLastError());
    return tm.mark();
}

static int forward_impl(aae_encoder* enc, const void* x, int x_dtype, int B, float* z_out, void* workspace,
                        size_t ws_bytes, void* stream_v, Timer& tm, const ExtraTicketPrep* extra = nullptr, bool* extra_prepared = nullptr,
                        bool* scan_done = nullptr) {
    if (extra_prepared) *extra_prepared = false;
    if (scan_done) *scan_done = false;
    if (!enc || !x || !z_out) return fail(AAE_ERR_INVALID, "aae_encoder_forward: null argument");
    if (B < 1) return fail(AAE_ERR_INVALID, "aae_encoder_forward: batch %d < 1", B);
    if (x_dtype != AAE_DTYPE_U8 && x_dtype != AAE_DTYPE_F32)
        return fail(AAE_ERR_INVALID, "aae_encoder_forward: x_dtype %d (want AAE_DTYPE_U8 or AAE_DTYPE_F32)", x_dtype);
    const Workspace ws = plan_workspace(enc, B);
    if (ws_bytes < ws.total) return fail(AAE_ERR_WORKSPACE, "workspace %zu B < required %zu B for batch %d", ws_bytes, ws.total, B);
    if (!workspace || ((uintptr_t)workspace & 255)) return fail(AAE_ERR_WORKSPACE, "workspace must be non-null and 256-B aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    unsigned char* base = static_cast<unsigned char*>(workspace);
    float* partial = reinterpret_cast<float*>(base + ws.partial_off);
    unsigned long long* tickets = reinterpret_cast<unsigned long long*>(base + ws.ticket_off);
    unsigned long long* gemv_tickets = tickets + kConvTicketBytes / 8;
    auto layer_tickets = [&](size_t li) { return tickets + li * kLayerTicketWords; };     // li == layers.size(): the dense layer
    RecordScope rec(enc);
    tm.stream = stream;
    if (int rc = tm.mark()) return rc;

    const void* cur = x;
    bool cur_u8 = (x_dtype == AAE_DTYPE_U8);
    t_x3h_last_slot = -1;
    if (runs_split(enc, B)) {
        // this forward's range flag: a ring slot, or -- while the stream is being captured into a graph -- a slot of its own
        hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream, &capture);
        int slot;
        if (capture != hipStreamCaptureStatusNone) {
            std::lock_guard<std::mutex> lk(enc->x3h_mu);
            if (!enc->x3h_free.empty()) {
                slot = enc->x3h_free.back();
                enc->x3h_free.pop_back();
            } else {
                if (enc->x3h_captured >= kX3hCaptured)         // (nothing is consumed by the failing call)
                    return fail(AAE_ERR_UNSUPPORTED, "more than %d f32x3h forwards live in HIP graphs on one encoder handle (aae_encoder_x3h_release_slot returns a destroyed graph's slot)", kX3hCaptured);
                slot = kX3hRing + enc->x3h_captured++;
            }
        } else {
            slot = (int)(enc->x3h_seq.fetch_add(1, std::memory_order_relaxed) % kX3hRing);
        }
        t_x3h_flag = enc->x3h_sat + slot;
        t_x3h_last_slot = slot;
        // f32x3h: conv1 (fp32 MFMA, K = 75) emits fp16 hi/lo planes, every later layer runs the
        // split-precision igemm on planes; only the latent z comes back as fp32.
        for (size_t li = 0; li < enc->layers.size(); ++li) {
            const Layer& L = enc->layers[li];
            float* out = reinterpret_cast<float*>(base + ws.act_off[li]);
            char name[16];
            snprintf(name, sizeof(name), "conv%zu", li + 1);
            int rc;
            if (li == 0) rc = launch_first(enc, L, cur, cur_u8, B, out, true, stream, tm);
            else rc = launch_igemm_x3h(enc, L, cur, B * L.Ho * L.Wo, out, false, partial, stream, tm, name, (int)li);
            if (rc) return rc;
            cur = out;
        }
        return launch_igemm_x3h(enc, enc->dense, cur, B, z_out, true, partial, stream, tm, "dense");
    }

    // ---- plan every layer first: the ticketed launches get their nonces now, so that the first kernel can install them
    const size_t nl = enc->layers.size();
    const Layer& D = enc->dense;
    std::vector<WaveKPlan> plans(nl + 1);
    std::vector<unsigned> nonces(nl + 1, 0u);
    for (size_t li = 0; li < nl; ++li) {
        const Layer& L = enc->layers[li];
        const bool first_mfma = li == 0 && L.kind == KIND_FIRST_MFMA;
        if (!first_mfma && L.kind == KIND_IGEMM && !(li == 0 && cur_u8)) plans[li] = plan_wavek(enc, L, (long long)B * L.Ho * L.Wo, false);
    }
    const bool dense_gemv = D.kind == KIND_IGEMM && B <= gemv_max_batch(enc) && enc->dense_gemv && D.K() % aae::kGemvChunk == 0;
    const bool gemv_ticket = dense_gemv && gemv_uses_ticket(enc, D);
    if (!dense_gemv && D.kind == KIND_IGEMM && enc->wavek_dense) plans[nl] = plan_wavek(enc, D, B, false);
    aae::TicketPrep prep;
    prep.n = 0;
    auto add_prep = [&](unsigned long long* words, int count, unsigned nonce) {
        if (prep.n >= aae::kMaxTicketPrep) return false;
        prep.words[prep.n] = words; prep.count[prep.n] = count; prep.nonce[prep.n] = nonce; ++prep.n;
        return true;
    };
    for (size_t li = 0; li <= nl; ++li)
        if (plans[li].use && (plans[li].gsplits > 1 || plans[li].tail_tiles > 0)) {
            nonces[li] = next_nonce();
            add_prep(layer_tickets(li), plans[li].tail_tiles > 0 ? plans[li].tail_tiles : plans[li].num_mt * plans[li].num_nt, nonces[li]);
        }
    unsigned gemv_nonce = 0;
    if (gemv_ticket) {
        gemv_nonce = next_nonce();
        add_prep(gemv_tickets, (D.CoutPad / 128) * aae::kTicketSlotWords, gemv_nonce);
    }
    const bool extra_listed = extra && extra->words && add_prep(extra->words, extra->count, extra->nonce);
    // the persistent per-detection launch: its grid-barrier words count monotonically inside a launch and must start from
    // (nonce, 0) -- also when a captured graph replays the launch with the SAME nonce.  The first kernel resets them with the
    // other ticket words; where it cannot, a memset in front of the launch does.
    const bool chain = chain_eligible(enc, B, plans, gemv_ticket);
    unsigned long long* barrier_words = tickets + (kConvTicketBytes + kGemvTicketBytes) / 8;
    const unsigned barrier_nonce = chain ? next_nonce() : 0u;
    const bool barrier_listed = chain && add_prep(barrier_words, aae::kGridBarrierWords, barrier_nonce);
    const bool can_prepare = enc->ticket_prep && enc->layers[0].kind == KIND_FIRST_MFMA && prep.n > 0;
    if (extra_prepared) *extra_prepared = can_prepare && extra_listed;

    // ---- per-detection batches: the first layer as its own launch, everything behind it in ONE persistent launch
    if (chain) {
        const Layer& L0 = enc->layers[0];
        float* out0 = reinterpret_cast<float*>(base + ws.act_off[0]);
        int rc;
        if (L0.kind == KIND_FIRST_MFMA) rc = launch_first(enc, L0, cur, cur_u8, B, out0, false, stream, tm, can_prepare ? &prep : nullptr);
        else if (L0.kind == KIND_IGEMM && !cur_u8) {
            if (plans[0].use)
                rc = launch_wavek(enc, L0, plans[0], static_cast<const float*>(cur), B * L0.Ho * L0.Wo, out0, partial, layer_tickets(0), nonces[0],
                                  stream, tm, "conv1", 0);
            else rc = launch_igemm(enc, L0, static_cast<const float*>(cur), B * L0.Ho * L0.Wo, out0, partial, stream, tm, "conv1", 0);
        } else rc = launch_generic(enc, L0, cur, cur_u8, B, out0, stream, tm, "conv1");
        if (rc) return rc;
        if (!(can_prepare && barrier_listed)) AAE_HIP_TRY(hipMemsetAsync(barrier_words, 0, (size_t)aae::kGridBarrierWords * 8, stream));
        rc = launch_detect_chain(enc, B, plans, nonces, gemv_nonce, barrier_nonce, out0, base, ws, tickets, z_out, extra, stream, tm);
        if (rc == AAE_OK && scan_done) *scan_done = extra && extra->scan_ready;
        return rc;
    }

    for (size_t li = 0; li < nl; ++li) {
        const Layer& L = enc->layers[li];
        float* out = reinterpret_cast<float*>(base + ws.act_off[li]);
        char name[16];
        snprintf(name, sizeof(name), "conv%zu", li + 1);
        int rc;
        if (li == 0 && L.kind == KIND_FIRST_MFMA) rc = launch_first(enc, L, cur, cur_u8, B, out, false, stream, tm, can_prepare ? &prep : nullptr);
        else if (L.kind == KIND_IGEMM && !cur_u8) {
            if (plans[li].use)
                rc = launch_wavek(enc, L, plans[li], static_cast<const float*>(cur), B * L.Ho * L.Wo, out, partial, layer_tickets(li), nonces[li],
                                  stream, tm, name, (int)li);
            else rc = launch_igemm(enc, L, static_cast<const float*>(cur), B * L.Ho * L.Wo, out, partial, stream, tm, name, (int)li);
        } else rc = launch_generic(enc, L, cur, cur_u8, B, out, stream, tm, name);
        if (rc) return rc;
        cur = out;
        cur_u8 = false;
    }
    if (dense_gemv)
        return launch_dense_gemv(enc, D, static_cast<const float*>(cur), B, z_out, partial, gemv_ticket ? gemv_tickets : nullptr, gemv_nonce, stream, tm);
    if (plans[nl].use)
        return launch_wavek(enc, D, plans[nl], static_cast<const float*>(cur), B, z_out, partial, layer_tickets(nl), nonces[nl], stream, tm, "dense", 0);
    if (D.kind == KIND_IGEMM) return launch_igemm(enc, D, static_cast<const float*>(cur), B, z_out, partial, stream, tm, "dense");
    return launch_generic(enc, D, cur, false, B, z_out, stream, tm, "dense");
}

// --------------------------------------------------------------- codebook side
struct ScanPlan {
    int nblk, Bpad, Bstride, Jpad, NT;
    bool gemv, stream;
    bool resident_ok;              // query-resident streaming kernel eligible (top-1, no similarity output, stride 1 decided at run time)
    int res_tiles_per_block, res_blocks, res_rh;
    bool topk_fused;               // top-k (2..8) inside the query-resident kernel: no [B][N] similarity matrix
    int cand_chunks;               // candidate lists per query that topk_merge_kernel merges
    size_t ticket_off, q_off, qp_off, pval_off, pidx_off, cs_off, cand_off, prune_off, total;
};

// answers of a top-1 stream scan that finishes inside its own launch (scan_ticket_finish)
struct ScanTicketOut {
    int64_t* idx_out = nullptr;
    float* score_out = nullptr;
    int idx_scale = 1;
    unsigned nonce = 0;            // != 0: the ticket words were prepared with this nonce by an earlier kernel on the stream
};

static ScanPlan plan_scan(const aae_codebook* cb, int B, int topk) {
    ScanPlan s;
    s.nblk = ceil_div(cb->N, 128);
    s.Jpad = 128;
    s.stream = B <= 4 && (cb->scan_mode == AAE_SCAN_STREAM || cb->scan_mode == AAE_SCAN_AUTO);
    s.gemv = B <= 4 && cb->scan_mode == AAE_SCAN_GEMV;
    s.NT = B <= 32 ? 1 : (B <= 64 ? 2 : 4);
    s.Bpad = (int)align_up((size_t)B, (size_t)(32 * s.NT));
    if (cb->dtype == AAE_DTYPE_BF16) {           // B <= 4: HBM-streaming kernel (256 rows per block); else 64 queries per MFMA pass
        s.gemv = false;
        s.stream = B <= 4 && cb->scan_mode != AAE_SCAN_MFMA;
        s.Bpad = (int)align_up((size_t)B, (size_t)aae::kScanBf16QC);
        if (s.stream) s.nblk = ceil_div(cb->N, 256);
    }
    s.Bstride = s.Bpad;
    // B > 4: queries resident in registers, codebook streamed (codebook_scan_resident.h); about one block (8 waves)
    // per CU: row ranges x 128-query chunks.  Measured against the tile-resident kernels (whole nn call): B=8 0.035 ->
    // 0.024 ms, B=32 0.036 -> 0.024, B=256 0.106 -> 0.063; bf16 4x codebook B=32 0.083 -> 0.034, B=256 0.25 -> 0.078
    s.resident_ok = false; s.res_tiles_per_block = 0; s.res_blocks = 0; s.res_rh = 2;
    if (cb->scan_mode == AAE_SCAN_AUTO && cb->J == 128 && !s.stream && !s.gemv && B > 4) {
        // B > 128: 256 queries per block, every wave all rows of a tile (the codebook streamed once per 256 queries);
        // B <= 32, arg-max: FOUR waves share the rows of a tile for the one query group (with two, two of the CU's four matrix pipes sat
        // idle: 20.7 us per query of the 47 MB default codebook at any B <= 32, now 15.8; with two query groups -- 33 ... 64 queries -- all
        // eight waves are busy either way and the 128-row fp32 tiles in two LDS images measured slower, 23.3 against 21.3)
        s.res_rh = s.Bpad > 128 ? 1 : ((B <= 32 && topk == 1 && cb->scan_rh4) ? 4 : 2);
        const int tile_rows = (cb->dtype == AAE_DTYPE_BF16 || s.res_rh == 4) ? 128 : 64;
        const int ntiles = ceil_div(cb->N, tile_rows);
        const int qchunks = ceil_div(s.Bpad, 256 / s.res_rh);
        int row_blocks = (cb->cu_count > 0 ? cb->cu_count : 256) / qchunks;
        if (row_blocks < 1) row_blocks = 1;
        s.res_tiles_per_block = ceil_div(ntiles, row_blocks);
        if (s.res_tiles_per_block < 128 / tile_rows) s.res_tiles_per_block = 128 / tile_rows;   // never more row blocks than nblk
        s.res_blocks = ceil_div(ntiles, s.res_tiles_per_block);
        s.resident_ok = s.res_blocks <= s.nblk;          // the partial buffers are sized for nblk row blocks
    }
    // top-k (2 <= k <= 8) on the query-resident kernel: per-lane sorted lists instead of the [B][N] similarity matrix
    s.topk_fused = topk >= 2 && topk <= 8 && s.resident_ok;     // (AAE_SCAN_MFMA keeps the similarity-matrix path for A/B)
    size_t off = 0;
    s.ticket_off = off; off += align_up((size_t)aae::kTicketSlotWords * 8, 256);   // block_ticket_arrive words of the single-launch stream scan
    s.q_off = off;    off += align_up((size_t)B * cb->J * sizeof(float), 256);
    s.qp_off = off;   off += align_up((size_t)s.Jpad * s.Bpad * 6, 256);   // fp32 packing: 4 B/elem; bf16: 3 terms x 2 B
    // block partials: one row per scan block -- or per block of the persistent per-detection launch, whose grid (one block per
    // CU, detect_chain.h) can exceed the block count of a small codebook
    const int partial_rows = s.stream ? std::max(s.nblk, kChainMaxBlocks) : s.nblk;
    s.pval_off = off; off += align_up((size_t)partial_rows * s.Bstride * sizeof(float), 256);
    s.pidx_off = off; off += align_up((size_t)partial_rows * s.Bstride * sizeof(int), 256);
    s.cs_off = off;
    if (topk > 1 && !s.topk_fused) off += align_up((size_t)B * cb->N * sizeof(float), 256);
    s.cand_off = off;
    s.cand_chunks = s.topk_fused ? s.res_blocks : ceil_div(cb->N, aae::kTopKChunk);
    if (topk > 1) off += 2 * align_up((size_t)B * s.cand_chunks * topk * sizeof(float), 256);
    s.prune_off = off;                                  // shared bound words of the pruned top-k scan
    if (s.topk_fused) off += align_up((size_t)aae::kPruneReplicas * s.Bpad * aae::kPruneGroups * sizeof(int), 256);
    s.total = off;
    return s;
}

template <int NT>
static void launch_scan_mfma_t(const aae::ScanArgs& a, bool upright, int nblk, hipStream_t stream) {
    constexpr int smem = aae::scan_mfma_smem<NT>();
    if (upright) {
        (void)hipFuncSetAttribute((const void*)aae::scan_mfma_kernel<NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        AAE_LAUNCH((aae::scan_mfma_kernel<NT, true>), dim3(nblk), dim3(256), smem, stream, a);
    } else {
        (void)hipFuncSetAttribute((const void*)aae::scan_mfma_kernel<NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        AAE_LAUNCH((aae::scan_mfma_kernel<NT, false>), dim3(nblk), dim3(256), smem, stream, a);
    }
}

// (the similarity output is a template parameter of the stream kernels: as a run-time branch inside the row loop it split the
// loop into 64 basic blocks and serialised the cross-lane reductions)
template <int NQ>
static void launch_scan_stream_t(const aae::ScanArgs& a, bool upright, int nblk, hipStream_t stream) {
    const int smem = NQ * 128 * (int)sizeof(float) + aae::kScanTicketSmem;
    if (a.cs) {
        if (upright) AAE_LAUNCH((aae::scan_stream_kernel<NQ, true, true>), dim3(nblk), dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::scan_stream_kernel<NQ, false, true>), dim3(nblk), dim3(256), smem, stream, a);
    } else {
        if (upright) AAE_LAUNCH((aae::scan_stream_kernel<NQ, true, false>), dim3(nblk), dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::scan_stream_kernel<NQ, false, false>), dim3(nblk), dim3(256), smem, stream, a);
    }
}
template <int NQ>
static void launch_scan_walk_t(const aae::ScanArgs& a, bool upright, int blocks, hipStream_t stream) {
    const int smem = 8 * NQ * (int)sizeof(float) + aae::kScanTicketSmem;
    if (a.cs) {
        if (upright) AAE_LAUNCH((aae::scan_stream_walk_kernel<NQ, true, true>), dim3(blocks), dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::scan_stream_walk_kernel<NQ, false, true>), dim3(blocks), dim3(256), smem, stream, a);
    } else {
        if (upright) AAE_LAUNCH((aae::scan_stream_walk_kernel<NQ, true, false>), dim3(blocks), dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::scan_stream_walk_kernel<NQ, false, false>), dim3(blocks), dim3(256), smem, stream, a);
    }
}
template <int NQ>
static void launch_scan_stream_bf16_t(const aae::ScanArgs& a, bool upright, int nblk, hipStream_t stream) {
    const int smem = NQ * 256 * (int)sizeof(float) + aae::kScanTicketSmem;
    if (a.cs) {
        if (upright) AAE_LAUNCH((aae::scan_stream_bf16_kernel<NQ, true, true>), dim3(nblk), dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::scan_stream_bf16_kernel<NQ, false, true>), dim3(nblk), dim3(256), smem, stream, a);
    } else {
        if (upright) AAE_LAUNCH((aae::scan_stream_bf16_kernel<NQ, true, false>), dim3(nblk), dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::scan_stream_bf16_kernel<NQ, false, false>), dim3(nblk), dim3(256), smem, stream, a);
    }
}

template <bool BF16, int K, int RH, bool NORM = false>
static void launch_scan_resident_t(const aae::ScanResidentArgs& a, dim3 grid, hipStream_t stream) {
    constexpr int smem = aae::scan_resident_smem<BF16, RH>();
    (void)hipFuncSetAttribute((const void*)aae::scan_resident_kernel<BF16, K, RH, NORM>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    AAE_LAUNCH((aae::scan_resident_kernel<BF16, K, RH, NORM>), grid, dim3(aae::kScanResidentThreads), smem, stream, a);
}
template <bool BF16, int RH>
static void launch_scan_resident_k(const aae::ScanResidentArgs& a, dim3 grid, hipStream_t stream) {
    if (a.k <= 1 && a.z) launch_scan_resident_t<BF16, 0, RH, true>(a, grid, stream);      // the block normalises its own queries
    else if (a.k <= 1) launch_scan_resident_t<BF16, 0, RH>(a, grid, stream);
    else if (a.k <= 2) launch_scan_resident_t<BF16, 2, RH>(a, grid, stream);    // list slots: the smallest instantiated K >= k
    else if (a.k <= 4) launch_scan_resident_t<BF16, 4, RH>(a, grid, stream);
    else if (a.k == 5) launch_scan_resident_t<BF16, 5, RH>(a, grid, stream);
    else launch_scan_resident_t<BF16, 8, RH>(a, grid, stream);
}

// topk == 1: block partials (pval, pidx) for argmax_reduce_kernel; topk 2..8: candidate lists for topk_merge_kernel
static int launch_scan_resident(const aae_codebook* cb, const void* qp, int B, const ScanPlan& s, unsigned char* base, hipStream_t stream,
                                int topk = 1, const float* raw_z = nullptr, const ScanTicketOut* fin = nullptr) {
    aae::ScanResidentArgs a;
    a.E = cb->E; a.e_bytes = (unsigned)((size_t)cb->N * cb->J * (cb->dtype == AAE_DTYPE_BF16 ? 2 : 4));
    a.qp = qp;
    a.z = raw_z;
    a.pval = reinterpret_cast<float*>(base + s.pval_off);
    a.pidx = reinterpret_cast<int*>(base + s.pidx_off);
    a.N = cb->N; a.B = B; a.Bpad = s.Bpad; a.Bstride = s.Bstride; a.tiles_per_block = s.res_tiles_per_block;
    const dim3 grid(s.res_blocks, ceil_div(s.Bpad, 256 / s.res_rh));
    a.k = topk > 1 ? topk : 0;
    if (fin && topk == 1 && grid.y == 1) {           // the last row block to arrive answers (no argmax_reduce launch)
        a.tickets = reinterpret_cast<unsigned long long*>(base + s.ticket_off); a.nonce = fin->nonce ? fin->nonce : next_nonce();
        a.idx_out = reinterpret_cast<long long*>(fin->idx_out); a.score_out = fin->score_out; a.idx_scale = fin->idx_scale;
    }
    if (topk > 1) {
        a.cand_v = reinterpret_cast<float*>(base + s.cand_off);
        a.cand_i = reinterpret_cast<int*>(base + s.cand_off + align_up((size_t)B * s.cand_chunks * topk * sizeof(float), 256));
        if (cb->topk_prune) a.prune = reinterpret_cast<int*>(base + s.prune_off);      // (reset by the normalise kernel in front)
    }
    const bool bf16 = cb->dtype == AAE_DTYPE_BF16;
    if (s.res_rh == 4) {                           // (arg-max only: plan_scan)
        if (bf16 && a.z) launch_scan_resident_t<true, 0, 4, true>(a, grid, stream);
        else if (bf16) launch_scan_resident_t<true, 0, 4>(a, grid, stream);
        else if (a.z) launch_scan_resident_t<false, 0, 4, true>(a, grid, stream);
        else launch_scan_resident_t<false, 0, 4>(a, grid, stream);
    } else if (bf16 && s.res_rh == 1) launch_scan_resident_k<true, 1>(a, grid, stream);
    else if (bf16) launch_scan_resident_k<true, 2>(a, grid, stream);
    else if (s.res_rh == 1) launch_scan_resident_k<false, 1>(a, grid, stream);
    else launch_scan_resident_k<false, 2>(a, grid, stream);
    AAE_HIP_TRY(hipGetLastError());
    return AAE_OK;
}

// *partial_rows: how many [Bstride]-rows of (pval, pidx) the arg-max reduce has to look at
static int run_scan(aae_codebook* cb, const float* z, int B, int col_stride, float* cs_out, const ScanPlan& s,
                    unsigned char* base, hipStream_t stream, int* partial_rows = nullptr, const ScanTicketOut* fin = nullptr, int topk = 1) {
    float* q = reinterpret_cast<float*>(base + s.q_off);
    float* qp = reinterpret_cast<float*>(base + s.qp_off);
    const bool resident = s.resident_ok && cs_out == nullptr && col_stride == 1;
    if (partial_rows) *partial_rows = resident ? s.res_blocks : s.nblk;
    // arg-max on the query-resident kernel: the scan normalises the queries itself (no l2norm_pack launch in front)
    const ScanTicketOut* rfin = (resident && topk == 1 && s.res_rh == 4) ? fin : nullptr;     // (nn_impl passes fin for these only when the mode asks)
    if (resident && topk == 1 && cb->scan_fused_norm && ((uintptr_t)z & 15) == 0) return launch_scan_resident(cb, nullptr, B, s, base, stream, 1, z, rfin);
    if (cb->dtype == AAE_DTYPE_BF16 && s.stream) {
        aae::ScanArgs a;
        a.z = z; a.e_bytes = (unsigned)((size_t)cb->N * cb->J * 2);
        a.E = cb->E; a.q = nullptr; a.qp = nullptr;
        a.pval = reinterpret_cast<float*>(base + s.pval_off);
        a.pidx = reinterpret_cast<int*>(base + s.pidx_off);
        a.cs = cs_out;
        a.N = cb->N; a.J = cb->J; a.Jpad = s.Jpad; a.B = B; a.Bpad = s.Bpad; a.Bstride = s.Bstride;
        a.col_stride = col_stride;
        if (fin) {
            a.tickets = reinterpret_cast<unsigned long long*>(base + s.ticket_off); a.nonce = fin->nonce ? fin->nonce : next_nonce();
            a.idx_out = reinterpret_cast<long long*>(fin->idx_out); a.score_out = fin->score_out; a.idx_scale = fin->idx_scale;
        }
        const bool up = col_stride > 1;
        if (B == 1) launch_scan_stream_bf16_t<1>(a, up, s.nblk, stream);
        else if (B == 2) launch_scan_stream_bf16_t<2>(a, up, s.nblk, stream);
        else launch_scan_stream_bf16_t<4>(a, up, s.nblk, stream);
        AAE_HIP_TRY(hipGetLastError());
        return AAE_OK;
    }
    if (cb->dtype == AAE_DTYPE_BF16) {
        aae::L2NormBf16Args n;
        n.z = z; n.qp3 = reinterpret_cast<unsigned short*>(qp); n.B = B; n.J = cb->J; n.Jpad = 128; n.Bpad = s.Bpad;
        if (resident && topk > 1 && cb->topk_prune) n.prune = reinterpret_cast<int*>(base + s.prune_off);
        AAE_LAUNCH((aae::l2norm_pack_bf16x3_kernel), dim3(ceil_div(s.Bpad, 4)), dim3(256), 0, stream, n);
        AAE_HIP_TRY(hipGetLastError());
        if (resident) return launch_scan_resident(cb, n.qp3, B, s, base, stream, topk, nullptr, rfin);
        aae::ScanBf16Args a;
        a.E = reinterpret_cast<const unsigned short*>(cb->E);
        a.e_bytes = (unsigned)((size_t)cb->N * cb->J * 2);
        a.qp3 = n.qp3;
        a.pval = reinterpret_cast<float*>(base + s.pval_off);
        a.pidx = reinterpret_cast<int*>(base + s.pidx_off);
        a.cs = cs_out;
        a.N = cb->N; a.B = B; a.Bpad = s.Bpad; a.Bstride = s.Bstride; a.col_stride = col_stride;
        if (col_stride > 1) {
            (void)hipFuncSetAttribute((const void*)aae::scan_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kScanBf16Smem);
            AAE_LAUNCH((aae::scan_bf16_kernel<true>), dim3(s.nblk), dim3(256), aae::kScanBf16Smem, stream, a);
        } else {
            (void)hipFuncSetAttribute((const void*)aae::scan_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kScanBf16Smem);
            AAE_LAUNCH((aae::scan_bf16_kernel<false>), dim3(s.nblk), dim3(256), aae::kScanBf16Smem, stream, a);
        }
        AAE_HIP_TRY(hipGetLastError());
        return AAE_OK;
    }
    if (!s.stream) {                     // the stream kernel normalises the queries itself
        aae::L2NormArgs n;
        n.z = z; n.q = q; n.qp = s.gemv ? nullptr : qp; n.B = B; n.J = cb->J; n.Jpad = s.Jpad; n.Bpad = s.gemv ? B : s.Bpad;
        if (resident && topk > 1 && cb->topk_prune) n.prune = reinterpret_cast<int*>(base + s.prune_off);
        AAE_LAUNCH((aae::l2norm_pack_kernel), dim3(ceil_div(n.Bpad, 4)), dim3(256), 0, stream, n);
        AAE_HIP_TRY(hipGetLastError());
    }
    if (resident) return launch_scan_resident(cb, qp, B, s, base, stream, topk, nullptr, rfin);

    aae::ScanArgs a;
    a.z = z; a.e_bytes = (unsigned)((size_t)cb->N * cb->J * sizeof(float));
    a.E = cb->E; a.q = q; a.qp = qp;
    a.pval = reinterpret_cast<float*>(base + s.pval_off);
    a.pidx = reinterpret_cast<int*>(base + s.pidx_off);
    a.cs = cs_out;
    a.N = cb->N; a.J = cb->J; a.Jpad = s.Jpad; a.B = B; a.Bpad = s.Bpad; a.Bstride = s.Bstride;
    a.col_stride = col_stride;
    if (fin && s.stream) {
        a.tickets = reinterpret_cast<unsigned long long*>(base + s.ticket_off); a.nonce = fin->nonce ? fin->nonce : next_nonce();
        a.idx_out = reinterpret_cast<long long*>(fin->idx_out); a.score_out = fin->score_out; a.idx_scale = fin->idx_scale;
    }
    const bool upright = col_stride > 1;
    if (s.stream && cb->scan_walk) {
        // one block per CU, never more blocks than 128-row groups (the partial buffers are sized for those)
        const int blocks = std::min(cb->cu_count > 0 ? cb->cu_count : 256, s.nblk);
        if (partial_rows) *partial_rows = blocks;
        if (B == 1) launch_scan_walk_t<1>(a, upright, blocks, stream);
        else if (B == 2) launch_scan_walk_t<2>(a, upright, blocks, stream);
        else if (B == 3) launch_scan_walk_t<3>(a, upright, blocks, stream);
        else launch_scan_walk_t<4>(a, upright, blocks, stream);
    } else if (s.stream) {
        if (B == 1) launch_scan_stream_t<1>(a, upright, s.nblk, stream);
        else if (B == 2) launch_scan_stream_t<2>(a, upright, s.nblk, stream);
        else if (B == 3) launch_scan_stream_t<3>(a, upright, s.nblk, stream);
        else launch_scan_stream_t<4>(a, upright, s.nblk, stream);
    } else if (s.gemv) {
        const int smem = 2 * 4 * 4 * (int)sizeof(float);
        if (upright) AAE_LAUNCH((aae::scan_gemv_kernel<4, true>), dim3(s.nblk), dim3(256), smem, stream, a);
        else AAE_LAUNCH((aae::scan_gemv_kernel<4, false>), dim3(s.nblk), dim3(256), smem, stream, a);
    } else if (s.NT == 1) launch_scan_mfma_t<1>(a, upright, s.nblk, stream);
    else if (s.NT == 2) launch_scan_mfma_t<2>(a, upright, s.nblk, stream);
    else launch_scan_mfma_t<4>(a, upright, s.nblk, stream);
    AAE_HIP_TRY(hipGetLastError());
    return AAE_OK;
}

}  // namespace aae_host

// =============================================================== C ABI
extern "C" {

int aae_abi_version(void) { return AAE_ABI_VERSION; }
const char* aae_last_error(void) { return aae_host::g_last_error.c_str(); }

int aae_encoder_create(const aae_encoder_desc* d, const void* const* hw, int n_weights, aae_encoder** out) {
    using namespace aae_host;
    if (!d || !hw || !out) return fail(AAE_ERR_INVALID, "aae_encoder_create: null argument");
    if (d->num_layers < 1 || d->num_layers > AAE_MAX_LAYERS)
        return fail(AAE_ERR_INVALID, "num_layers %d outside [1,%d]", d->num_layers, AAE_MAX_LAYERS);
    if (d->in_h < 1 || d->in_w < 1 || d->in_c < 1 || d->kernel_size < 1 || d->latent_size < 1)
        return fail(AAE_ERR_INVALID, "non-positive shape in encoder desc");
    const int per_layer = d->batch_norm ? 6 : 2;
    if (n_weights != d->num_layers * per_layer + 2)
        return fail(AAE_ERR_INVALID, "expected %d weight arrays, got %d", d->num_layers * per_layer + 2, n_weights);
    for (int i = 0; i < n_weights; ++i)
        if (!hw[i]) return fail(AAE_ERR_INVALID, "weight array %d is null", i);

    aae_encoder* enc = new aae_encoder();
    enc->desc = *d;
    auto bail = [&](int rc) { aae_encoder_destroy(enc); return rc; };

    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) enc->cu_count = cus;
    }
    float lut[256];
    for (int v = 0; v < 256; ++v) lut[v] = (float)((double)v / 255.0);   // float64 quotient, float32 feed cast
    if (int rc = upload(enc, lut, 256, &enc->lut)) return bail(rc);
    {
        const std::vector<float> zeros(kX3hRing + kX3hCaptured, 0.f);
        float* flags = nullptr;
        if (int rc = upload(enc, zeros.data(), zeros.size(), &flags)) return bail(rc);
        enc->x3h_sat = reinterpret_cast<int*>(flags);
    }

    int H = d->in_h, W = d->in_w, C = d->in_c, wi = 0;
    const float eps = d->bn_eps > 0.f ? d->bn_eps : 1e-3f;
    for (int li = 0; li < d->num_layers; ++li) {
        Layer L;
        L.index = li;
        L.H = H; L.W = W; L.Cin = C; L.Cout = d->num_filters[li]; L.KS = d->kernel_size; L.S = d->strides[li];
        if (L.Cout < 1 || L.S < 1) return bail(fail(AAE_ERR_INVALID, "layer %d: filters %d stride %d", li, L.Cout, L.S));
        same_pad(H, L.KS, L.S, &L.Ho, &L.pt);
        same_pad(W, L.KS, L.S, &L.Wo, &L.pl);
        L.CoutPad = (int)align_up((size_t)L.Cout, 128);
        const float* k = static_cast<const float*>(hw[wi++]);
        const float* b = static_cast<const float*>(hw[wi++]);
        if (int rc = upload(enc, k, (size_t)L.K() * L.Cout, &L.w_hwio)) return bail(rc);
        if (int rc = upload(enc, b, L.Cout, &L.bias)) return bail(rc);
        if (d->batch_norm) {
            const float* g = static_cast<const float*>(hw[wi++]);
            const float* be = static_cast<const float*>(hw[wi++]);
            const float* mu = static_cast<const float*>(hw[wi++]);
            const float* var = static_cast<const float*>(hw[wi++]);
            std::vector<float> sc(L.Cout), sh(L.Cout);
            for (int c = 0; c < L.Cout; ++c) {       // tf.nn.batch_normalization: inv = rsqrt(var+eps)*gamma
                const float inv = (1.0f / sqrtf(var[c] + eps)) * g[c];
                sc[c] = inv;
                sh[c] = be[c] - mu[c] * inv;
            }
            if (int rc = upload(enc, sc.data(), L.Cout, &L.bn_scale)) return bail(rc);
            if (int rc = upload(enc, sh.data(), L.Cout, &L.bn_shift)) return bail(rc);
        }
        if (li == 0 && first_layer_instantiated(L.KS, L.Cin)) {
            plan_first_layer(L);
            L.kind = (L.first_smem <= 160 * 1024 && L.first_packable) ? KIND_FIRST_MFMA : KIND_GENERIC;
        }
        if (L.kind == KIND_GENERIC && L.Cin % 32 == 0) {
            L.kind = KIND_IGEMM;
            const std::vector<float> packed = pack_weights(k, L.KS * L.KS, L.Cin, L.Cout, L.CoutPad);
            if (int rc = upload(enc, packed.data(), packed.size(), &L.wp)) return bail(rc);
            const std::vector<unsigned short> p16 = pack_weights_x3h(k, L.KS * L.KS, L.Cin, L.Cout, L.CoutPad, &L.w_shift);
            if (int rc = upload(enc, reinterpret_cast<const float*>(p16.data()), p16.size() / 2, reinterpret_cast<float**>(&L.wp16))) return bail(rc);
        }
        enc->layers.push_back(L);
        H = L.Ho; W = L.Wo; C = L.Cout;
    }
    Layer& D = enc->dense;
    D.H = D.W = D.Ho = D.Wo = 1; D.KS = 1; D.S = 1; D.pt = D.pl = 0; D.relu = 0;
    D.Cin = H * W * C;                       // tf.layers.flatten, NHWC row-major
    D.Cout = d->latent_size;
    D.CoutPad = (int)align_up((size_t)D.Cout, 128);
    {
        const float* k = static_cast<const float*>(hw[wi++]);
        const float* b = static_cast<const float*>(hw[wi++]);
        if (int rc = upload(enc, b, D.Cout, &D.bias)) return bail(rc);
        if (D.Cin % 32 == 0) {
            D.kind = KIND_IGEMM;
            const std::vector<float> packed = pack_weights(k, 1, D.Cin, D.Cout, D.CoutPad);
            if (int rc = upload(enc, packed.data(), packed.size(), &D.wp)) return bail(rc);
            const std::vector<unsigned short> p16 = pack_weights_x3h(k, 1, D.Cin, D.Cout, D.CoutPad, &D.w_shift);
            if (int rc = upload(enc, reinterpret_cast<const float*>(p16.data()), p16.size() / 2, reinterpret_cast<float**>(&D.wp16))) return bail(rc);
        } else {
            D.kind = KIND_GENERIC;
            if (int rc = upload(enc, k, (size_t)D.K() * D.Cout, &D.w_hwio)) return bail(rc);
        }
    }
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false, true, false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false, true, false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false, true, false, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false, true, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false, true, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<false, true, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_f32_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_kernel<aae::X3H_OUT_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_kernel<aae::X3H_OUT_PLANES>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_kernel<aae::X3H_OUT_PARTIAL>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::x3h_dma_smem<4>());
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::x3h_dma_smem<4>());
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::x3h_dma_smem<4>());
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PLANES, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    (void)hipFuncSetAttribute((const void*)aae::conv_igemm_x3h_dma_kernel<aae::X3H_OUT_PARTIAL>, hipFuncAttributeMaxDynamicSharedMemorySize, aae::kConvIgemmSmem);
    if (enc->layers[0].kind == KIND_FIRST_MFMA) {
        const int sm = enc->layers[0].first_smem;
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 3, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 3, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 1, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 1, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 3, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 1, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 3, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 1, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 3, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 3, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
        (void)hipFuncSetAttribute((const void*)aae::conv_first_f32_kernel<5, 1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sm);
    }
    *out = enc;
    return AAE_OK;
}

void aae_encoder_destroy(aae_encoder* enc) {
    if (!enc) return;
    for (void* p : enc->allocations) (void)hipFree(p);
    delete enc;
}

int aae_encoder_set_option(aae_encoder* enc, const char* name, int value) {
    using namespace aae_host;
    if (!enc || !name) return fail(AAE_ERR_INVALID, "aae_encoder_set_option: null argument");
    if (!strcmp(name, "splitk_min_base_blocks")) enc->splitk_min_base_blocks = value;
    else if (!strcmp(name, "splitk_target_blocks")) enc->splitk_target_blocks = value;
    else if (!strcmp(name, "reduce_small")) enc->reduce_small = value ? 1 : 0;
    else if (!strcmp(name, "igemm_stagger")) enc->igemm_stagger = value;
    else if (!strcmp(name, "x3h_dma")) enc->x3h_dma = value ? 1 : 0;
    else if (!strcmp(name, "x3h_wide256")) enc->x3h_wide256 = value ? 1 : 0;
    else if (!strcmp(name, "x3h_min_tiles")) enc->x3h_min_tiles = value < 0 ? 0 : value;
    else if (!strcmp(name, "x3h_wide256_min_blocks")) enc->x3h_wide256_min_blocks = value < 1 ? 1 : value;
    else if (!strcmp(name, "x3h_wide_min_blocks")) enc->x3h_wide_min_blocks = value < 0 ? 0 : value;
    else if (!strcmp(name, "igemm_dma")) enc->igemm_dma = value ? 1 : 0;
    else if (!strcmp(name, "igemm_breg")) enc->igemm_breg = value ? 1 : 0;
    else if (!strcmp(name, "dense_gemv")) enc->dense_gemv = value ? 1 : 0;
    else if (!strcmp(name, "dense_gemv_max_batch")) enc->dense_gemv_max_batch = value;
    else if (!strcmp(name, "wavek_tail_split")) enc->wavek_tail_split = value ? 1 : 0;
    else if (!strcmp(name, "planner_cost_min_batch")) enc->planner_cost_min_batch = value < 1 ? 1 : value;
    else if (!strcmp(name, "planner_cost_batch3")) enc->planner_cost_batch3 = value ? 1 : 0;
    else if (!strcmp(name, "wavek_eff64x32_pct")) enc->wavek_eff64x32_pct = value < 30 ? 30 : (value > 100 ? 100 : value);
    else if (!strcmp(name, "wavek_g_boost")) enc->wavek_g_boost = value < 1 ? 1 : (value > 4 ? 4 : value);
    else if (!strcmp(name, "wavek_force_tail_tiles")) enc->wavek_force_tail_tiles = value < 0 ? 0 : value;
    else if (!strcmp(name, "wavek_force_tail_g")) enc->wavek_force_tail_g = value < 2 ? 2 : value;
    else if (!strcmp(name, "gemv_ticket")) enc->gemv_ticket = value ? 1 : 0;
    else if (!strcmp(name, "wavek")) enc->wavek = value ? 1 : 0;
    else if (!strcmp(name, "wavek_dense")) enc->wavek_dense = value ? 1 : 0;
    else if (!strcmp(name, "wavek_ablate")) enc->wavek_ablate = value;
    else if (!strcmp(name, "wavek_balance")) enc->wavek_balance = value ? 1 : 0;
    else if (!strcmp(name, "planner_cost_model")) enc->planner_cost_model = value ? 1 : 0;
    else if (!strcmp(name, "ticket_prep")) enc->ticket_prep = value ? 1 : 0;
    else if (!strcmp(name, "multi_group_plan")) enc->multi_group_plan = value ? 1 : 0;
    else if (!strcmp(name, "multi_force_shape")) enc->multi_force_shape = value;
    else if (!strcmp(name, "multi_force_g")) enc->multi_force_g = value;
    else if (!strcmp(name, "detect_chain")) {
        if (value) {
            // the persistent launch's grid barrier needs EVERY block resident: refuse the option unless the runtime confirms that one
            // 256-thread block with the chain's LDS footprint fits a compute unit and the device's CU count is known (a plain launch
            // of an over-sized grid would spin until its bounded wait traps).  A CU mask smaller than the device is not detectable
            // from here: the option stays opt-in.
            if (enc->cu_count <= 0) return fail(AAE_ERR_UNSUPPORTED, "detect_chain: the device's compute-unit count is unknown");
            int per_cu = 0;
            const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)aae::detect_chain_kernel<1, 0, 0, 0>, 256, aae::kChainSmem);
            if (e != hipSuccess || per_cu < 1)
                return fail(AAE_ERR_UNSUPPORTED, "detect_chain: the runtime does not confirm residency of the persistent launch (%s, %d blocks per CU)",
                            e == hipSuccess ? "ok" : hipGetErrorString(e), per_cu);
        }
        enc->detect_chain = value ? 1 : 0;
    }
    else if (!strcmp(name, "detect_chain_blocks")) enc->detect_chain_blocks = value < 1 ? 1 : (value > aae_host::kChainMaxBlocks ? aae_host::kChainMaxBlocks : value);
    else if (!strcmp(name, "compact_workspace")) enc->compact_workspace = value ? 1 : 0;
    else if (!strcmp(name, "chain_timeline")) {
        if (value && !enc->wavek_timeline) {
            void* p = nullptr;
            AAE_HIP_TRY(hipMalloc(&p, 3 * 512 * 8 * sizeof(long long)));
            enc->allocations.push_back(p);
            enc->wavek_timeline = static_cast<long long*>(p);
        }
        enc->chain_timeline = value < 0 ? 0 : value;           // 1: phase edges of the launch; 1 + l: also the inner stamps of conv layer l (2 = conv2 ...)
        if (!value) enc->wavek_timeline = nullptr;
    }
    else if (!strcmp(name, "wavek_timeline")) {
        if (value && !enc->wavek_timeline) {
            void* p = nullptr;
            AAE_HIP_TRY(hipMalloc(&p, 3 * 512 * 8 * sizeof(long long)));
            enc->allocations.push_back(p);
            enc->wavek_timeline = static_cast<long long*>(p);
        }
        if (!value) enc->wavek_timeline = nullptr;     // (the buffer stays in `allocations` until the handle goes)
    }
    else if (!strcmp(name, "wavek_max_tiles")) enc->wavek_max_tiles = value < 0 ? 0 : (value > aae_host::kWaveKTileCap ? aae_host::kWaveKTileCap : value);
    else if (!strcmp(name, "wavek_narrow_max_tiles")) enc->wavek_narrow_max_tiles = value < 0 ? 0 : value;
    else if (!strcmp(name, "wavek_target_blocks")) enc->wavek_target_blocks = value < 0 ? 0 : (value > 2 * aae_host::kLayerTicketWords ? 2 * aae_host::kLayerTicketWords : value);
    else if (!strcmp(name, "wavek_tiny_max_tiles")) enc->wavek_tiny_max_tiles = value < 0 ? 0 : value;
    else if (!strcmp(name, "wavek_waves")) {
        if (value != 4 && value != 8) return fail(AAE_ERR_INVALID, "wavek_waves %d: 4 or 8", value);
        enc->wavek_waves = value;
    } else if (!strcmp(name, "wavek_pingpong")) enc->wavek_pingpong = value ? 1 : 0;
    else if (!strcmp(name, "wavek_spread")) enc->wavek_spread = value & 3;       // bit 0: 64 x 64 tiles, bit 1: 32 x 32 tiles (two accumulator chains)
    else if (!strcmp(name, "wavek_tiny_waves")) {
        if (value != 4 && value != 8) return fail(AAE_ERR_INVALID, "wavek_tiny_waves %d: 4 or 8", value);
        enc->wavek_tiny_waves = value;
    } else if (!strcmp(name, "wavek_depth")) {
        if (value != 2 && value != 3) return fail(AAE_ERR_INVALID, "wavek_depth %d: 2 or 3", value);
        enc->wavek_depth = value;
    }
    else if (!strcmp(name, "igemm_breg_wide")) enc->igemm_breg_wide = value ? 1 : 0;
    else if (!strcmp(name, "igemm_breg_wide_min_blocks")) enc->igemm_breg_wide_min_blocks = value;
    else if (!strcmp(name, "igemm_breg_min_blocks")) enc->igemm_breg_min_blocks = value;
    else if (!strcmp(name, "first_vec4")) enc->first_vec4 = value ? 1 : 0;
    else if (!strcmp(name, "first_group_split_max_tiles")) enc->first_group_split_max_tiles = value < 0 ? 0 : value;
    else if (!strcmp(name, "first_target_blocks")) enc->first_target_blocks = value < 1 ? 1 : value;
    else if (!strcmp(name, "first_max_tiles_per_block")) enc->first_max_tiles_per_block = value < 1 ? 1 : value;
    else if (!strcmp(name, "x3h_act_shift")) {
        if (value < -8 || value > 12) return fail(AAE_ERR_INVALID, "x3h_act_shift %d outside [-8, 12]", value);
        enc->x3h_act_shift = value;
    } else if (!strcmp(name, "precision")) {
        if (value < 0 || value > 2) return fail(AAE_ERR_INVALID, "precision %d: 0 = fp32, 1 = f32x3h, 2 = f32x3h where it is faster", value);
        if (value != 0) {
            bool ok = enc->layers[0].kind == KIND_FIRST_MFMA && enc->dense.kind == KIND_IGEMM;
            for (size_t i = 1; i < enc->layers.size(); ++i) ok = ok && enc->layers[i].kind == KIND_IGEMM;
            if (!ok) return fail(AAE_ERR_UNSUPPORTED, "f32x3h needs the matrix-core kernels on every layer (first layer 5x5 with C in {1,3}, later Cin %% 32 == 0)");
        }
        enc->precision = value;
    }
    else return fail(AAE_ERR_INVALID, "unknown encoder option '%s'", name);
    return AAE_OK;
}

size_t aae_encoder_workspace_bytes(const aae_encoder* enc, int B) {
    if (!enc || B < 1) return 0;
    return aae_host::plan_workspace(enc, B).total;
}

int aae_encoder_forward(aae_encoder* enc, const void* x, int x_dtype, int B, float* z_out, void* workspace,
                        size_t ws_bytes, void* stream) {
    aae_host::Timer tm;
    return aae_host::forward_impl(enc, x, x_dtype, B, z_out, workspace, ws_bytes, stream, tm);
}

int aae_encoder_forward_timed(aae_encoder* enc, const void* x, int x_dtype, int B, float* z_out, void* workspace,
                              size_t ws_bytes, void* stream, float* kernel_ms, int max_kernels, int* n_kernels) {
    using namespace aae_host;
    if (!kernel_ms || !n_kernels) return fail(AAE_ERR_INVALID, "aae_encoder_forward_timed: null output");
    Timer tm;
    tm.on = true;
    int rc = forward_impl(enc, x, x_dtype, B, z_out, workspace, ws_bytes, stream, tm);
    if (rc == AAE_OK && !tm.ev.empty()) {
        hipError_t e = hipEventSynchronize(tm.ev.back());
        if (e != hipSuccess) rc = fail(AAE_ERR_RUNTIME, "hipEventSynchronize: %s", hipGetErrorString(e));
    }
    int n = (int)tm.ev.size() - 1;
    if (n < 0) n = 0;
    if (rc == AAE_OK) {
        for (int i = 0; i < n && i < max_kernels; ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, tm.ev[i], tm.ev[i + 1]);
            kernel_ms[i] = ms;
        }
        *n_kernels = n;
    }
    for (hipEvent_t e : tm.ev) (void)hipEventDestroy(e);
    return rc;
}

const char* aae_encoder_kernel_label(const aae_encoder* enc, int i) {
    if (!enc || i < 0 || i >= (int)enc->records.size()) return "";
    return enc->records[i].label.c_str();
}

double aae_encoder_kernel_flops(const aae_encoder* enc, int i) {
    if (!enc || i < 0 || i >= (int)enc->records.size()) return 0.0;
    return enc->records[i].flops;
}

int aae_encoder_x3h_saturated(aae_encoder* enc, int* flag_out, void* stream_v) {
    using namespace aae_host;
    if (!enc || !flag_out) return fail(AAE_ERR_INVALID, "aae_encoder_x3h_saturated: null argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    int v[kX3hRing + kX3hCaptured];
    AAE_HIP_TRY(hipMemcpyAsync(v, enc->x3h_sat, sizeof(v), hipMemcpyDeviceToHost, stream));
    AAE_HIP_TRY(hipStreamSynchronize(stream));
    int any = 0;
    for (int i = 0; i < kX3hRing + kX3hCaptured; ++i) any |= v[i];
    if (any) AAE_HIP_TRY(hipMemsetAsync(enc->x3h_sat, 0, sizeof(v), stream));
    *flag_out = any ? 1 : 0;
    return AAE_OK;
}

int aae_encoder_x3h_last_slot(void) { return aae_host::t_x3h_last_slot; }

int aae_encoder_x3h_poll(aae_encoder* enc, const int* slots, int n, int* flags_out, void* stream_v) {
    using namespace aae_host;
    if (!enc || !slots || !flags_out || n < 0) return fail(AAE_ERR_INVALID, "aae_encoder_x3h_poll: bad argument");
    for (int i = 0; i < n; ++i)
        if (slots[i] < 0 || slots[i] >= kX3hRing + kX3hCaptured) return fail(AAE_ERR_INVALID, "aae_encoder_x3h_poll: slot %d out of range", slots[i]);
    if (n == 0) return AAE_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    int v[kX3hRing + kX3hCaptured];
    AAE_HIP_TRY(hipMemcpyAsync(v, enc->x3h_sat, sizeof(v), hipMemcpyDeviceToHost, stream));
    AAE_HIP_TRY(hipStreamSynchronize(stream));
    for (int i = 0; i < n; ++i) {
        flags_out[i] = v[slots[i]] ? 1 : 0;
        if (v[slots[i]]) {                                   // (rare) clear it for the slot's next user
            AAE_HIP_TRY(hipMemsetAsync(enc->x3h_sat + slots[i], 0, sizeof(int), stream));
            v[slots[i]] = 0;
        }
    }
    return AAE_OK;
}

int aae_encoder_x3h_release_slot(aae_encoder* enc, int slot, void* stream_v) {
    using namespace aae_host;
    if (!enc) return fail(AAE_ERR_INVALID, "aae_encoder_x3h_release_slot: null handle");
    if (slot < kX3hRing || slot >= kX3hRing + kX3hCaptured) return fail(AAE_ERR_INVALID, "aae_encoder_x3h_release_slot: slot %d is not a captured forward's", slot);
    std::lock_guard<std::mutex> lk(enc->x3h_mu);
    if (slot >= kX3hRing + enc->x3h_captured || std::find(enc->x3h_free.begin(), enc->x3h_free.end(), slot) != enc->x3h_free.end())
        return fail(AAE_ERR_INVALID, "aae_encoder_x3h_release_slot: slot %d is not in use", slot);
    // (the next owner starts from a lowered flag; asynchronous on the caller's stream: a synchronous null-stream memset is
    //  invalid while any stream of the process is being captured)
    AAE_HIP_TRY(hipMemsetAsync(enc->x3h_sat + slot, 0, sizeof(int), static_cast<hipStream_t>(stream_v)));
    enc->x3h_free.push_back(slot);
    return AAE_OK;
}

int aae_encoder_debug_timeline(aae_encoder* enc, long long* host_out) {
    using namespace aae_host;
    if (!enc || !host_out) return fail(AAE_ERR_INVALID, "aae_encoder_debug_timeline: null argument");
    if (!enc->wavek_timeline) return fail(AAE_ERR_INVALID, "aae_encoder_debug_timeline: option wavek_timeline is off");
    AAE_HIP_TRY(hipDeviceSynchronize());
    AAE_HIP_TRY(hipMemcpy(host_out, enc->wavek_timeline, 3 * 512 * 8 * sizeof(long long), hipMemcpyDeviceToHost));
    return AAE_OK;
}

int aae_encoder_split_precision_for_batch(const aae_encoder* enc, int B) { return (enc && B >= 1 && aae_host::runs_split(enc, B)) ? 1 : 0; }

int aae_encoder_activation_info(const aae_encoder* enc, int B, int layer, size_t* offset_bytes, size_t* count) {
    using namespace aae_host;
    if (!enc || !offset_bytes || !count || B < 1) return fail(AAE_ERR_INVALID, "aae_encoder_activation_info: bad argument");
    if (layer < 0 || layer >= (int)enc->layers.size()) return fail(AAE_ERR_INVALID, "layer %d out of range", layer);
    if (enc->compact_workspace && layer + 2 < (int)enc->layers.size())
        return fail(AAE_ERR_UNSUPPORTED, "compact_workspace: the output of layer %d has been overwritten by layer %d", layer, layer + 2);
    const Workspace ws = plan_workspace(enc, B);
    const Layer& L = enc->layers[layer];
    *offset_bytes = ws.act_off[layer];
    *count = (size_t)B * L.Ho * L.Wo * L.Cout;
    return AAE_OK;
}

int aae_codebook_create(const void* E, int N, int J, int dtype, int src_is_device, aae_codebook** out) {
    using namespace aae_host;
    if (!E || !out) return fail(AAE_ERR_INVALID, "aae_codebook_create: null argument");
    if (N < 1 || J < 1) return fail(AAE_ERR_INVALID, "codebook shape [%d,%d]", N, J);
    if (dtype != AAE_DTYPE_F32 && dtype != AAE_DTYPE_BF16)
        return fail(AAE_ERR_UNSUPPORTED, "codebook dtype %d: float32 (AAE_DTYPE_F32) and bfloat16 (AAE_DTYPE_BF16) are implemented", dtype);
    if (J % 4 != 0 || J > 128) return fail(AAE_ERR_UNSUPPORTED, "latent size %d: the scan kernels need J %% 4 == 0 and J <= 128", J);
    if (dtype == AAE_DTYPE_BF16 && J != 128) return fail(AAE_ERR_UNSUPPORTED, "latent size %d: the bf16 scan kernel is built for J == 128", J);
    if ((unsigned long long)N * J * sizeof(float) >= 0xFFFFFFF0ull) return fail(AAE_ERR_UNSUPPORTED, "codebook of %d x %d floats exceeds the 4 GiB buffer view", N, J);
    aae_codebook* cb = new aae_codebook();
    cb->N = N; cb->J = J; cb->dtype = dtype;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) cb->cu_count = cus;
    }
    const size_t bytes = (size_t)N * J * (dtype == AAE_DTYPE_BF16 ? 2 : 4);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { delete cb; return fail(AAE_ERR_RUNTIME, "hipMalloc(codebook): %s", hipGetErrorString(e)); }
    cb->E = static_cast<float*>(p);
    e = hipMemcpy(cb->E, E, bytes, src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
    if (e != hipSuccess) { aae_codebook_destroy(cb); return fail(AAE_ERR_RUNTIME, "hipMemcpy(codebook): %s", hipGetErrorString(e)); }
    *out = cb;
    return AAE_OK;
}

static int gather_upright_rows(const aae_codebook* cb, const aae_codebook* sub, int stride, hipStream_t stream) {
    using namespace aae_host;
    aae::GatherRowsArgs g;
    g.src = cb->E; g.dst = sub->E; g.rows_out = sub->N; g.stride = stride;
    g.pieces_per_row = cb->J * (cb->dtype == AAE_DTYPE_BF16 ? 2 : 4) / 16;
    long long blocks = ((long long)g.rows_out * g.pieces_per_row + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    AAE_LAUNCH((aae::gather_rows_kernel), dim3((unsigned)blocks), dim3(256), 0, stream, g);
    AAE_HIP_TRY(hipGetLastError());
    return AAE_OK;
}

int aae_codebook_prepare_upright(aae_codebook* cb, int col_stride, void* stream_v) {
    using namespace aae_host;
    if (!cb) return fail(AAE_ERR_INVALID, "aae_codebook_prepare_upright: null handle");
    if (col_stride < 2) return fail(AAE_ERR_INVALID, "aae_codebook_prepare_upright: col_stride %d < 2", col_stride);
    if ((cb->J * (cb->dtype == AAE_DTYPE_BF16 ? 2 : 4)) % 16 != 0) return AAE_OK;     // rows are not 16-byte pieces: the masked scan stays in use
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    if (!cb->upright || cb->upright_stride != col_stride) {
        aae_codebook* sub = nullptr;
        for (auto& c : cb->upright_copies)
            if (c.first == col_stride) sub = c.second;
        if (!sub) {
            sub = new (std::nothrow) aae_codebook();
            if (!sub) return fail(AAE_ERR_RUNTIME, "out of host memory");
            sub->N = ceil_div(cb->N, col_stride); sub->J = cb->J; sub->dtype = cb->dtype;
            void* p = nullptr;
            const hipError_t e = hipMalloc(&p, (size_t)sub->N * sub->J * (sub->dtype == AAE_DTYPE_BF16 ? 2 : 4));
            if (e != hipSuccess) { delete sub; return fail(AAE_ERR_RUNTIME, "hipMalloc(upright codebook): %s", hipGetErrorString(e)); }
            sub->E = static_cast<float*>(p);
            cb->upright_copies.push_back({col_stride, sub});
        }
        sub->scan_mode = cb->scan_mode; sub->scan_ticket = cb->scan_ticket; sub->topk_prune = cb->topk_prune; sub->cu_count = cb->cu_count;
        sub->scan_walk = cb->scan_walk; sub->scan_fused_norm = cb->scan_fused_norm; sub->scan_rh4 = cb->scan_rh4; sub->scan_resident_fin = cb->scan_resident_fin;
        cb->upright = sub; cb->upright_stride = col_stride;
    }
    if (int rc = gather_upright_rows(cb, cb->upright, cb->upright_stride, stream)) return rc;
    AAE_HIP_TRY(hipStreamSynchronize(stream));
    return AAE_OK;
}

int aae_codebook_update(aae_codebook* cb, const void* E, int src_is_device, void* stream) {
    using namespace aae_host;
    if (!cb || !E) return fail(AAE_ERR_INVALID, "aae_codebook_update: null argument");
    AAE_HIP_TRY(hipMemcpyAsync(cb->E, E, (size_t)cb->N * cb->J * (cb->dtype == AAE_DTYPE_BF16 ? 2 : 4),
                               src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    for (auto& c : cb->upright_copies)                                                          // keep every compacted copy in step
        if (int rc = gather_upright_rows(cb, c.second, c.first, static_cast<hipStream_t>(stream))) return rc;
    AAE_HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return AAE_OK;
}

void aae_codebook_destroy(aae_codebook* cb) {
    if (!cb) return;
    for (auto& c : cb->upright_copies) aae_codebook_destroy(c.second);
    if (cb->E) (void)hipFree(cb->E);
    delete cb;
}

int aae_codebook_set_scan_mode(aae_codebook* cb, int mode) {
    using namespace aae_host;
    if (!cb) return fail(AAE_ERR_INVALID, "aae_codebook_set_scan_mode: null handle");
    if (mode != AAE_SCAN_AUTO && mode != AAE_SCAN_GEMV && mode != AAE_SCAN_MFMA && mode != AAE_SCAN_STREAM && mode != AAE_SCAN_STREAM_2L &&
        mode != AAE_SCAN_AUTO_NO_PRUNE && mode != AAE_SCAN_STREAM_WALK && mode != AAE_SCAN_AUTO_PACKED && mode != AAE_SCAN_AUTO_RH2 && mode != AAE_SCAN_AUTO_FIN)
        return fail(AAE_ERR_INVALID, "scan mode %d", mode);
    cb->scan_ticket = mode == AAE_SCAN_STREAM_2L ? 0 : 1;
    cb->topk_prune = mode == AAE_SCAN_AUTO_NO_PRUNE ? 0 : 1;
    cb->scan_walk = mode == AAE_SCAN_STREAM_WALK ? 1 : 0;
    cb->scan_fused_norm = mode == AAE_SCAN_AUTO_PACKED ? 0 : 1;
    cb->scan_rh4 = mode == AAE_SCAN_AUTO_RH2 ? 0 : 1;
    cb->scan_resident_fin = mode == AAE_SCAN_AUTO_FIN ? 1 : 0;
    cb->scan_mode = (mode == AAE_SCAN_STREAM_2L || mode == AAE_SCAN_STREAM_WALK) ? AAE_SCAN_STREAM
                    : ((mode == AAE_SCAN_AUTO_NO_PRUNE || mode == AAE_SCAN_AUTO_PACKED || mode == AAE_SCAN_AUTO_RH2 || mode == AAE_SCAN_AUTO_FIN) ? AAE_SCAN_AUTO : mode);
    for (auto& c : cb->upright_copies) {
        c.second->scan_mode = cb->scan_mode; c.second->scan_ticket = cb->scan_ticket; c.second->topk_prune = cb->topk_prune; c.second->scan_walk = cb->scan_walk;
        c.second->scan_fused_norm = cb->scan_fused_norm; c.second->scan_rh4 = cb->scan_rh4; c.second->scan_resident_fin = cb->scan_resident_fin;
    }
    return AAE_OK;
}

size_t aae_codebook_workspace_bytes(const aae_codebook* cb, int B, int topk) {
    if (!cb || B < 1 || topk < 1) return 0;
    return aae_host::plan_scan(cb, B, topk).total;
}

// prepared_nonce != 0: the scan's ticket words (front of `workspace`) carry this nonce already
static int nn_impl(aae_codebook* cb, const float* z, int B, int topk, int col_stride, int64_t* idx_out,
                   float* score_out, void* workspace, size_t ws_bytes, void* stream_v, unsigned prepared_nonce) {
    using namespace aae_host;
    if (!cb || !z || !idx_out || !score_out) return fail(AAE_ERR_INVALID, "aae_codebook_nn: null argument");
    if (B < 1 || topk < 1 || topk > cb->N) return fail(AAE_ERR_INVALID, "aae_codebook_nn: B=%d topk=%d N=%d", B, topk, cb->N);
    if (col_stride < 1) return fail(AAE_ERR_INVALID, "col_stride %d < 1", col_stride);
    if (topk > 1 && col_stride != 1) return fail(AAE_ERR_INVALID, "upright (col_stride>1) is defined for topk == 1 only (codebook.py:65-66)");
    if (topk > 1 && B > 65535) return fail(AAE_ERR_UNSUPPORTED, "top-k for more than 65535 queries per call (got %d): split the batch", B);
    {
        const ScanPlan full = plan_scan(cb, B, topk);
        if (ws_bytes < full.total) return fail(AAE_ERR_WORKSPACE, "workspace %zu B < required %zu B", ws_bytes, full.total);
    }
    if (!workspace || ((uintptr_t)workspace & 255)) return fail(AAE_ERR_WORKSPACE, "workspace must be non-null and 256-B aligned");
    // upright: scan the prepared every-col_stride-th-row copy (1/col_stride of the work) and scale the row id back
    int idx_scale = 1;
    if (col_stride > 1 && cb->upright && cb->upright_stride == col_stride) {
        idx_scale = col_stride;
        cb = cb->upright;
        col_stride = 1;
    }
    const ScanPlan s = plan_scan(cb, B, topk);          // never larger than the plan of the full codebook
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    unsigned char* base = static_cast<unsigned char*>(workspace);
    float* cs = (topk > 1 && !s.topk_fused) ? reinterpret_cast<float*>(base + s.cs_off) : nullptr;
    int partial_rows = s.nblk;
    // B <= 4, top-1 on a stream kernel: the last block to arrive merges the block partials -- the query is one launch
    ScanTicketOut fin;
    fin.idx_out = idx_out; fin.score_out = score_out; fin.idx_scale = idx_scale; fin.nonce = prepared_nonce;
    // (opt-in, AAE_SCAN_AUTO_FIN: the same for the query-resident scan of at most 32 queries -- one row of row blocks)
    const bool resident_fin = topk == 1 && cb->scan_resident_fin && s.resident_ok && s.res_rh == 4 && col_stride == 1;
    const bool one_launch = topk == 1 && ((s.stream && cb->scan_ticket != 0) || resident_fin);
    if (int rc = run_scan(cb, z, B, col_stride, cs, s, base, stream, &partial_rows, one_launch ? &fin : nullptr, s.topk_fused ? topk : 1)) return rc;
    if (one_launch) return AAE_OK;
    if (topk == 1) {
        aae::ArgmaxReduceArgs r;
        r.pval = reinterpret_cast<float*>(base + s.pval_off);
        r.pidx = reinterpret_cast<int*>(base + s.pidx_off);
        r.idx_out = reinterpret_cast<long long*>(idx_out);
        r.score_out = score_out; r.nblk = partial_rows; r.B = B; r.Bstride = s.Bstride; r.idx_scale = idx_scale;
        AAE_LAUNCH((aae::argmax_reduce_kernel), dim3(B), dim3(256), 64, stream, r);
    } else {
        aae::TopKArgs t;
        t.cs = cs; t.idx_out = reinterpret_cast<long long*>(idx_out); t.score_out = score_out; t.N = cb->N; t.k = topk;
        t.chunks = s.cand_chunks;
        t.cand_v = reinterpret_cast<float*>(base + s.cand_off);
        t.cand_i = reinterpret_cast<int*>(base + s.cand_off + align_up((size_t)B * t.chunks * topk * sizeof(float), 256));
        if (!s.topk_fused) AAE_LAUNCH((aae::topk_chunks_kernel), dim3(t.chunks, B), dim3(256), 64, stream, t);   // (fused: the scan wrote the lists)
        if (t.chunks * topk <= 256 * aae::kTopKMergeSlots) AAE_LAUNCH((aae::topk_merge_kernel<true>), dim3(B), dim3(256), 64, stream, t);
        else AAE_LAUNCH((aae::topk_merge_kernel<false>), dim3(B), dim3(256), 64, stream, t);
    }
    AAE_HIP_TRY(hipGetLastError());
    return AAE_OK;
}

int aae_codebook_nn(aae_codebook* cb, const float* z, int B, int topk, int col_stride, int64_t* idx_out,
                    float* score_out, void* workspace, size_t ws_bytes, void* stream) {
    return nn_impl(cb, z, B, topk, col_stride, idx_out, score_out, workspace, ws_bytes, stream, 0u);
}

int aae_codebook_nn_timed(aae_codebook* cb, const float* z, int B, int topk, int col_stride, int64_t* idx_out,
                          float* score_out, void* workspace, size_t ws_bytes, void* stream_v, int reps, float* kernel_ms) {
    using namespace aae_host;
    if (!kernel_ms || reps < 1) return fail(AAE_ERR_INVALID, "aae_codebook_nn_timed: null output or reps < 1");
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    hipEvent_t e0, e1;
    AAE_HIP_TRY(hipEventCreate(&e0));
    AAE_HIP_TRY(hipEventCreate(&e1));
    int rc = AAE_OK;
    hipError_t e = hipEventRecord(e0, stream);
    if (e == hipSuccess) {
        for (int r = 0; r < reps && rc == AAE_OK; ++r) rc = nn_impl(cb, z, B, topk, col_stride, idx_out, score_out, workspace, ws_bytes, stream_v, 0u);
        e = hipEventRecord(e1, stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess && rc == AAE_OK) e = hipEventElapsedTime(kernel_ms, e0, e1);
        if (e == hipSuccess && rc == AAE_OK) *kernel_ms /= (float)reps;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc == AAE_OK && e != hipSuccess) rc = fail(AAE_ERR_RUNTIME, "aae_codebook_nn_timed: %s", hipGetErrorString(e));
    return rc;
}

int aae_encode_nn(aae_encoder* enc, aae_codebook* cb, const void* x, int x_dtype, int B, int col_stride, float* z_out,
                  int64_t* idx_out, float* score_out, void* enc_workspace, size_t enc_ws_bytes, void* cb_workspace,
                  size_t cb_ws_bytes, void* stream) {
    using namespace aae_host;
    if (!enc || !cb) return fail(AAE_ERR_INVALID, "aae_encode_nn: null handle");
    if (col_stride < 1) return fail(AAE_ERR_INVALID, "col_stride %d < 1", col_stride);
    if (!cb_workspace || ((uintptr_t)cb_workspace & 255)) return fail(AAE_ERR_WORKSPACE, "workspace must be non-null and 256-B aligned");
    if (B >= 1 && cb_ws_bytes < plan_scan(cb, B, 1).total) return fail(AAE_ERR_WORKSPACE, "codebook workspace %zu B too small", cb_ws_bytes);
    // B <= 4: the scan finishes inside its own launch; its ticket words sit at the front of the codebook workspace and
    // are prepared by the encoder's first kernel, several launches ahead on the same stream
    const aae_codebook* eff = (col_stride > 1 && cb->upright && cb->upright_stride == col_stride) ? cb->upright : cb;
    ExtraTicketPrep extra;
    if (B >= 1 && eff->scan_ticket >= 1 && plan_scan(eff, B, 1).stream) {
        extra.words = reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(cb_workspace) + plan_scan(eff, B, 1).ticket_off);
        extra.count = aae::kTicketSlotWords;
        extra.nonce = next_nonce();
    }
    // ... and when the whole query runs as conv1 + ONE persistent launch (detect_chain.h), the scan is that launch's last phase:
    // fp32 rows, stride 1 (the upright search on its compacted copy), answers written by the last block to arrive
    if (extra.words && B <= 4 && eff->dtype == AAE_DTYPE_F32 && (col_stride == 1 || eff != cb) && z_out && idx_out && score_out) {
        const ScanPlan sp = plan_scan(eff, B, 1);
        unsigned char* cbase = static_cast<unsigned char*>(cb_workspace);
        aae::ScanArgs& a = extra.scan;
        a.E = eff->E; a.e_bytes = (unsigned)((size_t)eff->N * eff->J * sizeof(float));
        a.q = nullptr; a.qp = nullptr; a.cs = nullptr; a.z = z_out;
        a.pval = reinterpret_cast<float*>(cbase + sp.pval_off);
        a.pidx = reinterpret_cast<int*>(cbase + sp.pidx_off);
        a.N = eff->N; a.J = eff->J; a.Jpad = sp.Jpad; a.B = B; a.Bpad = sp.Bpad; a.Bstride = sp.Bstride; a.col_stride = 1;
        a.tickets = extra.words; a.nonce = extra.nonce;
        a.idx_out = reinterpret_cast<long long*>(idx_out); a.score_out = score_out; a.idx_scale = eff != cb ? col_stride : 1;
        extra.scan_ready = true;
    }
    bool prepared = false, scan_done = false;
    Timer tm;
    if (int rc = forward_impl(enc, x, x_dtype, B, z_out, enc_workspace, enc_ws_bytes, stream, tm, extra.words ? &extra : nullptr, &prepared, &scan_done)) return rc;
    if (scan_done) return AAE_OK;
    return nn_impl(cb, z_out, B, 1, col_stride, idx_out, score_out, cb_workspace, cb_ws_bytes, stream, prepared ? extra.nonce : 0u);
}

int aae_codebook_similarity(aae_codebook* cb, const float* z, int B, float* cs_out, void* workspace, size_t ws_bytes,
                            void* stream_v) {
    using namespace aae_host;
    if (!cb || !z || !cs_out) return fail(AAE_ERR_INVALID, "aae_codebook_similarity: null argument");
    if (B < 1) return fail(AAE_ERR_INVALID, "aae_codebook_similarity: B=%d", B);
    const ScanPlan s = plan_scan(cb, B, 1);
    if (ws_bytes < s.total) return fail(AAE_ERR_WORKSPACE, "workspace %zu B < required %zu B", ws_bytes, s.total);
    if (!workspace || ((uintptr_t)workspace & 255)) return fail(AAE_ERR_WORKSPACE, "workspace must be non-null and 256-B aligned");
    return run_scan(cb, z, B, 1, cs_out, s, static_cast<unsigned char*>(workspace), static_cast<hipStream_t>(stream_v));
}

int aae_l2_normalize(const float* z, int B, int J, float* q_out, void* stream_v) {
    using namespace aae_host;
    if (!z || !q_out || B < 1 || J < 1) return fail(AAE_ERR_INVALID, "aae_l2_normalize: bad argument");
    aae::L2NormArgs n;
    n.z = z; n.q = q_out; n.qp = nullptr; n.B = B; n.J = J; n.Jpad = J; n.Bpad = B;
    AAE_LAUNCH((aae::l2norm_pack_kernel), dim3(ceil_div(B, 4)), dim3(256), 0, static_cast<hipStream_t>(stream_v), n);
    AAE_HIP_TRY(hipGetLastError());
    return AAE_OK;
}

int aae_pack_pairs(const int64_t* idx, const float* score, const int32_t* pos, int n, int stride, int64_t* packed, void* stream_v) {
    using namespace aae_host;
    if (!idx || !score || !packed) return fail(AAE_ERR_INVALID, "aae_pack_pairs: null argument");
    if (n < 0 || stride < 1) return fail(AAE_ERR_INVALID, "aae_pack_pairs: n=%d stride=%d", n, stride);
    if (n == 0) return AAE_OK;
    aae::PackPairsArgs a;
    a.idx = reinterpret_cast<const long long*>(idx); a.score = score; a.pos = pos; a.packed = reinterpret_cast<long long*>(packed);
    a.n = n; a.stride = stride;
    AAE_LAUNCH((aae::pack_pairs_kernel), dim3(ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_v), a);
    AAE_HIP_TRY(hipGetLastError());
    return AAE_OK;
}

int aae_unpack_pairs(const int64_t* gathered, const int32_t* owner, int n, int rows_per_rank, int64_t* idx_out, float* score_out,
                     void* stream_v) {
    using namespace aae_host;
    if (!gathered || !idx_out || !score_out) return fail(AAE_ERR_INVALID, "aae_unpack_pairs: null argument");
    if (n < 0 || rows_per_rank < n) return fail(AAE_ERR_INVALID, "aae_unpack_pairs: n=%d rows_per_rank=%d", n, rows_per_rank);
    if (n == 0) return AAE_OK;
    aae::UnpackPairsArgs a;
    a.gathered = reinterpret_cast<const long long*>(gathered); a.owner = owner; a.idx = reinterpret_cast<long long*>(idx_out);
    a.score = score_out; a.n = n; a.rows_per_rank = rows_per_rank;
    AAE_LAUNCH((aae::unpack_pairs_kernel), dim3(ceil_div(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_v), a);
    AAE_HIP_TRY(hipGetLastError());
    return AAE_OK;
}

int aae_crop_resize_u8(const void* img, int H, int W, int C, const int32_t* boxes, int D, int out_h, int out_w,
                       void* out, void* stream_v) {
    using namespace aae_host;
    if (!img || !boxes || !out) return fail(AAE_ERR_INVALID, "aae_crop_resize_u8: null argument");
    if (H < 1 || W < 1 || C < 1 || D < 1 || out_h < 1 || out_w < 1)
        return fail(AAE_ERR_INVALID, "aae_crop_resize_u8: image %dx%dx%d, %d boxes, output %dx%d", H, W, C, D, out_h, out_w);
    if (D > 65535) return fail(AAE_ERR_UNSUPPORTED, "aae_crop_resize_u8: at most 65535 boxes per call");
    aae::CropResizeArgs a;
    a.img = static_cast<const unsigned char*>(img); a.boxes = boxes; a.out = static_cast<unsigned char*>(out);
    a.H = H; a.W = W; a.C = C; a.D = D; a.OH = out_h; a.OW = out_w;
    AAE_LAUNCH((aae::crop_resize_bilinear_u8_kernel), dim3(ceil_div(out_h * out_w, 256), D), dim3(256), 0,
               static_cast<hipStream_t>(stream_v), a);
    AAE_HIP_TRY(hipGetLastError());
    return AAE_OK;
}

int aae_detect_nn(aae_encoder* enc, aae_codebook* cb, const void* img, int H, int W, int C, const int32_t* boxes, int n,
                  int col_stride, void* crops, float* z_out, int64_t* idx_out, float* score_out,
                  void* enc_workspace, size_t enc_ws_bytes, void* cb_workspace, size_t cb_ws_bytes, void* stream) {
    using namespace aae_host;
    if (!enc || !cb || !crops) return fail(AAE_ERR_INVALID, "aae_detect_nn: null argument");
    if (C != enc->desc.in_c) return fail(AAE_ERR_INVALID, "aae_detect_nn: image has %d channels, the encoder takes %d", C, enc->desc.in_c);
    if (int rc = aae_crop_resize_u8(img, H, W, C, boxes, n, enc->desc.in_h, enc->desc.in_w, crops, stream)) return rc;
    return aae_encode_nn(enc, cb, crops, AAE_DTYPE_U8, n, col_stride, z_out, idx_out, score_out, enc_workspace, enc_ws_bytes, cb_workspace,
                         cb_ws_bytes, stream);
}

}  // extern "C"

#include "aae_multi_impl.h"
#include "aae_decoder_impl.h"
