// gfx950 wave-level primitives used by the kernels under csrc/kernels/.
// Included (after <hip/hip_runtime.h>) by the product translation unit only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));

#define AAE_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]

namespace aae {

// v_mfma_f32_32x32x2_f32: D = A(32x2) * B(2x32) + C, exact fp32 fma chain.
// lane l: a = A[l&31][l>>5], b = B[l>>5][l&31];
// c/d reg r: row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31.
__device__ __forceinline__ f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// v_mfma_f32_32x32x16_f16: D = A(32x16) * B(16x32) + C with fp32 accumulation.
// lane l holds 8 halves of A row l&31 / B column l&31 for k = 8*(l>>5) .. 8*(l>>5)+7.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_32x32x16_f16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// fp32 -> (hi, lo) half pair: hi = rn16(v), lo = rn16(v - hi); hi + lo carries >= 22 bits of v.
// |v| beyond the fp16 range saturates hi (the remainder spills into lo) instead of producing inf.
__device__ __forceinline__ void split_f16(float v, unsigned short& hi, unsigned short& lo) {
    const float c = fminf(fmaxf(v, -65504.f), 65504.f);
    const _Float16 h = (_Float16)c;
    const float r = fminf(fmaxf(v - (float)h, -65504.f), 65504.f);
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, (_Float16)r);
}

// f32x3h activation format: element e = m*C + n of an [M][C] activation (C % 32 == 0) is the pair of halves at
// x3h_pair_index(e) (hi) and x3h_pair_index(e) + 32 (lo): every 32-channel chunk is one 128-byte line, 64 B of hi
// halves followed by the 64 B of their lo halves -- exactly the A rows of one K slab, so an operand row is ONE full
// L2 line.  (As two separate planes each row was two half lines in different places: the half-filled lines halved the
// useful L2 capacity and the tap-to-tap re-reads of the convolution window missed: 2.1 GB fetched for conv2 at B = 256.)
__host__ __device__ __forceinline__ long long x3h_pair_index(long long e) { return ((e >> 5) << 6) + (e & 31); }


// ... and note a value the (hi, lo) pair cannot carry at full accuracy: beyond the fp16 range hi saturates and the
// remainder lands in lo with the spacing of a large half (accuracy degrades gradually up to 2 x 65504, then the pair
// clamps).  sat: sticky device flag (nullptr = do not track); every writer stores the same 1, so the race is benign.
constexpr float kHalfPairLimit = 65504.f;
__device__ __forceinline__ void split_f16_checked(float v, unsigned short& hi, unsigned short& lo, int* sat) {
    split_f16(v, hi, lo);
    if (sat && !(fabsf(v) < kHalfPairLimit)) *sat = 1;         // (NaN counts as out of range too)
}

// v_mfma_f32_32x32x16_bf16, same fragment maps as the f16 form.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// fp32 -> three bf16 terms (round-to-nearest-even each): v = t0 + t1 + t2 to fp32 accuracy.
// (the conversion is one instruction on gfx950, v_cvt_pk_bf16_f32 -- two values at a time; as integer arithmetic on the bit
//  pattern it was four per value)
__device__ __forceinline__ unsigned bf16_rn_pack2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ unsigned short bf16_rn(float v) { return (unsigned short)(bf16_rn_pack2(v, 0.f) & 0xFFFFu); }
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ __forceinline__ void split_bf16x3(float v, unsigned short& t0, unsigned short& t1, unsigned short& t2) {
    t0 = bf16_rn(v);
    const float r1 = v - bf16_to_f32(t0);
    t1 = bf16_rn(r1);
    t2 = bf16_rn(r1 - bf16_to_f32(t1));
}

// Raw buffer view with hardware bounds checking: a 16-B load whose byte offset is >= the
// buffer size returns zeros.  The implicit-GEMM loader uses that for TF 'SAME' zero padding
// (out-of-image taps get offset kOobOffset) so the im2col gather has no branches.
typedef __amdgpu_buffer_rsrc_t buffer_rsrc;
constexpr uint32_t kOobOffset = 0xFFFFFFF0u;
__device__ __forceinline__ buffer_rsrc make_buffer(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buffer_load4(buffer_rsrc r, uint32_t byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
// ... with a wave-uniform part of the offset in a scalar register (no vector add per load).  The range check covers the lane offset
// only: an out-of-range lane offset returns zeros whatever the scalar part is.
__device__ __forceinline__ f32x4 buffer_load4_s(buffer_rsrc r, uint32_t lane_off, uint32_t uniform_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)uniform_off, 0));
}

// LDS-DMA: buffer_load_dwordx4 ... lds.  Lane l's 16 bytes go straight from the buffer view to
// LDS at lds_wave_base + 16*l (no VGPR staging, no ds_write); out-of-range lanes deposit zeros.
// lds_wave_base must be wave-uniform.  Completion is counted by vmcnt: a consumer may read the
// bytes only after every issuing wave has passed wait_dma_and_lds() and a barrier.
__device__ __forceinline__ void lds_dma16(buffer_rsrc r, uint32_t byte_off, float* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)byte_off, 0, 0, 0);
}
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// ... leaving the N youngest VMEM operations in flight (vmcnt retires loads in issue order)
template <int N>
__device__ __forceinline__ void wait_dma_keep_and_lds() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }
// bare s_barrier: unlike __syncthreads() it carries no fence, so it does not drain loads that are meant to stay in flight
__device__ __forceinline__ void block_barrier() {
    asm volatile("" ::: "memory");       // compiler-level fence on both sides: no LDS access may be moved across the barrier
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wait_dma_and_lds() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

// Cross-lane add through the DPP path (no LDS crossbar).  Lanes whose source is outside
// the row / masked out contribute 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// Sum over each 32-lane half of the wave.  The total is valid in lanes 16..31 (lanes 0..31)
// and 48..63 (lanes 32..63); the other lanes hold partial sums.
__device__ __forceinline__ float half_wave_sum(float v) {
    v += dpp_take<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
    v += dpp_take<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
    v += dpp_take<0x141, 0xF>(v);    // row_half_mirror
    v += dpp_take<0x140, 0xF>(v);    // row_mirror
    v += dpp_take<0x142, 0xA>(v);    // row_bcast15 -> rows 1 and 3
    return v;
}

// Sum over each 16-lane row of the wave (4 DPP adds); every lane of the row ends up with the total.
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_take<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
    v += dpp_take<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
    v += dpp_take<0x141, 0xF>(v);    // row_half_mirror
    v += dpp_take<0x140, 0xF>(v);    // row_mirror
    return v;
}

// ---- reduce-scatter: 16 quantities summed over the lanes, each total left in its own lane ---------------------------
// d[u], u = 0 ... 15: this lane's partial sums of 16 different quantities (the stream scans: 16 codebook rows against one
// query).  Sixteen separate butterfly trees (half_wave_sum) cost 16 x 5 cross-lane adds and leave every total in every
// lane; nobody needs that.  Here every level of the tree halves the quantities a lane still carries -- a lane keeps the
// upper or the lower half of them according to the lane bit that level reduces over -- so the work shrinks 16, 8, 4, 2, 1:
//   level 1, lane bit 4: v_permlane16_swap exchanges the odd 16-lane rows of d[u] with the even rows of d[u + 8]; one add
//            then gives (rows 0 / 2) d[u] and (rows 1 / 3) d[u + 8], each summed over lane bit 4          2 instr. per pair
//   level 2, lane bit 3: X + X[l ^ 8], Y + Y[l ^ 8] (DPP row_ror:8), banks 0-1 keep X, banks 2-3 keep Y   3 per pair
//   level 3, lane bit 2: X + X[l + 4] (row_shl:4) kept by banks 0 / 2, Y + Y[l - 4] (row_shr:4) by banks 1 / 3
//   level 4, lane bit 1: quad_perm [2,3,0,1], select by lane bit 1;   level 5, lane bit 0: quad_perm [1,0,3,2]
// = 16 + 12 + 6 + 3 + 1 = 38 instructions for the 16 quantities.  The two 32-lane halves of the wave are independent
// (nothing crosses lane bit 5).  Result: lane l holds the total of quantity (l >> 1) & 15 over its half-wave (both lanes
// of a pair the same value).  Every level adds a value to its partner's, so the summation tree is fixed and symmetric:
// tests/emu/hip_emu.h restates it.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_move(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, false));
}
__device__ __forceinline__ float half_wave_reduce_scatter16(const float (&d)[16]) {
    float t8[8], t4[4], t2[2];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, d[u]), __builtin_bit_cast(unsigned, d[u + 8]), false, false);
        const unsigned s0 = sw[0], s1 = sw[1];                   // (scalars first: a bit_cast of ONE vector element has been seen to pick element 0)
        t8[u] = __builtin_bit_cast(float, s0) + __builtin_bit_cast(float, s1);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float a1 = t8[u] + dpp_take<0x128, 0xF>(t8[u]);            // row_ror:8
        const float a2 = t8[u + 4] + dpp_take<0x128, 0xF>(t8[u + 4]);
        t4[u] = dpp_move<0xE4, 0xF, 0x3>(a2, a1);                         // banks 0, 1 (lane bit 3 = 0): a1, banks 2, 3: a2
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float a1 = t4[u] + dpp_take<0x104, 0xF>(t4[u]);            // row_shl:4: lane l reads l + 4 (used where lane bit 2 = 0)
        const float a2 = t4[u + 2] + dpp_take<0x114, 0xF>(t4[u + 2]);    // row_shr:4: lane l reads l - 4 (used where lane bit 2 = 1)
        t2[u] = dpp_move<0xE4, 0xF, 0x5>(a2, a1);                         // banks 0, 2: a1, banks 1, 3: a2
    }
    const float a1 = t2[0] + dpp_take<0x4E, 0xF>(t2[0]);                 // quad_perm [2,3,0,1]
    const float a2 = t2[1] + dpp_take<0x4E, 0xF>(t2[1]);
    const float t1 = (threadIdx.x & 2) ? a2 : a1;
    return t1 + dpp_take<0xB1, 0xF>(t1);                                  // quad_perm [1,0,3,2]
}

// Same idea inside each 16-lane row (the bf16 stream scan: a 256-byte row = 16 lanes x 16 B): d[u] summed over the 16
// lanes of the row, lane l of the row left with the total of quantity l (levels: lane bit 3, 2, 1, 0 -> 24 + 12 + 6 + 3 = 45
// instructions instead of 16 x 4 = 64).
__device__ __forceinline__ float row16_reduce_scatter16(const float (&d)[16]) {
    float t8[8], t4[4], t2[2];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const float a1 = d[u] + dpp_take<0x128, 0xF>(d[u]);
        const float a2 = d[u + 8] + dpp_take<0x128, 0xF>(d[u + 8]);
        t8[u] = dpp_move<0xE4, 0xF, 0x3>(a2, a1);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float a1 = t8[u] + dpp_take<0x104, 0xF>(t8[u]);
        const float a2 = t8[u + 4] + dpp_take<0x114, 0xF>(t8[u + 4]);
        t4[u] = dpp_move<0xE4, 0xF, 0x5>(a2, a1);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float a1 = t4[u] + dpp_take<0x4E, 0xF>(t4[u]);
        const float a2 = t4[u + 2] + dpp_take<0x4E, 0xF>(t4[u + 2]);
        t2[u] = (threadIdx.x & 2) ? a2 : a1;
    }
    const float a1 = t2[0] + dpp_take<0xB1, 0xF>(t2[0]);
    const float a2 = t2[1] + dpp_take<0xB1, 0xF>(t2[1]);
    return (threadIdx.x & 1) ? a2 : a1;
}

// Largest v of the wave and the LOWEST lane that holds it, both wave-uniform; NaNs rank as -inf (a wave of NaNs answers
// (-inf, lane 0)), -0 as +0.  The maximum is taken on integer keys that order like the floats (v_max_i32 fuses with its DPP
// operand and needs no NaN canonicalisation: one instruction per step where fmaxf costs four): row maxima by DPP, lane 63
// collects the four rows, the lane search is one compare + a scalar find-first-set -- np.argmax's first-index rule when
// lanes are ordered like rows.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_max_step(int k) {
    const int o = __builtin_amdgcn_update_dpp((int)0x80000000, k, CTRL, ROW_MASK, 0xF, false);     // disabled lanes: the identity of max (the combine then folds the move into v_max_i32_dpp)
    return o > k ? o : k;
}
__device__ __forceinline__ float wave_max_first_lane(float v, int& first_lane) {
    const float z = v + 0.f;                                              // -0 -> +0
    const float s = (z == z) ? z : -__builtin_huge_valf();
    const int b = __builtin_bit_cast(int, s);
    int k = b ^ ((b >> 31) & 0x7fffffff);                                 // orders like s
    k = dpp_max_step<0xB1, 0xF>(k);                                       // quad_perm [1,0,3,2]
    k = dpp_max_step<0x4E, 0xF>(k);                                       // quad_perm [2,3,0,1]
    k = dpp_max_step<0x141, 0xF>(k);                                      // row_half_mirror
    k = dpp_max_step<0x140, 0xF>(k);                                      // row_mirror: every lane of a row holds the row maximum
    k = dpp_max_step<0x142, 0xA>(k);                                      // row_bcast15 -> rows 1 and 3
    k = dpp_max_step<0x143, 0xC>(k);                                      // row_bcast31 -> rows 2 and 3: lane 63 has seen all four
    const int wk = __builtin_amdgcn_readlane(k, 63);
    const unsigned long long hit = __builtin_amdgcn_ballot_w64((b ^ ((b >> 31) & 0x7fffffff)) == wk);
    first_lane = (int)__builtin_ctzll(hit);                               // (never empty: the maximum is some lane's key)
    return __builtin_bit_cast(float, wk ^ ((wk >> 31) & 0x7fffffff));
}

// The same over NV values per lane, value j of lane l standing for position 64 j + l: largest value and the LOWEST position
// that holds it (one key maximum, then one compare + bit search per slot, first slot with a hit wins).
template <int NV>
__device__ __forceinline__ float wave_max_first_position(const float (&v)[NV], int& first_pos) {
    int key[NV];
    int k = (int)0x80000000;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float z = v[j] + 0.f;
        const float s = (z == z) ? z : -__builtin_huge_valf();
        const int b = __builtin_bit_cast(int, s);
        key[j] = b ^ ((b >> 31) & 0x7fffffff);
        k = key[j] > k ? key[j] : k;
    }
    k = dpp_max_step<0xB1, 0xF>(k);
    k = dpp_max_step<0x4E, 0xF>(k);
    k = dpp_max_step<0x141, 0xF>(k);
    k = dpp_max_step<0x140, 0xF>(k);
    k = dpp_max_step<0x142, 0xA>(k);
    k = dpp_max_step<0x143, 0xC>(k);
    const int wk = __builtin_amdgcn_readlane(k, 63);
    first_pos = 0;
    bool found = false;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const unsigned long long hit = __builtin_amdgcn_ballot_w64(key[j] == wk);
        if (!found && hit) { first_pos = 64 * j + (int)__builtin_ctzll(hit); found = true; }     // (wave-uniform)
    }
    return __builtin_bit_cast(float, wk ^ ((wk >> 31) & 0x7fffffff));
}

// makes v available HERE (the compiler must finish the load that produces it before this point instead of
// waiting in front of every later use in its own basic block)
__device__ __forceinline__ float pin_value(float v) { asm volatile("" : "+v"(v)); return v; }

// s_setprio: issue priority of this wave among the waves of its SIMD (0 = default ... 3)
template <int P>
__device__ __forceinline__ void wave_priority() { __builtin_amdgcn_s_setprio(P); }

// instruction-scheduling fence: nothing is moved across it by the compiler's scheduler
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// device-scope max on a word other blocks publish to as well (no return value wanted)
__device__ __forceinline__ void shared_word_max(int* word, int v) { (void)__hip_atomic_fetch_max(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// coarse start delay: n x 1024 shader cycles
__device__ __forceinline__ void sleep_kcycles(int n) {
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
}

// ---- cross-block hand-off inside ONE launch ("the last block to arrive finishes the job") -------------------------
// A split reduction then needs no second kernel.  Measured on MI355X (tools/ubench/sync_cost.hip, 721 blocks, one
// launch 3.3 us): agent-scope __threadfence() in every thread +44 us (an L2 write-back per block, serialised); a CAS
// loop on one word 1.7 ms; sc1 stores + fetch-add on one word +7 us; sc1 stores + two-level fetch-add +0.0 us.  So:
//   * partials that cross blocks are written with sc1 (device-scope, write-through) stores and read back with sc1
//     loads -- the instructions the compiler emits for agent-scope relaxed atomics, here in their 4/16-B buffer form,
//     which unlike atomic loads can be kept in flight by the dozen.  No L2 write-back or invalidate is ever issued;
//   * a block arrives with ONE fetch-add; above 32 arrivals the blocks first meet in 16 group words (128 B apart:
//     different channels) and the last of each group arrives at the top word.
// Ticket word = nonce << 32 | arrivals.  The nonce is unique per launch (never 0).  An arrival whose fetch-add
// returns another nonce met a foreign state -- a clean word (0), memory that was never initialised, the remains of
// a launch that died -- and installs (nonce, 1) by compare-and-swap (or, if another arrival got there first, adds
// again); adds that landed on a foreign state are discarded with it and each of those arrivals counts itself once
// more, so every block is counted exactly once.
// The last arriver leaves the word at 0.  A workspace therefore needs no memset, neither at first use nor between
// launches.
constexpr int kCoherent = 16;            // cache-policy bit sc1 (gfx940+): device scope
__device__ __forceinline__ void coherent_store4(buffer_rsrc r, uint32_t byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)byte_off, 0, kCoherent);
}
__device__ __forceinline__ f32x4 coherent_load4(buffer_rsrc r, uint32_t byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, kCoherent));
}
__device__ __forceinline__ void coherent_store2(buffer_rsrc r, uint32_t byte_off, uint32_t v0, uint32_t v1) {
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t v = {v0, v1};
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)byte_off, 0, kCoherent);
}
__device__ __forceinline__ void coherent_store1(buffer_rsrc r, uint32_t byte_off, uint32_t v) {
    __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)byte_off, 0, kCoherent);
}
__device__ __forceinline__ uint32_t coherent_load1(buffer_rsrc r, uint32_t byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, kCoherent);
}

constexpr int kTicketGroups = 16;
constexpr int kTicketGroupStride = 16;                         // 64-bit words between group words: 128 B
constexpr int kTicketSlotWords = (1 + kTicketGroups) * kTicketGroupStride;   // top word + 16 group words
constexpr unsigned kTicketSingleLevelMax = 32;

// arrivals at `word` in launch `nonce`, this one included
__device__ __forceinline__ unsigned ticket_count(unsigned long long* word, unsigned nonce) {
    const unsigned long long old = __hip_atomic_fetch_add(word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(old >> 32) == nonce) return (unsigned)old + 1u;         // the common case: one atomic
    // Foreign state.  From here on this arrival never adds to a foreign word again (arrivals that kept adding would
    // make every compare-and-swap of every other arrival fail: a livelock with ~40 blocks arriving together): it
    // either installs (nonce, 1) itself -- discarding the adds that landed on the foreign value, its own included --
    // or, once somebody else has installed the nonce, takes its number with a fetch-add.
    unsigned long long cur = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spin = 0;; ++spin) {
        if ((unsigned)(cur >> 32) == nonce)
            return (unsigned)__hip_atomic_fetch_add(word, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (__hip_atomic_compare_exchange_strong(word, &cur, ((unsigned long long)nonce << 32) | 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT))
            return 1u;
        if (spin > (1 << 16)) __builtin_trap();        // cannot happen (every failed swap means another arrival made progress); never hang the device
    }
}
__device__ __forceinline__ void ticket_clear(unsigned long long* word) {
    __hip_atomic_store(word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every thread of the block calls this after its coherent_store*() of the block's partial result; it returns true
// -- in every thread -- for exactly one block: the last of `total` arrivals (this block is arrival `id`, any unique
// number < total, used only to spread the blocks over the group words).  `words`: kTicketSlotWords words when
// total > kTicketSingleLevelMax, one word otherwise.  The winning block may then coherent_load*() all partials.
// Ordering: each thread waits for its own stores (vmcnt(0): an sc1 store is acknowledged once it is visible at device
// scope), then the block barrier, then thread 0's atomic; the winner's loads are issued after the second barrier.
// Early self-preparation.  One thread per ticket word, called by ONE block at the very start of a ticketed kernel:
// makes the word carry this launch's nonce before the arrivals come (they come microseconds later, after the block's
// real work), so that they all take the one-atomic path.  Safe in ANY order relative to the arrivals: it only ever
// replaces a foreign value, by compare-and-swap -- if an arrival installed the nonce first, nothing happens; if its
// swap discards optimistic adds that landed on the foreign value, those arrivals add again (ticket_count).
__device__ __forceinline__ void ticket_prepare_word(unsigned long long* word, unsigned nonce) {
    unsigned long long cur = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spin = 0; (unsigned)(cur >> 32) != nonce; ++spin) {
        if (__hip_atomic_compare_exchange_strong(word, &cur, (unsigned long long)nonce << 32, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT))
            break;
        if (spin > (1 << 16)) __builtin_trap();
    }
}
// the words of one ticket: a single word (total <= kTicketSingleLevelMax) or the top word + 16 group words
__device__ __forceinline__ void ticket_prepare_slot(unsigned long long* words, unsigned nonce, unsigned total) {
    const unsigned n = total <= kTicketSingleLevelMax ? 1u : 1u + kTicketGroups;
    if (threadIdx.x < n) ticket_prepare_word(words + threadIdx.x * kTicketGroupStride, nonce);
}

// first half: this block's coherent stores are complete and every thread knows it
__device__ __forceinline__ void block_ticket_publish() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
// second half: take the ticket
__device__ __forceinline__ bool block_ticket_take(unsigned long long* words, unsigned nonce, unsigned total, unsigned id, int* lds_flag) {
    if (threadIdx.x == 0) {
        bool last = false;
        if (total <= kTicketSingleLevelMax) {
            last = ticket_count(words, nonce) == total;
            if (last) ticket_clear(words);
        } else {
            const unsigned g = id % kTicketGroups, members = (total - g + kTicketGroups - 1) / kTicketGroups;
            unsigned long long* gw = words + (1 + g) * kTicketGroupStride;
            if (ticket_count(gw, nonce) == members) {
                ticket_clear(gw);
                last = ticket_count(words, nonce) == (unsigned)kTicketGroups;
                if (last) ticket_clear(words);
            }
        }
        *lds_flag = last ? 1 : 0;
    }
    __syncthreads();
    return *lds_flag != 0;
}
__device__ __forceinline__ bool block_ticket_arrive(unsigned long long* words, unsigned nonce, unsigned total, unsigned id, int* lds_flag) {
    block_ticket_publish();
    return block_ticket_take(words, nonce, total, id, lds_flag);
}

// ---- grid-wide barrier of a persistent launch -------------------------------------------------------------------------
// For a launch whose blocks are ALL resident (the host sizes the grid to the chip: one block per CU).  Data that crosses the
// barrier follows the same rule as the ticketed hand-offs above: written with device-coherent (sc1, write-through) stores,
// every buffer written at most once per launch and never read before its barrier -- so a reader can meet neither a stale
// L1 nor a stale L2 line, and no cache write-back or invalidate is ever issued.
// Words (64-bit, 128 B apart): top counter, 16 group counters, 16 group gates.  Counters are the ticket words of above
// used monotonically -- `nonce << 32 | arrivals so far`, never cleared inside a launch: barrier number `phase` (1, 2, ...)
// is complete when the top counter reads groups * phase; the last arriver then writes `nonce << 32 | phase` into every
// gate and each block's lane 0 polls the gate of its group (16 pollers per word).  A foreign nonce (uninitialised memory,
// an earlier launch) counts as empty / closed: no memset, ever.
// Split in two halves so that loads which do not depend on the other blocks (the next layer's weights, the first codebook
// rows) can be requested between arriving and waiting: they fly while the barrier closes.
constexpr int kGridBarrierWords = (1 + 2 * kTicketGroups) * kTicketGroupStride;
struct GridBarrier {
    unsigned long long* words;     // kGridBarrierWords
    unsigned nonce;                // unique per launch, never 0
};
__device__ __forceinline__ void grid_barrier_arrive(const GridBarrier& gb, unsigned nblocks, unsigned block, unsigned phase) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's coherent stores are visible at device scope
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned groups = nblocks < (unsigned)kTicketGroups ? nblocks : (unsigned)kTicketGroups;
        const unsigned g = block % groups, members = (nblocks - g + groups - 1) / groups;
        if (ticket_count(gb.words + (1 + g) * kTicketGroupStride, gb.nonce) == members * phase &&
            ticket_count(gb.words, gb.nonce) == groups * phase) {
            const unsigned long long open = ((unsigned long long)gb.nonce << 32) | phase;
            for (unsigned k = 0; k < groups; ++k)
                __hip_atomic_store(gb.words + (1 + kTicketGroups + k) * kTicketGroupStride, open, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__device__ __forceinline__ void grid_barrier_wait(const GridBarrier& gb, unsigned nblocks, unsigned block, unsigned phase) {
    if (threadIdx.x == 0) {
        const unsigned groups = nblocks < (unsigned)kTicketGroups ? nblocks : (unsigned)kTicketGroups;
        const unsigned long long* gate = gb.words + (1 + kTicketGroups + block % groups) * kTicketGroupStride;
        for (unsigned spin = 0;; ++spin) {
            const unsigned long long v = __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(v >> 32) == gb.nonce && (unsigned)v >= phase) break;
            __builtin_amdgcn_s_sleep(2);
            if (spin > (1u << 22)) __builtin_trap();       // seconds: a block of the grid is not resident -- never hang the device
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// shader-clock counter (s_memtime): kernel-internal timelines of the profiling tools
__device__ __forceinline__ long long clock_ticks() { return (long long)__builtin_readcyclecounter(); }
// constant 100 MHz counter (s_memrealtime): the same across CUs, independent of the shader clock
__device__ __forceinline__ long long wall_ticks() { return (long long)__builtin_amdgcn_s_memrealtime(); }

__device__ __forceinline__ float shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ int shfl_xor(int v, int mask) { return __shfl_xor(v, mask, 64); }
// true in every lane when the predicate holds in any lane of the wave (wave-uniform: usable as a scalar branch)
__device__ __forceinline__ bool wave_any(bool pred) { return __builtin_amdgcn_ballot_w64(pred) != 0; }

}  // namespace aae
