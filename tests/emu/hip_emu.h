// CPU fiber emulator for the HIP kernels under augmentedautoencoder_amd/csrc/.
//
// TEST INFRASTRUCTURE ONLY.  The product library (libaae_hip.so) is compiled by
// hipcc for gfx950 and never sees this header; the package loader refuses to
// load anything else.  This header lets `-m "not gpu"` tests run the *same*
// kernel sources and host launch logic on the CPU (one ucontext fiber per GPU
// thread, wave collectives = rendez-vous of 64 fibers) so that tile index
// math, MFMA fragment maps, masking and reductions are checked against the
// oracle before any GPU minute is spent.  It models functional behaviour only
// (no timing, no memory model).
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <functional>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t u16x4 __attribute__((ext_vector_type(4)));

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };

extern uint3_emu threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
extern unsigned char* aae_emu_dyn_smem;      // LDS image of the running block

#define __global__ static
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define AAE_DYN_SMEM(name) unsigned char* name = aae_emu_dyn_smem

void __syncthreads();

using std::max;
using std::min;

namespace aae_emu {
// All 64 lanes of the calling wave deposit `nbytes` (<= 64) and get back a view
// of every lane's deposit: ret[lane] is that lane's bytes.
typedef unsigned char lane_slot[64];
const lane_slot* wave_exchange(const void* mine, int nbytes);
void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
// all blocks alive side by side (persistent kernels with grid-wide waits); at most 16 blocks
void launch_resident(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
void spin_yield();          // inside a grid-wide wait: let the other blocks run
void note_progress();
}  // namespace aae_emu

namespace aae {

inline int lane_id() { return threadIdx.x & 63; }
inline long long clock_ticks() { return 0; }
inline long long wall_ticks() { return 0; }
inline void sleep_kcycles(int) {}
inline void sched_fence() {}
inline void shared_word_max(int* word, int v) { if (v > *word) *word = v; }
template <int P>
inline void wave_priority() {}
inline float pin_value(float v) { return v; }

struct buffer_rsrc { const unsigned char* base; uint32_t bytes; };
constexpr uint32_t kOobOffset = 0xFFFFFFF0u;
inline buffer_rsrc make_buffer(const void* base, uint32_t bytes) { return buffer_rsrc{static_cast<const unsigned char*>(base), bytes}; }
inline f32x4 buffer_load4(buffer_rsrc r, uint32_t byte_off) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((uint64_t)byte_off + 16 <= r.bytes) memcpy(&v, r.base + byte_off, 16);     // hardware range check
    return v;
}

inline f32x4 buffer_load4_s(buffer_rsrc r, uint32_t lane_off, uint32_t uniform_off) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((uint64_t)lane_off + 16 <= r.bytes) memcpy(&v, r.base + lane_off + uniform_off, 16);     // (the range check sees the lane offset only)
    return v;
}

// buffer_load_dwordx4 ... lds model: lane l's 16 bytes land at wave_base + 16*l (zeros when
// out of range).  The emulator completes it at issue; the landing order on hardware is
// covered by wait_dma_and_lds() + barrier in the kernels and by the GPU parity tests.
inline void lds_dma16(buffer_rsrc r, uint32_t byte_off, float* lds_wave_base) {
    const f32x4 v = buffer_load4(r, byte_off);
    memcpy(reinterpret_cast<unsigned char*>(lds_wave_base) + 16 * lane_id(), &v, 16);
}
template <int N>
inline void wait_dma_keep_and_lds() {}
inline void block_barrier() { __syncthreads(); }
inline void wait_dma_and_lds() {}
inline int wave_uniform(int v) { return v; }

inline f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    float ab[2] = {a, b};
    const aae_emu::lane_slot* all = aae_emu::wave_exchange(ab, sizeof(ab));
    const int lane = lane_id();
    const int col = lane & 31, hi = lane >> 5;
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = d[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            memcpy(&av, &all[row + 32 * k][0], 4);    // A[row][k] lives in lane row+32k
            memcpy(&bv, &all[col + 32 * k][4], 4);    // B[k][col] lives in lane col+32k
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}

// v_mfma_f32_32x32x16_f16 model: fp32 fma chain over k (the hardware's internal order is
// not specified; tests compare against the fp64 oracle with a tolerance, never bitwise).
inline f32x16 mfma_32x32x16_f16(u32x4 a, u32x4 b, f32x16 c) {
    struct { u32x4 a, b; } mine = {a, b};
    const aae_emu::lane_slot* all = aae_emu::wave_exchange(&mine, sizeof(mine));
    const int lane = lane_id();
    const int col = lane & 31, hi = lane >> 5;
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = d[r];
        for (int k = 0; k < 16; ++k) {
            _Float16 av, bv;
            memcpy(&av, &all[row + 32 * (k >> 3)][2 * (k & 7)], 2);         // A[row][k]: lane row+32*(k/8), half k%8
            memcpy(&bv, &all[col + 32 * (k >> 3)][16 + 2 * (k & 7)], 2);    // B[k][col]
            acc = fmaf((float)av, (float)bv, acc);
        }
        d[r] = acc;
    }
    return d;
}
inline unsigned short bf16_rn(float v) {             // v_cvt_pk_bf16_f32: round to nearest even, a NaN stays a (quiet) NaN
    unsigned u;
    memcpy(&u, &v, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x0040u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
inline unsigned bf16_rn_pack2(float lo, float hi) { return (unsigned)bf16_rn(lo) | ((unsigned)bf16_rn(hi) << 16); }
inline float bf16_to_f32(unsigned short h) {
    const unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
inline void split_bf16x3(float v, unsigned short& t0, unsigned short& t1, unsigned short& t2) {
    t0 = bf16_rn(v);
    const float r1 = v - bf16_to_f32(t0);
    t1 = bf16_rn(r1);
    t2 = bf16_rn(r1 - bf16_to_f32(t1));
}
inline f32x16 mfma_32x32x16_bf16(u32x4 a, u32x4 b, f32x16 c) {
    struct { u32x4 a, b; } mine = {a, b};
    const aae_emu::lane_slot* all = aae_emu::wave_exchange(&mine, sizeof(mine));
    const int lane = lane_id();
    const int col = lane & 31, hi = lane >> 5;
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = d[r];
        for (int k = 0; k < 16; ++k) {
            unsigned short av, bv;
            memcpy(&av, &all[row + 32 * (k >> 3)][2 * (k & 7)], 2);
            memcpy(&bv, &all[col + 32 * (k >> 3)][16 + 2 * (k & 7)], 2);
            acc = fmaf(bf16_to_f32(av), bf16_to_f32(bv), acc);
        }
        d[r] = acc;
    }
    return d;
}
inline void split_f16(float v, unsigned short& hi, unsigned short& lo) {
    const float c = fminf(fmaxf(v, -65504.f), 65504.f);
    const _Float16 h = (_Float16)c;
    const float r = fminf(fmaxf(v - (float)h, -65504.f), 65504.f);
    const _Float16 l = (_Float16)r;
    memcpy(&hi, &h, 2);
    memcpy(&lo, &l, 2);
}

// device_intrinsics.h cross-block hand-off: the emulator runs the blocks of a grid one after the other (in a
// selectable order), so the coherent accesses are plain ones and the ticket protocol (nonce << 32 | arrivals;
// foreign nonce = empty; two levels above 32 arrivals; last arriver leaves 0) is modelled without atomics.
inline void coherent_store4(buffer_rsrc r, uint32_t byte_off, f32x4 v) {
    if ((uint64_t)byte_off + 16 <= r.bytes) memcpy(const_cast<unsigned char*>(r.base) + byte_off, &v, 16);
}
inline f32x4 coherent_load4(buffer_rsrc r, uint32_t byte_off) { return buffer_load4(r, byte_off); }
inline void coherent_store2(buffer_rsrc r, uint32_t byte_off, uint32_t v0, uint32_t v1) {
    const uint32_t v[2] = {v0, v1};
    if ((uint64_t)byte_off + 8 <= r.bytes) memcpy(const_cast<unsigned char*>(r.base) + byte_off, v, 8);
}
inline void coherent_store1(buffer_rsrc r, uint32_t byte_off, uint32_t v) {
    if ((uint64_t)byte_off + 4 <= r.bytes) memcpy(const_cast<unsigned char*>(r.base) + byte_off, &v, 4);
}
inline uint32_t coherent_load1(buffer_rsrc r, uint32_t byte_off) {
    uint32_t v = 0;
    if ((uint64_t)byte_off + 4 <= r.bytes) memcpy(&v, r.base + byte_off, 4);
    return v;
}
constexpr int kTicketGroups = 16;
constexpr int kTicketGroupStride = 16;
constexpr int kTicketSlotWords = (1 + kTicketGroups) * kTicketGroupStride;
constexpr unsigned kTicketSingleLevelMax = 32;
inline unsigned ticket_count(unsigned long long* word, unsigned nonce) {
    *word += 1;                                                           // the optimistic add ...
    if ((unsigned)((*word - 1) >> 32) != nonce) *word = ((unsigned long long)nonce << 32) | 1ull;   // ... met a foreign state: install (nonce, 1)
    return (unsigned)*word;
}
inline void ticket_clear(unsigned long long* word) { *word = 0; }
inline void ticket_prepare_word(unsigned long long* word, unsigned nonce) {
    if ((unsigned)(*word >> 32) != nonce) *word = (unsigned long long)nonce << 32;
}
inline void ticket_prepare_slot(unsigned long long* words, unsigned nonce, unsigned total) {
    const unsigned n = total <= kTicketSingleLevelMax ? 1u : 1u + kTicketGroups;
    if (threadIdx.x < n) ticket_prepare_word(words + threadIdx.x * kTicketGroupStride, nonce);
}
inline void block_ticket_publish() { __syncthreads(); }
inline bool block_ticket_take(unsigned long long* words, unsigned nonce, unsigned total, unsigned id, int* lds_flag) {
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
inline bool block_ticket_arrive(unsigned long long* words, unsigned nonce, unsigned total, unsigned id, int* lds_flag) {
    block_ticket_publish();
    return block_ticket_take(words, nonce, total, id, lds_flag);
}

// grid-wide barrier of a persistent launch (device_intrinsics.h): same words, same protocol, plain accesses; a waiting
// thread hands the processor to the other blocks of the resident launch (aae_emu::launch_resident)
constexpr int kGridBarrierWords = (1 + 2 * kTicketGroups) * kTicketGroupStride;
struct GridBarrier {
    unsigned long long* words;
    unsigned nonce;
};
inline void grid_barrier_arrive(const GridBarrier& gb, unsigned nblocks, unsigned block, unsigned phase) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned groups = nblocks < (unsigned)kTicketGroups ? nblocks : (unsigned)kTicketGroups;
        const unsigned g = block % groups, members = (nblocks - g + groups - 1) / groups;
        if (ticket_count(gb.words + (1 + g) * kTicketGroupStride, gb.nonce) == members * phase &&
            ticket_count(gb.words, gb.nonce) == groups * phase) {
            for (unsigned k = 0; k < groups; ++k) gb.words[(1 + kTicketGroups + k) * kTicketGroupStride] = ((unsigned long long)gb.nonce << 32) | phase;
        }
        aae_emu::note_progress();
    }
}
inline void grid_barrier_wait(const GridBarrier& gb, unsigned nblocks, unsigned block, unsigned phase) {
    if (threadIdx.x == 0) {
        const unsigned groups = nblocks < (unsigned)kTicketGroups ? nblocks : (unsigned)kTicketGroups;
        const volatile unsigned long long* gate = gb.words + (1 + kTicketGroups + block % groups) * kTicketGroupStride;
        for (;;) {
            const unsigned long long v = *gate;
            if ((unsigned)(v >> 32) == gb.nonce && (unsigned)v >= phase) break;
            aae_emu::spin_yield();
        }
        aae_emu::note_progress();
    }
    __syncthreads();
}

// f32x3h activation pairs (device_intrinsics.h): 32-channel chunks of 32 hi halves + 32 lo halves
inline long long x3h_pair_index(long long e) { return ((e >> 5) << 6) + (e & 31); }
constexpr float kHalfPairLimit = 65504.f;
inline void split_f16_checked(float v, unsigned short& hi, unsigned short& lo, int* sat) {
    split_f16(v, hi, lo);
    if (sat && !(fabsf(v) < kHalfPairLimit)) *sat = 1;
}

template <typename T>
inline T shfl_xor(T v, int mask) {
    static_assert(sizeof(T) == 4, "32-bit shuffles only");
    const aae_emu::lane_slot* all = aae_emu::wave_exchange(&v, 4);
    T out;
    memcpy(&out, &all[lane_id() ^ mask][0], 4);
    return out;
}

inline bool wave_any(bool pred) {
    int v = pred ? 1 : 0;
    const aae_emu::lane_slot* all = aae_emu::wave_exchange(&v, 4);
    int any = 0;
    for (int l = 0; l < 64; ++l) { int o; memcpy(&o, &all[l][0], 4); any |= o; }
    return any != 0;
}

// DPP add tree of device_intrinsics.h::row16_sum, same partner order.
inline float row16_sum(float v) {
    const int lane = lane_id();
    const int partners[4] = {lane ^ 1, lane ^ 2, (lane & ~7) | (7 - (lane & 7)), (lane & ~15) | (15 - (lane & 15))};
    for (int s = 0; s < 4; ++s) {
        const aae_emu::lane_slot* all = aae_emu::wave_exchange(&v, 4);
        float o;
        memcpy(&o, &all[partners[s]][0], 4);
        v += o;
    }
    return v;
}

// DPP add tree of device_intrinsics.h::half_wave_sum, same partner order.
inline float half_wave_sum(float v) {
    const int lane = lane_id();
    auto step = [&](int partner, bool enabled) {
        const aae_emu::lane_slot* all = aae_emu::wave_exchange(&v, 4);
        float o;
        memcpy(&o, &all[partner][0], 4);
        if (enabled) v += o;
    };
    step(lane ^ 1, true);                                        // quad_perm [1,0,3,2]
    step(lane ^ 2, true);                                        // quad_perm [2,3,0,1]
    step((lane & ~7) | (7 - (lane & 7)), true);                  // row_half_mirror
    step((lane & ~15) | (15 - (lane & 15)), true);               // row_mirror
    step(((lane >> 4) - 1) * 16 + 15 < 0 ? 0 : ((lane >> 4) - 1) * 16 + 15, ((lane >> 4) & 1) == 1);   // row_bcast15, rows 1 and 3
    return v;
}

// device_intrinsics.h::half_wave_reduce_scatter16: the same tree, level by level (lane bit 4, 3, 2, 1, 0); at every level a
// lane keeps the lower half of its quantities when that lane bit is 0, the upper half when it is 1, and adds its partner's
// value of the same quantity.
inline float reduce_scatter_levels(const float (&d)[16], const int* bits, int nbits, int last_bit) {
    const int lane = lane_id();
    float t[16];
    for (int u = 0; u < 16; ++u) t[u] = d[u];
    int n = 16;
    for (int s = 0; s < nbits; ++s) {
        const aae_emu::lane_slot* all = aae_emu::wave_exchange(t, 64);
        const int bit = bits[s], half = n / 2, up = (lane & bit) ? half : 0;
        float out[16];
        for (int u = 0; u < half; ++u) {
            float o;
            memcpy(&o, &all[lane ^ bit][4 * (u + up)], 4);
            out[u] = t[u + up] + o;
        }
        for (int u = 0; u < half; ++u) t[u] = out[u];
        n = half;
    }
    if (last_bit) {
        const aae_emu::lane_slot* all = aae_emu::wave_exchange(t, 4);
        float o;
        memcpy(&o, &all[lane ^ last_bit][0], 4);
        t[0] += o;
    }
    return t[0];
}
inline float half_wave_reduce_scatter16(const float (&d)[16]) {
    static const int bits[4] = {16, 8, 4, 2};
    return reduce_scatter_levels(d, bits, 4, 1);
}
inline float row16_reduce_scatter16(const float (&d)[16]) {
    static const int bits[4] = {8, 4, 2, 1};
    return reduce_scatter_levels(d, bits, 4, 0);
}
inline float wave_max_first_lane(float v, int& first_lane) {
    // device_intrinsics.h: NaNs rank as -inf, -0 as +0; lowest lane among the holders of the maximum
    const float z = v + 0.f;
    const float s = (z == z) ? z : -INFINITY;
    const aae_emu::lane_slot* all = aae_emu::wave_exchange(&s, 4);
    float m = -INFINITY;
    for (int l = 0; l < 64; ++l) { float o; memcpy(&o, &all[l][0], 4); if (o > m) m = o; }
    first_lane = 0;
    for (int l = 63; l >= 0; --l) { float o; memcpy(&o, &all[l][0], 4); if (o == m) first_lane = l; }
    return m;
}

template <int NV>
inline float wave_max_first_position(const float (&v)[NV], int& first_pos) {
    // device_intrinsics.h: value j of lane l stands for position 64 j + l; NaNs rank as -inf, -0 as +0
    float s[NV];
    for (int j = 0; j < NV; ++j) { const float z = v[j] + 0.f; s[j] = (z == z) ? z : -INFINITY; }
    const aae_emu::lane_slot* all = aae_emu::wave_exchange(s, 4 * NV);
    float m = -INFINITY;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < NV; ++j) { float o; memcpy(&o, &all[l][4 * j], 4); if (o > m) m = o; }
    first_pos = 0;
    for (int j = NV - 1; j >= 0; --j)
        for (int l = 63; l >= 0; --l) { float o; memcpy(&o, &all[l][4 * j], 4); if (o == m) first_pos = 64 * j + l; }
    return m;
}

}  // namespace aae

// ---- the sliver of the HIP runtime API the host-side launch code uses ----
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
// the emulated "device" has three compute units: persistent launches size their grid to it (aae_emu::launch_resident)
enum { hipDeviceAttributeMultiprocessorCount = 63 };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 3; return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
struct hipEvent_emu {};
typedef hipEvent_emu* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipEvent_emu; return hipSuccess; }
constexpr unsigned hipEventDisableTiming = 2;
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipEvent_emu; return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }      // (launches run to completion inside the call)
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 1; return hipSuccess; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

#define AAE_LAUNCH(kernel, grid, block, smem, stream, ...) \
    aae_emu::launch((grid), (block), (smem), [&]() { kernel(__VA_ARGS__); })
// a launch whose blocks must all be resident (grid-wide waits inside the kernel)
#define AAE_LAUNCH_RESIDENT(kernel, grid, block, smem, stream, ...) \
    aae_emu::launch_resident((grid), (block), (smem), [&]() { kernel(__VA_ARGS__); })
