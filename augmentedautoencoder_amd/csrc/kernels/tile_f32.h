// Shared fp32-MFMA tile machinery (v_mfma_f32_32x32x2_f32) for the implicit-GEMM
// convolution, the split-K dense layer and the codebook scan.
//
// K-slab convention.  A K-slab is 32 consecutive k.  In LDS it is held as 8
// "slots" of 4 consecutive k (16 B):
//   A slab : [128 rows][8 slots][4]  row-major, slot XOR-swizzled by ((row>>1)&7)
//            so that the 16 lanes of a ds_read_b128 group (16 distinct rows at
//            one slot) land on 16 distinct 16-B bank slots.
//   B slab : [8 slots][ncols][4]     (weights / queries are pre-packed in global
//            memory as [K/4][ncols][4], so the slab is a straight 16-B copy).
// One ds_read_b128 at slot (2c+h) gives lane (i, h = lane>>5) the four k values
// 4*(2c+h)+q, q=0..3.  MFMA step (c,q) therefore contracts k in
// {8c+q, 8c+4+q}: every k of the slab is visited exactly once, A and B use the
// same pairing, and each output is one k-ordered fp32 fma chain per slab.
#pragma once

namespace aae {

constexpr int kBM = 128;          // tile rows
constexpr int kBK = 32;           // k per slab
constexpr int kSlabFloatsA = kBM * kBK;

__device__ __forceinline__ int a_slab_off(int row, int slot) {
    return row * kBK + ((slot ^ ((row >> 1) & 7)) << 2);
}

__device__ __forceinline__ f32x4 lds_read4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void lds_write4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// XCD-aware bijective remap of a 1-D grid: physical block p runs on XCD p%8
// (observed dispatch; a wrong guess costs speed only).  Each XCD gets one
// contiguous chunk of the logical order so neighbouring tiles share its L2.
__device__ __forceinline__ int xcd_remap(int p, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = p & 7, within = p >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + within;
}

// One K-slab of MFMAs for a wave that owns MT x NT tiles of 32x32:
//   rows  a_row0 + 32*mi + (lane&31)            of the A slab
//   cols  b_col0 + 32*ni + (lane&31)            of the B slab (b_cols wide)
template <int MT, int NT>
__device__ __forceinline__ void mfma_slab(const float* As, const float* Bs, int b_cols, int a_row0, int b_col0,
                                          int lane, f32x16 (&acc)[MT][NT]) {
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int slot = 2 * c + h;
        f32x4 a[MT], b[NT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) a[mi] = lds_read4(As + a_slab_off(a_row0 + 32 * mi + i, slot));
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) b[ni] = lds_read4(Bs + (slot * b_cols + b_col0 + 32 * ni + i) * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = mfma_32x32x2(a[mi][q], b[ni][q], acc[mi][ni]);
    }
}

// The same slab split into its 4 k-groups, so a caller can keep two fragment sets in
// registers (reads for group c+1 in flight under the 16 MFMAs of group c) and slot other
// work (LDS stores of the next slab) between groups.
template <int MT, int NT>
__device__ __forceinline__ void frag_load(const float* As, const float* Bs, int b_cols, int a_row0, int b_col0, int lane,
                                          int c, f32x4 (&a)[MT], f32x4 (&b)[NT]) {
    const int i = lane & 31, slot = 2 * c + (lane >> 5);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) a[mi] = lds_read4(As + a_slab_off(a_row0 + 32 * mi + i, slot));
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) b[ni] = lds_read4(Bs + (slot * b_cols + b_col0 + 32 * ni + i) * 4);
}

template <int MT, int NT>
__device__ __forceinline__ void frag_mfma(const f32x4 (&a)[MT], const f32x4 (&b)[NT], f32x16 (&acc)[MT][NT]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = mfma_32x32x2(a[mi][q], b[ni][q], acc[mi][ni]);
}

// one q-step (MT*NT MFMAs) of a k-group
template <int MT, int NT>
__device__ __forceinline__ void frag_mfma_q(const f32x4 (&a)[MT], const f32x4 (&b)[NT], int q, f32x16 (&acc)[MT][NT]) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) acc[mi][ni] = mfma_32x32x2(a[mi][q], b[ni][q], acc[mi][ni]);
}

// accumulator register r of a 32x32 tile -> row inside the tile
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Scores as integers that order like the scores (for integer max on words shared between blocks); kScoreKeyEmpty is
// below every score's key, -inf included.
constexpr int kScoreKeyEmpty = (int)0x80000000;
__device__ __forceinline__ int score_key(float v) { const int b = __builtin_bit_cast(int, v); return b >= 0 ? b : b ^ 0x7fffffff; }
__device__ __forceinline__ float score_of_key(int k) {
    return k == kScoreKeyEmpty ? -__builtin_huge_valf() : __builtin_bit_cast(float, k >= 0 ? k : k ^ 0x7fffffff);
}
constexpr int kPruneGroups = 16;      // shared words per query: block b publishes to word b % 16 ...
constexpr int kPruneReplicas = 8;     // ... of EVERY replica [replica][query][word]; block b reads replica b % 8 -- 241 blocks re-reading
                                      // the same 64-byte line with device-coherent loads queue up behind each other (12 us per refresh at
                                      // config 5), 30 do not, and the few publications that happen cost nothing measurable (top-5 kernel with 1 / 2 / 4 / 8 / 16
                                      // replicas: 97.9 / 93.0 / 90.4 / 91.6 / 101.5 us)

// (score, index) ordering of np.argmax: higher score wins, lower index wins ties.
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) {
    return (v > bv) || (v == bv && i < bi);
}

}  // namespace aae
