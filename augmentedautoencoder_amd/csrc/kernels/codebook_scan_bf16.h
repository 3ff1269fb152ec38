// Codebook scan against a bf16 codebook (BASELINE config 5: 4x rows, bf16 storage, batched
// queries on the bf16 matrix cores).  Same contract as codebook_scan_f32.h
// (/root/reference/auto_pose/ae/codebook.py:27,50,64-71) -- the codebook rows are bf16, the
// cosine stays inside the 1e-5 contract: each normalised query is split into bf16 terms
// q = t0 + t1 + t2 (24 significant bits) of which the matrix-core kernels use the first
// kBf16QueryTerms = 2 -- one v_mfma_f32_32x32x16_bf16 per term and 32x32x16 tile, fp32
// accumulation; the bf16 rows are used exactly as stored.
// Error of two terms: |q_j - t0 - t1| <= 2^-18 |q_j| (each rounding to bf16's 8-bit significand is
// within 2^-9 of what it rounds), so |cos - cos_exact| <= 2^-18 sum |q_j||e_j| <= 2^-18 |q||e| = 3.8e-6
// for unit vectors (Cauchy-Schwarz), typically ten times less; fp32 accumulation adds ~1e-6.  The third
// term bought 2^-26 at 50 % more MFMAs: config 5 (B = 256, 368928 rows) is bound by the bf16 matrix pipe --
// arg-max 82 -> 68 us at the time (round 3, profiles/r11; 50 us after the other changes of codebook_scan_resident.h).
// The B <= 4 streaming kernel keeps full fp32 queries (vector ALU).
//
// Block = 128 codebook rows (32 KB, staged once in LDS with coalesced 16-B loads, XOR
// swizzled so the fragment reads are conflict-free) x passes of 64 queries; the running
// (max, first row) reduction and the partial-result format are those of the fp32 scan.
#pragma once

namespace aae {

constexpr int kBf16QueryTerms = 2;       // bf16 terms of a query the MFMA kernels multiply (of the three that are packed)

struct L2NormBf16Args {
    const float* z;          // [B][J]
    unsigned short* qp3;     // [3 terms][Jpad/8 slots][Bpad][8] bf16
    int B, J, Jpad, Bpad;
    int* prune = nullptr;    // optional [kPruneReplicas][Bpad][kPruneGroups]: reset for the top-k scan that follows (codebook_scan_resident.h)
};

// one wave per query row b < Bpad (rows >= B are written as zeros)
__global__ __launch_bounds__(256) void l2norm_pack_bf16x3_kernel(const L2NormBf16Args p) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= p.Bpad) return;
    if (p.prune)
        for (int w = lane; w < kPruneReplicas * kPruneGroups; w += 64) p.prune[((w / kPruneGroups) * p.Bpad + b) * kPruneGroups + w % kPruneGroups] = kScoreKeyEmpty;
    const bool real = b < p.B;
    float ss = 0.f;
    if (real)
        for (int j = lane; j < p.J; j += 64) { const float v = p.z[(long long)b * p.J + j]; ss = fmaf(v, v, ss); }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) ss += shfl_xor(ss, m);
    const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    const long long plane = (long long)(p.Jpad / 8) * p.Bpad * 8;
    for (int j = lane; j < p.Jpad; j += 64) {
        const float v = (real && j < p.J) ? p.z[(long long)b * p.J + j] * inv : 0.f;
        unsigned short t0, t1, t2;
        split_bf16x3(v, t0, t1, t2);
        const long long o = ((long long)(j >> 3) * p.Bpad + b) * 8 + (j & 7);
        p.qp3[o] = t0;
        p.qp3[plane + o] = t1;
        p.qp3[2 * plane + o] = t2;
    }
}

struct ScanBf16Args {
    const unsigned short* E;   // [N][J] bf16 row-major, J == 128
    unsigned e_bytes;
    const unsigned short* qp3; // [3][16][Bpad][8]
    float* pval;
    int* pidx;
    float* cs;                 // optional [B][N]
    int N, B, Bpad, Bstride;
    int col_stride;
};

constexpr int kScanBf16QC = 64;                                   // queries per pass
constexpr int kScanBf16Smem = 128 * 256 + 3 * 16 * kScanBf16QC * 16 + 2 * 4 * kScanBf16QC * 4;

__device__ __forceinline__ int e16_tile_off(int row, int slot) { return row * 64 + ((slot ^ (row & 15)) << 2); }   // dwords

template <bool UPRIGHT>
__global__ __launch_bounds__(256) void scan_bf16_kernel(const ScanBf16Args p) {
    constexpr int QC = kScanBf16QC;
    AAE_DYN_SMEM(smem_raw);
    float* Et = reinterpret_cast<float*>(smem_raw);                // [128 rows][16 slots x 16 B]
    float* Qt = Et + 128 * 64;                                     // [3][16 slots][QC][16 B]
    float* red_v = Qt + 3 * 16 * QC * 4;                           // [4][QC]
    int* red_i = reinterpret_cast<int*>(red_v + 4 * QC);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * 128;

    const buffer_rsrc ebuf = make_buffer(p.E, p.e_bytes);
#pragma unroll
    for (int u = 0; u < 8; ++u) {                                  // 128 rows x 16 slots = 2048 16-B pieces
        const int idx = tid + 256 * u;
        const int r = idx >> 4, slot = idx & 15;
        const int row = row0 + r;
        const f32x4 v = buffer_load4(ebuf, row < p.N ? (unsigned)(row * 256 + slot * 16) : kOobOffset);
        lds_write4(Et + e16_tile_off(r, slot), v);
    }
    const long long qplane = (long long)16 * p.Bpad * 8;           // halves per term plane
    for (int qt = 0; qt < p.Bpad; qt += QC) {
        __syncthreads();
        for (int idx = tid; idx < kBf16QueryTerms * 16 * QC; idx += 256) {
            const int term = idx / (16 * QC), rem = idx - term * (16 * QC);
            const int slot = rem / QC, c = rem - slot * QC;
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.qp3 + term * qplane + ((long long)slot * p.Bpad + qt + c) * 8);
            lds_write4(Qt + idx * 4, v);
        }
        __syncthreads();

        f32x16 acc[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
#pragma unroll 2
        for (int s = 0; s < 8; ++s) {
            const int slot = 2 * s + h;
            const u32x4 a = __builtin_bit_cast(u32x4, lds_read4(Et + e16_tile_off(wave * 32 + i, slot)));
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                for (int term = kBf16QueryTerms - 1; term >= 0; --term) {   // smallest term first
                    const u32x4 b = __builtin_bit_cast(u32x4, lds_read4(Qt + ((term * 16 + slot) * QC + ni * 32 + i) * 4));
                    acc[ni] = mfma_32x32x16_bf16(a, b, acc[ni]);
                }
            }
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int query = qt + ni * 32 + i;
            float bv = kNegInf;
            int bi = row0 + wave * 32 + acc_row(0, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + wave * 32 + acc_row(r, lane);
                const float v = acc[ni][r];
                bool cand = row < p.N;
                if (UPRIGHT) cand = cand && (row % p.col_stride == 0);
                if (p.cs && row < p.N && query < p.B) p.cs[(long long)query * p.N + row] = v;
                if (cand && v > bv) { bv = v; bi = row; }
            }
            const float ov = shfl_xor(bv, 32);
            const int oi = shfl_xor(bi, 32);
            if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
            if (h == 0) { red_v[wave * QC + ni * 32 + i] = bv; red_i[wave * QC + ni * 32 + i] = bi; }
        }
        __syncthreads();
        if (tid < QC && qt + tid < p.B) {
            float v = red_v[tid];
            int ix = red_i[tid];
            for (int w = 1; w < 4; ++w)
                if (better(red_v[w * QC + tid], red_i[w * QC + tid], v, ix)) { v = red_v[w * QC + tid]; ix = red_i[w * QC + tid]; }
            p.pval[(long long)blockIdx.x * p.Bstride + qt + tid] = v;
            p.pidx[(long long)blockIdx.x * p.Bstride + qt + tid] = ix;
        }
    }
}

// ---------------------------------------------------------------- scan_stream_bf16
// B <= 4 against a bf16 codebook: the HBM-streaming form (cf. scan_stream_kernel in codebook_scan_f32.h).
// A row is 256 B = 16 lanes x 16 B, so one 1-KiB wave load covers 4 rows; a wave owns 64 consecutive rows -- the 16-lane
// group g reads rows 16 g ... 16 g + 15, one per load -- and puts all 16 loads in flight (behind the query loads) before
// it touches any.  The bf16 elements are widened exactly (<< 16), the dot products are fp32 fma chains against the
// fp32-normalised queries (no query splitting needed off the matrix cores); row16_reduce_scatter16 sums the 16 rows of a
// group over its 16 lanes in 45 cross-lane instructions (16 separate four-step trees: 64) and leaves row l of the wave's
// 64 in lane l; the block's scores meet in LDS and wave b finds (max, first row) of query b (scan_block_argmax_store).
template <int NQ, bool UPRIGHT, bool WITH_CS>
__global__ __launch_bounds__(256) void scan_stream_bf16_kernel(const ScanArgs p) {
    AAE_DYN_SMEM(smem_raw);
    float* sc = reinterpret_cast<float*>(smem_raw);              // [NQ][256] scores of the block's rows

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rs = lane >> 4, kq = lane & 15;                    // 16-lane group, 8-element column group
    const int row_first = blockIdx.x * 256 + wave * 64;

    f32x4 z0[NQ], z1[NQ];
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
        z0[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        z1[b] = z0[b];
        if (b < p.B) {
            z0[b] = *reinterpret_cast<const f32x4*>(p.z + (long long)b * 128 + kq * 8);
            z1[b] = *reinterpret_cast<const f32x4*>(p.z + (long long)b * 128 + kq * 8 + 4);
        }
    }
    const buffer_rsrc ebuf = make_buffer(p.E, p.e_bytes);
    u32x4 e[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int r = row_first + 16 * rs + u;
        e[u] = __builtin_bit_cast(u32x4, buffer_load4(ebuf, r < p.N ? (unsigned)(r * 256 + kq * 16) : kOobOffset));
    }

    if (p.tickets && blockIdx.x == 0) ticket_prepare_slot(p.tickets, p.nonce, gridDim.x);   // loads in flight; arrivals come later

    // tf.nn.l2_normalize(z, 1) per query, 8 columns per lane, replicated in every 16-lane group
    float qv[NQ][8];
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
        float ss = z0[b].x * z0[b].x;
        ss = fmaf(z0[b].y, z0[b].y, ss); ss = fmaf(z0[b].z, z0[b].z, ss); ss = fmaf(z0[b].w, z0[b].w, ss);
        ss = fmaf(z1[b].x, z1[b].x, ss); ss = fmaf(z1[b].y, z1[b].y, ss); ss = fmaf(z1[b].z, z1[b].z, ss); ss = fmaf(z1[b].w, z1[b].w, ss);
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) ss += shfl_xor(ss, m);
        const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        qv[b][0] = z0[b].x * inv; qv[b][1] = z0[b].y * inv; qv[b][2] = z0[b].z * inv; qv[b][3] = z0[b].w * inv;
        qv[b][4] = z1[b].x * inv; qv[b][5] = z1[b].y * inv; qv[b][6] = z1[b].z * inv; qv[b][7] = z1[b].w * inv;
    }

    const int row = row_first + lane;
    bool cand = row < p.N;
    if (UPRIGHT) cand = cand && (row % p.col_stride == 0);
    float d[NQ][16];                                             // the lane's share of row 16 rs + u against query b
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const u32x4 eu = e[u];                                    // (whole vector first: see device_intrinsics.h on bit_casts of single elements)
        const uint32_t w0 = eu[0], w1 = eu[1], w2 = eu[2], w3 = eu[3];
        float ev[8];                                              // element 2w (low half), 2w + 1 (high half), widened exactly
        ev[0] = __builtin_bit_cast(float, w0 << 16); ev[1] = __builtin_bit_cast(float, w0 & 0xFFFF0000u);
        ev[2] = __builtin_bit_cast(float, w1 << 16); ev[3] = __builtin_bit_cast(float, w1 & 0xFFFF0000u);
        ev[4] = __builtin_bit_cast(float, w2 << 16); ev[5] = __builtin_bit_cast(float, w2 & 0xFFFF0000u);
        ev[6] = __builtin_bit_cast(float, w3 << 16); ev[7] = __builtin_bit_cast(float, w3 & 0xFFFF0000u);
#pragma unroll
        for (int b = 0; b < NQ; ++b) {
            float t = ev[0] * qv[b][0];
#pragma unroll
            for (int j = 1; j < 8; ++j) t = fmaf(ev[j], qv[b][j], t);
            d[b][u] = t;
        }
    }
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
        const float dot = row16_reduce_scatter16(d[b]);          // lane l: row row_first + l
        if (WITH_CS) {
            if (row < p.N && b < p.B) p.cs[(long long)b * p.N + row] = dot;
        }
        sc[b * 256 + wave * 64 + lane] = cand ? dot : kNegInf;
    }
    __syncthreads();
    scan_block_argmax_store<NQ, 256>(p, sc, blockIdx.x * 256, blockIdx.x, gridDim.x);   // wave b: (max, first row) of query b over the block's 256 rows
    if (p.tickets) scan_ticket_finish<NQ>(p, sc + NQ * 256, blockIdx.x, gridDim.x);
}

}  // namespace aae
